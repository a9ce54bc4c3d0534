/* hz_fast.h - the vertex transform of hz_num.h once more, for the marching
 * kernel, with every IEEE division and square root written as the instruction
 * sequence hipcc itself emits for `a/b` and `sqrtf(x)` - minus the parts of that
 * sequence that only exist for operands near the ends of the float32 range.
 *
 * Why: k_march is bound by the SIMDs' vector issue rate (profiles/valu_issue.json:
 * v_fma/v_mul/v_add_f32 2.7-2.9 cycles per wave64 instruction, most others 4.2-4.6,
 * v_rcp/v_rsq/v_sqrt 8), and with hipcc's own sequences seven divisions (46 cycles
 * each) and two square roots (56) were half of what its transform cost per row of
 * 64 vertices.  Round 3: those sequences without their range handling (below).
 * Round 5: shorter sequences still, each proved by trying every operand (further
 * below) - 164 -> 127 instructions per vertex, profiles/r5_ab_short_sequences.txt.
 *
 * hipcc's float32 division (default, correctly rounded) is
 *     ds = v_div_scale(b,b,a)   as = v_div_scale(a,b,a)      [pre-scaling by 2^+-64]
 *     r  = v_rcp(ds); e = fma(-ds,r,1); r = fma(e,r,r)
 *     q  = as*r; e = fma(-ds,q,as); q = fma(e,r,q); e = fma(-ds,q,as)
 *     q  = v_div_fmas(e,r,q)                                  [fma, then undo the scaling]
 *     q  = v_div_fixup(q,b,a)                                 [zeros, infinities, NaNs]
 * v_div_scale returns its operand unchanged, v_div_fmas is a plain fma and
 * v_div_fixup passes q through whenever (CDNA4 ISA, V_DIV_SCALE_F32 /
 * V_DIV_FIXUP_F32): a and b are finite, b is normal with 2^-125 <= |b| <= 2^126,
 * a = 0 or |a| >= 2^-103, the exponents of a and b differ by less than 96 and
 * the quotient is normal.  Inside that range the sequence below IS the IEEE
 * sequence, operation for operation, so the result is the same bit pattern by
 * construction - nothing is approximated and no theorem about roundings is
 * needed.  The same holds for the square root: hipcc's sequence is v_sqrt
 * (1 ulp), then the two neighbours s-1ulp and s+1ulp tried with an fma
 * residual; the input scaling only acts below 2^-96 and the class test only on
 * zero and infinity.
 *
 * One difference remains inside the range: a numerator of -0 comes out as +0
 * (v_div_fixup would restore the sign).  No numerator of the transform can be
 * -0: they are differences and sums of finite floats, which round-to-nearest
 * makes +0 when they vanish, or absolute values.
 *
 * The range conditions are established once per draw (hzf_draw_ok, host), once
 * per strip (the east offsets) and once per row (the north offset) - see
 * k_march; whatever fails them takes hz_transform_en(), the unabridged code.
 * With a divisor that is a per-draw constant the reciprocal refinement is done
 * once per wave (hzf_setup) and a division costs five instructions.
 *
 * Round 5: where the operand is ONE float - a reciprocal, a square root, a
 * quotient by 2 pi - "the same bits as `/` and sqrtf" can be had for every
 * operand by trying them all, and tools/exact_seq.hip did, on the MI355X, for
 * shorter sequences than hipcc's (profiles/r5_exact_sequences.txt):
 *     1/b         v_rcp + one Newton step (3 instructions, not 7): the correctly
 *                 rounded reciprocal of every float with 2^-62 <= |b| <= 2^62
 *     sqrt(x)     v_rsq, s = x*y, h = y/2, s + (x - s*s)*h (5, not 9): the
 *                 correctly rounded root of every float from 2^-96 on
 *     a/(2 pi)    one correction step (3, not 5): every numerator that is zero or
 *                 has 2^-62 <= |a| <= 2^30
 * and the transform's remaining general division, min(t,1)/max(t,1), is t itself
 * or the reciprocal of t.  These are statements about gfx950's v_rcp_f32 and
 * v_rsq_f32, not about IEEE arithmetic: tests/test_gpu_fastmath.py tries every
 * operand again on whatever device the library runs on.
 *
 * tests/test_gpu_fastmath.py runs the abridged sequences against `/` and
 * sqrtf on the device: every float32 bit pattern for the reciprocal, the
 * square root and the quotients by a draw's constants, 2^32 seeded pairs for
 * the general division (hzf_div: no longer part of the transform).
 */
#pragma once

#include "hz_num.h"

#ifdef __HIPCC__

/* operands of the abridged sequences must lie in [HZF_LO, HZF_HI] (or, where
 * noted, be zero): 2^-30 .. 2^30 */
#define HZF_LO 9.31322575e-10f
#define HZF_HI 1073741824.0f

__device__ static inline int hzf_in_range(float x) { const float a = hz_abs(x); return a >= HZF_LO && a <= HZF_HI; }

/* r = v_rcp(c) refined once: what the division sequence multiplies with */
__device__ static inline float hzf_refined_rcp(float c)
{
    const float r = __builtin_amdgcn_rcpf(c);
    const float e = __builtin_fmaf(-c, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
/* a/c, given rr = hzf_refined_rcp(c) */
__device__ static inline float hzf_div_by(float a, float c, float rr)
{
    float q = a*rr;
    float e = __builtin_fmaf(-c, q, a);
    q = __builtin_fmaf(e, rr, q);
    e = __builtin_fmaf(-c, q, a);
    return __builtin_fmaf(e, rr, q);
}
__device__ static inline float hzf_div(float a, float b) { return hzf_div_by(a, b, hzf_refined_rcp(b)); }
/* 1.0f/b for 2^-62 <= |b| <= 2^62: the refined reciprocal IS the correctly rounded one (every such b tried) */
__device__ static inline float hzf_rcp(float b)          { return hzf_refined_rcp(b); }
/* a/c for c = 2 pi, rr = hzf_refined_rcp(c), a zero or 2^-62 <= |a| <= 2^30: one correction step (every such a tried) */
__device__ static inline float hzf_div_by_two_pi(float a, float rr)
{
    const float q = a*rr;
    const float e = __builtin_fmaf(-HZ_TWO_PI, q, a);
    return __builtin_fmaf(e, rr, q);
}
/* min(t,1)/max(t,1) for t = 0 or 2^-62 <= t <= 2^62: t/1 or 1/t */
__device__ static inline float hzf_fold_to_unit(float t) { return (1.0f < t) ? hzf_rcp(t) : t; }

/* sqrtf(x), 2^-96 <= x < inf (every such x tried): Newton's step from x*rsq(x) with rsq(x)/2 for 1/(2 sqrt x) */
__device__ static inline float hzf_sqrt(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x*y;
    const float h = 0.5f*y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}

/* per-wave constants of a draw */
typedef struct { float rr_two_pi, rr_zrange, rr_crange, zrange, crange; } hzf_const_t;

__device__ static inline hzf_const_t hzf_setup(const hz_xform_t* u)
{
    hzf_const_t c;
    c.zrange = u->zfar - u->znear;
    c.crange = u->zfar_color - u->znear_color;
    /* (the divisor goes through a register the compiler cannot see into: a
     * constant-folded reciprocal would be the correctly rounded 1/(2 pi), not
     * necessarily what v_rcp_f32 returns - and "the same sequence as `/`" means
     * the hardware's) */
    float two_pi = HZ_TWO_PI;
    asm volatile("" : "+v"(two_pi));
    c.rr_two_pi = hzf_refined_rcp(two_pi);
    c.rr_zrange = hzf_refined_rcp(c.zrange);
    c.rr_crange = hzf_refined_rcp(c.crange);
    return c;
}

/* hz_atan2() for 2^-30 <= |t| <= 2^30 (t: the denominator it picks) and s = 0 or
 * in that range too: the 1/4 scaling of huge denominators (>= 1e18) never acts,
 * t*1 = t and s*1 = s exactly.  XPOS: x > 0 is known (the elevation angle, whose
 * x is a distance). */
/* (rcp: 1.0f/t of the denominator t the function picks - x's magnitude, or y where x <= 0 -, which the marching wave
 * has ahead of time for the azimuth: its y is the same for every row of a strip, its x the same for every lane of a row) */
template<bool XPOS>
__device__ static inline float hzf_atan2_r(float y, float x, float rcp)
{
    const bool  flip = XPOS ? false : (0.f >= x);
    const float ax   = hz_abs(x);
    const float s    = flip ? ax : y;
    const float sot  = s * rcp;
    const float tn   = (ax == hz_abs(y)) ? 1.0f : hz_abs(sot);

    const float u  = hzf_fold_to_unit(tn);
    const float u2 = u*u;
    const float u3 = u2*u;
    const float u5 = u3*u2;
    const float u7 = u5*u2;
    const float u9 = u7*u2;
    float p = u*0.9999793128310355f + u3*-0.3326756418091246f;
    p = p + u5*0.1938924977115610f;
    p = p + u7*-0.1173503194786851f;
    p = p + u9*0.0536813784310406f;
    p = p + (u9*-0.0121323213173444f)*u2;
    const float a   = (1.0f < tn) ? ((p*-2.0f + HZ_HALF_PI) + p) : p;
    const float arc = flip ? (HZ_HALF_PI + a) : a;
    /* (min(y, rcp) < 0 ? -arc : arc) - arc is +0 or positive (p >= 0 on [0, 1]), y is not -0 (see above) and rcp is a
     * number other than zero: the result's sign bit is the OR of theirs */
    const uint32_t sign = ((uint32_t)__float_as_int(y) | (uint32_t)__float_as_int(rcp)) & 0x80000000u;
    return __int_as_float((int)((uint32_t)__float_as_int(arc) | sign));
}
template<bool XPOS>
__device__ static inline float hzf_atan2(float y, float x)
{
    const float t = (XPOS ? false : (0.f >= x)) ? y : hz_abs(x);
    return hzf_atan2_r<XPOS>(y, x, hzf_rcp(t));
}

/* hz_transform_en() under the range conditions: e, n in [2^-30, 2^30] in
 * magnitude (not zero), h = fz - viewer_z zero or in that range, the depth and
 * colour extents and their spans in that range (hzf_draw_ok) */
/* (in two halves, as hz_num.h's: hzf_polar_en() is what depends on the viewer's position alone, hzf_finish() the rest) */
/* (rcp_az: 1.0f/e where n <= 0, 1.0f/|n| elsewhere - hzf_rcp() of it) */
__device__ static inline hz_polar_t hzf_polar_en_r(const hz_xform_t* u, float e, float n, float fz, float rcp_az)
{
    hz_polar_t q;
    const float h = fz - u->viewer_z;
    const float nn = n*n, ee = e*e;
    q.d_ne  = hzf_sqrt(nn + ee);
    q.az    = hzf_atan2_r<false>(e, n, rcp_az);
    q.el    = hzf_atan2<true>(h, q.d_ne);
    q.d_enh = hzf_sqrt(h*h + nn + ee);
    return q;
}
__device__ static inline hz_polar_t hzf_polar_en(const hz_xform_t* u, float e, float n, float fz)
{
    return hzf_polar_en_r(u, e, n, fz, hzf_rcp((0.f >= n) ? e : hz_abs(n)));
}
__device__ static inline hz_vertex_t hzf_finish(const hz_xform_t* u, const hzf_const_t* c, hz_polar_t q)
{
    hz_vertex_t v;
    const float d = hzf_div_by_two_pi(q.az + -u->az_center, c->rr_two_pi);
    v.x = (HZ_TWO_PI*(d - hz_roundeven(d))) * u->az_ndc_per_rad;
    v.y = q.el * u->aspect * u->az_ndc_per_rad;
    v.z = hzf_div_by(q.d_enh - u->znear, c->zrange, c->rr_zrange) * 2.0f + -1.0f;

    const float r = hzf_div_by(q.d_ne - u->znear_color, c->crange, c->rr_crange);
    v.red = hz_min(hz_max(r, 0.0f), 1.0f);
    return v;
}
__device__ static inline hz_vertex_t hzf_transform_en(const hz_xform_t* u, const hzf_const_t* c, float e, float n, float fz)
{
    return hzf_finish(u, c, hzf_polar_en(u, e, n, fz));
}
__device__ static inline hz_vertex_t hzf_transform_en_r(const hz_xform_t* u, const hzf_const_t* c, float e, float n, float fz, float rcp_az)
{
    return hzf_finish(u, c, hzf_polar_en_r(u, e, n, fz, rcp_az));
}

#endif /* __HIPCC__ */

/* host and device: may a draw with these uniforms use the abridged sequences?
 * (everything finite; viewer height not so small that an elevation minus it
 * could be tiny without being zero; extents and their spans in range; the
 * azimuth centre zero or not tiny) */
HZ_HD int hzf_draw_ok(const hz_xform_t* u)
{
    const float lo = 9.31322575e-10f, hi = 1073741824.0f;
    const float vz = hz_abs(u->viewer_z), ce = hz_abs(u->az_center);
    const float zr = hz_abs(u->zfar - u->znear), cr = hz_abs(u->zfar_color - u->znear_color);
    if(!(vz == 0.0f || (vz >= 0.015625f && vz <= 536870912.0f))) return 0;
    if(!(ce == 0.0f || (ce >= 8.8817842e-16f && ce <= 1048576.0f))) return 0;
    if(!(u->znear >= lo && u->znear <= hi && hz_abs(u->zfar) >= lo && hz_abs(u->zfar) <= hi && zr >= lo && zr <= hi)) return 0;
    if(!(hz_abs(u->znear_color) >= lo && hz_abs(u->znear_color) <= hi &&
         hz_abs(u->zfar_color) >= lo && hz_abs(u->zfar_color) <= hi && cr >= lo && cr <= hi)) return 0;
    if(!(hz_abs(u->aspect) <= hi && hz_abs(u->az_ndc_per_rad) <= hi)) return 0;
    return 1;
}
