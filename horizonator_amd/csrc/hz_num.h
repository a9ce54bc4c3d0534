/* hz_num.h - the float32 arithmetic of the vertex stage, written once and used
 * by the HIP kernels (device) and by the C host (per-draw constants).
 *
 * Rules, so that results do not depend on who compiles this:
 *   - float32 only, every operation written out; build with -ffp-contract=off
 *   - only + - * / sqrt (IEEE, correctly rounded on x86 and on gfx950 with
 *     hipcc's default correctly-rounded divide/sqrt) and round-to-nearest-even
 *   - no libm transcendental: atan is the polynomial below
 *
 * The operation order is the one the reference's vertex shader (reference
 * vertex.glsl:30-38, 111-162) has AFTER Mesa's GLSL compiler, i.e. what the
 * reference actually computes when it runs on llvmpipe: Rearth*pi folded into
 * one constant, atan(y,x) as Mesa's degree-11 polynomial with s*(1/t),
 * length() summed as n*n + e*e (+ h*h first), the azimuth unwrap simplified.
 * With it the vertex stage reproduces the reference's gl_Position and colour
 * bit for bit (DESIGN.md "Numerics").
 */
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
  #define HZ_HD __host__ __device__ static inline
#else
  #define HZ_HD static inline
#endif

#define HZ_REARTH_PI  20015088.0f       /* float(6371000 * 3.14159265358979)        */
#define HZ_PI         3.14159274f       /* float(3.14159265358979), vertex.glsl:31  */
#define HZ_TWO_PI     6.28318548f
#define HZ_HALF_PI    1.57079637f
#define HZ_DEG2RAD    0.0174532924f     /* GLSL radians()                           */

HZ_HD float hz_roundeven(float x) { return __builtin_rintf(x); }
HZ_HD float hz_sqrt(float x)      { return __builtin_sqrtf(x); }
HZ_HD float hz_abs(float x)       { return __builtin_fabsf(x); }
HZ_HD float hz_min(float a, float b) { return a < b ? a : b; }
HZ_HD float hz_max(float a, float b) { return a > b ? a : b; }

/* GLSL atan(y,x) (reference vertex.glsl:134,153) as Mesa lowers it.
 * Result in [-pi, pi].  No input produces a NaN: y = x = 0 yields 3*pi/4,
 * which is also what the reference's shader returns on llvmpipe for a viewer
 * standing exactly on a grid sample. */
HZ_HD float hz_atan2(float y, float x)
{
    const int   flip  = (0.f >= x);
    const float ax    = hz_abs(x);
    const float s     = flip ? ax : y;
    const float t     = flip ? y  : ax;
    /* Mesa scales huge denominators by 1/4 before the reciprocal (kept as a
     * multiplication by a selected constant: branch-free) */
    const float scale = (hz_abs(t) >= 1e18f) ? 0.25f : 1.0f;
    const float rcp   = 1.0f / (t*scale);
    const float sot   = (s*scale) * rcp;
    const float tn    = (ax == hz_abs(y)) ? 1.0f : hz_abs(sot);

    /* atan(tn), tn >= 0, through atan(min(tn,1)/max(tn,1)) */
    const float u  = hz_min(tn, 1.0f) / hz_max(tn, 1.0f);
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
    /* Mesa: a = b2f(tn > 1)*(p*-2 + pi/2) + p, then a *= sign(tn), then
     * arc = b2f(flip)*pi/2 + a.  With finite p >= 0 these multiplications by
     * 0.0 / 1.0 are exact selections (and tn = 0 implies p = 0), so the
     * selects below produce the same bits with fewer instructions. */
    const float a   = (1.0f < tn) ? ((p*-2.0f + HZ_HALF_PI) + p) : p;
    const float arc = flip ? (HZ_HALF_PI + a) : a;
    return (hz_min(y, rcp) < 0.f) ? -arc : arc;
}

/* per-draw constants, reference vertex.glsl:139-150 */
HZ_HD void hz_frame_from_az(float az_deg0, float az_deg1, float* az_center, float* az_ndc_per_rad)
{
    const float az0 = az_deg0 * HZ_DEG2RAD;
    float       az1 = az_deg1 * HZ_DEG2RAD;
    /* unwrap_near_rad(az1-az0, pi) (vertex.glsl:34-38,143).  round() is
     * round-half-even: an exactly-360-degree view spans 2*pi, not 0 */
    const float d    = ((az1 + -HZ_PI) + -az0) / HZ_TWO_PI;
    const float span = HZ_TWO_PI*(d - hz_roundeven(d)) + HZ_PI;
    az1 = span + az0;
    *az_center      = (az0 + az1) / 2.0f;
    *az_ndc_per_rad = 2.0f / span;
}

/* gl_Position.xyz and rgb.r of one vertex */
typedef struct { float x, y, z, red; } hz_vertex_t;

typedef struct
{
    float viewer_cell_i, viewer_cell_j, viewer_z, cos_viewer_lat, deg_per_cell;
    float aspect, znear, zfar, znear_color, zfar_color;
    float az_center, az_ndc_per_rad;
} hz_xform_t;

/* fi,fj: grid indices as float; fz: elevation as float (the reference feeds
 * them as GLshort attributes, reference horizonator-lib.c:424) */
/* east / north offset of a grid column / row, vertex.glsl:128-130.  They
 * depend on one index each: a kernel that walks the grid computes them once
 * per column / row. */
HZ_HD float hz_east(const hz_xform_t* u, float fi)
{
    return (fi - u->viewer_cell_i) * HZ_REARTH_PI * u->deg_per_cell / 180.0f * u->cos_viewer_lat;
}
HZ_HD float hz_north(const hz_xform_t* u, float fj)
{
    return (fj - u->viewer_cell_j) * HZ_REARTH_PI * u->deg_per_cell / 180.0f;
}

/* The transform in two halves.  What is expensive in it - two atan, two square roots (vertex.glsl:133-134, 154, 156) -
 * depends on where the viewer stands (e, n, h) and on nothing else: not on the azimuth extents, the aspect ratio or
 * the depth and colour extents.  hz_polar_en() is that half; hz_finish() turns its four numbers into gl_Position and
 * the colour for a given view.  A context may keep the four numbers of every vertex from one draw to the next
 * (hz_draw.cpp: the vertex cache - 16 bytes per vertex) and redo only the second half while the viewer stays put: the
 * same operations in the same order, hence the same bits. */
typedef struct { float az, d_ne, el, d_enh; } hz_polar_t;

HZ_HD hz_polar_t hz_polar_en(const hz_xform_t* u, float e, float n, float fz)
{
    hz_polar_t q;
    const float h = fz - u->viewer_z;                       /* vertex.glsl:131 */
    /* vertex.glsl:133-134 */
    const float nn = n*n, ee = e*e;
    q.d_ne  = hz_sqrt(nn + ee);
    q.az    = hz_atan2(e, n);
    q.el    = hz_atan2(h, q.d_ne);                          /* vertex.glsl:154 */
    q.d_enh = hz_sqrt(h*h + nn + ee);                       /* vertex.glsl:156 */
    return q;
}

HZ_HD hz_vertex_t hz_finish(const hz_xform_t* u, hz_polar_t q)
{
    hz_vertex_t v;
    /* vertex.glsl:148-156 */
    const float d = (q.az + -u->az_center) / HZ_TWO_PI;
    v.x = (HZ_TWO_PI*(d - hz_roundeven(d))) * u->az_ndc_per_rad;
    v.y = q.el * u->aspect * u->az_ndc_per_rad;
    v.z = (q.d_enh - u->znear) / (u->zfar - u->znear) * 2.0f + -1.0f;

    /* vertex.glsl:159-160 */
    const float r = (q.d_ne - u->znear_color) / (u->zfar_color - u->znear_color);
    v.red = hz_min(hz_max(r, 0.0f), 1.0f);
    return v;
}

HZ_HD hz_vertex_t hz_transform_en(const hz_xform_t* u, float e, float n, float fz)
{
    return hz_finish(u, hz_polar_en(u, e, n, fz));
}

/* fi,fj: grid indices as float; fz: elevation as float (the reference feeds
 * them as GLshort attributes, reference horizonator-lib.c:424) */
HZ_HD hz_vertex_t hz_transform(const hz_xform_t* u, float fi, float fj, float fz)
{
    return hz_transform_en(u, hz_east(u, fi), hz_north(u, fj), fz);
}
