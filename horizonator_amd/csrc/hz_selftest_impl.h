/* hz_selftest_impl.h - part of hz_kernels.hip, compiled only into libhorizonator_selftest.so (-DHZ_SELFTEST: the
 * library's sources plus what follows; the library that ships has none of it): the device-side self-checks of the
 * arithmetic shortcuts (hz_fast.h, hz_tri_hidden, the cull of whole cells, the smallest depth of a rectangle) and
 * the diagnostics entry points of tools/ (include/hz_selftest.h). */
#pragma once
/* ------------------------------------------------------------------------ */
/* self-check of hz_fast.h: the abridged sequences against `/` and sqrtf       */

__device__ static inline unsigned long long hz_mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* a float with seeded mantissa and sign and an exponent in [elo, ehi] (biased) */
__device__ static inline float hz_seeded_float(unsigned long long r, int elo, int ehi)
{
    const uint32_t mant = (uint32_t)r & 0x7FFFFFu, sign = (uint32_t)(r >> 23) & 1u;
    const uint32_t ex = (uint32_t)elo + (uint32_t)((r >> 24) % (unsigned long long)(ehi - elo + 1));
    return __uint_as_float((sign << 31) | (ex << 23) | mant);
}

__global__ __launch_bounds__(256)
void k_check_fastmath(int what, unsigned long long seed, unsigned long long n, unsigned long long* mismatches, float* first_bad)
{
    unsigned long long bad = 0;
    for(unsigned long long k = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; k < n; k += (unsigned long long)gridDim.x*blockDim.x)
    {
        float a = 0.f, b = 0.f, want = 0.f, got = 0.f;
        bool in_range = true;
        if(what == 0)                       /* reciprocal: every bit pattern (k = the pattern), those in range checked */
        {
            b = __uint_as_float((uint32_t)k);
            in_range = hz_abs(b) >= 2.16840434e-19f && hz_abs(b) <= 4.61168602e18f;        /* 2^-62 .. 2^62 */
            if(in_range) { want = 1.0f / b; got = hzf_rcp(b); }
        }
        else if(what == 1)                  /* square root: every bit pattern from 2^-96 up to the largest finite float */
        {
            b = __uint_as_float((uint32_t)k);
            in_range = b >= 1.26217745e-29f && b <= 3.40282347e38f;
            if(in_range) { want = __builtin_sqrtf(b); got = hzf_sqrt(b); }
        }
        else if(what == 2)                  /* division: seeded pairs, numerator 0 or 2^-60..2^60, denominator 2^-31..2^31 */
        {
            const unsigned long long r1 = hz_mix64(seed + 2*k), r2 = hz_mix64(seed + 2*k + 1);
            a = hz_seeded_float(r1, 127-60, 127+60);
            b = hz_seeded_float(r2, 127-31, 127+31);
            /* the shapes the transform divides: any pair, min/max of a pair and 1, a shared mantissa, zero */
            if((r2 >> 61) == 1) { const float t = hz_abs(a); a = hz_min(t, 1.0f); b = hz_max(t, 1.0f); }
            if((r2 >> 61) == 2) a = __uint_as_float((__float_as_uint(a) & 0xFF800000u) | (__float_as_uint(b) & 0x7FFFFFu));
            if((r1 >> 60) == 0) a = 0.0f;            /* +0 */
            want = a / b; got = hzf_div(a, b);
        }
        else if(what == 4)                  /* hz_rcp_f64: every divisor 1 <= d < 2^31 (k = d - 1): within 2^-50 of 1/d? */
        {
            const double dd = (double)(k + 1);
            const double r = hz_rcp_f64(dd);
            /* d*r - 1 is the relative error of r; the fma gives it without cancellation */
            const double rel = __builtin_fabs(__builtin_fma(dd, r, -1.0));
            a = (float)dd; b = (float)rel; want = 0.f; got = rel < 8.8817841970012523e-16 ? 0.f : 1.f;      /* 2^-50 */
        }
        else if(what == 5)                  /* hz_floor_div against 64-bit integer division: seeded n (|n| < 2^55), d (1 <= d < 2^31) */
        {
            const unsigned long long r1 = hz_mix64(seed + 2*k), r2 = hz_mix64(seed + 2*k + 1);
            /* divisors of every magnitude; numerators of every magnitude and both signs, and the
             * hard ones: multiples of d and their neighbours */
            const int32_t d = (int32_t)(((r2 >> 8) & 0x7FFFFFFFull) >> (r2 & 31)) | 1;
            int64_t n = (int64_t)(r1 >> 9) >> ((r1 >> 3) & 63);
            if(r1 & 1) n = -n;
            if((r1 & 6) == 2) n = (n / d)*(int64_t)d + (int64_t)((r2 >> 40) % 3) - 1;
            int64_t q = n / d;                                  /* truncates */
            if((n % d) != 0 && n < 0) q--;                      /* floor */
            const int32_t f = hz_floor_div(n, d, hz_rcp_f64((double)d));
            a = (float)n; b = (float)d;
            /* beyond +-2^30 any value beyond is right (see hz_floor_div) */
            const bool ok = (q > 1073741824ll) ? f >= 1073741824 : (q < -1073741824ll) ? f <= -1073741824 : (int64_t)f == q;
            want = 0.f; got = ok ? 0.f : 1.f;
        }
        else if(what == 6)                  /* the quotient by 2 pi in one correction step: k = numerator pattern, zero or 2^-100 .. 2^30 */
        {
            float two_pi = HZ_TWO_PI;
            asm volatile("" : "+v"(two_pi));
            b = two_pi;
            a = __uint_as_float((uint32_t)k);
            const float aa = hz_abs(a);
            in_range = __float_as_uint(a) == 0u || (aa >= 7.88860905e-31f && aa <= 1073741824.0f);
            if(in_range) { want = a / b; got = hzf_div_by_two_pi(a, hzf_refined_rcp(two_pi)); }
        }
        else if(what == 7)                  /* min(t,1)/max(t,1) as t or 1/t: k = the pattern of t, zero or 2^-62 .. 2^62 */
        {
            a = __uint_as_float((uint32_t)k);
            in_range = __float_as_uint(a) == 0u || (a >= 2.16840434e-19f && a <= 4.61168602e18f);
            if(in_range) { want = hz_min(a, 1.0f) / hz_max(a, 1.0f); got = hzf_fold_to_unit(a); }
        }
        else if(what == 8 || what == 9)     /* the abridged arc tangent against hz_atan2: seeded (y, x), y zero or 2^-30 .. 2^30 of either
                                             * sign, x likewise but never zero; 9: x > 0 (the elevation angle's: a distance, up to 2^30.5) */
        {
            const unsigned long long r1 = hz_mix64(seed + 2*k), r2 = hz_mix64(seed + 2*k + 1);
            a = hz_seeded_float(r1, 127-30, 127+29);
            b = hz_seeded_float(r2, 127-30, 127+29);
            if((r1 >> 60) == 0 && what == 9) a = 0.0f;     /* (the azimuth's y is an east offset in range: never zero) */
            if((r1 >> 60) == 1) a = __uint_as_float((__float_as_uint(a) & 0x80000000u) | (__float_as_uint(b) & 0x7FFFFFFFu));    /* |y| = |x| */
            if((r1 >> 60) == 2) a = __uint_as_float(__float_as_uint(a) & 0xFFFFFF00u);                                          /* short mantissas */
            if((r2 >> 60) == 2) b = __uint_as_float(__float_as_uint(b) & 0xFFFFFF00u);
            if(what == 9) b = hz_abs(b) * (((r2 >> 59) & 1) ? 1.41421354f : 1.0f);
            want = hz_atan2(a, b);
            got  = what == 9 ? hzf_atan2<true>(a, b) : hzf_atan2<false>(a, b);
        }
        else                                /* division by a per-draw constant through hzf_div_by: k = numerator pattern */
        {
            b = __uint_as_float((uint32_t)seed);
            a = __uint_as_float((uint32_t)k);
            const float aa = hz_abs(a);
            /* (+0 only: a negative zero would come out positive - see hz_fast.h on why none gets here) */
            in_range = __float_as_uint(a) == 0u || (aa >= 7.88860905e-31f && aa <= 1.15292150e18f);     /* 2^-100 .. 2^60 */
            if(in_range) { want = a / b; got = hzf_div_by(a, b, hzf_refined_rcp(b)); }
        }
        if(in_range && __float_as_uint(want) != __float_as_uint(got))
        {
            if(bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = a; first_bad[1] = b; first_bad[2] = want; first_bad[3] = got; }
            bad++;
        }
    }
    if(bad) atomicAdd(mismatches, bad);
}

extern "C" int hz_hip_check_fastmath(int device, int what, unsigned long long seed, unsigned long long n,
                                     unsigned long long* mismatches, float* first_bad)
{
    hz_device_guard device_guard_(device);
    if(!device_guard_.ok) return -1;
    if(what < 0 || what > 9) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_check_fastmath: what = %d", what); return -1; }
    if(what == 0 || what == 1 || what == 3 || what == 6 || what == 7) n = 1ull << 32;
    if(what == 4) n = (1ull << 31) - 1;
    unsigned long long* d_bad = NULL; float* d_first = NULL;
    HZ_CHECK(hipMalloc(&d_bad, 2*sizeof(unsigned long long)));
    HZ_CHECK(hipMalloc(&d_first, 4*sizeof(float)));
    HZ_CHECK(hipMemset(d_bad, 0, 2*sizeof(unsigned long long)));
    HZ_CHECK(hipMemset(d_first, 0, 4*sizeof(float)));
    hipLaunchKernelGGL(k_check_fastmath, dim3(256*32), dim3(256), 0, 0, what, seed, n, d_bad, d_first);
    HZ_CHECK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HZ_CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
    if(first_bad) HZ_CHECK(hipMemcpy(first_bad, d_first, 4*sizeof(float), hipMemcpyDeviceToHost));
    *mismatches = h[0];
    (void)hipFree(d_bad); (void)hipFree(d_first);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* self-checks of the two shortcuts the marching kernel takes on the strength of  */
/* an argument rather than of the reference's arithmetic (hz_tri_hidden, the      */
/* cull of whole cells): on seeded inputs around every border of the argument     */

/* k_check_hidden: one triangle per thread.  Vertices as the rasteriser has them
 * (window position, depth, snapped position); families: 0 far-field (a pixel or two
 * across), 1 slivers (third vertex a hair off the line through the other two:
 * |area| down to 2^-12 px^2, the depth plane extrapolated over the snapping
 * distance is what hz_tri_hidden's slack is for), 2 grazing (depth gradients up to
 * 10^5 LSB per pixel), 3 anything up to 16 pixels across.  For each triangle that
 * survives the cull the LARGEST stored depth zs for which hz_tri_hidden() still
 * answers "hidden" is found by bisection (the answer is monotone in zs) and every
 * pixel centre the triangle covers (hz_tri_covers) is drawn (hz_tri_planes,
 * hz_tri_fragment): no fragment may pass GL_LESS against zs, i.e. have a depth
 * <= zs.  out[0] triangles tested, [1] of them hidden for some zs, [2] fragments
 * drawn, [3] violations, [4] the smallest (fragment depth - zs) seen, + 2^32. */
__global__ __launch_bounds__(256)
void k_check_hidden(unsigned long long seed, unsigned long long n, int W, int H, unsigned long long* out)
{
    const float z_guard = 1.0f/500.0f + (float)(W > H ? W : H) * (1.0f/4194304.0f);
    const float kk = 1.03f * z_guard * 16777215.f;
    unsigned long long tested = 0, hidden = 0, frags = 0, bad = 0, margin = ~0ull;
    for(unsigned long long t = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; t < n; t += (unsigned long long)gridDim.x*blockDim.x)
    {
        const unsigned long long r0 = hz_mix64(seed + 4*t), r1 = hz_mix64(seed + 4*t + 1), r2 = hz_mix64(seed + 4*t + 2), r3 = hz_mix64(seed + 4*t + 3);
        auto unit = [](unsigned long long r, int shift) { return (float)((r >> shift) & 0xFFFFFFull) * (1.0f/16777216.0f); };   /* [0,1) */
        const int family = (int)(r0 & 3);
        /* where: anywhere in the image, often next to its right / top border (large coordinates: coarse float spacing) */
        float bx = unit(r0, 8) * (float)W, by = unit(r0, 32) * (float)H;
        if(((r0 >> 2) & 7) == 0) bx = (float)W - 4.0f*unit(r1, 0);
        if(((r0 >> 5) & 7) == 0) by = (float)H - 4.0f*unit(r1, 24);
        float x[3], y[3], z[3];
        const float ext = family == 3 ? 16.0f*unit(r3, 40) : family == 0 ? 2.5f : 4.0f;
        x[0] = bx; y[0] = by;
        x[1] = bx + (unit(r1, 8) - 0.5f)*ext;  y[1] = by + (unit(r1, 36) - 0.5f)*ext;
        x[2] = bx + (unit(r2, 0) - 0.5f)*ext;  y[2] = by + (unit(r2, 24) - 0.5f)*ext;
        if(family == 1)
        {
            /* the third vertex on the line through the first two, then off it by 2^-1 .. 2^-14 pixels */
            const float tt = unit(r2, 0)*1.5f - 0.25f, off = __builtin_ldexpf(1.0f, -1 - (int)((r2 >> 40) % 14)) * ((r2 >> 63) ? 1.0f : -1.0f);
            const float dx = x[1] - x[0], dy = y[1] - y[0], len = hz_sqrt(dx*dx + dy*dy) + 1e-6f;
            x[2] = x[0] + tt*dx - off*dy/len; y[2] = y[0] + tt*dy + off*dx/len;
        }
        /* depth: a base anywhere in (0,1), differences from 10^-8 (a few LSB) to 10^-2 (10^5 LSB) */
        const float zb = 0.02f + 0.96f*unit(r3, 0);
        const float dz = (family == 2 ? 6e-3f : 1e-3f) * __builtin_ldexpf(1.0f, -(int)((r3 >> 24) % 18));
        z[0] = zb; z[1] = zb + (unit(r3, 30) - 0.5f)*dz; z[2] = zb + (unit(r2, 40) - 0.5f)*dz;
        hz_wvert_t v[3];
        bool inside = true;
        for(int k=0; k<3; k++)
        {
            /* as mr_window(): the survivors of the cull are inside the view volume */
            v[k].xn = 0.f; v[k].wx = x[k]; v[k].wy = y[k]; v[k].zw = z[k]; v[k].red = 0.25f*(float)k; v[k].cmask = 0;
            inside = inside && x[k] >= 0.f && x[k] <= (float)W && y[k] >= 0.f && y[k] <= (float)H && z[k] >= 0.f && z[k] <= 1.f;
            v[k].xs = (int32_t)hz_roundeven((x[k] - 0.5f)*256.f);
            v[k].ys = (int32_t)hz_roundeven((y[k] - 0.5f)*256.f);
        }
        if(!inside) continue;
        hz_box_t box;
        /* either winding: the back face of one is the front face of the other */
        if(!hz_tri_cull_window(&box, &v[0], &v[1], &v[2], 0, W-1, 0, H-1))
        {
            const hz_wvert_t tmp = v[1]; v[1] = v[2]; v[2] = tmp;
            if(!hz_tri_cull_window(&box, &v[0], &v[1], &v[2], 0, W-1, 0, H-1)) continue;
        }
        tested++;
        if(!hz_tri_hidden(&v[0], &v[1], &v[2], kk, 0u)) continue;              /* not even behind depth 0 */
        uint32_t lo = 0, hi = HZ_Z24_MAX;                                       /* hidden at lo, not (or untested) at hi */
        if(hz_tri_hidden(&v[0], &v[1], &v[2], kk, hi)) lo = hi;
        while(hi - lo > 1u)
        {
            const uint32_t mid = lo + (hi - lo)/2u;
            if(hz_tri_hidden(&v[0], &v[1], &v[2], kk, mid)) lo = mid; else hi = mid;
        }
        hidden++;
        hz_tri_t tri;
        hz_tri_planes(&tri, &v[0], &v[1], &v[2]);
        for(int py = box.py0; py <= box.py1; py++)
            for(int px = box.px0; px <= box.px1; px++)
            {
                if(!hz_tri_covers(&tri, px, py)) continue;
                uint32_t zi, r8;
                if(!hz_tri_fragment(&tri, px, py, &zi, &r8)) continue;          /* at or beyond the cleared depth: never drawn */
                frags++;
                if(zi <= lo) bad++;
                const unsigned long long m = (unsigned long long)((long long)zi - (long long)lo + (1ll << 32));
                if(m < margin) margin = m;
            }
    }
    atomicAdd(&out[0], tested); atomicAdd(&out[1], hidden); atomicAdd(&out[2], frags); atomicAdd(&out[3], bad);
    atomicMin(&out[4], margin);
}

/* k_check_cull: one wave per case = two vertex rows of 64 as k_march sees them, NDC
 * positions seeded around the borders of mr_simple_cull()'s argument: cells about
 * a sixteenth of the image wide (quad_max_dx), rows that touch the edges of the view
 * volume (|x|,|y|,|z| = 1 and a float beyond), positions that snap onto pixel
 * centres, the +-180 degree seam (x jumping from one image border to the other),
 * back faces and empty boxes of every kind.  Wherever the wave would take the
 * short way, its verdict must be hz_tri_cull()'s.  out[0] cases, [1] cases that
 * took the short way, [2] cells compared, [3] disagreements, [4] triangles kept. */
__global__ __launch_bounds__(64)
void k_check_cull(unsigned long long seed, unsigned long long ncases, int W, int H, int col0, int col1, unsigned long long* out)
{
    hz_params_t p;
    memset(&p, 0, sizeof(p));
    p.halfW = (float)W*0.5f; p.halfH = (float)H*0.5f; p.W = W; p.H = H; p.col0 = col0; p.col1 = col1; p.SW = col1 - col0;
    p.quad_max_dx = W >= 64 && W <= 65535 && H <= 65535 ? 256*(W/16 - 1) : 0;
    const int lane = threadIdx.x;
    const bool has_cell = lane < MR_COLS;
    unsigned long long cases = 0, shortway = 0, cells = 0, bad = 0, kept = 0;
    for(unsigned long long t = blockIdx.x; t < ncases; t += gridDim.x)
    {
        const unsigned long long rc = hz_mix64(seed + 977*t);                   /* per case (wave-uniform) */
        auto unit = [](unsigned long long r, int shift) { return (float)((r >> shift) & 0xFFFFFFull) * (1.0f/16777216.0f); };
        const int kind = (int)(rc & 7);
        /* the step between neighbouring vertices, in NDC x: a fraction of a pixel to 8 pixels; kinds 0-2: ONE cell of the
         * row about a sixteenth of the image wide (the threshold quad_max_dx, +-3 %), at a seeded lane */
        float step = 2.0f/(float)W * (0.1f + 8.0f*unit(rc, 8));
        if(kind == 5) step = -step;                                             /* back faces */
        const float jump = kind <= 2 ? 2.0f/16.0f * (0.97f + 0.06f*unit(rc, 44)) : 0.0f;
        const int   jump_lane = (int)((rc >> 3) & 63);
        const float x_first = kind == 1 ? -1.0f : kind == 2 ? 1.0f - 63.0f*step - jump : -1.0f + (2.0f - 64.0f*hz_abs(step) - jump)*unit(rc, 32);
        const float y_base  = ((rc >> 56) & 3) == 0 ? 1.0f - 4.0f/(float)H*unit(rc, 40) : -1.0f + 2.0f*unit(rc, 40);
        hz_vertex_t vt[2];
        for(int row=0; row<2; row++)
        {
            const unsigned long long r = hz_mix64(seed + 977*t + 131*(unsigned long long)(row + 1) + 7*(unsigned long long)lane);
            float xn = x_first + (float)lane*step + (lane > jump_lane ? jump : 0.0f) + (unit(r, 0) - 0.5f)*hz_abs(step)*0.6f;
            float yn = y_base + (row ? 1 : 0)*(2.0f/(float)H)*(0.2f + 4.0f*unit(rc, 16)) + (unit(r, 24) - 0.5f)*(2.0f/(float)H);
            float zn = -1.0f + 2.0f*unit(r, 40);
            if(kind == 4 && ((r >> 60) & 3) == 0)                               /* onto a pixel centre / a pixel border, exactly */
                xn = ((float)(int)(unit(r, 8)*(float)W) + (((r >> 59) & 1) ? 0.5f : 0.0f))/p.halfW - 1.0f;
            if(kind == 6 && lane >= 32) xn -= 1.9f*unit(rc, 20);                /* the seam: the row jumps back towards the other border */
            if(kind == 7 && ((r >> 58) & 15) == 0)                              /* a vertex on / just beyond a face of the view volume */
            {
                const float edge[4] = { 1.0f, -1.0f, 1.00000012f, -1.00000012f };
                if((r >> 62) & 1) xn = edge[(r >> 56) & 3]; else if((r >> 63) & 1) yn = edge[(r >> 56) & 3]; else zn = edge[(r >> 56) & 3];
            }
            vt[row].x = xn; vt[row].y = yn; vt[row].z = zn; vt[row].red = 0.5f;
        }
        mr_rowstate_t st[2];
        hz_wvert_t wv[2];
        bool simple[2];
        for(int row=0; row<2; row++)
        {
            bool in_volume, in_guard;
            wv[row] = mr_window(vt[row], p, &in_volume, &in_guard);
            simple[row] = __all(in_guard && in_volume);
            if(!simple[row]) mr_window_flags(wv[row], vt[row], in_guard);
            st[row] = mr_rowstate_of(wv[row], p);
        }
        cases++;
        bool keep0 = false, keep1 = false;
        if(!(simple[0] && simple[1] && mr_simple_cull(st[0], st[1], has_cell, p, &keep0, &keep1))) continue;
        shortway++;
        /* the long way, as k_march takes it */
        hz_wvert_t v00 = {}, v01 = {}, v10 = {}, v11 = {};
        v00.xn = st[0].xn; v00.xs = st[0].xs; v00.ys = st[0].ys; v00.cmask = st[0].cmask;
        v01.xn = st[1].xn; v01.xs = st[1].xs; v01.ys = st[1].ys; v01.cmask = st[1].cmask;
        v10.xn = mr_from_east(st[0].xn); v10.xs = mr_from_east(st[0].xs); v10.ys = mr_from_east(st[0].ys); v10.cmask = (uint32_t)mr_from_east((int32_t)st[0].cmask);
        v11.xn = mr_from_east(st[1].xn); v11.xs = mr_from_east(st[1].xs); v11.ys = mr_from_east(st[1].ys); v11.cmask = (uint32_t)mr_from_east((int32_t)st[1].cmask);
        if(has_cell)
        {
            hz_box_t box;
            const bool want0 = hz_tri_cull(&box, &v00, &v11, &v01, p.col0, p.col1-1, 0, p.H-1) == HZ_TRI_DRAW;
            const bool want1 = hz_tri_cull(&box, &v00, &v10, &v11, p.col0, p.col1-1, 0, p.H-1) == HZ_TRI_DRAW;
            cells++;
            if(want0 != keep0) bad++;
            if(want1 != keep1) bad++;
            kept += (keep0 ? 1 : 0) + (keep1 ? 1 : 0);
        }
    }
    atomicAdd(&out[0], lane == 0 ? cases : 0ull); atomicAdd(&out[1], lane == 0 ? shortway : 0ull);
    atomicAdd(&out[2], cells); atomicAdd(&out[3], bad); atomicAdd(&out[4], kept);
}

/* k_check_rect: hiz_rect_min_depth() (hz_k_hiz.h: the smallest depth a rectangle of pixel centres can get is the
 * smallest of its corners') against the minimum over every pixel centre of the rectangle, one thread per case: depth
 * planes from flat to 10^6 per pixel, signs of every kind, origins inside and outside [0, 1] (clamped depths), values
 * that overflow, infinities and NaNs; rectangles of up to 48 x 48 anywhere in a W x H image.  out[0] cases, [1] cases
 * whose corners are all numbers, [2] pixel centres evaluated, [3] disagreements (a different minimum, or a pixel
 * without a depth inside a rectangle whose corners have one), [4] cases with a depth that is not a number. */
__global__ __launch_bounds__(256)
void k_check_rect(unsigned long long seed, unsigned long long ncases, int W, int H, unsigned long long* out)
{
    unsigned long long cases = 0, numbers = 0, pixels = 0, bad = 0, nans = 0;
    for(unsigned long long t = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; t < ncases; t += (unsigned long long)gridDim.x*blockDim.x)
    {
        const unsigned long long r0 = hz_mix64(seed + 3*t), r1 = hz_mix64(seed + 3*t + 1), r2 = hz_mix64(seed + 3*t + 2);
        auto unit = [](unsigned long long r, int shift) { return (float)((r >> shift) & 0xFFFFFFull) * (1.0f/16777216.0f); };
        auto slope = [&](unsigned long long r) -> float
        {
            const int kind = (int)(r & 15);
            if(kind == 0) return 0.0f;
            if(kind == 1) return (r >> 4) & 1 ? __builtin_inff() : -__builtin_inff();
            if(kind == 2) return __builtin_nanf("");
            if(kind == 3) return ((r >> 4) & 1 ? 1.0f : -1.0f) * 3.0e38f * unit(r, 8);             /* overflows once multiplied */
            const float mag = exp2f(-40.0f + 60.0f*unit(r, 8));                                   /* 1e-12 .. 1e6 per pixel */
            return ((r >> 4) & 1) ? mag : -mag;
        };
        hz_tri_t tri;
        memset(&tri, 0, sizeof(tri));
        tri.dzdx = slope(r0); tri.dzdy = slope(r1);
        tri.z_org = ((r2 & 7) == 0) ? -3.0f + 7.0f*unit(r2, 8) : unit(r2, 8);
        if((r2 & 63) == 1) tri.z_org = __builtin_inff();
        /* (steep planes: an origin that puts the rectangle's depths near [0, 1] rather than far outside) */
        const int bw = 1 + (int)((r2 >> 32) % 48u), bh = 1 + (int)((r2 >> 40) % 48u);
        const int x0 = (int)((r0 >> 32) % (unsigned long long)(W - bw + 1)), y0 = (int)((r1 >> 32) % (unsigned long long)(H - bh + 1));
        if((r2 >> 48) & 1) tri.z_org = 0.5f - tri.dzdx*(float)(x0 + bw/2) - tri.dzdy*(float)(y0 + bh/2);
        uint32_t qmin = 0;
        const bool ok = hiz_rect_min_depth(tri, x0, x0 + bw - 1, y0, y0 + bh - 1, &qmin);
        uint32_t m = 0xFFFFFFFFu;
        bool nan_inside = false;
        for(int py = y0; py < y0 + bh; py++)
            for(int px = x0; px < x0 + bw; px++)
            {
                /* hz_tri_fragment()'s depth, without its two early returns */
                float z = __builtin_fmaf(tri.dzdy, (float)py, __builtin_fmaf(tri.dzdx, (float)px, tri.z_org));
                if(!(z == z)) { nan_inside = true; continue; }
                z = hz_min(hz_max(z, 0.f), 1.f);
                const uint32_t q = (uint32_t)hz_roundeven(z * 16777215.f);
                uint32_t zi, r8;
                const int drawn = hz_tri_fragment(&tri, px, py, &zi, &r8);
                if(drawn && zi != q) bad++;                                    /* (this loop is what hz_tri_fragment computes) */
                m = m < q ? m : q;
            }
        cases++; pixels += (unsigned long long)bw*bh;
        if(nan_inside) nans++;
        if(ok) { numbers++; if(nan_inside || m != qmin) bad++; }
    }
    atomicAdd(&out[0], cases); atomicAdd(&out[1], numbers); atomicAdd(&out[2], pixels); atomicAdd(&out[3], bad); atomicAdd(&out[4], nans);
}

/* what: 0 = hz_tri_hidden (n triangles), 1 = the cull of whole cells (n cases of two rows of 64 vertices), 2 = the
 * smallest depth of a rectangle (n rectangles); image W x H (and, for 1, the drawn columns [col0,col1)); out: 5 words,
 * see the kernels */
extern "C" int hz_hip_check_exactness(int device, int what, unsigned long long seed, unsigned long long n,
                                      int W, int H, int col0, int col1, unsigned long long* out)
{
    hz_device_guard device_guard_(device);
    if(!device_guard_.ok) return -1;
    if(what < 0 || what > 2 || W < 1 || H < 1 || col0 < 0 || col1 > W || col0 >= col1 || (what == 2 && (W < 48 || H < 48)))
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_check_exactness: bad arguments");
        return -1;
    }
    unsigned long long* d_out = NULL;
    HZ_CHECK(hipMalloc(&d_out, 5*sizeof(unsigned long long)));
    const unsigned long long init[5] = { 0, 0, 0, 0, what == 0 ? ~0ull : 0ull };
    HZ_CHECK(hipMemcpy(d_out, init, sizeof(init), hipMemcpyHostToDevice));
    if(what == 0)      hipLaunchKernelGGL(k_check_hidden, dim3(256*32), dim3(256), 0, 0, seed, n, W, H, d_out);
    else if(what == 1) hipLaunchKernelGGL(k_check_cull, dim3(256*64), dim3(64), 0, 0, seed, n, W, H, col0, col1, d_out);
    else               hipLaunchKernelGGL(k_check_rect, dim3(256*16), dim3(256), 0, 0, seed, n, W, H, d_out);
    HZ_CHECK(hipGetLastError());
    HZ_CHECK(hipMemcpy(out, d_out, 5*sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_out);
    return 0;
}

/* diagnostics: the large-triangle queue of the last draw (set 0: its only or
 * second round, set 1: the first round of a two-round draw): counters[6] and,
 * for the first min(max_rec, counters[0]) records, px0 py0 bw bh + the three
 * edge vectors dx, dy in 1/256 pixel (10 int32 each) */
extern "C" int hz_hip_debug_bigqueue(hz_dev_t* d, int set, unsigned int* counters, int max_rec, int32_t* recs)
{
    HZ_ON_DEVICE(d);
    if(hz_hip_sync(d) != 0) return -1;
    const int k = (set ? HZ_NFB : 0) + d->fbi;
    /* (a conversion that cleared the framebuffer emptied the queues too and left a copy of the counters - of the big
     * triangles' shards their sums: the records themselves can then no longer be told from the slots in between) */
    std::vector<unsigned int> all(HZ_NCOUNTERS);
    HZ_CHECK(hipMemcpy(all.data(), d->d_big_counters_s[k], HZ_NCOUNTERS*sizeof(unsigned int), hipMemcpyDeviceToHost));
    unsigned int per_shard[HZ_QSHARDS], longest = 0;
    const int sl = d->last_qshards_log2;          /* (of the last draw: hz_draw_impl) */
    if(d->fb_consumed) { for(int c=0; c<6; c++) counters[c] = all[HZ_CNT_LAST + c]; for(int s=0; s<HZ_QSHARDS; s++) per_shard[s] = 0; }
    else
    {
        counters[0] = counters[1] = 0; counters[2] = 0; counters[3] = 0; counters[5] = 0;
        for(int s=0; s<HZ_QSHARDS; s++)
        {
            const unsigned int* c = all.data() + HZ_QSHARD0 + s*HZ_QSHARD_STRIDE;
            per_shard[s] = c[0] < HZ_QSHARD_ROOM(d->bigrec_capacity, sl) ? c[0] : HZ_QSHARD_ROOM(d->bigrec_capacity, sl);
            counters[0] += c[0]; counters[1] += c[1]; if(c[2]) counters[2] = c[2];
            counters[3] += c[4]; if(c[5]) counters[5] = c[5];
            if(per_shard[s] > longest) longest = per_shard[s];
        }
        counters[4] = all[4];
    }
    counters[2] = ~counters[2]; counters[5] = ~counters[5];     /* as documented: the first invalid index (of one of the shards) */
    if(longest == 0 || recs == NULL || max_rec <= 0) return 0;
    /* which record of which shard a slot holds: HZ_QSLOT (hz_types.h) taken apart */
    const size_t slots = (size_t)((longest + HZ_QBLOCK-1)/HZ_QBLOCK*HZ_QBLOCK) << sl;
    hz_bigrec_t* h = (hz_bigrec_t*)malloc(slots*sizeof(hz_bigrec_t));
    if(!h) return -1;
    HZ_CHECK(hipMemcpy(h, d->d_bigrec_s[k], slots*sizeof(hz_bigrec_t), hipMemcpyDeviceToHost));
    int n = 0;
    for(size_t g=0; g<slots && n<max_rec; g++)
    {
        const size_t block = g >> HZ_QBLOCK_LOG2;
        if((((block >> sl) << HZ_QBLOCK_LOG2) | (g & (HZ_QBLOCK-1))) >= per_shard[block & (((size_t)1 << sl) - 1)]) continue;
        int32_t* o = recs + (size_t)n*10;
        o[0] = h[g].r.px0; o[1] = h[g].r.py0; o[2] = h[g].r.bw; o[3] = h[g].bh;
        for(int m=0; m<3; m++) { o[4+m] = h[g].r.e.dx[m]; o[7+m] = -h[g].r.e.ndy[m]; }
        n++;
    }
    free(h);
    return 0;
}

/* diagnostics / tests, no device needed: the work list draw_impl would build for a context of
 * N samples per axis and a W x H image drawing columns [col0,col1) of `view`.  round: 0 the
 * only round of a one-round draw, 1 / 2 the rounds of a two-round draw.  out: 3 int32 per
 * item (strip column, first cell row, cell row behind the last); returns the number of items
 * (also when capacity_items is smaller: then only that many were written), -1 if such a
 * draw would launch the whole grid instead (full circle). */
extern "C" long hz_hip_debug_worklist(int N, int W, int H, const hz_view_t* view, int col0, int col1, int round,
                                      int32_t* out, size_t capacity_items)
{
    hz_dev_t* d = (hz_dev_t*)calloc(1, sizeof(*d));
    if(!d) return -1;
    d->env = hz_options_from_env();
    d->N = N; d->W = W; d->H = H; d->col0 = col0; d->col1 = col1;
    d->seg_stride = (W + HZ_SEG-1)/HZ_SEG;
    hz_params_t p = hz_make_params(d, view);
    (void)hz_plan_rounds(d, view, p);
    const mr_zones_t zn = hz_make_zones(p, (round & 3) != 0);
    free(d);
    p.pass = round & 3;
    double a0 = 0, a1 = 0;
    const bool every_strip = (round & 256) != 0;
    if(!every_strip && !hz_azimuths_of_columns(p, &a0, &a1)) return -1;
    std::vector<uint32_t> items;
    hz_list_items(p, zn, a0, a1, items, every_strip);
    for(size_t k=0; k<items.size() && k<capacity_items; k++)
    {
        int jbeg, jend;
        mr_segment_rows(zn, (int)(items[k] >> MR_ITEM_SX_BITS), &jbeg, &jend);
        out[3*k] = (int32_t)(items[k] & ((1u << MR_ITEM_SX_BITS) - 1u)); out[3*k+1] = jbeg; out[3*k+2] = jend;
    }
    return (long)items.size();
}

/* diagnostics (tools/wave_timing.py): draws `view` once more with the instance of k_march
 * that counts, and returns, per wave of its second (or only) round's launch, 4 words:
 * duration in shader clock cycles, flushes<<32 | triangles set up, to k_big<<32 | to k_mid,
 * hidden by the early depth test<<32 | pixel centres tested in the wave.  out: room for
 * capacity_words; grid[2] = the launch grid (a work-list launch is grid[0] x 1). */
extern "C" int hz_hip_debug_wave_timing(hz_dev_t* d, const hz_view_t* view, unsigned long long* out, size_t capacity_words, unsigned int* grid)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hz_sync_all(d));
    unsigned long long* d_cycles = NULL;
    HZ_CHECK(hipMalloc(&d_cycles, capacity_words*sizeof(unsigned long long)));
    HZ_CHECK(hipMemset(d_cycles, 0, capacity_words*sizeof(unsigned long long)));
    d->wave_timing.d_cycles = d_cycles; d->wave_timing.capacity = capacity_words;
    d->wave_timing.grid_x = d->wave_timing.grid_y = 0;
    int rc = hz_draw_impl(d, view);
    d->wave_timing.d_cycles = NULL; d->wave_timing.capacity = 0;
    if(rc == 0 && hz_sync_all(d) != hipSuccess) rc = -1;
    grid[0] = d->wave_timing.grid_x; grid[1] = d->wave_timing.grid_y;
    if(rc == 0 && hipMemcpy(out, d_cycles, (size_t)grid[0]*grid[1]*4*sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    (void)hipFree(d_cycles);
    return rc;
}

