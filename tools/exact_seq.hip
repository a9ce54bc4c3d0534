/* exact_seq.hip - which SHORTER instruction sequences give the correctly rounded reciprocal, square root and quotient
 * by a constant on this GPU for EVERY operand hz_fast.h admits?  (round 5: k_march is bound by vector issue, and of the
 * 141 instructions of its transform 40 are the refinement steps of four reciprocals and two square roots.)
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math tools/exact_seq.hip -o exact_seq && ./exact_seq
 *
 * The reference is the device's own IEEE `/` and sqrtf (correctly rounded: -fno-fast-math); every float32 bit pattern
 * is tried, so a count of zero below is a proof for this hardware's v_rcp_f32 / v_rsq_f32 / v_sqrt_f32. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while(0)

/* operands: hz_fast.h admits e, n, h in [2^-30, 2^30]; a tangent of two of them and its reciprocal lie in [2^-61, 2^61],
 * an angle may be as small as 2^-61 */
#define LO 2.168404345e-19f     /* 2^-62 */
#define HI 4.611686018e18f      /* 2^62 */

#define NCAND 8
struct res_t { unsigned long long tried, bad[NCAND]; unsigned int first[NCAND]; };

__device__ static inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* ---- reciprocal ---- */
__device__ static inline float rcp_A(float x) { float r = __builtin_amdgcn_rcpf(x); float e = fma_(-x, r, 1.0f); return fma_(e, r, r); }
__device__ static inline float rcp_B(float x) { float r = rcp_A(x); float e = fma_(-x, r, 1.0f); return fma_(e, r, r); }
__device__ static inline float rcp_C(float x)       /* hz_fast.h's hzf_rcp as of round 4 */
{
    const float rr = rcp_A(x);
    float q = 1.0f*rr;
    float e = fma_(-x, q, 1.0f);
    q = fma_(e, rr, q);
    e = fma_(-x, q, 1.0f);
    return fma_(e, rr, q);
}
__device__ static inline float rcp_D(float x)       /* one step of the long form: residual against the refined value, corrected with the raw one */
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    float e = fma_(-x, r0, 1.0f);
    const float r1 = fma_(e, r0, r0);
    e = fma_(-x, r1, 1.0f);
    return fma_(e, r0, r1);
}

__global__ void k_rcp(res_t* out)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x;
    const unsigned long long step = (unsigned long long)gridDim.x*blockDim.x;
    unsigned long long tried = 0, bad[NCAND] = {0};
    for(unsigned long long b = gid; b < (1ull << 32); b += step)
    {
        const float x = __uint_as_float((unsigned int)b);
        const float a = fabsf(x);
        if(!(a >= LO && a <= HI)) continue;
        tried++;
        const float want = 1.0f / x;
        const float got[5] = { __builtin_amdgcn_rcpf(x), rcp_A(x), rcp_B(x), rcp_C(x), rcp_D(x) };
        for(int k=0; k<5; k++)
            if(__float_as_uint(got[k]) != __float_as_uint(want)) { if(!bad[k]) atomicCAS(&out->first[k], 0u, (unsigned int)b); bad[k]++; }
    }
    atomicAdd(&out->tried, tried);
    for(int k=0; k<NCAND; k++) if(bad[k]) atomicAdd(&out->bad[k], bad[k]);
}

/* ---- square root ---- */
__device__ static inline float sqrt_cur(float x)    /* hz_fast.h's hzf_sqrt as of round 4 */
{
    const float s    = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1);
    const float r_dn = fma_(-s_dn, s, x);
    const float r_up = fma_(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}
__device__ static inline float sqrt_S1(float x) { const float s = __builtin_amdgcn_sqrtf(x); const float h = 0.5f*__builtin_amdgcn_rsqf(x); const float r = fma_(-s, s, x); return fma_(r, h, s); }
__device__ static inline float sqrt_S2(float x) { const float y = __builtin_amdgcn_rsqf(x); const float s = x*y; const float h = 0.5f*y; const float r = fma_(-s, s, x); return fma_(r, h, s); }
__device__ static inline float sqrt_S3(float x) { const float y = __builtin_amdgcn_rsqf(x); float s = x*y; const float h = 0.5f*y; float r = fma_(-s, s, x); s = fma_(r, h, s); r = fma_(-s, s, x); return fma_(r, h, s); }
__device__ static inline float sqrt_S4(float x)     /* Goldschmidt: h refined too */
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x*y, h = 0.5f*y;
    const float r = fma_(-h, g, 0.5f);
    g = fma_(g, r, g); h = fma_(h, r, h);
    const float d = fma_(-g, g, x);
    return fma_(d, h, g);
}
__device__ static inline float sqrt_S5(float x) { const float s = __builtin_amdgcn_sqrtf(x); const float h = 0.5f*__builtin_amdgcn_rcpf(s); const float r = fma_(-s, s, x); return fma_(r, h, s); }

__global__ void k_sqrt(res_t* out)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x;
    const unsigned long long step = (unsigned long long)gridDim.x*blockDim.x;
    unsigned long long tried = 0, bad[NCAND] = {0};
    for(unsigned long long b = gid; b < (1ull << 31); b += step)
    {
        const float x = __uint_as_float((unsigned int)b);
        if(!(x >= 1.262177448e-29f /* 2^-96 */ && x < INFINITY)) continue;
        tried++;
        const float want = sqrtf(x);
        const float got[7] = { __builtin_amdgcn_sqrtf(x), sqrt_cur(x), sqrt_S1(x), sqrt_S2(x), sqrt_S3(x), sqrt_S4(x), sqrt_S5(x) };
        for(int k=0; k<7; k++)
            if(__float_as_uint(got[k]) != __float_as_uint(want)) { if(!bad[k]) atomicCAS(&out->first[k], 0u, (unsigned int)b); bad[k]++; }
    }
    atomicAdd(&out->tried, tried);
    for(int k=0; k<NCAND; k++) if(bad[k]) atomicAdd(&out->bad[k], bad[k]);
}

/* ---- quotient by a constant: every numerator that is zero or in [2^-62, 2^30] ---- */
__global__ void k_divc(res_t* out, float c)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x;
    const unsigned long long step = (unsigned long long)gridDim.x*blockDim.x;
    unsigned long long tried = 0, bad[NCAND] = {0};
    float cc = c; asm volatile("" : "+v"(cc));
    const float rr = rcp_A(cc);                 /* hzf_refined_rcp */
    const float rx = 1.0f / cc;                 /* the correctly rounded reciprocal */
    for(unsigned long long b = gid; b < (1ull << 32); b += step)
    {
        const float a = __uint_as_float((unsigned int)b);
        const float m = fabsf(a);
        if(!(m == 0.0f || (m >= LO && m <= 1073741824.0f))) continue;
        if(a == 0.0f && (b >> 31)) continue;            /* -0: not a numerator of the transform (hz_fast.h) */
        tried++;
        const float want = a / cc;
        float got[4];
        { float q = a*rr; float e = fma_(-cc, q, a); q = fma_(e, rr, q); e = fma_(-cc, q, a); got[0] = fma_(e, rr, q); }     /* as of round 4 */
        { float q = a*rr; float e = fma_(-cc, q, a); got[1] = fma_(e, rr, q); }                                            /* one step */
        { float q = a*rx; float e = fma_(-cc, q, a); got[2] = fma_(e, rx, q); }                                            /* one step, exact reciprocal */
        got[3] = a*rx;
        for(int k=0; k<4; k++)
            if(__float_as_uint(got[k]) != __float_as_uint(want)) { if(!bad[k]) atomicCAS(&out->first[k], 0u, (unsigned int)b); bad[k]++; }
    }
    atomicAdd(&out->tried, tried);
    for(int k=0; k<NCAND; k++) if(bad[k]) atomicAdd(&out->bad[k], bad[k]);
}

static void report(const char* what, const res_t* r, const char* const* names, int n)
{
    printf("%s: %llu operands\n", what, r->tried);
    for(int k=0; k<n; k++)
    {
        float f; memcpy(&f, &r->first[k], 4);
        if(r->bad[k]) printf("   %-58s %12llu wrong (first: 0x%08x = %.9g)\n", names[k], r->bad[k], r->first[k], f);
        else          printf("   %-58s            0 wrong\n", names[k]);
    }
    fflush(stdout);
}

int main(int argc, char** argv)
{
    res_t* d; CHECK(hipMalloc(&d, sizeof(res_t)));
    res_t h;
    const int grid = 256*32, block = 256;

    CHECK(hipMemset(d, 0, sizeof(res_t)));
    hipLaunchKernelGGL(k_rcp, dim3(grid), dim3(block), 0, 0, d);
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    { const char* n[] = { "v_rcp_f32 alone", "A: rcp + one Newton step (3 instructions)", "B: rcp + two Newton steps (5)", "C: hzf_rcp of round 4 (7)", "D: rcp, one step, residual of that corrected with rcp (5)" };
      report("reciprocal, every float with 2^-62 <= |x| <= 2^62", &h, n, 5); }

    CHECK(hipMemset(d, 0, sizeof(res_t)));
    hipLaunchKernelGGL(k_sqrt, dim3(grid), dim3(block), 0, 0, d);
    CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    { const char* n[] = { "v_sqrt_f32 alone", "hzf_sqrt of round 4 (sqrt + 8)", "S1: sqrt, rsq, mul, fma, fma", "S2: rsq, mul, mul, fma, fma", "S3: S2 + fma, fma", "S4: rsq + Goldschmidt (mul mul fma fma fma fma fma)", "S5: sqrt, rcp, mul, fma, fma" };
      report("square root, every float from 2^-96 on", &h, n, 7); }

    float cs[64]; int nc = 0;
    const float defaults[] = { 6.28318548f, 599900.0f, 39900.0f, 29900.0f, 3.0f, 1e-3f, 12345.678f, 99900.0f, 199000.0f, 1.0f, 7.0f, 49999.0f };
    for(unsigned i=0; i<sizeof(defaults)/sizeof(defaults[0]); i++) cs[nc++] = defaults[i];
    srand(12345);
    const int nrand = argc > 1 ? atoi(argv[1]) : 24;
    for(int i=0; i<nrand && nc<64; i++) { const double u = rand()/(double)RAND_MAX, v = rand()/(double)RAND_MAX; cs[nc++] = (float)(exp(u*14.0)*(1.0 + v)); }
    unsigned long long tot[4] = {0};
    for(int i=0; i<nc; i++)
    {
        CHECK(hipMemset(d, 0, sizeof(res_t)));
        hipLaunchKernelGGL(k_divc, dim3(grid), dim3(block), 0, 0, d, cs[i]);
        CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("a / %.9g, every numerator: round 4's five %llu wrong | mul fma fma %llu | the same with the exact reciprocal %llu | one mul %llu\n",
               cs[i], h.bad[0], h.bad[1], h.bad[2], h.bad[3]);
        for(int k=0; k<4; k++) tot[k] += h.bad[k];
    }
    printf("over %d divisors: %llu | %llu | %llu | %llu\n", nc, tot[0], tot[1], tot[2], tot[3]);
    return 0;
}
