/* valu_issue.hip - what does one wave64 vector instruction cost a gfx950 SIMD?
 *
 * DESIGN.md prices k_march (a pure VALU kernel) against the SIMDs' issue rate,
 * so that rate has to be a measurement, not a belief: this program runs streams
 * of INDEPENDENT instructions of one kind at 1, 2, 4 and 8 waves per SIMD and
 * reports SIMD cycles per wave-instruction,
 *
 *     cycles/instr = (kernel time * shader clock) / (instructions per wave * waves per SIMD)
 *
 * with the kernel time from HIP events and, beside it, from s_memtime inside
 * the waves (so that the clock the chip really held shows).  Placement is
 * checked, not assumed: every wave records HW_ID (CU, SIMD) and the program
 * prints how many waves really shared a SIMD.
 *
 *   hipcc --offload-arch=gfx950 -O2 -o valu_issue tools/valu_issue.hip
 *   ./valu_issue > profiles/valu_issue.json
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while(0)

#define UNROLL 64            /* instructions per loop body */
#define ITERS  2000

/* eight independent accumulators, rotated: no instruction depends on one of the
 * seven before it */
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

struct rec_t { unsigned long long cycles; unsigned int hw_id, xcc_id; };

#define KERNEL(NAME, BODY)                                                                      \
__global__ __launch_bounds__(256) void NAME(rec_t* out, float seed, int iters)                   \
{                                                                                               \
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f,                 \
          a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                           \
    float b = seed * 0.5f + 1.0f, c = seed + 0.25f;                                             \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
    for(int k=0; k<iters; k++) { BODY }                                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                            \
    if(s == 12345.678f) out[0].cycles = 0;               /* keeps the accumulators alive */     \
    if((threadIdx.x & 63) == 0)                                                                 \
    {                                                                                           \
        unsigned int hw, xcc;                                                                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                      \
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;                              \
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;                                             \
    }                                                                                           \
}

#define A(n) a##n
#define FMA(n)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define MUL(n)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define ADD(n)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define RCP(n)   asm volatile("v_rcp_f32 %0, %0" : "+v"(A(n)));
#define RSQ(n)   asm volatile("v_rsq_f32 %0, %0" : "+v"(A(n)));
#define SQRT(n)  asm volatile("v_sqrt_f32 %0, %0" : "+v"(A(n)));
#define RNDNE(n) asm volatile("v_rndne_f32 %0, %0" : "+v"(A(n)));
#define CVTI(n)  asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(A(n)));
#define MINF(n)  asm volatile("v_min_f32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define CMP(n)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(A(n)), "v"(b) : "vcc");
#define CNDM(n)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
#define DPPMOV(n) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(A(n)) : "v"(b));
#define MULLO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define MULHI(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define ADDU(n)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define LSHL(n)  asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(A(n)));
#define DIVFIX(n) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define DIVFMAS(n) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c) : "vcc");
#define DIVSCALE(n) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c) : "vcc");
#define READLANE(n) { int t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(A(n))); asm volatile("" :: "s"(t_)); }
#define BPERM(n) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(A(n)) : "v"(idx));

#define MINI(n)  asm volatile("v_min_i32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define MAXI(n)  asm volatile("v_max_i32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define MIN3I(n) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define MAX3F(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define MED3I(n) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define SUBU(n)  asm volatile("v_sub_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define ASHR(n)  asm volatile("v_ashrrev_i32 %0, 8, %0" : "+v"(A(n)));
#define ANDB(n)  asm volatile("v_and_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define ORB(n)   asm volatile("v_or_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define XORB(n)  asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define CMPI(n)  asm volatile("v_cmp_lt_i32 vcc, %0, %1" :: "v"(A(n)), "v"(b) : "vcc");
#define CMPI64(n) asm volatile("v_cmp_lt_i64 vcc, %0, %1" :: "v"(q##n), "v"(qb) : "vcc");
#define PKMINI16(n) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define PKSUBI16(n) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define MULI24(n) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define MADI24(n) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define MADU24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define BFEI(n)  asm volatile("v_bfe_i32 %0, %0, 8, 16" : "+v"(A(n)));
#define MINIDPP(n) asm volatile("v_min_i32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(A(n)) : "v"(b));
#define SUBUDPP(n) asm volatile("v_sub_u32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(A(n)) : "v"(b));
#define MAXFDPP(n) asm volatile("v_max_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(A(n)) : "v"(b));
#define MOVV(n)  asm volatile("v_mov_b32 %0, %1" : "=v"(A(n)) : "v"(b));
#define ADD3(n)  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
#define LSHLADD(n) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(A(n)) : "v"(b));
#define ADDLSHL(n) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(A(n)) : "v"(b));
#define MBCNT(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(A(n)) : "v"(b));
#define CNDM2(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "s"(cond));
#define SUBF(n)  asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
#define CVTF(n)  asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(A(n)));
#define CMPFE64(n) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(cond) : "v"(A(n)), "v"(b));
#define DSREAD(n) asm volatile("ds_read_b32 %0, %1" : "=v"(A(n)) : "v"(idx));
#define DSREAD128(n) asm volatile("ds_read_b128 %0, %1" : "=v"(w##n) : "v"(idx));
#define DSWRITE(n) asm volatile("ds_write_b32 %0, %1" :: "v"(idx), "v"(A(n)));

KERNEL(k_mini,    R64(MINI))
KERNEL(k_maxi,    R64(MAXI))
KERNEL(k_min3i,   R64(MIN3I))
KERNEL(k_max3f,   R64(MAX3F))
KERNEL(k_med3i,   R64(MED3I))
KERNEL(k_subu,    R64(SUBU))
KERNEL(k_ashr,    R64(ASHR))
KERNEL(k_and,     R64(ANDB))
KERNEL(k_or,      R64(ORB))
KERNEL(k_xor,     R64(XORB))
KERNEL(k_cmpi,    R64(CMPI))
KERNEL(k_pkmini16,R64(PKMINI16))
KERNEL(k_pksubi16,R64(PKSUBI16))
KERNEL(k_muli24,  R64(MULI24))
KERNEL(k_madi24,  R64(MADI24))
KERNEL(k_madu24,  R64(MADU24))
KERNEL(k_bfei,    R64(BFEI))
KERNEL(k_minidpp, R64(MINIDPP))
KERNEL(k_subudpp, R64(SUBUDPP))
KERNEL(k_maxfdpp, R64(MAXFDPP))
KERNEL(k_movv,    R64(MOVV))
KERNEL(k_add3,    R64(ADD3))
KERNEL(k_lshladd, R64(LSHLADD))
KERNEL(k_addlshl, R64(ADDLSHL))
KERNEL(k_mbcnt,   R64(MBCNT))
KERNEL(k_subf,    R64(SUBF))
KERNEL(k_cvtf,    R64(CVTF))
KERNEL(k_fma,     R64(FMA))
KERNEL(k_mul,     R64(MUL))
KERNEL(k_add,     R64(ADD))
KERNEL(k_rcp,     R64(RCP))
KERNEL(k_rsq,     R64(RSQ))
KERNEL(k_sqrt,    R64(SQRT))
KERNEL(k_rndne,   R64(RNDNE))
KERNEL(k_cvti,    R64(CVTI))
KERNEL(k_min,     R64(MINF))
KERNEL(k_cmp,     R64(CMP))
KERNEL(k_cndmask, R64(CNDM))
KERNEL(k_dppmov,  R64(DPPMOV))
KERNEL(k_mullo,   R64(MULLO))
KERNEL(k_mulhi,   R64(MULHI))
KERNEL(k_addu,    R64(ADDU))
KERNEL(k_lshl,    R64(LSHL))
KERNEL(k_divfix,  R64(DIVFIX))
KERNEL(k_divfmas, R64(DIVFMAS))
KERNEL(k_divscale,R64(DIVSCALE))
KERNEL(k_readlane,R64(READLANE))

/* 64-bit and packed forms need register pairs */
__global__ __launch_bounds__(256) void k_pkfma(rec_t* out, float seed, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed, seed+1}, a1 = a0+1.f, a2 = a0+2.f, a3 = a0+3.f, a4 = a0+4.f, a5 = a0+5.f, a6 = a0+6.f, a7 = a0+7.f;
    f2 b = {seed*0.5f+1.f, seed*0.25f+1.f}, c = {seed+0.25f, seed+0.5f};
    a0 += (float)threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #define PKFMA(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
    for(int k=0; k<iters; k++) { R64(PKFMA) }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s.x + s.y == 12345.678f) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}
__global__ __launch_bounds__(256) void k_pkmul(rec_t* out, float seed, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed, seed+1}, a1 = a0+1.f, a2 = a0+2.f, a3 = a0+3.f, a4 = a0+4.f, a5 = a0+5.f, a6 = a0+6.f, a7 = a0+7.f;
    f2 b = {seed*0.5f+1.f, seed*0.25f+1.f};
    a0 += (float)threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #define PKMUL(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
    for(int k=0; k<iters; k++) { R64(PKMUL) }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s.x + s.y == 12345.678f) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}
__global__ __launch_bounds__(256) void k_mad64(rec_t* out, float seed, int iters)
{
    unsigned long long a0 = (unsigned long long)seed + threadIdx.x, a1 = a0+1, a2 = a0+2, a3 = a0+3, a4 = a0+4, a5 = a0+5, a6 = a0+6, a7 = a0+7;
    int b = (int)seed + 3, c = (int)seed + 5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #define MAD64(n) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(A(n)) : "v"(b), "v"(c) : "vcc");
    for(int k=0; k<iters; k++) { R64(MAD64) }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s == 12345678ull) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}
__global__ __launch_bounds__(256) void k_bpermute(rec_t* out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0+1, a2 = a0+2, a3 = a0+3, a4 = a0+4, a5 = a0+5, a6 = a0+6, a7 = a0+7;
    int idx = ((threadIdx.x + 1) & 63) * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #define BPERM2(n) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(A(n)) : "v"(idx));
    for(int k=0; k<iters; k++) { R64(BPERM2) asm volatile("s_waitcnt lgkmcnt(0)"); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s == 12345.678f) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}

#define KERNEL_EX(NAME, DECL, BODY)                                                               \
__global__ __launch_bounds__(256) void NAME(rec_t* out, float seed, int iters)                   \
{                                                                                               \
    __shared__ float lds[2048];                                                                 \
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f,                 \
          a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                           \
    float b = seed * 0.5f + 1.0f;                                                               \
    int idx = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 2048;                               \
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;                                         \
    DECL                                                                                        \
    __syncthreads();                                                                            \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
    for(int k=0; k<iters; k++) { BODY }                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[(threadIdx.x*7) & 2047];              \
    if(s == 12345.678f) out[0].cycles = 0;                                                      \
    if((threadIdx.x & 63) == 0)                                                                 \
    {                                                                                           \
        unsigned int hw, xcc;                                                                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                      \
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;                              \
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;                                             \
    }                                                                                           \
}
KERNEL_EX(k_cndmask2, unsigned long long cond = __builtin_amdgcn_ballot_w64(((int)a0 & 1) != 0); , R64(CNDM2))
KERNEL_EX(k_cmpfe64,  unsigned long long cond = 0; , R64(CMPFE64) asm volatile("" :: "s"(cond));)
KERNEL_EX(k_cmpi64,   long long q0 = (long long)a0; long long q1 = q0+1; long long q2 = q0+2; long long q3 = q0+3; long long q4 = q0+4; long long q5 = q0+5; long long q6 = q0+6; long long q7 = q0+7; long long qb = q0+9; , R64(CMPI64))
KERNEL_EX(k_dsread,   (void)b; , R64(DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");)
KERNEL_EX(k_dswrite,  (void)b; , R64(DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)");)
typedef float f4_t __attribute__((ext_vector_type(4)));
KERNEL_EX(k_dsread128, f4_t w0; f4_t w1; f4_t w2; f4_t w3; f4_t w4; f4_t w5; f4_t w6; f4_t w7; idx = (threadIdx.x & 63)*16 + (threadIdx.x >> 6)*1024; (void)b; , R64(DSREAD128) asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));)

/* the IEEE float32 division as hipcc emits it (-fno-fast-math), 64 in a row on
 * independent operands: the unit k_march pays seven times per vertex */
__global__ __launch_bounds__(256) void k_ieee_div(rec_t* out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0+1, a2 = a0+2, a3 = a0+3, a4 = a0+4, a5 = a0+5, a6 = a0+6, a7 = a0+7;
    float b = seed*0.5f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int k=0; k<iters; k++)
    {
        #pragma unroll
        for(int u=0; u<8; u++)
        {
            a0 = a0 / b; a1 = a1 / b; a2 = a2 / b; a3 = a3 / b; a4 = a4 / b; a5 = a5 / b; a6 = a6 / b; a7 = a7 / b;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s == 12345.678f) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}
__global__ __launch_bounds__(256) void k_ieee_sqrt(rec_t* out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0+1, a2 = a0+2, a3 = a0+3, a4 = a0+4, a5 = a0+5, a6 = a0+6, a7 = a0+7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int k=0; k<iters; k++)
    {
        #pragma unroll
        for(int u=0; u<8; u++)
        {
            a0 = __builtin_sqrtf(a0); a1 = __builtin_sqrtf(a1); a2 = __builtin_sqrtf(a2); a3 = __builtin_sqrtf(a3);
            a4 = __builtin_sqrtf(a4); a5 = __builtin_sqrtf(a5); a6 = __builtin_sqrtf(a6); a7 = __builtin_sqrtf(a7);
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if(s == 12345.678f) out[0].cycles = 0;
    if((threadIdx.x & 63) == 0)
    {
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec_t r; r.cycles = t1 - t0; r.hw_id = hw; r.xcc_id = xcc;
        out[blockIdx.x*4 + (threadIdx.x >> 6)] = r;
    }
}

typedef void (*kern_t)(rec_t*, float, int);
struct entry_t { const char* name; kern_t k; int instr_per_body; };

int main(int argc, char** argv)
{
    (void)argc; (void)argv;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    int clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);

    const entry_t entries[] = {
        {"v_fma_f32", k_fma, UNROLL}, {"v_mul_f32", k_mul, UNROLL}, {"v_add_f32", k_add, UNROLL},
        {"v_pk_fma_f32", k_pkfma, UNROLL}, {"v_pk_mul_f32", k_pkmul, UNROLL},
        {"v_min_f32", k_min, UNROLL}, {"v_cmp_lt_f32", k_cmp, UNROLL}, {"v_cndmask_b32", k_cndmask, UNROLL},
        {"v_rndne_f32", k_rndne, UNROLL}, {"v_cvt_i32_f32", k_cvti, UNROLL},
        {"v_add_u32", k_addu, UNROLL}, {"v_lshlrev_b32", k_lshl, UNROLL},
        {"v_mul_lo_u32", k_mullo, UNROLL}, {"v_mul_hi_u32", k_mulhi, UNROLL}, {"v_mad_i64_i32", k_mad64, UNROLL},
        {"v_rcp_f32", k_rcp, UNROLL}, {"v_rsq_f32", k_rsq, UNROLL}, {"v_sqrt_f32", k_sqrt, UNROLL},
        {"v_div_scale_f32", k_divscale, UNROLL}, {"v_div_fmas_f32", k_divfmas, UNROLL}, {"v_div_fixup_f32", k_divfix, UNROLL},
        {"v_mov_b32_dpp wave_shl:1", k_dppmov, UNROLL}, {"v_readlane_b32", k_readlane, UNROLL},
        {"ds_bpermute_b32", k_bpermute, UNROLL},
        {"v_min_i32", k_mini, UNROLL}, {"v_max_i32", k_maxi, UNROLL}, {"v_min3_i32", k_min3i, UNROLL}, {"v_max3_f32", k_max3f, UNROLL},
        {"v_med3_i32", k_med3i, UNROLL}, {"v_sub_u32", k_subu, UNROLL}, {"v_sub_f32", k_subf, UNROLL}, {"v_ashrrev_i32", k_ashr, UNROLL},
        {"v_and_b32", k_and, UNROLL}, {"v_or_b32", k_or, UNROLL}, {"v_xor_b32", k_xor, UNROLL}, {"v_mov_b32", k_movv, UNROLL},
        {"v_cmp_lt_i32", k_cmpi, UNROLL}, {"v_cmp_lt_i64", k_cmpi64, UNROLL}, {"v_cmp_lt_f32_e64 (sgpr dst)", k_cmpfe64, UNROLL},
        {"v_cndmask_b32_e64 (sgpr cond)", k_cndmask2, UNROLL},
        {"v_pk_min_i16", k_pkmini16, UNROLL}, {"v_pk_sub_i16", k_pksubi16, UNROLL},
        {"v_mul_i32_i24", k_muli24, UNROLL}, {"v_mad_i32_i24", k_madi24, UNROLL}, {"v_mad_u32_u24", k_madu24, UNROLL}, {"v_bfe_i32", k_bfei, UNROLL},
        {"v_min_i32_dpp wave_shl:1", k_minidpp, UNROLL}, {"v_sub_u32_dpp wave_shl:1", k_subudpp, UNROLL}, {"v_max_f32_dpp wave_shl:1", k_maxfdpp, UNROLL},
        {"v_add3_u32", k_add3, UNROLL}, {"v_lshl_add_u32", k_lshladd, UNROLL}, {"v_add_lshl_u32", k_addlshl, UNROLL}, {"v_mbcnt_lo_u32_b32", k_mbcnt, UNROLL},
        {"v_cvt_f32_i32", k_cvtf, UNROLL},
        {"ds_read_b32 (conflict-free)", k_dsread, UNROLL}, {"ds_read_b128 (conflict-free)", k_dsread128, UNROLL}, {"ds_write_b32 (conflict-free)", k_dswrite, UNROLL},
        {"IEEE a/b (hipcc sequence), per division", k_ieee_div, UNROLL}, {"IEEE sqrtf (hipcc sequence), per sqrt", k_ieee_sqrt, UNROLL},
    };
    const int waves_per_simd[] = {1, 2, 4, 8};

    rec_t* d_out; CHECK(hipMalloc(&d_out, sizeof(rec_t)*4*ncu*8));
    std::vector<rec_t> h(4*ncu*8);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"wall_clock_rate_khz\": %d,\n", prop.gcnArchName, ncu, clock_khz, wall_khz);
    printf("  \"method\": \"streams of %d independent instructions x %d iterations per wave; blocks of 256 threads (one wave per SIMD), w blocks per CU; cycles per wave-instruction on one SIMD = elapsed / (instructions per wave * waves sharing the SIMD)\",\n", UNROLL, ITERS);
    printf("  \"rows\": [\n");
    bool first = true;
    for(const entry_t& en : entries)
        for(int w : waves_per_simd)
        {
            const int nblocks = ncu*w;
            CHECK(hipMemset(d_out, 0, sizeof(rec_t)*4*nblocks));
            hipLaunchKernelGGL(en.k, dim3(nblocks), dim3(256), 0, 0, d_out, 1.5f, 10);     /* warm */
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.k, dim3(nblocks), dim3(256), 0, 0, d_out, 1.5f, ITERS);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(rec_t)*4*nblocks, hipMemcpyDeviceToHost));
            /* how many waves really shared each SIMD */
            std::vector<int> per_simd(16*8*2*16*4, 0);
            std::vector<unsigned long long> cyc;
            for(int k=0; k<4*nblocks; k++)
            {
                const unsigned int hw = h[k].hw_id;
                const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const int xcc = h[k].xcc_id & 15;
                per_simd[(((xcc*8 + se)*2 + sh)*16 + cu)*4 + simd]++;
                cyc.push_back(h[k].cycles);
            }
            int used = 0, maxw = 0; double sumw = 0;
            for(int v : per_simd) if(v) { used++; sumw += v; maxw = std::max(maxw, v); }
            std::sort(cyc.begin(), cyc.end());
            const unsigned long long med = cyc[cyc.size()/2];
            const double ninstr = (double)en.instr_per_body*ITERS;
            /* s_memtime: if it ticks at the shader clock, med/(ninstr*waves) is the figure directly */
            const double per_instr_memtime = (double)med/(ninstr*(sumw/used));
            const double per_instr_wall_at_nominal = (double)ms*1e-3*(double)clock_khz*1e3/(ninstr*(sumw/used));
            printf("%s    {\"instr\": \"%s\", \"waves_per_simd_asked\": %d, \"simds_used\": %d, \"waves_per_simd_mean\": %.2f, \"waves_per_simd_max\": %d, "
                   "\"kernel_ms\": %.4f, \"memtime_ticks_median_wave\": %llu, \"ticks_per_wave_instr\": %.3f, \"cycles_per_wave_instr_at_nominal_clock\": %.3f}",
                   first ? "" : ",\n", en.name, w, used, sumw/used, maxw, ms, med, per_instr_memtime, per_instr_wall_at_nominal);
            first = false;
        }
    printf("\n  ]\n}\n");
    return 0;
}
