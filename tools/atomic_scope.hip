/* atomic_scope.hip - what does a 64-bit atomic minimum into the framebuffer cost, and where is it executed?
 *
 * k_big is bound by its atomics on zoomed views (131 M fragments of a first round in 0.85 ms = 154 G atomics/s =
 * 1.2 TB/s of 8-byte words: DESIGN.md appendix C).  An MI355X has eight XCDs with an L2 each; an atomic of agent
 * (device) scope carries sc1 and is executed where all XCDs see it - behind the L2s -, one of workgroup scope is
 * executed in the issuing XCD's L2.  This program measures both, on a buffer the size of cfg3's framebuffer
 * (16000 x 4000 x 8 B = 512 MB), with the access pattern of a span rasteriser: a wave = 64 consecutive words of one
 * row, rows and columns pseudo-random, and
 *   - "anywhere":  every wave may hit any pixel (what k_big does now; workgroup scope is then WRONG across XCDs -
 *                  timed all the same, the result is not looked at),
 *   - "own stripes": a wave only hits columns of stripes its XCD owns (stripe = 256 columns, owner = stripe % 8,
 *                  XCC_ID read from the hardware register): workgroup scope is then exact, and checked - the
 *                  buffer after the run must equal a host-side replay.
 *
 *   hipcc --offload-arch=gfx950 -O2 -o tools/build/atomic_scope tools/atomic_scope.hip
 *   tools/build/atomic_scope > profiles/r4_atomic_scope.json
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while(0)

#define W 16000
#define H 4000
#define STRIPE 256
#define NSTRIPES ((W + STRIPE-1)/STRIPE)

__device__ __forceinline__ unsigned int mix(unsigned int x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

/* MODE 0: agent scope, anywhere   1: workgroup scope, anywhere (inexact)   2: agent scope, own stripes   3: workgroup scope, own stripes
 * 4: plain 8-byte stores, anywhere (the ceiling of writing that many words)   5: agent-scope loads, anywhere */
template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(unsigned long long* fb, int iters, unsigned int seed, unsigned int* xcc_of_wg)
{
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x*blockDim.x + threadIdx.x) >> 6;
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if(threadIdx.x == 0 && xcc_of_wg) xcc_of_wg[blockIdx.x] = xcc;
    unsigned long long sink = 0;
    for(int it=0; it<iters; it++)
    {
        const unsigned int r = mix(seed + wave*7919u + (unsigned int)it*104729u);
        const unsigned int y = r % H;
        unsigned int x0;
        if(MODE == 2 || MODE == 3)
        {
            /* a stripe this XCD owns, a 64-word span inside it */
            const unsigned int nown = (NSTRIPES - xcc + 7u)/8u;
            const unsigned int s = xcc + 8u*((r >> 12) % nown);
            unsigned int room = STRIPE - 64; if(s*STRIPE + STRIPE > W) room = W - s*STRIPE - 64;
            x0 = s*STRIPE + (r >> 20) % (room + 1);
        }
        else
            x0 = (r >> 12) % (W - 64);
        unsigned long long* p = fb + (size_t)y*W + x0 + lane;
        const unsigned long long key = ((unsigned long long)(r & 0xFFFFFFu) << 40) | ((unsigned long long)wave << 8) | (unsigned int)lane;
        if(MODE == 0 || MODE == 2)      __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if(MODE == 1 || MODE == 3) __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if(MODE == 4)              *p = key;
        else                            sink += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if(MODE == 5 && sink == 1) fb[0] = sink;
}

typedef void (*kern_t)(unsigned long long*, int, unsigned int, unsigned int*);

int main(void)
{
    const size_t words = (size_t)W*H;
    unsigned long long* fb;
    CHECK(hipMalloc(&fb, words*8));
    const int wgs = 4096, iters = 64;               /* 16384 waves (k_big's launch) x 64 spans x 64 words = 67 M atomics per run */
    unsigned int* d_xcc; CHECK(hipMalloc(&d_xcc, wgs*sizeof(unsigned int)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const kern_t kerns[6] = { k_atomics<0>, k_atomics<1>, k_atomics<2>, k_atomics<3>, k_atomics<4>, k_atomics<5> };
    const char* names[6] = { "agent scope, anywhere", "workgroup scope, anywhere (inexact across XCDs; timing only)", "agent scope, own stripes",
                             "workgroup scope, own stripes", "plain stores, anywhere", "agent-scope loads, anywhere" };
    std::vector<unsigned long long> want(words), got(words);
    std::vector<unsigned int> xcc(wgs);
    printf("{\"what\": \"64-bit atomic minimum, a wave = 64 consecutive words of a row of a 16000x4000 buffer (512 MB), 16384 waves x %d spans\", \"rows\": [\n", iters);
    for(int m=0; m<6; m++)
    {
        float best = 1e30f;
        for(int rep=0; rep<4; rep++)
        {
            CHECK(hipMemset(fb, 0xFF, words*8));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kerns[m], dim3(wgs), dim3(256), 0, 0, fb, iters, 12345u, d_xcc);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if(rep && ms < best) best = ms;
        }
        const double n = (double)wgs*4*iters*64;
        int exact = -1;
        if(m == 2 || m == 3)
        {
            /* replay on the host with the XCC ids the workgroups really had */
            CHECK(hipMemcpy(got.data(), fb, words*8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(xcc.data(), d_xcc, wgs*sizeof(unsigned int), hipMemcpyDeviceToHost));
            std::fill(want.begin(), want.end(), ~0ull);
            for(unsigned int wave=0; wave<(unsigned int)wgs*4; wave++)
                for(int it=0; it<iters; it++)
                {
                    unsigned int x = 12345u + wave*7919u + (unsigned int)it*104729u;
                    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
                    const unsigned int r = x, y = r % H, c = xcc[wave/4];
                    const unsigned int nown = (NSTRIPES - c + 7u)/8u, s = c + 8u*((r >> 12) % nown);
                    unsigned int room = STRIPE - 64; if(s*STRIPE + STRIPE > W) room = W - s*STRIPE - 64;
                    const unsigned int x0 = s*STRIPE + (r >> 20) % (room + 1);
                    for(int lane=0; lane<64; lane++)
                    {
                        const unsigned long long key = ((unsigned long long)(r & 0xFFFFFFu) << 40) | ((unsigned long long)wave << 8) | (unsigned int)lane;
                        unsigned long long& w = want[(size_t)y*W + x0 + lane];
                        if(key < w) w = key;
                    }
                }
            exact = memcmp(want.data(), got.data(), words*8) == 0 ? 1 : 0;
        }
        unsigned int seen = 0;
        if(m == 0) { CHECK(hipMemcpy(xcc.data(), d_xcc, wgs*sizeof(unsigned int), hipMemcpyDeviceToHost)); for(int k=0; k<wgs; k++) seen |= 1u << xcc[k]; }
        printf("  {\"mode\": \"%s\", \"ms\": %.4f, \"G_per_s\": %.1f, \"TB_per_s_of_words\": %.3f, \"equals_host_replay\": %s%s}%s\n", names[m], best, n/best/1e6, n*8/best/1e9,
               exact < 0 ? "null" : exact ? "true" : "false", m == 0 ? (seen == 0xFFu ? ", \"xcc_ids_seen\": \"0-7\"" : ", \"xcc_ids_seen\": \"not all of 0-7\"") : "", m == 5 ? "" : ",");
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}
