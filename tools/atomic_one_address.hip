/* atomic_one_address.hip - how many RETURNING 64-bit atomic adds per second does one address take, from every CU at once,
 * and how does that change with the number of addresses the waves are spread over?  (round 5: a zoomed view's marching
 * waves append to the draw's big-triangle queue with one such atomic per flush, all on one counter - hz_k_march.h.)
 *
 *   hipcc --offload-arch=gfx950 -O3 tools/atomic_one_address.hip -o atomic_one_address && ./atomic_one_address */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while(0)

/* one wave per block; lane 0 adds, the wave waits for the old value (as the queue append does) and does a little work */
__global__ __launch_bounds__(64) void k(unsigned long long* counters, int shards, int stride_words, int per_wave, unsigned long long* sink)
{
    const int wave = blockIdx.x;
    unsigned long long acc = 0;
    for(int k=0; k<per_wave; k++)
    {
        unsigned long long old = 0;
        if(threadIdx.x == 0) old = atomicAdd(&counters[(size_t)((wave + k) % shards)*stride_words], 0x100000001ull);
        old = __shfl(old, 0);
        acc += old;
        /* ~200 instructions of something else between appends */
        float f = (float)(acc & 1023);
        #pragma unroll 1
        for(int m=0; m<50; m++) f = f*1.0001f + 0.5f;
        acc += (unsigned long long)f;
    }
    if(acc == 0x1234567) sink[0] = acc;
}

int main()
{
    unsigned long long *d, *sink;
    CHECK(hipMalloc(&d, 64*256*sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 8));
    const int waves = 16384, per_wave = 16;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int shard_counts[] = { 1, 2, 4, 8, 16, 64 };
    for(int stride : { 16, 256 })           /* counters 128 bytes / 2 KB apart */
        for(int s : shard_counts)
        {
            CHECK(hipMemset(d, 0, 64*256*sizeof(unsigned long long)));
            for(int rep=0; rep<2; rep++)
            {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, s, stride, per_wave, sink);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%2d address(es) %4d bytes apart: %d waves x %d returning atomics in %.3f ms = %.0f M/s (%.1f ns each)\n",
                   s, stride*8, waves, per_wave, ms, waves*(double)per_wave/ms*1e-3, ms*1e6/(waves*(double)per_wave));
        }
    return 0;
}
