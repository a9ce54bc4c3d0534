// tools/pinned_alloc.hip - what does the pinned landing area of a panorama cost horizonator_init(), and is there a cheaper way
// to get one?  (round 6: "host path: threads, pinned memory" is 79-93 ms of cfg3's init and 278 ms of cfg5's, HZ_INIT_TIMES=1)
//   (a) hipHostMalloc(bytes)                                       what hz_hostpath.cpp does
//   (b) mmap + MADV_HUGEPAGE, touched by T threads, hipHostRegister pages of 2 MB for the driver to pin instead of 4 KB ones
//   (c) as (b) without the touch (the driver faults the pages in)
// ... and what the copy engine then makes of each: device-to-host copies of 16 MB, GB/s.
//   hipcc --offload-arch=gfx950 -O2 -o pinned_alloc tools/pinned_alloc.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double copy_rate(void* h, const void* d, size_t bytes, hipStream_t s)
{
    const size_t piece = (size_t)16 << 20;
    double best = 0;
    for(int rep=0; rep<3; rep++)
    {
        const double t0 = now_ms();
        for(size_t o=0; o<bytes; o+=piece) CK(hipMemcpyAsync((char*)h + o, (const char*)d + o, o + piece <= bytes ? piece : bytes - o, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const double gbps = (double)bytes/((now_ms() - t0)*1e6);
        if(gbps > best) best = gbps;
    }
    return best;
}

static void touch(char* p, size_t bytes, int nthreads)
{
    std::vector<std::thread> ts;
    for(int k=0; k<nthreads; k++)
        ts.emplace_back([=] { const size_t lo = bytes*k/nthreads, hi = bytes*(k+1)/nthreads; for(size_t o=lo; o<hi; o+=4096) p[o] = 0; });
    for(auto& t : ts) t.join();
}

int main(int argc, char** argv)
{
    const int nthreads = argc > 1 ? atoi(argv[1]) : 32;
    CK(hipSetDevice(0));
    void* d = NULL;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    { void* w; CK(hipHostMalloc(&w, 1 << 20, hipHostMallocDefault)); CK(hipHostFree(w)); }      // (the first pinned allocation of a process: not what is measured)
    for(size_t mb : { (size_t)277, (size_t)1100 })
    {
        const size_t bytes = mb << 20;
        CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0x5A, bytes)); CK(hipDeviceSynchronize());
        for(int rep=0; rep<2; rep++)
        {
            // (a)
            double t0 = now_ms();
            void* h = NULL; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
            const double t_a = now_ms() - t0;
            const double r_a = copy_rate(h, d, bytes, s);
            t0 = now_ms(); CK(hipHostFree(h)); const double t_af = now_ms() - t0;
            // (b)
            t0 = now_ms();
            char* m = (char*)mmap(NULL, bytes + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if(m == MAP_FAILED) { printf("mmap failed\n"); return 1; }
            char* al = (char*)(((uintptr_t)m + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1));
            (void)madvise(al, bytes, MADV_HUGEPAGE);
            touch(al, bytes, nthreads);
            const double t_touch = now_ms() - t0;
            CK(hipHostRegister(al, bytes, hipHostRegisterDefault));
            const double t_b = now_ms() - t0;
            const double r_b = copy_rate(al, d, bytes, s);
            t0 = now_ms(); CK(hipHostUnregister(al)); munmap(m, bytes + ((size_t)2 << 20)); const double t_bf = now_ms() - t0;
            // (c)
            t0 = now_ms();
            m = (char*)mmap(NULL, bytes + ((size_t)2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            al = (char*)(((uintptr_t)m + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1));
            (void)madvise(al, bytes, MADV_HUGEPAGE);
            CK(hipHostRegister(al, bytes, hipHostRegisterDefault));
            const double t_c = now_ms() - t0;
            const double r_c = copy_rate(al, d, bytes, s);
            CK(hipHostUnregister(al)); munmap(m, bytes + ((size_t)2 << 20));
            printf("%4zu MB: hipHostMalloc %7.1f ms (copies %5.1f GB/s, free %5.1f ms) | huge pages touched by %d threads %6.1f ms + hipHostRegister = %7.1f ms (copies %5.1f GB/s, unregister+unmap %5.1f ms) | "
                   "hipHostRegister of untouched huge pages %7.1f ms (copies %5.1f GB/s)\n", mb, t_a, r_a, t_af, nthreads, t_touch, t_b, r_b, t_bf, t_c, r_c);
            fflush(stdout);
        }
        CK(hipFree(d));
    }
    return 0;
}
