// tools/zero_copy.hip - how fast does a KERNEL write into pinned host memory over PCIe, compared with the copy engine?
// (round 5: results for host memory - should k_pack_host write its blobs straight into host memory instead of into HBM, from
// where hipMemcpyAsync fetches them in chunks?)   hipcc --offload-arch=gfx950 -O2 -o zero_copy tools/zero_copy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void k_write1(uint32_t* dst, const uint32_t* src, size_t n)
{
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) dst[i] = src[i];
}
__global__ void k_write4(uint4* dst, const uint4* src, size_t n)
{
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) dst[i] = src[i];
}
__global__ void k_write4_nt(uint4* dst, const uint4* src, size_t n)
{
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x)
    {
        const uint4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}
// a load generator: something that keeps every SIMD busy while the copies run (the marching kernel's stand-in)
__global__ void k_busy(float* out, int iters)
{
    float a = threadIdx.x*1e-3f, b = 1.0001f;
    for(int i=0; i<iters; i++) { a = a*b + 0.5f; b = b*0.99999f + 1e-6f; }
    if(a == 12345.f) out[0] = a;
}

int main()
{
    const size_t bytes = (size_t)100 << 20, words = bytes/4;
    uint32_t *d_src, *h_pinned, *h_pinned_nc; float* d_out;
    CK(hipMalloc(&d_src, bytes)); CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_src, 0x5A, bytes));
    CK(hipHostMalloc((void**)&h_pinned, bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_pinned_nc, bytes, hipHostMallocNonCoherent));
    memset(h_pinned, 0, bytes); memset(h_pinned_nc, 0, bytes);
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* what, auto fn)
    {
        float best = 1e9f;
        for(int rep=0; rep<5; rep++)
        {
            CK(hipEventRecord(e0, s)); fn(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
        }
        printf("%-70s %7.3f ms  %6.1f GB/s\n", what, best, bytes/best/1e6);
    };
    for(size_t chunk : { (size_t)4 << 20, (size_t)16 << 20, bytes })
    {
        char what[128]; snprintf(what, sizeof(what), "hipMemcpyAsync D2H, chunks of %zu MB", chunk >> 20);
        timeit(what, [&] { for(size_t o=0; o<bytes; o+=chunk) CK(hipMemcpyAsync((char*)h_pinned + o, (char*)d_src + o, o + chunk <= bytes ? chunk : bytes - o, hipMemcpyDeviceToHost, s)); });
    }
    for(int blocks : { 64, 256, 1024, 4096 })
    {
        char what[128];
        snprintf(what, sizeof(what), "kernel, dword stores to coherent pinned memory, %d blocks", blocks);
        timeit(what, [&] { hipLaunchKernelGGL(k_write1, dim3(blocks), dim3(256), 0, s, h_pinned, d_src, words); });
        snprintf(what, sizeof(what), "kernel, dwordx4 stores to coherent pinned memory, %d blocks", blocks);
        timeit(what, [&] { hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(256), 0, s, (uint4*)h_pinned, (const uint4*)d_src, words/4); });
        snprintf(what, sizeof(what), "kernel, non-temporal dword stores to coherent pinned memory, %d blocks", blocks);
        timeit(what, [&] { hipLaunchKernelGGL(k_write4_nt, dim3(blocks), dim3(256), 0, s, (uint4*)h_pinned, (const uint4*)d_src, words/4); });
        snprintf(what, sizeof(what), "kernel, dwordx4 stores to NON-coherent pinned memory, %d blocks", blocks);
        timeit(what, [&] { hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(256), 0, s, (uint4*)h_pinned_nc, (const uint4*)d_src, words/4); });
    }
    if(memcmp(h_pinned, h_pinned_nc, bytes) != 0 || h_pinned[12345] != 0x5A5A5A5Au) printf("!! the bytes did not arrive\n");
    // ... and beside a kernel that fills the chip (on another stream)
    for(int mode=0; mode<2; mode++)
    {
        hipLaunchKernelGGL(k_busy, dim3(256*16), dim3(256), 0, s2, d_out, 400000);
        if(mode == 0) timeit("beside a busy chip: hipMemcpyAsync D2H, chunks of 16 MB", [&] { for(size_t o=0; o<bytes; o+=(size_t)16<<20) CK(hipMemcpyAsync((char*)h_pinned + o, (char*)d_src + o, o + ((size_t)16<<20) <= bytes ? (size_t)16<<20 : bytes - o, hipMemcpyDeviceToHost, s)); });
        else timeit("beside a busy chip: kernel, dwordx4 stores to coherent pinned memory, 256 blocks", [&] { hipLaunchKernelGGL(k_write4, dim3(256), dim3(256), 0, s, (uint4*)h_pinned, (const uint4*)d_src, words/4); });
        CK(hipStreamSynchronize(s2));
    }
    return 0;
}
