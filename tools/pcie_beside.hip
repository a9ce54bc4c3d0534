// tools/pcie_beside.hip - what does device-to-host traffic cost the kernels that run beside it?  (round 6)
// The host path ships a panorama's terrain pixels over PCIe while the next panorama (or sector) is drawn; the draws
// beside a transfer took 1.3-2.4 times their time alone (profiles/r5_host_inclusive.txt, profiles/r6_host_path.txt).
// Which resource do they share?  Victims: pure arithmetic, streaming HBM reads, streaming HBM writes, random 64-bit
// atomic minima on 512 MB (the framebuffer's access pattern).  Traffic: none, the copy engine (hipMemcpyAsync D2H in
// 16 MB copies), a kernel's stores into pinned host memory with 64 / 8 / 2 workgroups, the same on one XCD, and - the
// control - the same kernel storing into HBM.
//   hipcc --offload-arch=gfx950 -O2 -o pcie_beside tools/pcie_beside.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void v_alu(float* out, int iters)
{
    float a = threadIdx.x*1e-3f, b = 1.0001f;
    for(int i=0; i<iters; i++) { a = a*b + 0.5f; b = b*0.99999f + 1e-6f; }
    if(a == 12345.f) out[0] = a;
}
__global__ void v_read(const uint4* src, size_t n, uint32_t* out)
{
    uint32_t acc = 0;
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if(acc == 0x12345u) out[0] = acc;
}
__global__ void v_write(uint4* dst, size_t n)
{
    for(size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) dst[i] = make_uint4(1, 2, 3, (uint32_t)i);
}
__global__ void v_atomic(unsigned long long* fb, size_t nwords, int per_thread)
{
    uint32_t s = (blockIdx.x*blockDim.x + threadIdx.x)*2654435761u + 12345u;
    for(int k=0; k<per_thread; k++)
    {
        s = s*1664525u + 1013904223u;
        const size_t at = ((size_t)s*64u + (threadIdx.x & 63u)) % nwords;      // a wave's lanes: 64 consecutive words, like a row of pixels
        atomicMin(&fb[at], ((unsigned long long)s << 32) | k);
    }
}
// the traffic: src (HBM) -> dst (pinned host memory, or HBM for the control), for ever until *stop
__global__ void t_copy(uint4* dst, const uint4* src, size_t n, int one_xcd, volatile int* stop, unsigned long long* moved)
{
    unsigned int bid = blockIdx.x, nb = gridDim.x;
    if(one_xcd) { if(bid & 7u) return; bid >>= 3; nb >>= 3; }
    unsigned long long count = 0;
    for(int pass=0; pass<400 && !*stop; pass++)        // (bounded: never a kernel that outlives its host)
    {
        for(size_t i = (size_t)bid*blockDim.x + threadIdx.x; i < n; i += (size_t)nb*blockDim.x*4)
        {
            uint4 x[4];
            #pragma unroll
            for(int k=0; k<4; k++) { const size_t j = i + (size_t)k*nb*blockDim.x; x[k] = j < n ? src[j] : make_uint4(0,0,0,0); }
            #pragma unroll
            for(int k=0; k<4; k++) { const size_t j = i + (size_t)k*nb*blockDim.x; if(j < n) dst[j] = x[k]; }
            count += 4;
            if((count & 0xFF) == 0 && *stop) break;
        }
    }
    if(threadIdx.x == 0) atomicAdd(moved, count*blockDim.x);
}

int main()
{
    const size_t bytes = (size_t)128 << 20;
    uint4 *d_src, *d_dst, *h_pinned; float* d_out; unsigned long long* d_fb; uint4* d_big; int* h_stop; unsigned long long* d_moved;
    CK(hipMalloc(&d_src, bytes)); CK(hipMalloc(&d_dst, bytes)); CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&d_fb, (size_t)512 << 20)); CK(hipMalloc(&d_big, (size_t)2 << 30)); CK(hipMalloc(&d_moved, 8));
    CK(hipMemset(d_src, 0x5A, bytes)); CK(hipMemset(d_fb, 0xFF, (size_t)512 << 20)); CK(hipMemset(d_big, 1, (size_t)2 << 30));
    CK(hipHostMalloc((void**)&h_pinned, bytes, hipHostMallocDefault)); memset(h_pinned, 0, bytes);
    CK(hipHostMalloc((void**)&h_stop, 4, hipHostMallocDefault));
    hipStream_t sv, st; CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    struct victim_t { const char* name; int id; } victims[] = { {"arithmetic (every SIMD, 4 waves)", 0}, {"HBM streaming read, 2 GB", 1}, {"HBM streaming write, 2 GB", 2}, {"64-bit atomic minima, random rows of 512 MB", 3} };
    auto launch_victim = [&](int id)
    {
        switch(id)
        {
        case 0: hipLaunchKernelGGL(v_alu, dim3(256*4), dim3(256), 0, sv, d_out, 600000); break;
        case 1: hipLaunchKernelGGL(v_read, dim3(256*8), dim3(256), 0, sv, (const uint4*)d_big, ((size_t)2 << 30)/16, (uint32_t*)d_out); break;
        case 2: hipLaunchKernelGGL(v_write, dim3(256*8), dim3(256), 0, sv, d_big, ((size_t)2 << 30)/16); break;
        case 3: hipLaunchKernelGGL(v_atomic, dim3(256*16), dim3(256), 0, sv, d_fb, ((size_t)512 << 20)/8, 64); break;
        }
    };
    struct traffic_t { const char* name; int kind, blocks, one_xcd; } traffic[] = {
        {"alone", 0, 0, 0}, {"beside the copy engine (16 MB copies D2H)", 1, 0, 0},
        {"beside a kernel's stores to host memory, 64 workgroups", 2, 64, 0}, {"... 8 workgroups", 2, 8, 0}, {"... 2 workgroups", 2, 2, 0},
        {"... 16 workgroups on one XCD", 2, 16*8, 1},
        {"control: the same kernel storing into HBM, 64 workgroups", 3, 64, 0} };
    for(const victim_t& v : victims)
    {
        printf("== victim: %s\n", v.name);
        float alone = 0;
        for(const traffic_t& t : traffic)
        {
            float best = 1e9f; double gbs = 0;
            for(int rep=0; rep<3; rep++)
            {
                *h_stop = 0; CK(hipMemsetAsync(d_moved, 0, 8, st)); CK(hipStreamSynchronize(st));
                hipEvent_t c0, c1; CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
                CK(hipEventRecord(c0, st));
                if(t.kind == 1) for(int k=0; k<40; k++) for(size_t o=0; o<bytes; o += (size_t)16 << 20) CK(hipMemcpyAsync((char*)h_pinned + o, (char*)d_src + o, (size_t)16 << 20, hipMemcpyDeviceToHost, st));
                if(t.kind == 2) hipLaunchKernelGGL(t_copy, dim3(t.blocks), dim3(256), 0, st, h_pinned, (const uint4*)d_src, bytes/16, t.one_xcd, (volatile int*)h_stop, d_moved);
                if(t.kind == 3) hipLaunchKernelGGL(t_copy, dim3(t.blocks), dim3(256), 0, st, d_dst, (const uint4*)d_src, bytes/16, t.one_xcd, (volatile int*)h_stop, d_moved);
                CK(hipEventRecord(e0, sv)); launch_victim(v.id); CK(hipEventRecord(e1, sv)); CK(hipEventSynchronize(e1));
                *h_stop = 1;
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if(t.kind >= 2)
                {
                    CK(hipEventRecord(c1, st)); CK(hipStreamSynchronize(st));
                    float tms; CK(hipEventElapsedTime(&tms, c0, c1));
                    unsigned long long moved = 0; CK(hipMemcpy(&moved, d_moved, 8, hipMemcpyDeviceToHost));
                    gbs = (double)moved*16.0/tms/1e6;
                }
                else CK(hipStreamSynchronize(st));     // (the queued copies run out: minutes of copies would be a long wait - 40 x 128 MB = 0.1 s)
                if(ms < best) best = ms;
                CK(hipEventDestroy(c0)); CK(hipEventDestroy(c1));
            }
            if(t.kind == 0) alone = best;
            printf("   %-62s %8.3f ms  x%.2f", t.name, best, best/alone);
            if(t.kind >= 2) printf("   (traffic %.1f GB/s)", gbs);
            printf("\n");
        }
    }
    return 0;
}
