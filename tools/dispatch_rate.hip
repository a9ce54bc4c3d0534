// tools/dispatch_rate.hip - how fast does the chip start workgroups?  (round 6: k_march is one wave per workgroup, 36 K to
// 100 K of them per launch; configs[1]'s launch takes 210 us where its waves' durations, list-scheduled onto 4096 slots,
// need 154.)  A kernel whose waves spin for a given time, launched as N workgroups of 1, 2 and 4 waves: waves started per
// microsecond, and the launch's duration against N x T / slots.
//   hipcc --offload-arch=gfx950 -O2 -o dispatch_rate tools/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

template<int VGPRS>
__global__ void k_spin(unsigned long long cycles, float* out)
{
    // (VGPRS: registers held, so that as many waves fit a SIMD as k_march's do - 104 registers: four)
    float keep[VGPRS];
    #pragma unroll
    for(int k=0; k<VGPRS; k++) keep[k] = (float)(threadIdx.x + k);
    const unsigned long long t0 = wall_clock64();            // the constant 100 MHz counter
    while(wall_clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(1); }
    float s = 0.f;
    #pragma unroll
    for(int k=0; k<VGPRS; k++) s += keep[k];
    if(s == 12345.678f) out[0] = s;
}

int main()
{
    float* d_out; CK(hipMalloc(&d_out, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double memtime_mhz = 100.0;       // wall_clock64(): 100 MHz
    for(int us : { 0, 5, 15, 40 })
        for(int waves_per_wg : { 1, 2, 4 })
            for(int nwaves : { 36000, 100000 })
            {
                const unsigned long long cycles = (unsigned long long)(us*memtime_mhz);
                const int nwg = nwaves/waves_per_wg;
                float best = 1e9f;
                for(int rep=0; rep<4; rep++)
                {
                    CK(hipEventRecord(e0, st));
                    hipLaunchKernelGGL(k_spin<96>, dim3(nwg), dim3(64*waves_per_wg), 0, st, cycles, d_out);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
                }
                printf("waves spin %2d us, %d wave(s) per workgroup, %6d waves: launch %8.1f us = %6.0f waves/us; N x T / 4096 slots = %7.1f us\n",
                       us, waves_per_wg, nwaves, best*1e3, nwaves/(best*1e3), (double)nwaves*us/4096.0);
            }
    return 0;
}
