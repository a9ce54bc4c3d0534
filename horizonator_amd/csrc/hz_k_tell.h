/* hz_k_tell.h - the device tells the host what k_pack_host produced, without the host asking.
 *
 * horizonator_render_offscreen() hands its results over in host memory (reference horizonator-lib.c:936-1048); what
 * travels is k_pack_host's stream of blobs (hz_k_resolve.h, hz_scatter.c), moved by the copy engine.  Before the host
 * can issue those copies it has to know how long the stream is.  Until round 5 it asked: a 16-byte copy of the cursor
 * words and an event per sector, waited for.  Now k_tell runs in stream order behind k_pack_host and stores into pinned
 * host memory, where the host polls:
 *   info[0..2]  the stream's words, blobs, overflow flag;  info[3] = epoch (never 0): they are there
 *   present[]   one bit per tile of the sector (4 rows x 2048 columns): a blob for it is in the stream - the host
 *               fills the sky of the other tiles at once, not when it has walked the whole stream
 * A few KB per sector.  (The stream itself does NOT go this way: round 6 tried - k_ship, 64 workgroups storing the
 * stream into pinned memory at the link's 55 GB/s - and measured what that costs the kernels beside it: a kernel's
 * stores that wait for PCIe hold the memory pipeline every other kernel needs; streaming HBM reads beside them took
 * 100 times their time, the framebuffer's atomics 5.5 times, the draws of a series 1.5-2.4 times.  The copy engine
 * beside the same kernels: 1.00.  tools/pcie_beside.hip, profiles/r6_pcie_beside.txt.) */
#pragma once

#define TELL_THREADS 256

__global__ __launch_bounds__(TELL_THREADS)
void k_tell(hz_tell_t s)
{
    const unsigned int tid = threadIdx.x;
    for(unsigned int i = tid; i < s.npresent; i += TELL_THREADS) s.h_present[i] = s.present[i];
    __threadfence_system();
    __syncthreads();
    if(tid == 0)
    {
        unsigned int n = s.cursor[0];
        if(n > s.capacity) n = s.capacity;      /* (an overflowed stream: the host reports it, info[2]) */
        s.h_info[0] = n & ~3u;                  /* (blobs and voids are multiples of four words) */
        s.h_info[1] = s.cursor[1]; s.h_info[2] = s.cursor[2];
        __hip_atomic_store(&s.h_info[3], s.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
