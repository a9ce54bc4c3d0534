/* hz_k_resolve.h - part of hz_kernels.hip (included there, in this order; one translation unit):
 * readback conversion (k_resolve4, k_resolve) and the packed / sparse strips of the multi-GPU gather. */
#pragma once

/* ------------------------------------------------------------------------ */
/* resolve: framebuffer words -> BGR8, range, primitive id, z24; flips rows  */

/* CLEAR: the kernel is the last reader of this draw: it leaves the framebuffer
 * as glClear would (reference horizonator-lib.c:896), storing all ones behind
 * itself where a triangle had written - the words of the sky (62 % of the
 * benchmark image) are all ones already and are not written again */
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_resolve(unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
               unsigned char* __restrict__ bgr, float* __restrict__ ranges,
               int32_t* __restrict__ index, uint32_t* __restrict__ z24,
               int SW, int H, float znear, float zfar, unsigned int* qa, unsigned int* qb)
{
    /* (the framebuffer's queue sets are emptied with it: hz_counters_consume) */
    if(CLEAR && blockIdx.x == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    const size_t npix = (size_t)SW*H;
    for(size_t o = (size_t)blockIdx.x*blockDim.x + threadIdx.x; o < npix; o += (size_t)gridDim.x*blockDim.x)
    {
        const int yo  = (int)(o / SW);          /* output row, 0 = top            */
        const int x   = (int)(o - (size_t)yo*SW);
        const int row = H-1 - yo;               /* GL row, reference horizonator-lib.c:949-958 */
        const unsigned long long key = fb[(size_t)row*SW + x];
        if(CLEAR && key != HZ_FB_CLEAR) fb[(size_t)row*SW + x] = HZ_FB_CLEAR;
        const uint32_t zi = (uint32_t)(key >> 40);
        const bool sky = (zi == HZ_Z24_MAX);
        if(bgr)
        {
            /* reference horizonator-lib.c:185 clear colour (0,0,1) -> B=255;
             * reference fragment.glsl:15-16 terrain = (red,0,0) -> R */
            bgr[o*3+0] = sky ? 255 : 0;
            bgr[o*3+1] = 0;
            bgr[o*3+2] = sky ? 0 : (unsigned char)(key & 0xFF);
        }
        if(index) index[o] = sky ? -1 : (int32_t)(uint32_t)((key >> 8) & 0xFFFFFFFFull);
        if(z24)   z24[o]   = zi;
        if(ranges)
        {
            /* reference horizonator-lib.c:1013-1025 */
            float r = -1.0f;
            if(!sky)
            {
                const float depth = (float)((double)zi * (1.0/16777215.0));
                const float len   = depth * (zfar-znear) + znear;
                const float zt    = tanel[row] * len;
                r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);  /* = hypotf */
            }
            ranges[o] = r;
        }
    }
}

/* the same for sector widths that are a multiple of 4 and 16-byte aligned
 * buffers (the normal case): a thread takes four neighbouring pixels of one
 * row - two 16-byte loads, one store per output - and the row/column come from
 * the launch grid instead of a 64-bit division per pixel */
__device__ static inline float hz_range_from_z24(uint32_t zi, float tan_row, float znear, float zfar)
{
    /* reference horizonator-lib.c:1013-1025 */
    const float depth = (float)((double)zi * (1.0/16777215.0));
    const float len   = depth * (zfar-znear) + znear;
    const float zt    = tan_row * len;
    return (float)sqrt((double)len*(double)len + (double)zt*(double)zt);  /* = hypotf */
}

template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_resolve4(unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
                unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                int32_t* __restrict__ index, uint32_t* __restrict__ z24,
                int SW, int H, float znear, float zfar,
                unsigned char* __restrict__ touched, int seg_stride, unsigned int* qa, unsigned int* qb,
                int yo0, int yo1, int nt)
{
    /* (nt: the results leave with non-temporal stores - written once, read by nobody on this chip before they are
     * complete - so that 448 MB of them per panorama do not push the framebuffer lines of the marching kernel that runs
     * beside this one out of L2: 0.984 -> 0.959, 0.994 -> 0.979 ms per pipelined render in two A/B pairs on one box;
     * HZ_RESOLVE_NT=0 for plain stores.  Reading the framebuffer non-temporally as well, or the DEM in k_march: nothing.) */
    /* (output rows [yo0,yo1): the conversion may run in bands, so that the first results can leave for the host
     * while the rest is converted; qa: only the band that comes first empties the queue sets) */
    if(CLEAR && qa && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    /* a wave = 64 lanes x 4 pixels = one HZ_SEG-pixel segment of a row */
    static_assert(HZ_SEG == 256, "k_resolve4: one wave converts one segment");
    const int x = (int)(blockIdx.x*blockDim.x + threadIdx.x)*4;
    if(x >= SW) return;
    for(int yo = yo0 + (int)blockIdx.y; yo < yo1; yo += gridDim.y)
    {
        const int row = H-1 - yo;               /* GL row, reference horizonator-lib.c:949-958 */
        ulonglong2* src = (ulonglong2*)(fb + (size_t)row*SW + x);
        unsigned char* flag = touched + (size_t)row*seg_stride + (x >> HZ_SEG_LOG2);
        ulonglong2 k01 = { HZ_FB_CLEAR, HZ_FB_CLEAR }, k23 = k01;
        const size_t o = (size_t)yo*SW + x;
        const bool drawn = *flag != 0;          /* (the same byte for the whole wave) */
        if(!drawn)
        {
            /* Nothing was drawn into this segment - 60 % of the benchmark's: sky, without reading it, and without the
             * hundred instructions that unpack, convert and pack four words that are known to be all ones.  (Round 3
             * measured no gain from this: the period of a series of renders was then set by the first round's chain of
             * kernels; now it is the marching kernel beside which this one runs, and the conversion's instructions are
             * that kernel's.)  HZ_FRESH: the constants are made where they are stored - kept in registers across the
             * loop they took the kernel from 44 to 58, and two of its waves no longer fitted beside four marching waves. */
            #define HZ_FRESH(T, name, value) T name = (value); asm volatile("" : "+v"(name))
            if(bgr)
            {
                /* B,G,R = 255,0,0 four times (reference horizonator-lib.c:185) */
                HZ_FRESH(uint32_t, w0, 0xFF0000FFu); HZ_FRESH(uint32_t, w1, 0x00FF0000u); HZ_FRESH(uint32_t, w2, 0x0000FF00u);
                uint32_t* q = (uint32_t*)(bgr + o*3);
                if(nt) { __builtin_nontemporal_store(w0, q); __builtin_nontemporal_store(w1, q+1); __builtin_nontemporal_store(w2, q+2); }
                else { q[0] = w0; q[1] = w1; q[2] = w2; }
            }
            if(index)
            {
                HZ_FRESH(int32_t, m, -1);
                int32_t* q = index + o;
                if(nt) { __builtin_nontemporal_store(m, q); __builtin_nontemporal_store(m, q+1); __builtin_nontemporal_store(m, q+2); __builtin_nontemporal_store(m, q+3); }
                else *(int4*)q = int4{ m, m, m, m };
            }
            if(z24)
            {
                HZ_FRESH(uint32_t, m, HZ_Z24_MAX);
                uint32_t* q = z24 + o;
                if(nt) { __builtin_nontemporal_store(m, q); __builtin_nontemporal_store(m, q+1); __builtin_nontemporal_store(m, q+2); __builtin_nontemporal_store(m, q+3); }
                else *(uint4*)q = uint4{ m, m, m, m };
            }
            if(ranges)
            {
                HZ_FRESH(float, m, -1.0f);
                float* q = ranges + o;
                if(nt) { __builtin_nontemporal_store(m, q); __builtin_nontemporal_store(m, q+1); __builtin_nontemporal_store(m, q+2); __builtin_nontemporal_store(m, q+3); }
                else *(float4*)q = float4{ m, m, m, m };
            }
            #undef HZ_FRESH
            continue;
        }
        {
            k01 = src[0]; k23 = src[1];
            if(CLEAR)
            {
                const ulonglong2 ones = { HZ_FB_CLEAR, HZ_FB_CLEAR };
                if((k01.x & k01.y) != HZ_FB_CLEAR) src[0] = ones;
                if((k23.x & k23.y) != HZ_FB_CLEAR) src[1] = ones;
                if((x & (HZ_SEG-1)) == 0) *flag = 0;
            }
        }
        const unsigned long long key[4] = { k01.x, k01.y, k23.x, k23.y };
        uint32_t zi[4], pix[4];
        #pragma unroll
        for(int k=0; k<4; k++)
        {
            zi[k] = (uint32_t)(key[k] >> 40);
            /* reference horizonator-lib.c:185 clear colour (0,0,1) -> B=255; fragment.glsl:15-16 terrain = (red,0,0) -> R;
             * the three bytes B,G,R as the low 24 bits */
            pix[k] = zi[k] == HZ_Z24_MAX ? 0x0000FFu : (((uint32_t)key[k] & 0xFFu) << 16);
        }
        if(bgr)
        {
            uint3 w;
            w.x = pix[0] | (pix[1] << 24);
            w.y = (pix[1] >> 8) | (pix[2] << 16);
            w.z = (pix[2] >> 16) | (pix[3] << 8);
            if(nt) { uint32_t* q = (uint32_t*)(bgr + o*3); __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q+1); __builtin_nontemporal_store(w.z, q+2); }
            else *(uint3*)(bgr + o*3) = w;
        }
        if(index)
        {
            int4 w;
            w.x = zi[0] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[0] >> 8);
            w.y = zi[1] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[1] >> 8);
            w.z = zi[2] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[2] >> 8);
            w.w = zi[3] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[3] >> 8);
            if(nt) { int32_t* q = index + o; __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q+1); __builtin_nontemporal_store(w.z, q+2); __builtin_nontemporal_store(w.w, q+3); }
            else *(int4*)(index + o) = w;
        }
        if(z24)
        {
            uint4 w = { zi[0], zi[1], zi[2], zi[3] };
            if(nt) { uint32_t* q = z24 + o; __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q+1); __builtin_nontemporal_store(w.z, q+2); __builtin_nontemporal_store(w.w, q+3); }
            else *(uint4*)(z24 + o) = w;
        }
        if(ranges)
        {
            float4 w = { -1.0f, -1.0f, -1.0f, -1.0f };
            {
                const float tr = tanel[row];
                w.x = zi[0] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[0], tr, znear, zfar);
                w.y = zi[1] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[1], tr, znear, zfar);
                w.z = zi[2] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[2], tr, znear, zfar);
                w.w = zi[3] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[3], tr, znear, zfar);
            }
            if(nt) { float* q = ranges + o; __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q+1); __builtin_nontemporal_store(w.z, q+2); __builtin_nontemporal_store(w.w, q+3); }
            else *(float4*)(ranges + o) = w;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* results for the caller's HOST memory, without the sky                     */
/*
 * horizonator_render_offscreen() hands its results over in host memory (reference horizonator-lib.c:936-1048),
 * and what lies between this kernel and the caller's buffers is PCIe: 448 MB per 16000 x 4000 panorama at ~55 GB/s
 * was 8 of the 10.9 ms a call took.  62 % of those bytes say "sky".  This conversion writes the terrain pixels only,
 * as a stream of blobs (hz_scatter.h / hz_scatter.c: a blob = the terrain pixels of 4 rows x <= 2048 columns, with a
 * mask; tiles without terrain send nothing), 4 bytes per terrain pixel - z24<<8 | red8, the word the multi-GPU strips
 * carry too: the host threads make the BGR bytes, the depth and the float32 range of it (hz_scatter.c, the same IEEE
 * operations as k_resolve4's; round 4 sent range + shade, 5 bytes) and put them in their places between the sky
 * constants they filled in while the draw was running.  col_off: what to add to a column of the framebuffer (a
 * sector's) to get the column of the caller's image.
 * One workgroup per blob, one wave per row, the tile's words kept in registers between the count and the write-out
 * (as k_pack_sparse); a blob takes its place in the stream with one atomic add (and another if the place straddles
 * two of the chunks the stream travels in). */

template<bool CLEAR>
__global__ __launch_bounds__(64*HZ_BLOB_ROWS)
void k_pack_host(unsigned long long* __restrict__ fb, hz_hostpack_t o, int SW, int H, int col_off,
                 unsigned char* __restrict__ touched, int seg_stride, unsigned int* qa, unsigned int* qb)
{
    static_assert(HZ_SEG == 256 && HZ_BLOB_COLS == 8*HZ_SEG && HZ_BLOB_ROWS == 4, "k_pack_host: a wave = a row of eight segments");
    __shared__ uint32_t s_count[HZ_BLOB_ROWS], s_start;
    if(CLEAR && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = (int)blockIdx.x*HZ_BLOB_COLS, n = min(HZ_BLOB_COLS, SW - x0), mw = (n + 31) >> 5, nit = (n + 255) >> 8;
    const int yo0 = (int)blockIdx.y*HZ_BLOB_ROWS, yo = yo0 + wave;
    const bool have = yo < H;
    const int glrow = H-1 - (have ? yo : 0);         /* GL row, reference horizonator-lib.c:949-958 */
    unsigned long long* row = fb + (size_t)glrow*SW + x0;
    unsigned char* segflag = touched + (size_t)glrow*seg_stride + (x0 >> HZ_SEG_LOG2);
    const bool wide = (SW & 3) == 0;

    /* The words of the tile are read twice - once for the terrain bits, once more (out of L2: 64 KB a workgroup) when they
     * are written out - instead of being kept in 64 registers in between (round 4): with 96 registers a wave of this kernel
     * found room on a SIMD only when a marching wave had left it, and a call that draws its panorama in sectors converts sector
     * s while the marching kernel of sector s+1 fills the chip (profiles/r5_host_inclusive.txt: the conversions of the first
     * sectors waited for the last draw).  Now two of its waves fit beside four marching waves, like k_resolve4's. */
    auto load4 = [&](int it, unsigned long long key[4])
    {
        const int c = (it << 8) + 4*lane;
        key[0] = key[1] = key[2] = key[3] = HZ_FB_CLEAR;
        if(c + 3 < n && wide)
        {
            const ulonglong2 a = *(const ulonglong2*)(row + c), b = *(const ulonglong2*)(row + c + 2);
            key[0] = a.x; key[1] = a.y; key[2] = b.x; key[3] = b.y;
        }
        else
        {
            #pragma unroll
            for(int k=0; k<4; k++) if(c + k < n) key[k] = row[c + k];
        }
    };
    uint32_t nibs = 0, flagged = 0;
    #pragma unroll
    for(int it=0; it<8; it++)
        if(it < nit && have && segflag[it])                 /* (the same byte for the whole wave) */
        {
            flagged |= 1u << it;
            unsigned long long key[4];
            load4(it, key);
            #pragma unroll
            for(int k=0; k<4; k++) nibs |= ((uint32_t)(key[k] >> 40) != HZ_Z24_MAX ? 1u : 0u) << (4*it + k);
        }
    uint32_t count = (uint32_t)__popc(nibs);
    #pragma unroll
    for(int m=32; m>=1; m>>=1) count += __shfl_xor(count, m);
    if(lane == 0) s_count[wave] = count;
    __syncthreads();
    const uint32_t words_per_pixel = ((o.flags & HZ_BLOB_PACKED) ? 1u : 0u) + ((o.flags & HZ_BLOB_INDEX) ? 1u : 0u);
    const uint32_t total = s_count[0] + s_count[1] + s_count[2] + s_count[3];
    if(threadIdx.x == 0)
    {
        uint32_t start = HP_NONE;
        if(total)
        {
            uint32_t size = HZ_BLOB_HDR + (uint32_t)(HZ_BLOB_ROWS*mw) + total*words_per_pixel + ((o.flags & HZ_BLOB_RED) ? (total + 3u) >> 2 : 0u);
            size = (size + 3u) & ~3u;
            /* Its place in the stream: one atomic add.  (A compare-and-swap that also stepped over chunk boundaries was
             * tried first: with two thousand workgroups after the one word, one CAS succeeds per round trip to the atomic
             * unit and the rest start over - 9 ms for 4143 blobs.)  A place that straddles a chunk boundary is given up -
             * marked as a void the reader steps over, hz_scatter.c - and another is taken. */
            for(;;)
            {
                const uint32_t at = atomicAdd(&o.cursor[0], size);
                if((unsigned long long)at + size > o.capacity) { atomicExch(&o.cursor[2], 1u); break; }
                if(at / o.chunk_words == (at + size - 1u) / o.chunk_words) { start = at; break; }
                o.out[at] = HZ_BLOB_VOID; o.out[at + 1] = size;
            }
            if(start != HP_NONE) atomicAdd(&o.cursor[1], 1u);
            if(start != HP_NONE)
            {
                if(o.present) { const uint32_t tile = blockIdx.y*gridDim.x + blockIdx.x; atomicOr(&o.present[tile >> 5], 1u << (tile & 31u)); }
                uint32_t* b = o.out + start;
                b[0] = (uint32_t)yo0 | (o.flags << 16); b[1] = (uint32_t)(x0 + col_off);
                b[2] = s_count[0]; b[3] = s_count[1]; b[4] = s_count[2]; b[5] = s_count[3];
                b[6] = size; b[7] = (uint32_t)n;
            }
        }
        s_start = start;
    }
    __syncthreads();
    const uint32_t start = s_start;
    if(start != HP_NONE)
    {
        uint32_t* blob = o.out + start;
        uint32_t* mask = blob + HZ_BLOB_HDR + wave*mw;
        uint32_t* p = blob + HZ_BLOB_HDR + HZ_BLOB_ROWS*mw;
        uint32_t* d_pk = NULL; int32_t* d_idx = NULL; unsigned char* d_red = NULL;
        if(o.flags & HZ_BLOB_PACKED) { d_pk  = p;           p += total; }
        if(o.flags & HZ_BLOB_INDEX)  { d_idx = (int32_t*)p; p += total; }
        if(o.flags & HZ_BLOB_RED)    { d_red = (unsigned char*)p; }
        uint32_t at = 0;
        #pragma unroll
        for(int w=0; w<HZ_BLOB_ROWS; w++) if(w < wave) at += s_count[w];
        const unsigned long long lt = (1ull << lane) - 1ull;
        #pragma unroll
        for(int it=0; it<8; it++)
            if(it < nit)
            {
                const uint32_t nib = (nibs >> (4*it)) & 0xFu;
                /* mask word w of the segment = the nibbles of lanes 8w..8w+7 (all zero where nothing was drawn, and in rows beyond the image) */
                uint32_t v = nib << (4*(lane & 7));
                v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4);
                if((lane & 7) == 0 && (it << 8) + 4*lane < n) mask[(it << 3) + (lane >> 3)] = v;
                if(!((flagged >> it) & 1u)) continue;
                unsigned long long key[4];
                load4(it, key);
                if(CLEAR)
                {
                    /* glClear behind the last reader (reference horizonator-lib.c:896), as k_resolve4<true> */
                    const int c = (it << 8) + 4*lane;
                    if(c + 3 < n && wide)
                    {
                        const ulonglong2 ones = { HZ_FB_CLEAR, HZ_FB_CLEAR };
                        if((key[0] & key[1]) != HZ_FB_CLEAR) *(ulonglong2*)(row + c)     = ones;
                        if((key[2] & key[3]) != HZ_FB_CLEAR) *(ulonglong2*)(row + c + 2) = ones;
                    }
                    else
                    {
                        #pragma unroll
                        for(int k=0; k<4; k++) if(c + k < n && key[k] != HZ_FB_CLEAR) row[c + k] = HZ_FB_CLEAR;
                    }
                    if(lane == 0) segflag[it] = 0;
                }
                unsigned long long b[4];
                #pragma unroll
                for(int k=0; k<4; k++) b[k] = __ballot((nib >> k) & 1u);
                uint32_t q = at + (uint32_t)(__popcll(b[0] & lt) + __popcll(b[1] & lt) + __popcll(b[2] & lt) + __popcll(b[3] & lt));
                #pragma unroll
                for(int k=0; k<4; k++)
                    if((nib >> k) & 1u)
                    {
                        const unsigned long long kk = key[k];
                        if(d_pk)  d_pk[q]  = ((uint32_t)(kk >> 40) << 8) | (uint32_t)(kk & 0xFFu);     /* z24 << 8 | red8 */
                        if(d_idx) d_idx[q] = (int32_t)(uint32_t)(kk >> 8);
                        if(d_red) d_red[q] = (unsigned char)(kk & 0xFFu);       /* reference fragment.glsl:15-16: colour = (red,0,0) */
                        q++;
                    }
                at += (uint32_t)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
            }
    }
    if(CLEAR && have && start == HP_NONE)
    {
        /* (nothing was sent - a stale flag, or no room in the stream, which the host reports: the framebuffer is left as
         * glClear would leave it all the same) */
        #pragma unroll
        for(int it=0; it<8; it++)
            if((flagged >> it) & 1u)
            {
                const int c = (it << 8) + 4*lane;
                #pragma unroll
                for(int k=0; k<4; k++) if(c + k < n && row[c + k] != HZ_FB_CLEAR) row[c + k] = HZ_FB_CLEAR;
                if(lane == 0) segflag[it] = 0;
            }
    }
}

/* ------------------------------------------------------------------------ */
/* packed strips for the multi-GPU gather                                    */
/*
 * A finished strip as BGR8 + float32 range is 7 bytes per pixel, and with N
 * GPUs (N-1)/N of the panorama has to reach the gathering rank through its
 * xGMI links: at N = 2 that is 224 MB over ONE link per panorama, more time
 * than the render itself.  Everything the readback conversion needs is the
 * 24-bit depth and the 8-bit shade, so a rank ships z24<<8 | red8 (4 bytes per
 * pixel, top row first) and the gathering rank runs the conversion
 * (reference horizonator-lib.c:936-1048) on what arrives: same bytes out.
 */
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_pack(unsigned long long* __restrict__ fb, uint32_t* __restrict__ packed, int SW, int H, unsigned int* qa, unsigned int* qb)
{
    if(CLEAR && blockIdx.x == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    const size_t npix = (size_t)SW*H;
    for(size_t o = (size_t)blockIdx.x*blockDim.x + threadIdx.x; o < npix; o += (size_t)gridDim.x*blockDim.x)
    {
        const int yo = (int)(o / SW), x = (int)(o - (size_t)yo*SW);
        const unsigned long long key = fb[(size_t)(H-1 - yo)*SW + x];
        if(CLEAR && key != HZ_FB_CLEAR) fb[(size_t)(H-1 - yo)*SW + x] = HZ_FB_CLEAR;
        packed[o] = ((uint32_t)(key >> 40) << 8) | (uint32_t)(key & 0xFF);
    }
}

/* packed[H][stride] (columns 0..ncols-1 used) -> columns out_col0.. of the
 * full-width outputs bgr[H][out_W][3], ranges[H][out_W]; rows top first */
__global__ __launch_bounds__(256)
void k_resolve_packed(const uint32_t* __restrict__ packed, int stride, int ncols,
                      const float* __restrict__ tanel,
                      unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                      int out_W, int out_col0, int H, float znear, float zfar)
{
    const size_t npix = (size_t)ncols*H;
    for(size_t k = (size_t)blockIdx.x*blockDim.x + threadIdx.x; k < npix; k += (size_t)gridDim.x*blockDim.x)
    {
        const int yo = (int)(k / ncols), x = (int)(k - (size_t)yo*ncols);
        const uint32_t w  = packed[(size_t)yo*stride + x];
        const uint32_t zi = w >> 8;
        const bool sky = (zi == HZ_Z24_MAX);
        const size_t o = (size_t)yo*out_W + out_col0 + x;
        if(bgr)
        {
            bgr[o*3+0] = sky ? 255 : 0;
            bgr[o*3+1] = 0;
            bgr[o*3+2] = sky ? 0 : (unsigned char)(w & 0xFF);
        }
        if(ranges)
        {
            /* reference horizonator-lib.c:1013-1025, as k_resolve */
            float r = -1.0f;
            if(!sky)
            {
                const float depth = (float)((double)zi * (1.0/16777215.0));
                const float len   = depth * (zfar-znear) + znear;
                const float zt    = tanel[H-1 - yo] * len;
                r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);
            }
            ranges[o] = r;
        }
    }
}

/* Sparse strips: most of a panorama is sky (62 % of the benchmark image), and a
 * sky pixel carries no information.  A strip as a stream of uint32:
 *   [0]                    number of terrain pixels T
 *   [1 .. 1+H)             row_base[yo]: where row yo's words start in the data
 *   [1+H .. HDR)           terrain mask, mask_stride words per row, bit c%32 of word c/32
 *   [HDR .. HDR+T)         z24<<8 | red8 of the terrain pixels, row by row, left to right
 * with HDR = 1 + H + H*mask_stride, rows top first.  Rows may be laid out in
 * any order in the data (row_base says where): one block per row, one atomic
 * per row for its base.  The buffer must hold HDR + H*SW words; [0] must be 0
 * on entry. */
/* One WAVE packs one row, four pixels per lane and step (a step = one 256-pixel
 * segment, as in k_resolve4): all loads of a row of up to 2048 pixels are in
 * flight at once and the words stay in registers between the count and the
 * write-out - a kernel that walked a row 256 pixels at a time, one pixel per
 * thread, with a dependent load and two barriers per step, took 65 us for a
 * 2000 x 4000 strip (1 TB/s); wider rows are read twice (the second time from
 * L2).  Segments nothing was drawn into (`touched`) are sky without being read.
 * The four rows of a workgroup take their place in the data with ONE atomic
 * (same-address atomics are serialised at ~10 ns each). */

/* one step of one row: loads, terrain bits (nibble per lane), mask words */
__device__ static inline uint32_t sp_step(const unsigned long long* row, int SW, int it, int lane, bool flagged,
                                          unsigned long long key[4], uint32_t* mask)
{
    const int c = (it << 8) + 4*lane;
    key[0] = key[1] = key[2] = key[3] = HZ_FB_CLEAR;
    if(flagged)
    {
        if(c + 3 < SW && (SW & 3) == 0)
        {
            const ulonglong2 a = *(const ulonglong2*)(row + c), b = *(const ulonglong2*)(row + c + 2);
            key[0] = a.x; key[1] = a.y; key[2] = b.x; key[3] = b.y;
        }
        else
        {
            #pragma unroll
            for(int k=0; k<4; k++) if(c + k < SW) key[k] = row[c + k];
        }
    }
    uint32_t nib = 0;
    #pragma unroll
    for(int k=0; k<4; k++) nib |= ((uint32_t)(key[k] >> 40) != HZ_Z24_MAX ? 1u : 0u) << k;
    /* mask word w of the segment = the nibbles of lanes 8w..8w+7 */
    uint32_t v = nib << (4*(lane & 7));
    v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4);
    if((lane & 7) == 0 && (it << 8) + 4*lane < SW) mask[(it << 3) + (lane >> 3)] = v;
    return nib;
}

/* the words of one step: `at` = where the step's first word goes */
template<bool CLEAR>
__device__ static inline uint32_t sp_emit(unsigned long long* row, int SW, int it, int lane, uint32_t nib,
                                          const unsigned long long key[4], uint32_t* data, uint32_t at)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long b[4];
    #pragma unroll
    for(int k=0; k<4; k++) b[k] = __ballot((nib >> k) & 1u);
    uint32_t o = at + (uint32_t)(__popcll(b[0] & lt) + __popcll(b[1] & lt) + __popcll(b[2] & lt) + __popcll(b[3] & lt));
    #pragma unroll
    for(int k=0; k<4; k++)
        if((nib >> k) & 1u) data[o++] = ((uint32_t)(key[k] >> 40) << 8) | (uint32_t)(key[k] & 0xFF);
    if(CLEAR)
    {
        const int c = (it << 8) + 4*lane;
        if(c + 3 < SW && (SW & 3) == 0)
        {
            const ulonglong2 ones = { HZ_FB_CLEAR, HZ_FB_CLEAR };
            if((key[0] & key[1]) != HZ_FB_CLEAR) *(ulonglong2*)(row + c)     = ones;
            if((key[2] & key[3]) != HZ_FB_CLEAR) *(ulonglong2*)(row + c + 2) = ones;
        }
        else
        {
            #pragma unroll
            for(int k=0; k<4; k++) if(c + k < SW && key[k] != HZ_FB_CLEAR) row[c + k] = HZ_FB_CLEAR;
        }
    }
    return at + (uint32_t)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
}

template<bool CLEAR>
__global__ __launch_bounds__(64*SP_WAVES)
void k_pack_sparse(unsigned long long* __restrict__ fb, uint32_t* __restrict__ out,
                   int SW, int H, int mask_stride, unsigned char* __restrict__ touched, int seg_stride,
                   unsigned int* qa, unsigned int* qb)
{
    static_assert(HZ_SEG == 256, "k_pack_sparse: one step of a wave = one segment");
    __shared__ uint32_t row_count[SP_WAVES], row_base[SP_WAVES];
    if(CLEAR && blockIdx.x == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* data = out + 1 + (size_t)H + (size_t)H*mask_stride;
    const int nit = (SW + 255) >> 8;
    const bool cached = nit <= SP_STEPS;
    for(int y0 = blockIdx.x*SP_WAVES; y0 < H; y0 += gridDim.x*SP_WAVES)
    {
        const int yo = y0 + wave;
        const bool have = yo < H;
        unsigned long long* row = fb + (size_t)(H-1 - (have ? yo : 0))*SW;
        unsigned char* flags = touched + (size_t)(H-1 - (have ? yo : 0))*seg_stride;
        uint32_t* mask = out + 1 + H + (size_t)(have ? yo : 0)*mask_stride;
        unsigned long long key[SP_STEPS][4];
        uint32_t nibs = 0;                                  /* the nibbles of the cached steps */
        unsigned long long flagged = 0;                     /* bit it: segment `it` of the row has been drawn into (the first 64
                                                             * segments; the ones beyond - rows of more than 16384 pixels - are read) */
        auto is_flagged = [&](int it) -> bool { return it >= 64 || ((flagged >> it) & 1ull); };
        uint32_t count = 0;
        if(have)
        {
            if(lane < nit && flags[lane]) flagged = 1ull << lane;
            #pragma unroll
            for(int m=32; m>=1; m>>=1) flagged |= __shfl_xor(flagged, m);
            if(cached)
            {
                #pragma unroll
                for(int it=0; it<SP_STEPS; it++)
                    if(it < nit)
                    {
                        const uint32_t nib = sp_step(row, SW, it, lane, is_flagged(it), key[it], mask);
                        nibs |= nib << (4*it);
                    }
                count = (uint32_t)__popc(nibs);
            }
            else
                for(int it=0; it<nit; it++)
                {
                    unsigned long long k4[4];
                    count += (uint32_t)__popc(sp_step(row, SW, it, lane, is_flagged(it), k4, mask));
                }
            #pragma unroll
            for(int m=32; m>=1; m>>=1) count += __shfl_xor(count, m);
        }
        if(lane == 0) row_count[wave] = count;
        __syncthreads();
        if(threadIdx.x == 0)
        {
            uint32_t total = 0;
            #pragma unroll
            for(int w=0; w<SP_WAVES; w++) { row_base[w] = total; total += row_count[w]; }
            const uint32_t base = atomicAdd(&out[0], total);
            #pragma unroll
            for(int w=0; w<SP_WAVES; w++) row_base[w] += base;
        }
        __syncthreads();
        if(have)
        {
            uint32_t at = row_base[wave];
            if(lane == 0) out[1 + yo] = at;
            if(cached)
            {
                #pragma unroll
                for(int it=0; it<SP_STEPS; it++)
                    if(it < nit && is_flagged(it))
                        at = sp_emit<CLEAR>(row, SW, it, lane, (nibs >> (4*it)) & 0xFu, key[it], data, at);
            }
            else
                for(int it=0; it<nit; it++)
                    if(is_flagged(it))
                    {
                        /* (read again: the row has just been through this XCD's L2) */
                        unsigned long long k4[4];
                        const int c = (it << 8) + 4*lane;
                        uint32_t nib = 0;
                        #pragma unroll
                        for(int k=0; k<4; k++)
                        {
                            k4[k] = c + k < SW ? row[c + k] : HZ_FB_CLEAR;
                            nib |= ((uint32_t)(k4[k] >> 40) != HZ_Z24_MAX ? 1u : 0u) << k;
                        }
                        at = sp_emit<CLEAR>(row, SW, it, lane, nib, k4, data, at);
                    }
            if(CLEAR) for(int it=lane; it<nit; it+=64) if(is_flagged(it)) flags[it] = 0;
        }
        __syncthreads();                                    /* row_count / row_base are reused */
    }
}

/* the readback conversion on sparse strips: columns [0,ncols) of strip k go to
 * columns col0[k].. of the full-width outputs.  All strips of a panorama in one
 * launch (blockIdx.y = strip); a thread takes four neighbouring OUTPUT pixels
 * whose first column is a multiple of four, so that its results leave as one
 * 12-byte and one 16-byte store whatever column the strip starts at. */

__global__ __launch_bounds__(256)
void k_resolve_sparse(hz_strips_t st, int mask_stride,
                      const float* __restrict__ tanel,
                      unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                      int out_W, int H, float znear, float zfar)
{
    __shared__ uint32_t wave_count[2][4];
    const uint32_t* __restrict__ in = st.in[blockIdx.y];
    const int ncols = st.ncols[blockIdx.y], out_col0 = st.col0[blockIdx.y];
    if(ncols <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HDR = 1 + (size_t)H + (size_t)H*mask_stride;
    const bool aligned = (out_W & 3) == 0 && (((uintptr_t)bgr | (uintptr_t)ranges) & 15u) == 0;
    const int shift = out_col0 & 3;             /* strip column of a thread's first pixel = 4*k - shift */
    for(int yo = blockIdx.x; yo < H; yo += gridDim.x)
    {
        const uint32_t* mask = in + 1 + H + (size_t)yo*mask_stride;
        uint32_t run = in[1 + yo];
        const float tan_row = tanel[H-1 - yo];
        int flip = 0;
        for(int c0 = -shift; c0 < ncols; c0 += 1024, flip ^= 1)
        {
            const int c = c0 + 4*(int)threadIdx.x;
            /* which of the four are columns of the strip, which of those are terrain */
            uint32_t valid = 0, terrain = 0;
            #pragma unroll
            for(int k=0; k<4; k++)
            {
                const int col = c + k;
                if(col >= 0 && col < ncols)
                {
                    valid |= 1u << k;
                    terrain |= ((mask[col >> 5] >> (col & 31)) & 1u) << k;
                }
            }
            const uint32_t mine = (uint32_t)__popc(terrain);
            const uint32_t incl = mr_scan(mine, lane);
            if(lane == 63) wave_count[flip][wave] = incl;
            __syncthreads();                                            /* (two sets of counts in turn: one barrier per step) */
            uint32_t before = 0, total = 0;
            #pragma unroll
            for(int w=0; w<4; w++) { const uint32_t v = wave_count[flip][w]; total += v; if(w < wave) before += v; }
            uint32_t at = (uint32_t)HDR + run + before + incl - mine;
            run += total;
            if(!valid) continue;
            uint32_t pix[4]; float rng[4];
            #pragma unroll
            for(int k=0; k<4; k++)
            {
                pix[k] = 0x0000FFu; rng[k] = -1.0f;                    /* sky: B = 255 (reference horizonator-lib.c:185), range -1 */
                if((terrain >> k) & 1u)
                {
                    const uint32_t w = in[at++];
                    pix[k] = (w & 0xFFu) << 16;                         /* terrain: R = shade (fragment.glsl:15-16) */
                    rng[k] = hz_range_from_z24(w >> 8, tan_row, znear, zfar);
                }
            }
            const size_t o = (size_t)yo*out_W + out_col0 + c;
            if(valid == 0xFu && aligned)
            {
                if(bgr)
                {
                    uint3 w;
                    w.x = pix[0] | (pix[1] << 24);
                    w.y = (pix[1] >> 8) | (pix[2] << 16);
                    w.z = (pix[2] >> 16) | (pix[3] << 8);
                    /* (non-temporal, as k_resolve4's: nobody on this chip reads the panorama while it is assembled) */
                    uint32_t* q = (uint32_t*)(bgr + o*3);
                    __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q+1); __builtin_nontemporal_store(w.z, q+2);
                }
                if(ranges)
                {
                    float* q = ranges + o;
                    __builtin_nontemporal_store(rng[0], q); __builtin_nontemporal_store(rng[1], q+1);
                    __builtin_nontemporal_store(rng[2], q+2); __builtin_nontemporal_store(rng[3], q+3);
                }
            }
            else
            {
                #pragma unroll
                for(int k=0; k<4; k++)
                    if((valid >> k) & 1u)
                    {
                        if(bgr) { bgr[(o+k)*3+0] = (unsigned char)pix[k]; bgr[(o+k)*3+1] = 0; bgr[(o+k)*3+2] = (unsigned char)(pix[k] >> 16); }
                        if(ranges) ranges[o+k] = rng[k];
                    }
            }
        }
        __syncthreads();
    }
}
