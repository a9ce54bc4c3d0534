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
                unsigned char* __restrict__ touched, int seg_stride, unsigned int* qa, unsigned int* qb)
{
    if(CLEAR && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    /* a wave = 64 lanes x 4 pixels = one HZ_SEG-pixel segment of a row */
    static_assert(HZ_SEG == 256, "k_resolve4: one wave converts one segment");
    const int x = (int)(blockIdx.x*blockDim.x + threadIdx.x)*4;
    if(x >= SW) return;
    for(int yo = blockIdx.y; yo < H; yo += gridDim.y)
    {
        const int row = H-1 - yo;               /* GL row, reference horizonator-lib.c:949-958 */
        ulonglong2* src = (ulonglong2*)(fb + (size_t)row*SW + x);
        unsigned char* flag = touched + (size_t)row*seg_stride + (x >> HZ_SEG_LOG2);
        ulonglong2 k01 = { HZ_FB_CLEAR, HZ_FB_CLEAR }, k23 = k01;
        if(*flag)                               /* (the same byte for the whole wave) */
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
        const size_t o = (size_t)yo*SW + x;
        if(bgr)
        {
            uint3 w;
            w.x = pix[0] | (pix[1] << 24);
            w.y = (pix[1] >> 8) | (pix[2] << 16);
            w.z = (pix[2] >> 16) | (pix[3] << 8);
            *(uint3*)(bgr + o*3) = w;
        }
        if(index)
        {
            int4 w;
            w.x = zi[0] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[0] >> 8);
            w.y = zi[1] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[1] >> 8);
            w.z = zi[2] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[2] >> 8);
            w.w = zi[3] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[3] >> 8);
            *(int4*)(index + o) = w;
        }
        if(z24) { uint4 w = { zi[0], zi[1], zi[2], zi[3] }; *(uint4*)(z24 + o) = w; }
        if(ranges)
        {
            const float tr = tanel[row];
            float4 w;
            w.x = zi[0] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[0], tr, znear, zfar);
            w.y = zi[1] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[1], tr, znear, zfar);
            w.z = zi[2] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[2], tr, znear, zfar);
            w.w = zi[3] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[3], tr, znear, zfar);
            *(float4*)(ranges + o) = w;
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
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_pack_sparse(unsigned long long* __restrict__ fb, uint32_t* __restrict__ out,
                   int SW, int H, int mask_stride, unsigned int* qa, unsigned int* qb)
{
    __shared__ uint32_t wave_count[4];
    __shared__ uint32_t row_base_s;
    if(CLEAR && blockIdx.x == 0 && threadIdx.x == 0) hz_counters_consume(qa, qb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HDR = 1 + (size_t)H + (size_t)H*mask_stride;
    for(int yo = blockIdx.x; yo < H; yo += gridDim.x)
    {
        unsigned long long* row = fb + (size_t)(H-1 - yo)*SW;
        uint32_t* mask = out + 1 + H + (size_t)yo*mask_stride;
        /* pass 1: mask and count */
        uint32_t mine = 0;
        for(int c0 = 0; c0 < SW; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            const bool terrain = c < SW && (uint32_t)(row[c] >> 40) != HZ_Z24_MAX;
            const unsigned long long b = __ballot(terrain);
            if(lane == 0  && c0 + wave*64      < SW) mask[(c0 >> 5) + wave*2]     = (uint32_t)b;
            if(lane == 32 && c0 + wave*64 + 32 < SW) mask[(c0 >> 5) + wave*2 + 1] = (uint32_t)(b >> 32);
            mine += (uint32_t)__popcll(b);                  /* the same in every lane of the wave */
        }
        if(lane == 0) wave_count[wave] = mine;
        __syncthreads();
        if(threadIdx.x == 0)
        {
            const uint32_t total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            const uint32_t base = atomicAdd(&out[0], total);
            out[1 + yo] = base;
            row_base_s = base;
        }
        __syncthreads();
        /* pass 2: the words (the row is in L2 now) */
        uint32_t run = row_base_s;
        for(int c0 = 0; c0 < SW; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            unsigned long long key = 0;
            bool terrain = false;
            if(c < SW)
            {
                key = row[c];
                terrain = (uint32_t)(key >> 40) != HZ_Z24_MAX;
                if(CLEAR && key != HZ_FB_CLEAR) row[c] = HZ_FB_CLEAR;
            }
            const unsigned long long b = __ballot(terrain);
            __syncthreads();
            if(lane == 0) wave_count[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0;
            for(int w=0; w<wave; w++) before += wave_count[w];
            if(terrain)
                out[HDR + run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] =
                    ((uint32_t)(key >> 40) << 8) | (uint32_t)(key & 0xFF);
            run += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        }
        __syncthreads();
    }
}

/* the readback conversion on a sparse strip: columns [0,ncols) of the strip go
 * to columns out_col0.. of the full-width outputs */
__global__ __launch_bounds__(256)
void k_resolve_sparse(const uint32_t* __restrict__ in, int mask_stride, int ncols,
                      const float* __restrict__ tanel,
                      unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                      int out_W, int out_col0, int H, float znear, float zfar)
{
    __shared__ uint32_t wave_count[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HDR = 1 + (size_t)H + (size_t)H*mask_stride;
    for(int yo = blockIdx.x; yo < H; yo += gridDim.x)
    {
        const uint32_t* mask = in + 1 + H + (size_t)yo*mask_stride;
        uint32_t run = in[1 + yo];
        const float tan_row = tanel[H-1 - yo];
        for(int c0 = 0; c0 < ncols; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            const bool terrain = c < ncols && ((mask[c >> 5] >> (c & 31)) & 1u);
            const unsigned long long b = __ballot(terrain);
            __syncthreads();
            if(lane == 0) wave_count[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0;
            for(int w=0; w<wave; w++) before += wave_count[w];
            if(c < ncols)
            {
                const size_t o = (size_t)yo*out_W + out_col0 + c;
                uint32_t w = 0;
                if(terrain) w = in[HDR + run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))];
                if(bgr)
                {
                    bgr[o*3+0] = terrain ? 0 : 255;
                    bgr[o*3+1] = 0;
                    bgr[o*3+2] = terrain ? (unsigned char)(w & 0xFF) : 0;
                }
                if(ranges)
                {
                    float r = -1.0f;
                    if(terrain)
                    {
                        const float depth = (float)((double)(w >> 8) * (1.0/16777215.0));
                        const float len   = depth * (zfar-znear) + znear;
                        const float zt    = tan_row * len;
                        r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);
                    }
                    ranges[o] = r;
                }
            }
            run += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        }
        __syncthreads();
    }
}
