/* hz_k_hiz.h - part of hz_kernels.hip (included there; one translation unit):
 * coarse depth for the early depth test of boxes larger than the 4 x 2 pixels hz_k_march.h reads itself.
 *
 * The reference leaves hidden-surface removal to GL_LESS per fragment (horizonator-lib.c:183-185, 896-897); every
 * triangle dropped before that has to be one that could not have won a single pixel.  hz_tri_hidden() (hz_raster.h)
 * decides that from a number zs that is >= the depth stored at every pixel centre of the triangle's box - for small
 * boxes the marching wave reads the eight words itself.  A zoomed view has few small boxes (a 45 degree view of
 * 16000 columns shows a cell 300 cells away 68 pixels wide), and reading a large box costs what rasterising it costs.
 * So, for those views: two levels of "largest depth in this tile", 8 x 4 and 32 x 16 pixels, one 32-bit word per tile
 * (the upper half of the largest framebuffer word, i.e. z24 << 8 | bits of the triangle id - compared after >> 8).
 *
 * Depths in the framebuffer only ever decrease during a draw (atomicMin), so a maximum taken over a tile at ANY
 * earlier moment of the draw is still >= every depth of the tile: the summary may be as stale as it likes, it needs
 * no synchronisation with the kernels that draw.  k_hiz is the sweep that takes it: once per draw, behind the first
 * round (whose reach decides what the tables are worth: HZ_NEAR_CELLS_MAX); the second round waits for it.  Every tile of
 * the drawn columns is written by every sweep, nothing is left over from the draw before.
 *
 * Exactness: the same hz_tri_hidden() with a zs that is merely larger (a superset of pixels, read earlier) than
 * the one the 4 x 2 test would use; tests/test_gpu_parity.py draws with and without (HZ_HIZ=1 / 0), the golden
 * scenes run under HZ_HIZ=1 in tools/gpu_modes.sh. */
#pragma once

/* level 2 of the tables a draw's parameters point to */
__device__ static inline const uint32_t* hiz_level2(const hz_params_t& p) { return p.hiz + hiz_w1(p.SW)*hiz_h1(p.H); }

/* largest of a value over the 2 / 8 lanes of an aligned group (all lanes get it) */
__device__ static inline uint32_t hiz_max_xor(uint32_t v, int mask)
{
    const uint32_t o = (uint32_t)__shfl_xor((int)v, mask);
    return v > o ? v : o;
}

__global__ __launch_bounds__(256)
void k_hiz(const unsigned long long* __restrict__ fb, const unsigned char* __restrict__ touched, int seg_stride,
           int SW, int H, hz_hiz_t hz, unsigned int nunits)
{
    const int lane = threadIdx.x & 63;
    const unsigned int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6));
    const unsigned int nwaves = gridDim.x*(blockDim.x/64);
    for(unsigned int u = wave_global; u < nunits; u += nwaves)
    {
        const int s = (int)(u % (unsigned int)seg_stride), g = (int)(u / (unsigned int)seg_stride);
        const int x = s*HZ_SEG + lane*4, y0 = g*HIZ_UNIT_ROWS;
        uint32_t m1[HIZ_UNIT_ROWS >> HIZ1_H_LOG2];
        #pragma unroll
        for(int q=0; q<(HIZ_UNIT_ROWS >> HIZ1_H_LOG2); q++) m1[q] = 0u;
        #pragma unroll
        for(int r=0; r<HIZ_UNIT_ROWS; r++)
        {
            const int y = y0 + r;
            uint32_t v = 0u;            /* columns and rows outside the framebuffer: no pixel centre there, nothing to hide behind */
            if(y < H)
            {
                /* (wave-uniform; a stale zero reads as "nothing drawn here yet": all ones, which hides nothing) */
                const unsigned char t = touched[(size_t)y*seg_stride + s];
                if(!t) v = x < SW ? 0xFFFFFFFFu : 0u;
                else
                {
                    const unsigned long long* w = fb + (size_t)y*SW + x;
                    #pragma unroll
                    for(int k=0; k<4; k++)
                        if(x + k < SW)
                        {
                            const uint32_t hi = __hip_atomic_load((const uint32_t*)&w[k] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v = v > hi ? v : hi;
                        }
                }
            }
            m1[r >> HIZ1_H_LOG2] = m1[r >> HIZ1_H_LOG2] > v ? m1[r >> HIZ1_H_LOG2] : v;
        }
        /* level 1: 8 columns = this lane's four and its neighbour's */
        uint32_t m2 = 0u;
        #pragma unroll
        for(int q=0; q<(HIZ_UNIT_ROWS >> HIZ1_H_LOG2); q++)
        {
            const uint32_t t1 = hiz_max_xor(m1[q], 1);
            const int ty = (y0 >> HIZ1_H_LOG2) + q;
            if(!(lane & 1) && x < SW && (ty << HIZ1_H_LOG2) < H)
                hz.l1[(size_t)ty*hz.w1 + (x >> HIZ1_W_LOG2)] = t1;
            m2 = m2 > t1 ? m2 : t1;
        }
        /* level 2: 32 columns = eight lanes, all 16 rows */
        m2 = hiz_max_xor(m2, 2);
        m2 = hiz_max_xor(m2, 4);
        if(!(lane & 7) && x < SW)
            hz.l2[(size_t)(y0 >> HIZ2_H_LOG2)*hz.w2 + (x >> HIZ2_W_LOG2)] = m2;
    }
}

/* The smallest 24-bit depth hz_tri_fragment() gives any pixel centre of columns [x0, x1] x rows [y0, y1]: the depth is
 * round(dzdy*py + round(dzdx*px + z_org)), clamped to [0, 1], scaled and rounded - every step monotone in px and in
 * py (roundings are), the direction in px the same on every row - so it is the smallest of the four corners',
 * exactly.  false: a corner's depth is not a number (hz_tri_fragment drops such fragments; nothing is concluded).
 * tests/test_gpu_exactness.py compares it with the minimum over all the pixels of seeded rectangles and planes. */
__device__ static inline bool hiz_rect_min_depth(const hz_tri_t& tri, int x0, int x1, int y0, int y1, uint32_t* qmin)
{
    bool numbers = true;
    uint32_t m = 0xFFFFFFFFu;
    #pragma unroll
    for(int c=0; c<4; c++)
    {
        const float fpx = (float)((c & 1) ? x1 : x0), fpy = (float)((c & 2) ? y1 : y0);
        float z = __builtin_fmaf(tri.dzdy, fpy, __builtin_fmaf(tri.dzdx, fpx, tri.z_org));      /* as hz_tri_fragment */
        if(!(z == z)) numbers = false;
        z = hz_min(hz_max(z, 0.f), 1.f);
        const uint32_t q = (uint32_t)hz_roundeven(z * 16777215.f);
        m = m < q ? m : q;
    }
    *qmin = m;
    return numbers;
}

/* k_big's own look (one wave = up to 64 pixel rows of one set-up triangle, hz_k_scatter.h): can any fragment of the
 * columns [px0, px0+bw) x rows [y0, y0+nrows) of `tri` still win?  Not if the smallest depth any of them can get
 * (hiz_rect_min_depth: exact, no slack) is above every depth the level-2 tiles under the rectangle hold (each >= the
 * pixels' own): every fragment then loses GL_LESS.  Lanes read the tiles. */
__device__ static inline bool hiz_chunk_hidden(const hz_tri_t& tri, const hz_params_t& p, int px0, int bw, int y0, int nrows, int lane)
{
    const int x0 = px0 - p.col0, x1 = x0 + bw - 1, y1 = y0 + nrows - 1;
    const int tx0 = x0 >> HIZ2_W_LOG2, ntx = (x1 >> HIZ2_W_LOG2) - tx0 + 1;
    const int ty0 = y0 >> HIZ2_H_LOG2, ty1 = y1 >> HIZ2_H_LOG2;
    uint32_t zs = 0u;
    const uint32_t* l2 = hiz_level2(p);
    const size_t w2 = hiz_w2(p.SW);
    /* (not unrolled, 32-bit offsets: the registers this takes are k_big's, two waves of which have to fit beside four marching waves) */
    #pragma unroll 1
    for(int ty = ty0; ty <= ty1; ty++)
    {
        #pragma unroll 1
        for(int t = lane; t < ntx; t += 64)
        {
            const uint32_t v = l2[(uint32_t)ty*(uint32_t)w2 + (uint32_t)(tx0 + t)];
            zs = zs > v ? zs : v;
        }
    }
    zs = (uint32_t)__builtin_amdgcn_readlane((int)mr_scan_max(zs), 63) >> 8;      /* (the largest over the wave: a running maximum's last lane) */
    uint32_t qmin;
    return hiz_rect_min_depth(tri, px0, px0 + bw - 1, y0, y1, &qmin) && qmin > zs;
}
