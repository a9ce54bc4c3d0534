/* hz_k_tile.h - part of hz_kernels.hip (included there, after hz_k_scatter.h; one translation unit):
 * the large triangles of a round rasterised by screen tile, each tile's depth in LDS.
 *
 * BASELINE north_star's "tile-binned software rasteriser ... per-bin depth": the triangles k_march and k_clip
 * queued for k_big are binned by the 64 x 32 pixel tiles of the image they can cover; one workgroup OWNS a
 * tile - it is the only writer of those pixels while it runs (the kernels that wrote them before are earlier
 * on its stream, the second round and the conversion wait for it) -, takes the tile's 2048 framebuffer words
 * into LDS, draws the tile's triangles with LDS atomic minima, and writes the tile back with plain stores: one
 * store per pixel where k_big issues one device-scope atomic per FRAGMENT into a framebuffer that is larger
 * than the last-level cache.  Same fragments (hz_row_span, hz_tri_fragment), same keys, a minimum either way:
 * the same bytes (the GPU parity suite is green under HZ_TILES=1).
 *
 * NOT THE DEFAULT (HZ_TILES=1 switches it on; profiles/r3_experiments.json): as built it is slower than k_big
 * wherever it was measured - 16000x4000, every kernel alone: k_tile_bin 42-58 us + k_tile_raster 417 us for the
 * first round where k_big takes 310, 246 us for the second round where k_big takes 62; pipelined 1.00 -> 1.28 ms
 * per render; 40 km far clip 0.68 -> 0.75-0.83; a 45 degree view 2.85 -> 3.45 - and the most a perfect one could
 * gain is what plain stores instead of k_big's atomics gain (2.6 % of the headline render, 19 % with the 40 km
 * far clip: same file).  Why: a tile is one workgroup's job and the tiles are not alike - those along the horizon
 * hold a thousand and more small triangles (256 per tile was not enough for the benchmark scene), each costing the
 * wave its row spans and a pass of its own, four waves working through them one after the other, while k_big
 * deals its row chunks over the whole chip; the second round's large triangles are few and scattered, so that
 * most of a tile's 32 KB round trip carries nothing.  A rasteriser of this kind that wins needs a second path
 * for small triangles inside a tile (a lane each) and larger tiles for the largest - k_big with LDS in front of
 * it, not instead of it.  Kept as the measured answer to "why not LDS depth bins".
 *
 * k_tile_bin     one pass over the round's queue (a lane per triangle, the wave for those that touch many tiles):
 *                the triangle's number goes into the list of every tile of its box that an edge test does not
 *                rule out.  A tile's list holds TL_LIST (2048) numbers; one tile with more sends the whole round back to
 *                k_big (flag) - no counting pass, no prefix sum (a single-workgroup scan over the 31 K tiles of a
 *                16000x4000 image alone took 82 us per round)
 * k_tile_raster  one workgroup per tile with at least one triangle
 */
#pragma once

#define TL_W 64
#define TL_H 32

#define TL_LIST 2048                /* triangles a tile's list holds */

/* what the tile kernels share: per queue set, allocated with the context */
struct tl_bins_t
{
    unsigned int* cursor;           /* [ntiles]: triangles listed for the tile (all zero between rounds: k_tile_raster zeroes its own) */
    unsigned int* pairs;            /* [ntiles][TL_LIST]: record numbers                                                          */
    unsigned int* state;            /* [0] 1 = some tile's list overflowed: k_tile_raster stands down, k_big draws the round;
                                     * [1] tiles with a list (both zeroed in front of k_tile_bin)                                 */
    unsigned int* busy;             /* [ntiles]: the numbers of the tiles with a list (an empty workgroup still costs its dispatch:
                                     * 31 K of them - one per tile of a 16000x4000 image - 0.25 ms)                              */
    int           tiles_x, tiles_y;
    unsigned int  list_cap;         /* <= TL_LIST (tests make it small: HZ_TILE_LIST) */
};

/* can edge m have a covered pixel centre inside the tile [x0,x1] x [y0,y1]?  (its function g + dx*py - dy*px is
 * linear: the maximum over the tile is at a corner) */
__device__ static inline bool tl_edge_reaches(const hz_edges_t& e, int m, int x0, int x1, int y0, int y1)
{
    const int64_t dx = e.dx[m], ndy = e.ndy[m];
    const int64_t best = hz_edges_g(&e, m) + dx*(int64_t)(dx >= 0 ? y1 : y0) + ndy*(int64_t)(ndy >= 0 ? x1 : x0);
    return best >= 0;
}

/* One pass over the round's queue, listing every (triangle, tile) pair.  The records
 * are reached through their FIRST work item (chunk 0): every valid record has exactly one - items and records beyond the
 * first queue overflow were drawn by their producer and have none.  A lane per item: a triangle that touches up to
 * TL_SOLO tiles is walked by its own lane, a larger one by the whole wave, one after the other. */
#define TL_SOLO 8
__device__ static inline void tl_visit(const hz_bigrec_t& br, int k, int tx0, int ty0, int ntx, const tl_bins_t& tb, const hz_params_t& p,
                                       unsigned int rec)
{
    const int px0 = br.r.px0 - p.col0, py0 = br.r.py0, bw = br.r.bw, bh = br.bh;       /* columns relative to the sector */
    const int ty = ty0 + k/ntx, tx = tx0 + k%ntx;
    const int x0 = max(tx*TL_W, px0) + p.col0, x1 = min(tx*TL_W + TL_W-1, px0 + bw - 1) + p.col0;
    const int y0 = max(ty*TL_H, py0), y1 = min(ty*TL_H + TL_H-1, py0 + bh - 1);
    if(!(tl_edge_reaches(br.r.e, 0, x0, x1, y0, y1) && tl_edge_reaches(br.r.e, 1, x0, x1, y0, y1) && tl_edge_reaches(br.r.e, 2, x0, x1, y0, y1)))
        return;
    const int tile = ty*tb.tiles_x + tx;
    const unsigned int at = atomicAdd(&tb.cursor[tile], 1u);
    if(at == 0) tb.busy[atomicAdd(&tb.state[1], 1u)] = (unsigned int)tile;
    if(at < tb.list_cap) tb.pairs[(size_t)tile*TL_LIST + at] = rec;
    else tb.state[0] = 1u;                                      /* (any number of writers, one value) */
}

__global__ __launch_bounds__(256)
void k_tile_bin(const hz_bigrec_t* __restrict__ bigrec, const hz_bigitem_t* __restrict__ bigitem,
                const unsigned int* __restrict__ big_counters, unsigned int bigrec_capacity, tl_bins_t tb, hz_params_t p)
{
    const unsigned int nitems = min(big_counters[1], ~big_counters[2]);
    const int lane = threadIdx.x & 63;
    (void)bigrec_capacity;
    for(unsigned int base = (blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6))*64u; base < nitems; base += gridDim.x*(blockDim.x/64)*64u)
    {
        const unsigned int it = base + lane;
        bool mine = false;
        unsigned int rec = 0;
        int tx0 = 0, ty0 = 0, ntx = 1, nt = 0;
        if(it < nitems)
        {
            const hz_bigitem_t item = bigitem[it];
            if(item.chunk == 0)
            {
                mine = true; rec = item.rec;
                const hz_bigrec_t& br = bigrec[rec];
                const int px0 = br.r.px0 - p.col0, py0 = br.r.py0;
                tx0 = px0/TL_W; ty0 = py0/TL_H;
                ntx = (px0 + br.r.bw - 1)/TL_W - tx0 + 1;
                nt  = ntx*((py0 + br.bh - 1)/TL_H - ty0 + 1);
            }
        }
        if(mine && nt <= TL_SOLO)
            for(int k=0; k<nt; k++) tl_visit(bigrec[rec], k, tx0, ty0, ntx, tb, p, rec);
        unsigned long long wide = __ballot(mine && nt > TL_SOLO);
        while(wide)
        {
            const int src = (int)__builtin_ctzll(wide);
            wide &= wide - 1;
            const unsigned int r = (unsigned int)__builtin_amdgcn_readlane((int)rec, src);
            const int wtx0 = __builtin_amdgcn_readlane(tx0, src), wty0 = __builtin_amdgcn_readlane(ty0, src);
            const int wntx = __builtin_amdgcn_readlane(ntx, src), wnt = __builtin_amdgcn_readlane(nt, src);
            for(int k = lane; k < wnt; k += 64) tl_visit(bigrec[r], k, wtx0, wty0, wntx, tb, p, r);
        }
    }
}

__global__ __launch_bounds__(256)
void k_tile_raster(unsigned long long* __restrict__ fb, const hz_bigrec_t* __restrict__ bigrec, tl_bins_t tb, hz_params_t p)
{
    __shared__ unsigned long long tile[TL_H][TL_W];
    const unsigned int nbusy = tb.state[1];
    const bool stand_down = tb.state[0] != 0;                     /* some tile's list overflowed: k_big draws this round */
    for(unsigned int b = blockIdx.x; b < nbusy; b += gridDim.x)
    {
        const int t = (int)tb.busy[b];
        const unsigned int n = tb.cursor[t];
        __syncthreads();                                            /* (everybody has read the count; the tile of the turn before is stored) */
        if(threadIdx.x == 0) tb.cursor[t] = 0u;                     /* the list is empty again for the round that takes this queue set next */
        if(stand_down) continue;
        const size_t first = (size_t)t*TL_LIST;
        const int tx = t % tb.tiles_x, ty = t / tb.tiles_x;
        const int X0 = tx*TL_W, Y0 = ty*TL_H;                       /* tile origin: column relative to the sector, GL row */
        const int wcols = min(TL_W, p.SW - X0), hrows = min(TL_H, p.H - Y0);
        for(int k = threadIdx.x; k < TL_W*TL_H; k += 256)
        {
            const int r = k/TL_W, c = k%TL_W;
            tile[r][c] = (r < hrows && c < wcols) ? fb[(size_t)(Y0 + r)*p.SW + X0 + c] : HZ_FB_CLEAR;
        }
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        /* (the next record is requested before the current one is drawn: its latency hides behind the pixel work) */
        hz_bigrec_t rec_next = {};
        if((unsigned int)wave < n) rec_next = bigrec[__builtin_amdgcn_readfirstlane(tb.pairs[first + wave])];
        for(unsigned int k = wave; k < n; k += 4)
        {
            const hz_bigrec_t br = rec_next;
            if(k + 4 < n) rec_next = bigrec[__builtin_amdgcn_readfirstlane(tb.pairs[first + k + 4])];
            hz_tri_t tri;
            hz_planes_from_rec(tri, br.r);
            const uint32_t prim = br.r.prim;
            /* lane = row of the tile (the upper half of the wave has none) */
            const int row = Y0 + (lane & (TL_H-1));
            const int xlo = max(br.r.px0, X0 + p.col0), xhi = min(br.r.px0 + br.r.bw - 1, X0 + p.col0 + wcols - 1);
            int32_t x0 = xlo;
            const uint32_t span = hz_row_span(br.r.e, row, xlo, xhi, &x0);
            const uint32_t count = (lane < TL_H && row >= br.r.py0 && row < br.r.py0 + br.bh && row < Y0 + hrows) ? span : 0u;
            /* lane = pixel, as k_big */
            const uint32_t incl  = mr_scan(count, lane);
            const uint32_t excl  = incl - count;
            const uint32_t total = __shfl(incl, 63);
            for(uint32_t base = 0; base < total; base += 64)
            {
                const uint32_t q = base + lane;
                int own = 0;
                #pragma unroll
                for(int step=TL_H/2; step>=1; step>>=1)
                {
                    const uint32_t v = __shfl(excl, own + step);
                    if(v <= q) own += step;
                }
                const int px = __shfl(x0, own) + (int)(q - __shfl(excl, own));
                const int py = Y0 + own;
                if(q < total)
                {
                    uint32_t zi, r8;
                    if(hz_tri_fragment(&tri, px, py, &zi, &r8))
                        atomicMin(&tile[own][px - p.col0 - X0], hz_pack(zi, prim, r8));
                }
            }
        }
        __syncthreads();
        for(int k = threadIdx.x; k < TL_W*TL_H; k += 256)
        {
            const int r = k/TL_W, c = k%TL_W;
            if(r < hrows && c < wcols)
            {
                const unsigned long long v = tile[r][c];
                if(v != HZ_FB_CLEAR) fb[(size_t)(Y0 + r)*p.SW + X0 + c] = v;
            }
        }
        /* something was (or may have been) drawn into these rows' segment: the conversion has to look */
        if(threadIdx.x < hrows) p.touched[(size_t)(Y0 + threadIdx.x)*p.seg_stride + (X0 >> HZ_SEG_LOG2)] = 1;
    }
}
