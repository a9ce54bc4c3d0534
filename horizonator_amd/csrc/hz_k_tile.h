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
 * wherever it was measured - 16000x4000: first round alone 0.39 -> 0.47 ms, second round's queue kernels 0.08 ->
 * 0.55, pipelined 1.00 -> 1.25 ms per render; 40 km far clip 0.68 -> 0.88; a 45 degree view 2.8 -> 4.5 - and the
 * most a perfect one could gain is what plain stores instead of k_big's atomics gain (2 % of the headline render,
 * 19 % with the 40 km far clip: same file).  It costs k_big's instructions per fragment (the same row spans, the
 * same owner search), the spans again for every tile a triangle touches, four more launches per round and 32 KB
 * of framebuffer traffic per tile with anything in it.  Kept as the measured answer to "why not LDS depth bins".
 *
 * k_tile_bin<false>  per triangle (one wave each): the tiles of its box that an edge test does not rule out,
 *                    counted per tile
 * k_tile_scan        exclusive prefix over the tiles' counts; the grand total decides: more pairs than there
 *                    is room for -> the round falls back to k_big (flag)
 * k_tile_bin<true>   the same walk, writing the triangle's number into its tiles' lists
 * k_tile_raster      one workgroup per tile with at least one triangle
 */
#pragma once

#define TL_W 64
#define TL_H 32

/* what the tile kernels share: per queue set, allocated with the context */
struct tl_bins_t
{
    unsigned int* count;            /* [ntiles]: triangles per tile (all zero between rounds: the scan zeroes what it has read) */
    unsigned int* offset;           /* [ntiles + 1]: where a tile's list starts in `pairs` (the scan's result)          */
    unsigned int* cursor;           /* [ntiles]: fill positions (zeroed by the scan)                                    */
    unsigned int* pairs;            /* [capacity]: record numbers, tile by tile                                         */
    unsigned int* state;            /* [0] total pairs [1] 1 = binned, k_tile_raster draws; 0 = did not fit, k_big does  */
    unsigned int  capacity;
    int           tiles_x, tiles_y;
};

/* can edge m have a covered pixel centre inside the tile [x0,x1] x [y0,y1]?  (its function g + dx*py - dy*px is
 * linear: the maximum over the tile is at a corner) */
__device__ static inline bool tl_edge_reaches(const hz_edges_t& e, int m, int x0, int x1, int y0, int y1)
{
    const int64_t dx = e.dx[m], ndy = e.ndy[m];
    const int64_t best = hz_edges_g(&e, m) + dx*(int64_t)(dx >= 0 ? y1 : y0) + ndy*(int64_t)(ndy >= 0 ? x1 : x0);
    return best >= 0;
}

/* number of queued records this round (items and records beyond the first overflow were drawn by their producer:
 * the records below min(counters[0], capacity) exist, but only those whose items all fitted are complete - k_big's
 * rule is by ITEM, the tiles' by RECORD: a record is binned iff its last item is valid) */
template<bool FILL>
__global__ __launch_bounds__(256)
void k_tile_bin(const hz_bigrec_t* __restrict__ bigrec, const hz_bigitem_t* __restrict__ bigitem,
                const unsigned int* __restrict__ big_counters, unsigned int bigrec_capacity, tl_bins_t tb, hz_params_t p)
{
    const unsigned int nitems = min(big_counters[1], ~big_counters[2]);
    const int lane = threadIdx.x & 63;
    const unsigned int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6));
    const unsigned int nwaves = gridDim.x*(blockDim.x/64);
    if(FILL && tb.state[1] == 0) return;
    /* the records are reached through their FIRST item (chunk 0): every valid record has one, no record twice */
    for(unsigned int it = wave_global; it < nitems; it += nwaves)
    {
        const hz_bigitem_t item = bigitem[it];
        if(item.chunk != 0) continue;
        const hz_bigrec_t& br = bigrec[item.rec];
        const int px0 = br.r.px0 - p.col0, py0 = br.r.py0, bw = br.r.bw, bh = br.bh;       /* columns relative to the sector */
        /* (a record whose later items did not fit is drawn by k_big's producer-side fallback for the missing rows only -
         * no: the producer draws the WHOLE triangle when its reservation fails, and then writes no item at all) */
        const int tx0 = px0/TL_W, tx1 = (px0 + bw - 1)/TL_W, ty0 = py0/TL_H, ty1 = (py0 + bh - 1)/TL_H;
        const int ntx = tx1 - tx0 + 1, nt = ntx*(ty1 - ty0 + 1);
        for(int k = lane; k < nt; k += 64)
        {
            const int ty = ty0 + k/ntx, tx = tx0 + k%ntx;
            const int x0 = max(tx*TL_W, px0) + p.col0, x1 = min(tx*TL_W + TL_W-1, px0 + bw - 1) + p.col0;
            const int y0 = max(ty*TL_H, py0), y1 = min(ty*TL_H + TL_H-1, py0 + bh - 1);
            if(!(tl_edge_reaches(br.r.e, 0, x0, x1, y0, y1) && tl_edge_reaches(br.r.e, 1, x0, x1, y0, y1) && tl_edge_reaches(br.r.e, 2, x0, x1, y0, y1)))
                continue;
            const int tile = ty*tb.tiles_x + tx;
            if(!FILL) atomicAdd(&tb.count[tile], 1u);
            else
            {
                const unsigned int at = tb.offset[tile] + atomicAdd(&tb.cursor[tile], 1u);
                tb.pairs[at] = item.rec;
            }
        }
    }
    (void)bigrec_capacity;
}

/* offset[t] <- sum of count[0..t), offset[ntiles] <- the total; counts and cursors zeroed; state[1] <- does it fit */
__global__ __launch_bounds__(1024)
void k_tile_scan(tl_bins_t tb)
{
    __shared__ unsigned int part[1024];
    const int n = tb.tiles_x*tb.tiles_y;
    const int per = (n + 1023)/1024;
    const int lo = min((int)threadIdx.x*per, n), hi = min(lo + per, n);
    unsigned int sum = 0;
    for(int k=lo; k<hi; k++) sum += tb.count[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    /* (1024 partial sums: a plain scan by one wave's worth of work is enough) */
    if(threadIdx.x == 0)
    {
        unsigned int run = 0;
        for(int k=0; k<1024; k++) { const unsigned int v = part[k]; part[k] = run; run += v; }
        tb.state[0] = run;
        tb.state[1] = run <= tb.capacity ? 1u : 0u;
        tb.offset[n] = run;
    }
    __syncthreads();
    unsigned int run = part[threadIdx.x];
    for(int k=lo; k<hi; k++) { const unsigned int v = tb.count[k]; tb.offset[k] = run; run += v; tb.count[k] = 0u; tb.cursor[k] = 0u; }
}

__global__ __launch_bounds__(256)
void k_tile_raster(unsigned long long* __restrict__ fb, const hz_bigrec_t* __restrict__ bigrec, tl_bins_t tb, hz_params_t p)
{
    __shared__ unsigned long long tile[TL_H][TL_W];
    if(tb.state[1] == 0) return;                                /* too many pairs: k_big draws this round */
    const int t = blockIdx.x;
    const unsigned int first = tb.offset[t], n = tb.offset[t+1] - first;
    if(n == 0) return;
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
    for(unsigned int k = wave; k < n; k += 4)
    {
        const unsigned int ri = __builtin_amdgcn_readfirstlane(tb.pairs[first + k]);
        const hz_bigrec_t& br = bigrec[ri];
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
