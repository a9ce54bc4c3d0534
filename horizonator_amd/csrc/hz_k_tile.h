/* hz_k_tile.h - part of hz_kernels.hip (included there, after hz_k_scatter.h; one translation unit):
 * the large triangles of a round rasterised by screen tile, each tile's depth in LDS.
 *
 * BASELINE north_star's "tile-binned software rasteriser ... per-bin depth".  k_big issues one 64-bit atomic minimum per
 * FRAGMENT, and the chip does 173 G of those a second whatever their scope (tools/atomic_scope.hip,
 * profiles/r4_atomic_scope.json): where a round's large triangles lie hills behind hills - the first round of a zoomed
 * view: 131 M fragments for 54 M pixels - k_big runs at 90 % of that rate and three times as long as its arithmetic
 * needs (0.88 ms; 0.29 with plain stores, 0.25 with the fragments dropped: profiles/r4_tile_batches.txt).  Here the
 * triangles queued for k_big are binned by the 64 x 64 pixel tiles of the image they can cover; a workgroup takes a
 * BATCH of up to TL_BATCH triangles of one tile, draws them into 32 KB of LDS with LDS atomic minima (the tile starts
 * out cleared: nothing is read from the framebuffer) and merges what it drew into the framebuffer with one atomic
 * minimum per touched PIXEL.  Same fragments (hz_row_span, hz_tri_fragment), same keys, a minimum of minima: the
 * same bytes (the GPU parity suite under HZ_TILES=1).
 *
 * Round 3's version gave a whole tile to one workgroup, which owned it (plain stores, the tile read first): the tiles
 * along the horizon hold a thousand and more triangles, the others a dozen, and it lost to k_big everywhere (header
 * of that version: git history; 126 + 770 us against 854 for the first round above).  Batches make the units of work
 * alike and need no ownership - any kernel may draw into the same pixels meanwhile - at the price of atomics for the
 * merge; a tile's batches resolve the overdraw among their own triangles only.
 * Where it runs (draw_impl; HZ_TILES): the first round of views zoomed so far that a cell at that round's reach is
 * HZ_TILES_MIN_PX = 35 pixels wide - 71 + 522 us against k_big's 876 on the 45 degree view towards the east, seven such
 * views 10.5 -> 9.8 ms in sum (9.0 with the reach following the draws before).  Not second rounds (their large
 * triangles are few and scattered: +5..140 %), not the first round of a whole panorama (small triangles along the
 * horizon: 0.92 -> 1.14 ms).  What is left in k_tile_raster: 1.33 vector instructions per fragment - k_big's own
 * arithmetic, and a triangle's row spans once per tile it touches; the merge is 50 of its 507 us.
 *
 * k_tile_bin     one pass over the round's queue (a lane per triangle, the wave for those that touch many tiles):
 *                the triangle's number goes into the list of every tile of its box that an edge test does not
 *                rule out; the first entry of every batch of a list registers a unit of work (tile, batch).  A
 *                tile's list holds TL_LIST numbers, the unit list TL_UNITS_PER_TILE x tiles; one overflow sends
 *                the whole round back to k_big (flag) - no counting pass, no prefix sum
 * k_tile_raster  one workgroup per unit of work, the units dealt round the launch
 */
#pragma once


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
    if(at < tb.list_cap)
    {
        tb.pairs[(size_t)tile*TL_LIST + at] = rec;
        if(at % TL_BATCH == 0)                                  /* the first of a batch: a unit of work */
        {
            const unsigned int u = atomicAdd(&tb.state[1], 1u);
            if(u < tb.units_cap) tb.busy[u] = (unsigned int)tile | ((at / TL_BATCH) << 24);
            else tb.state[0] = 1u;
        }
    }
    else tb.state[0] = 1u;                                      /* (any number of writers, one value) */
}

__global__ __launch_bounds__(256)
void k_tile_bin(const hz_bigrec_t* __restrict__ bigrec, const hz_bigitem_t* __restrict__ bigitem,
                const unsigned int* __restrict__ big_counters, unsigned int bigrec_capacity, tl_bins_t tb, hz_params_t p)
{
    const unsigned int nitems = hz_queue_span(big_counters, p.qshards_log2);        /* item slots, not all of them in use: hz_types.h, HZ_QSHARDS */
    const int lane = threadIdx.x & 63;
    (void)bigrec_capacity;
    for(unsigned int base = (blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6))*64u; base < nitems; base += gridDim.x*(blockDim.x/64)*64u)
    {
        const unsigned int it = base + lane;
        bool mine = false;
        unsigned int rec = 0;
        int tx0 = 0, ty0 = 0, ntx = 1, nt = 0;
        if(it < nitems && hz_queue_item_valid(big_counters, it, p.qshards_log2))
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
    /* per wave, as in k_big: which row's span starts at pixel `base + k` of the current pass, each row's first column
     * minus its exclusive prefix */
    __shared__ uint32_t s_start[256/64][64];
    __shared__ int32_t  s_delta[256/64][64];
    if(tb.state[0] != 0) return;                                    /* an overflow: k_big draws this round */
    const unsigned int nunits = tb.state[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_start[wave][lane] = 0u;
    for(unsigned int b = blockIdx.x; b < nunits; b += gridDim.x)
    {
        const unsigned int unit = tb.busy[b];
        const int t = (int)(unit & 0xFFFFFFu);
        const unsigned int k0 = (unit >> 24)*TL_BATCH;
        const unsigned int n = min(tb.cursor[t], tb.list_cap);
        const unsigned int k1 = min(n, k0 + TL_BATCH);
        const size_t first = (size_t)t*TL_LIST;
        const int tx = t % tb.tiles_x, ty = t / tb.tiles_x;
        const int X0 = tx*TL_W, Y0 = ty*TL_H;                       /* tile origin: column relative to the sector, GL row */
        const int wcols = min(TL_W, p.SW - X0), hrows = min(TL_H, p.H - Y0);
        const int xt = X0 + p.col0;                                 /* image column of the tile's column 0 */
        __syncthreads();                                            /* (the tile of the turn before is merged) */
        {
            ulonglong2* t2 = (ulonglong2*)&tile[0][0];
            const ulonglong2 ones = { HZ_FB_CLEAR, HZ_FB_CLEAR };
            for(int k = threadIdx.x; k < TL_W*TL_H/2; k += 256) t2[k] = ones;
        }
        __syncthreads();
        /* (the next record is requested before the current one is drawn: its latency hides behind the pixel work) */
        hz_bigrec_t rec_next = {};
        if(k0 + wave < k1) rec_next = bigrec[__builtin_amdgcn_readfirstlane(tb.pairs[first + k0 + wave])];
        for(unsigned int k = k0 + wave; k < k1; k += 4)
        {
            const hz_bigrec_t br = rec_next;
            if(k + 4 < k1) rec_next = bigrec[__builtin_amdgcn_readfirstlane(tb.pairs[first + k + 4])];
            hz_tri_t tri;
            hz_planes_from_rec(tri, br.r);
            const uint32_t prim = br.r.prim;
            /* lane = row of the tile */
            const int row = Y0 + lane;
            const int xlo = max(br.r.px0, X0 + p.col0), xhi = min(br.r.px0 + br.r.bw - 1, X0 + p.col0 + wcols - 1);
            int32_t x0 = xlo;
            const uint32_t span = hz_row_span(br.r.e, row, xlo, xhi, &x0);
            const uint32_t count = (row >= br.r.py0 && row < br.r.py0 + br.bh && lane < hrows) ? span : 0u;
            /* lane = pixel, as k_big */
            const uint32_t incl  = mr_scan(count, lane);
            const uint32_t excl  = incl - count;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            unsigned long long rows_left = __ballot(count > 0u);
            if(total >= (uint32_t)TL_ROW_MIN*(uint32_t)__popcll(rows_left))
            {
                /* long spans (a tile holds at most 64 pixels of a row): row by row, the lanes side by side */
                while(rows_left)
                {
                    const int r = (int)__builtin_ctzll(rows_left);
                    rows_left &= rows_left - 1ull;
                    const int rx0 = __builtin_amdgcn_readlane(x0, r);
                    const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)count, r);
                    if((uint32_t)lane < rc)
                    {
                        const int px = rx0 + lane, py = Y0 + r;
                        uint32_t zi, r8;
                        if(hz_tri_fragment(&tri, px, py, &zi, &r8)) atomicMin(&tile[r][px - xt], hz_pack(zi, prim, r8));
                    }
                }
                continue;
            }
            KB_LDS_ORDER();                                         /* (the reads of the triangle before) */
            s_delta[wave][lane] = x0 - (int32_t)excl;
            uint32_t carry = 0u;
            for(uint32_t base = 0; base < total; base += 64)
            {
                const uint32_t q = base + lane;
                if(count > 0u && excl - base < 64u) s_start[wave][excl - base] = (uint32_t)lane + 1u;
                KB_LDS_ORDER();
                uint32_t own1 = s_start[wave][lane];
                KB_LDS_ORDER();
                if(own1) s_start[wave][lane] = 0u;
                own1 = mr_scan_max(lane == 0 ? (own1 > carry ? own1 : carry) : own1);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)own1, 63);
                const int own = (int)own1 - 1;
                const int px = (int)q + s_delta[wave][own & 63];
                const int py = Y0 + own;
                if(q < total)
                {
                    uint32_t zi, r8;
                    if(hz_tri_fragment(&tri, px, py, &zi, &r8)) atomicMin(&tile[own & 63][px - xt], hz_pack(zi, prim, r8));
                }
            }
        }
        __syncthreads();
        /* what the batch drew, into the framebuffer: a minimum of minima */
        for(int k = threadIdx.x; k < TL_W*TL_H; k += 256)
        {
            const int r = k/TL_W, c = k%TL_W;
            const unsigned long long v = tile[r][c];
            if(v != HZ_FB_CLEAR) hz_fb_min<HZ_WHO_BIG>(fb, p, xt + c, Y0 + r, v);
        }
    }
}
