/* hz_k_scatter.h - part of hz_kernels.hip (included there, in this order; one translation unit):
 * k_scatter (first design, kept as the second rasteriser), wave prefix sum, exact span division, k_big. */
#pragma once

/* ------------------------------------------------------------------------ */
/* scatter rasteriser                                                        */
/*
 * block = 64 x 4 DEM cells (one wave = one 128-byte row segment of the mosaic)
 *   phase 0  the block's 65 x 5 vertices are transformed once into LDS (2-D
 *            staging of the (i,j) (i+1,j) (i,j+1) (i+1,j+1) neighbourhood)
 *   phase 1a thread = cell: the two triangles of the cell (reference
 *            horizonator-lib.c:500-506) go through every pixel-free rejection
 *            (discard rule, guard band, back face, empty pixel box, depth
 *            range); ~78% of all triangles end here.  Survivors are compacted
 *            into an LDS list with wave ballots.
 *   phase 1b thread = surviving triangle: attribute planes; boxes above
 *            HZ_INLINE_MAX_PIX pixels go to the HBM queue of k_big
 *   phase 2  thread = one pixel centre of one survivor's box, found through a
 *            block-wide prefix sum of the box sizes: every lane tests a pixel,
 *            whatever the mix of box sizes (a per-triangle pixel loop ran at
 *            ~15% lane utilisation here)
 */

#define SC_REC_STRIDE 23

static_assert(sizeof(hz_rec_t) == SC_REC_STRIDE*4, "record layout");

__global__ __launch_bounds__(SC_THREADS)
void k_scatter(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb,
               mr_queue_t q, hz_params_t p)
{
    hz_bigrec_t* const bigrec = q.bigrec;
    hz_bigitem_t* const bigitem = q.bigitem;
    unsigned int* const big_counters = q.counters;
    const unsigned int bigrec_capacity = q.bigrec_capacity, bigitem_capacity = q.bigitem_capacity;
    __shared__ float   s_xn [SC_VY][SC_VX];
    __shared__ float   s_fx [SC_VY][SC_VX];
    __shared__ float   s_fy [SC_VY][SC_VX];
    __shared__ float   s_zw [SC_VY][SC_VX];
    __shared__ float   s_red[SC_VY][SC_VX];
    __shared__ int32_t s_xs [SC_VY][SC_VX];
    __shared__ int32_t s_ys [SC_VY][SC_VX];
    __shared__ uint32_t s_cm[SC_VY][SC_VX];
    __shared__ unsigned short s_cand[2*SC_THREADS];
    __shared__ uint32_t s_rec[SC_THREADS*SC_REC_STRIDE];
    __shared__ uint32_t s_prefix[SC_THREADS+1];
    __shared__ uint32_t s_wavesum[SC_THREADS/64];
    __shared__ uint32_t s_ncand;

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i0   = blockIdx.x*SC_CX;
    const int j0   = blockIdx.y*SC_CY;

    /* ---- phase 0: vertices ------------------------------------------------ */
    if(tid == 0) s_ncand = 0;
    int some_not_near = 0, some_not_far = 0;
    for(int v = tid; v < SC_VX*SC_VY; v += SC_THREADS)
    {
        const int vy = v / SC_VX, vx = v - vy*SC_VX;
        const int i = i0 + vx, j = j0 + vy;
        if(i < p.N && j < p.N)
        {
            const hz_wvert_t w = hz_vertex_at(p, mosaic, i, j);
            s_xn [vy][vx] = w.xn;  s_fx [vy][vx] = w.wx;  s_fy[vy][vx] = w.wy;
            s_zw [vy][vx] = w.zw;  s_red[vy][vx] = w.red;
            s_xs [vy][vx] = w.xs;  s_ys [vy][vx] = w.ys;  s_cm[vy][vx] = w.cmask;
            some_not_near |= !(w.zw < 0.f);
            some_not_far  |= !(w.zw > 1.f);
        }
    }
    /* block-wide early out: every vertex in front of the near sphere, or
     * every vertex beyond the far one.  hz_tri_cull() drops exactly those
     * triangles anyway (with the default zfar = 40 km most of a large mosaic
     * goes this way). */
    some_not_near = __syncthreads_or(some_not_near);
    some_not_far  = __syncthreads_or(some_not_far);
    if(!some_not_near || !some_not_far) return;

    /* ---- phase 1a: pixel-free rejection, compaction ----------------------- */
    {
        const int cx = lane, cy = wave;
        const int i = i0 + cx, j = j0 + cy;
        int keep0 = 0, keep1 = 0;
        if(i < p.N-1 && j < p.N-1)
        {
            #define LDV(vy,vx) hz_wvert_t{ s_xn[vy][vx], s_fx[vy][vx], s_fy[vy][vx], s_zw[vy][vx], s_red[vy][vx], s_xs[vy][vx], s_ys[vy][vx], s_cm[vy][vx] }
            const hz_wvert_t v00 = LDV(cy,   cx  );
            const hz_wvert_t v10 = LDV(cy,   cx+1);
            const hz_wvert_t v01 = LDV(cy+1, cx  );
            const hz_wvert_t v11 = LDV(cy+1, cx+1);
            hz_box_t box;
            keep0 = hz_tri_cull(&box, &v00, &v11, &v01, p.col0, p.col1-1, 0, p.H-1);
            keep1 = hz_tri_cull(&box, &v00, &v10, &v11, p.col0, p.col1-1, 0, p.H-1);
        }
        /* triangles that cross the view volume's planes go through k_clip */
        {
            const uint32_t prim0 = (uint32_t)(((size_t)j*(p.N-1) + i)*2);
            hz_queue_clip(q, keep0 == HZ_TRI_CLIP, prim0,   lane);
            hz_queue_clip(q, keep1 == HZ_TRI_CLIP, prim0+1, lane);
            keep0 = keep0 == HZ_TRI_DRAW; keep1 = keep1 == HZ_TRI_DRAW;
        }
        const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
        const unsigned int n0 = __popcll(m0), n1 = __popcll(m1);
        unsigned int base = 0;
        if(lane == 0 && n0+n1) base = atomicAdd(&s_ncand, n0+n1);
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned short id = (unsigned short)((cy << 7) | (cx << 1));
        if(keep0) s_cand[base      + __popcll(m0 & below)] = id;
        if(keep1) s_cand[base + n0 + __popcll(m1 & below)] = id | 1;
    }
    __syncthreads();
    const unsigned int ncand = s_ncand;

    for(unsigned int batch = 0; batch < ncand; batch += SC_THREADS)
    {
        /* ---- phase 1b: attribute planes, one thread per survivor ----------- */
        uint32_t npix = 0;
        const unsigned int k = batch + tid;
        if(k < ncand)
        {
            const unsigned int id = s_cand[k];
            const int t = id & 1, cx = (id >> 1) & 63, cy = id >> 7;
            const hz_wvert_t a = LDV(cy, cx);
            const hz_wvert_t b = t == 0 ? LDV(cy+1, cx+1) : LDV(cy,   cx+1);
            const hz_wvert_t c = t == 0 ? LDV(cy+1, cx  ) : LDV(cy+1, cx+1);
            hz_box_t box;
            hz_tri_cull_window(&box, &a, &b, &c, p.col0, p.col1-1, 0, p.H-1);     /* known to pass: recomputes the box */
            hz_tri_t tri;
            hz_tri_planes(&tri, &a, &b, &c);
            hz_rec_t r;
            hz_rec_from_tri(r, tri);
            r.px0 = box.px0; r.py0 = box.py0; r.bw = box.px1 - box.px0 + 1;
            r.inv_bw = 1.0f / (float)r.bw;
            r.prim = (uint32_t)(((size_t)(j0+cy)*(p.N-1) + (i0+cx))*2 + t);
            const int bh = box.py1 - box.py0 + 1;
            const long long n = (long long)r.bw*bh;
            if(n <= HZ_INLINE_MAX_PIX)
            {
                npix = (uint32_t)n;
                uint32_t* dst = &s_rec[tid*SC_REC_STRIDE];
                const uint32_t* src = (const uint32_t*)&r;
                #pragma unroll
                for(int q=0; q<SC_REC_STRIDE; q++) dst[q] = src[q];
            }
            else
            {
                /* large: hand over to k_big, 64 tiles per work item */
                const unsigned int chunks = hz_big_chunks(r.bw, bh);
                uint32_t ri = 0, ii = 0;
                mr_queue_t qq = {};
                qq.counters = big_counters; qq.bigrec_capacity = bigrec_capacity; qq.bigitem_capacity = bigitem_capacity;
                const int sl = p.qshards_log2, shard = (int)(blockIdx.x & ((1u << sl) - 1u));
                const bool queued = hz_queue_reserve(qq, shard, sl, 1u, chunks, &ri, &ii);
                if(queued)
                {
                    const uint32_t rslot = HZ_QSLOT(ri, shard, sl);
                    bigrec[rslot].r = r; bigrec[rslot].bh = bh;
                    for(unsigned int c2=0; c2<chunks; c2++) { const uint32_t g = HZ_QSLOT(ii + c2, shard, sl); bigitem[g].rec = rslot; bigitem[g].chunk = c2; }
                }
                else
                {
                    /* queue full (never seen; the capacities are sized for 32k-wide
                     * panoramas): rasterise here, slowly but correctly */
                    for(int py = box.py0; py <= box.py1; py++)
                        for(int px = box.px0; px <= box.px1; px++)
                            hz_emit(fb, p, tri, r.prim, px, py);
                }
            }
        }
        #undef LDV

        /* ---- exclusive prefix sum of the box sizes over the block ---------- */
        uint32_t incl = npix;
        #pragma unroll
        for(int d=1; d<64; d<<=1)
        {
            const uint32_t up = __shfl_up(incl, d);
            if(lane >= d) incl += up;
        }
        if(lane == 63) s_wavesum[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
        #pragma unroll
        for(int w=0; w<SC_THREADS/64; w++)
        {
            const uint32_t ws = s_wavesum[w];
            if(w < wave) wave_base += ws;
            total += ws;
        }
        s_prefix[tid] = wave_base + incl - npix;
        if(tid == 0) s_prefix[SC_THREADS] = total;
        __syncthreads();

        /* ---- phase 2: one thread per pixel centre --------------------------- */
        for(uint32_t it = tid; it < total; it += SC_THREADS)
        {
            /* record holding item `it`: last k with prefix[k] <= it */
            int lo = 0, hi = SC_THREADS;
            #pragma unroll
            for(int step=0; step<9; step++)       /* the range [lo,hi) of 256 shrinks to empty in 9 halvings */
            {
                const int mid = (lo + hi) >> 1;
                if(s_prefix[mid+1] <= it) lo = mid+1; else hi = mid;
            }
            const uint32_t* src = &s_rec[lo*SC_REC_STRIDE];
            hz_rec_t r;
            uint32_t* dst = (uint32_t*)&r;
            #pragma unroll
            for(int q=0; q<SC_REC_STRIDE; q++) dst[q] = src[q];
            const uint32_t local = it - s_prefix[lo];
            const int ry = (int)(((float)local + 0.5f) * r.inv_bw);
            const int rx = (int)local - ry*r.bw;
            hz_emit_rec<true>(fb, p, r, r.px0 + rx, r.py0 + ry);
        }
        __syncthreads();
    }
}

/* 1/d for 1 <= d < 2^31 to within 2^-50 relative: v_rcp_f64's estimate and two
 * Newton steps (each squares the relative error; the estimate is good to 2^-20
 * at the very least).  Not the correctly rounded quotient - hz_floor_div() does
 * not need it - and a third of the instructions of 1.0/d. */
__device__ static inline double hz_rcp_f64(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}

/* floor(n / d) for d > 0: a double-precision estimate (r = hz_rcp_f64(d),
 * computed by the caller once per edge), then the remainder decides - exactly.
 * |n| < 2^55, d < 2^31; results beyond +-2^30 come back clamped (the caller
 * only compares them with pixel columns).  Inside that range the estimate
 * n*r differs from n/d by less than 2^30 * (2^-50 + 2 * 2^-53) < 2^-19, so its
 * floor is the true one or a neighbour of it: one step each way. */
__device__ static inline int32_t hz_floor_div(int64_t n, int32_t d, double r)
{
    double qd = __builtin_floor((double)n * r);
    qd = qd < -1073741824.0 ? -1073741824.0 : (qd > 1073741824.0 ? 1073741824.0 : qd);
    int32_t q = (int32_t)qd;
    int64_t rem = n - (int64_t)q*(int64_t)d;
    if(rem < 0)  { q--; rem += d; }
    if(rem >= d) { q++; rem -= d; }
    return q;
}

/* The covered pixel centres of row `row` of a set-up triangle: a span [x0, x1] inside the columns [xlo, xhi] given -
 * each edge function is linear in px, so each edge bounds the span from one side, at a column that an integer division
 * gives exactly (the ownership of zeros included).  Returns the number of pixels (0: none). */
__device__ static inline uint32_t hz_row_span(const hz_edges_t& e, int row, int32_t xlo, int32_t xhi, int32_t* first)
{
    int32_t x0 = xlo, x1 = xhi;
    bool any = true;
    #pragma unroll
    for(int m=0; m<3; m++)
    {
        /* edge m covers px in this row iff g + dx*row - dy*px >= 0 (hz_edges_t), g and
         * the deltas wave-uniform: a bound on px from one side, by an exact division */
        const int32_t dx = e.dx[m], dy = -e.ndy[m];
        const int64_t n8 = hz_edges_g(&e, m) + (int64_t)dx*(int64_t)row;
        if(dy > 0)
        {
            /* dy*px <= n8  <=>  px <= floor(n8 / dy) */
            const int32_t q = hz_floor_div(n8, dy, hz_rcp_f64((double)dy));
            x1 = x1 < q ? x1 : q;
        }
        else if(dy < 0)
        {
            /* |dy|*px >= -n8  <=>  px >= ceil(-n8 / |dy|) = -floor(n8 / |dy|) */
            const int32_t q = hz_floor_div(n8, -dy, hz_rcp_f64((double)(-dy)));
            x0 = x0 > -q ? x0 : -q;
        }
        else if(n8 < 0) any = false;                /* a horizontal edge: the whole row is on one side */
    }
    *first = x0;
    return (any && x1 >= x0) ? (uint32_t)(x1 - x0 + 1) : 0u;
}

#ifndef KB_ROW_MIN
#define KB_ROW_MIN 48                   /* k_big: average pixels per non-empty row from which a chunk is drawn row by row */
#endif
/* large triangles: one wave per work item = 64 pixel rows of a queued triangle.
 * Lane = row: the covered pixel centres of a row are a span [x0, x1] - each
 * edge function is linear in px, so each edge bounds the span from one side, at
 * a column that an integer division gives exactly (the ownership of zeros
 * included).  Then lane = pixel: the spans of the 64 rows are laid end to end
 * (wave prefix sum) and every lane takes one covered pixel per pass, whatever
 * the shape of the triangle - the long thin slivers next to the viewer cover a
 * quarter of their boxes. */
#define KB_LDS_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while(0)
/* SHARDS: the queue was filled through HZ_QSHARDS counters (zoomed views: p.qshards_log2, hz_types.h); else through one, and
 * this is the loop it had before there were shards (the general one cost a render of a series 1 %) */
template<bool SHARDS>
__global__ __launch_bounds__(256)
void k_big(unsigned long long* __restrict__ fb,
           const hz_bigrec_t* __restrict__ bigrec, const hz_bigitem_t* __restrict__ bigitem,
           const unsigned int* __restrict__ big_counters,
           unsigned int bigrec_capacity, unsigned int bigitem_capacity, hz_params_t p, const unsigned int* tile_state,
           unsigned int* report)
{
    /* per wave: which row's span starts at pixel `base + k` of the current pass (row + 1, 0: none), and each row's
     * first column minus its exclusive prefix (a pixel's column = its number + that).  A wave's own LDS traffic is
     * served in program order; KB_LDS_ORDER keeps the compiler from moving or merging it across the points where one
     * lane reads what another wrote - no barrier, no wait for the atomics in flight.  (Not `volatile` through a
     * pointer: that loses the address space - flat loads and stores with a wait for every outstanding memory
     * operation behind each, 660 -> 750 us for the first round's launch beside a marching kernel.) */
    __shared__ uint32_t s_start[256/64][64];
    __shared__ int32_t  s_delta[256/64][64];
    /* (report: pinned host memory - what this round queued, for the host's choice of the next first round's reach: hz_draw.cpp, adapt) */
    if(report && blockIdx.x == 0 && threadIdx.x == 0) { unsigned int records, items; hz_queue_totals(big_counters, &records, &items); report[0] = records; report[1] = items; }
    const int wv = threadIdx.x >> 6;
    s_start[wv][threadIdx.x & 63] = 0u;
    KB_LDS_ORDER();
    /* (tile_state: the round's triangles were binned and drawn by screen tile - hz_k_tile.h - unless there were too many) */
    if(tile_state && tile_state[0] == 0) return;
    /* items at and beyond the first overflow were rasterised inline by their producer */
    /* (item slots [0, nitems): a slot is in use if the shard it belongs to got that far - hz_types.h, HZ_QSLOT) */
    const int sl = SHARDS ? HZ_QSHARDS_LOG2 : 0;
    const unsigned int nitems = SHARDS ? hz_queue_span(big_counters, sl) : hz_queue_nitems_of(big_counters, 0);
    (void)bigrec_capacity; (void)bigitem_capacity;
    const int lane = threadIdx.x & 63;
    const unsigned int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6));
    const unsigned int nwaves = gridDim.x*(blockDim.x/64);
    /* item and record come through the scalar cache (wave-uniform addresses);
     * the next item's are requested before the current one is rasterised, so
     * their latency hides behind the pixel work */
    hz_bigitem_t item_next = {};
    hz_bigrec_t  rec_next  = {};
    /* (how far each shard got: lane s holds shard s's count, an item's validity is one v_readlane - no memory access in the loop) */
    unsigned int shard_items = 0;
    if(SHARDS) shard_items = hz_queue_nitems_of(big_counters, lane & (HZ_QSHARDS-1));
    auto slot_in_use = [&](unsigned int g) -> bool
    {
        if(!SHARDS) return true;
        const unsigned int block = g >> HZ_QBLOCK_LOG2;
        const unsigned int l = ((block >> sl) << HZ_QBLOCK_LOG2) | (g & (HZ_QBLOCK-1));
        return l < (unsigned int)__builtin_amdgcn_readlane((int)shard_items, (int)(block & ((1u << sl) - 1u)));
    };
    bool valid_next = wave_global < nitems && slot_in_use(wave_global);
    if(valid_next) { item_next = bigitem[wave_global]; rec_next = bigrec[item_next.rec]; }
    for(unsigned int it = wave_global; it < nitems; it += nwaves)
    {
        const hz_bigitem_t item = item_next;
        const hz_bigrec_t  br   = rec_next;
        const bool valid = valid_next;
        valid_next = it + nwaves < nitems && slot_in_use(it + nwaves);
        if(valid_next) { item_next = bigitem[it + nwaves]; rec_next = bigrec[item_next.rec]; }
        if(!valid) continue;
        hz_tri_t tri;
        hz_planes_from_rec(tri, br.r);
        const int px0 = br.r.px0, py0 = br.r.py0, bw = br.r.bw, bh = br.bh;
        const uint32_t prim = br.r.prim;

        /* lane = row */
        const int rows_log2 = hz_big_rows_log2(bw);
        const int row_first = py0 + ((int)item.chunk << rows_log2);
        /* draws that keep coarse depth (second rounds of zoomed views, hz_k_hiz.h): nothing of these rows can win? */
        if(p.hiz && hiz_chunk_hidden(tri, p, px0, bw, row_first, min(1 << rows_log2, py0 + bh - row_first), lane)) continue;
        const int row = row_first + lane;
        int32_t x0 = px0;
        const uint32_t span = hz_row_span(br.r.e, row, px0, px0 + bw - 1, &x0);
        const uint32_t count = (lane < (1 << rows_log2) && row < py0 + bh) ? span : 0u;

        /* lane = pixel */
        const uint32_t incl  = mr_scan(count, lane);
        const uint32_t excl  = incl - count;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        /* Long spans (the triangles next to the viewer: hundreds of pixels a row): row by row, the lanes side by side
         * along the span - row and first column are scalars, no search for the row that owns a pixel (six dependent
         * ds_bpermute per pass below: 40 % of this kernel's instructions).  Worth it from an average of 48 pixels per
         * non-empty row on (a pass then has at most a quarter of its lanes idle at the spans' ends). */
        unsigned long long rows_left = __ballot(count > 0u);
        if(total >= (uint32_t)KB_ROW_MIN*(uint32_t)__popcll(rows_left))
        {
            while(rows_left)
            {
                const int r = (int)__builtin_ctzll(rows_left);
                rows_left &= rows_left - 1ull;
                const int rx0 = __builtin_amdgcn_readlane(x0, r);
                const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)count, r);
                const int py = row_first + r;
                for(uint32_t o = 0; o < rc; o += 64u)
                {
                    const uint32_t k = o + (uint32_t)lane;
                    if(k < rc)
                    {
                        const int px = rx0 + (int)k;
                        uint32_t zi, r8;
                        if(hz_tri_fragment(&tri, px, py, &zi, &r8))
                        {
                            const unsigned long long key = hz_pack(zi, prim, r8);
                            hz_fb_min<HZ_WHO_BIG>(fb, p, px, py, key);
                        }
                    }
                }
            }
            continue;
        }
        /* Which row holds pixel k?  The last non-empty row whose exclusive prefix is <= k.  (Round 3 searched for it:
         * six dependent ds_bpermute per pass, 40 % of the kernel's instructions.)  Every row whose span starts inside
         * the pass says so at its start's slot - distinct slots, spans of non-empty rows start at distinct pixels -,
         * every lane reads its slot, and a running maximum over the lanes (the rows come in rising order) carries the
         * latest start to the pixels behind it; the maximum of a pass carries over to the next. */
        KB_LDS_ORDER();                                 /* (the reads of the item before) */
        s_delta[wv][lane] = x0 - (int32_t)excl;
        uint32_t carry = 0u;
        for(uint32_t base = 0; base < total; base += 64)
        {
            const uint32_t k = base + lane;
            if(count > 0u && excl - base < 64u) s_start[wv][excl - base] = (uint32_t)lane + 1u;
            KB_LDS_ORDER();
            uint32_t own1 = s_start[wv][lane];
            KB_LDS_ORDER();
            if(own1) s_start[wv][lane] = 0u;            /* (clean for the next pass, the next item) */
            own1 = mr_scan_max(lane == 0 ? (own1 > carry ? own1 : carry) : own1);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)own1, 63);
            const int own = (int)own1 - 1;
            const int px = (int)k + s_delta[wv][own & 63];
            const int py = row_first + own;
            if(k < total)
            {
                uint32_t zi, r8;
                if(hz_tri_fragment(&tri, px, py, &zi, &r8))
                {
                    const unsigned long long key = hz_pack(zi, prim, r8);
                    /* (a look at the word before the atomic - leave it out where the fragment cannot win - lost here
                     * on every scene: rounds 3 and 4, removed in round 5) */
                    hz_fb_min<HZ_WHO_BIG>(fb, p, px, py, key);
                }
            }
        }
    }
}
