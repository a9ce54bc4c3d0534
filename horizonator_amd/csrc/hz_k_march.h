/* hz_k_march.h - part of hz_kernels.hip (included there, in this order; one translation unit):
 * the marching rasteriser: k_march, its flush, pixel spreading, k_mid. */
#pragma once

/* ------------------------------------------------------------------------ */
/* marching rasteriser                                                       */
/*
 * One wave walks one strip of the DEM, 63 cells wide and 4..64 cell rows long
 * (short near the viewer, where a cell covers many pixels and a wave would
 * otherwise carry the whole near field; see mr_zones_t), from south to north,
 * lane = grid column:
 *   - the east offset e(i) is computed once per strip, the elevation of the
 *     next row is in flight while the current row is transformed
 *   - each vertex is transformed once (64 vertices per row for 63 cells);
 *     a cell takes its right-hand vertices from the neighbouring lane
 *     (cross-lane reads, no LDS staging, no workgroup barrier anywhere)
 *   - triangles that survive every pixel-free rejection are appended to a
 *     per-wave LDS ring (ballot compaction); whenever 64 are waiting they are
 *     set up one per lane and their pixel centres are spread over the lanes
 *     through a wave prefix sum, as in k_scatter
 * Workgroup = one wave, so nothing ever waits for another wave.
 */

#define MR_CAP    128               /* pending-triangle ids, ring (power of two)    */
#define MR_RSLOTS 4                 /* vertex rows kept in LDS (power of two)       */
#define MR_FIELDS 6                 /* wx wy zw red xs ys                            */

/* LDS of one wave: the last MR_RSLOTS vertex rows (structure of arrays: one
 * conflict-free 256-byte store per field and row) and a ring of ids of the
 * triangles waiting for set-up.  id = (cell row - first row of the segment)<<7
 * | lane<<1 | t.  Only 6 + 2 LDS stores per row of 126 triangles. */
struct mr_lds_t
{
    uint32_t rows[MR_RSLOTS][MR_FIELDS][64];
    uint32_t ids[MR_CAP];
    uint32_t clip[64];              /* ids on their way to k_clip's queue (mr_clip_note) */
    uint32_t nclip;                 /* ... how many: kept here, not in a register (one scalar less to carry through the marching loop) */
};

/* Triangles that cross a plane of the view volume: their ids go to k_clip (as hz_queue_clip), but collected here
 * first, 64 to one atomic.  hz_queue_clip's one atomic per call was two returning atomics per row for every strip
 * along the image's border, on the address every wave of the draw appends to - in a zoomed view, whose border runs
 * through hundreds of strips, the waves spent most of their time queueing for it (tools/wave_timing.py: a strip
 * with two flushes took 700 us). */
__device__ static inline void mr_clip_flush(mr_lds_t& L, const mr_queue_t& q, int lane)
{
    const unsigned int nclip = (unsigned int)__builtin_amdgcn_readfirstlane((int)L.nclip);
    if(!nclip) return;
    uint32_t base = 0;
    if(lane == 0) base = atomicAdd(&q.counters[4], nclip);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if((unsigned int)lane < nclip)
    {
        const uint32_t at = base + (uint32_t)lane;
        if(at < q.clip_capacity) q.clip[at] = L.clip[lane];     /* ids that do not fit are counted, not stored: k_clip then rescans */
    }
    __syncthreads();                /* (one wave: the reads above before the writes that follow) */
    if(lane == 0) L.nclip = 0;
    __syncthreads();
}
__device__ static inline void mr_clip_note(mr_lds_t& L, const mr_queue_t& q, bool want, uint32_t prim, int lane)
{
    const unsigned long long m = __ballot(want);
    if(!m) return;
    const unsigned int n = (unsigned int)__popcll(m);
    unsigned int nclip = (unsigned int)__builtin_amdgcn_readfirstlane((int)L.nclip);
    if(nclip + n > 64u) { mr_clip_flush(L, q, lane); nclip = 0; }
    if(want) L.clip[nclip + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = prim;
    __syncthreads();
    if(lane == 0) L.nclip = nclip + n;
    __syncthreads();
}

__device__ static inline void mr_store_row(mr_lds_t& L, int slot, int lane, const hz_wvert_t& v)
{
    L.rows[slot][0][lane] = __float_as_uint(v.wx);  L.rows[slot][1][lane] = __float_as_uint(v.wy);
    L.rows[slot][2][lane] = __float_as_uint(v.zw);  L.rows[slot][3][lane] = __float_as_uint(v.red);
    L.rows[slot][4][lane] = (uint32_t)v.xs;         L.rows[slot][5][lane] = (uint32_t)v.ys;
}
__device__ static inline hz_wvert_t mr_load_vert(const mr_lds_t& L, int slot, int lane)
{
    hz_wvert_t v;
    v.xn  = 0.f;
    v.wx  = __uint_as_float(L.rows[slot][0][lane]); v.wy  = __uint_as_float(L.rows[slot][1][lane]);
    v.zw  = __uint_as_float(L.rows[slot][2][lane]); v.red = __uint_as_float(L.rows[slot][3][lane]);
    v.xs  = (int32_t)L.rows[slot][4][lane];         v.ys  = (int32_t)L.rows[slot][5][lane];
    v.cmask = 0;
    return v;
}

/* ... without the colour */
__device__ static inline hz_wvert_t mr_load_vert_pos(const mr_lds_t& L, int slot, int lane)
{
    hz_wvert_t v;
    v.xn  = 0.f; v.red = 0.f; v.cmask = 0;
    v.wx  = __uint_as_float(L.rows[slot][0][lane]); v.wy  = __uint_as_float(L.rows[slot][1][lane]);
    v.zw  = __uint_as_float(L.rows[slot][2][lane]);
    v.xs  = (int32_t)L.rows[slot][4][lane];         v.ys  = (int32_t)L.rows[slot][5][lane];
    return v;
}


/* value held by the lane one to the east (lane+1): DPP wave shift, one VALU
 * move instead of an LDS-crossbar permute (gfx9 family: wave_shl:1).  Lane 63,
 * which has no source and no cell, reads 0 (bound_ctrl: no `old` operand to set up). */
__device__ static inline int32_t mr_from_east(int32_t v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}
__device__ static inline float mr_from_east(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}



__device__ static inline int32_t hz_imin(int32_t a, int32_t b) { return a < b ? a : b; }
__device__ static inline int32_t hz_imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* what k_march keeps of a vertex row for the cells between it and the next (mr_rowstate_of) */
struct mr_rowstate_t
{
    float    xn;                        /* NDC x (discard rule)                              */
    int32_t  xs, ys;                    /* snapped position                                  */
    uint32_t cmask;                     /* clip mask (0 in rows that are wholly inside)      */
    uint32_t c, f1;                     /* first pixel column | row << 16 at or beyond the vertex; 1 + the last up to it; clipped to the scissor */
    uint32_t e_c, e_f1;                 /* those of the eastern neighbour */
    uint32_t h_c, h_f1;                 /* ... and over the vertex and its eastern neighbour */
    int32_t  h_dx, h_dy;                /* eastern neighbour's snapped position minus this vertex's */
};

/* window position of a transformed vertex as hz_to_window() computes it, and two
 * facts about it that decide how the cells of its row are culled: inside the view
 * volume (clip mask 0: with xn + 1 < 0 <=> xn < -1 for every float, "inside" is
 * |x|,|y|,|z| <= 1; fmaxf skips a NaN, as the six comparisons of hz_clip_mask() do:
 * all false) and inside the guard band (a NaN is outside) */
__device__ static inline hz_wvert_t mr_window(const hz_vertex_t& vtx, const hz_params_t& p, bool* in_volume, bool* in_guard)
{
    hz_wvert_t cur;
    cur.xn  = vtx.x;
    cur.wx  = vtx.x*p.halfW + p.halfW;
    cur.wy  = vtx.y*p.halfH + p.halfH;
    cur.zw  = vtx.z*0.5f + 0.5f;
    cur.red = vtx.red;
    const float fxw = cur.wx - 0.5f, fyw = cur.wy - 0.5f;
    *in_guard  = hz_abs(fxw) <= HZ_GUARD_PX && hz_abs(fyw) <= HZ_GUARD_PX;
    *in_volume = __builtin_fmaxf(__builtin_fmaxf(hz_abs(vtx.x), hz_abs(vtx.y)), hz_abs(vtx.z)) <= 1.0f;
    cur.xs = (int32_t)hz_roundeven(fxw*256.f);
    cur.ys = (int32_t)hz_roundeven(fyw*256.f);
    cur.cmask = 0;
    return cur;
}
/* ... what a row that is not "simple" (some vertex outside the volume or the band) adds */
__device__ static inline void mr_window_flags(hz_wvert_t& cur, const hz_vertex_t& vtx, bool in_guard)
{
    cur.cmask = hz_clip_mask(vtx.x, vtx.y, vtx.z);
    if(!in_guard) { cur.xs = HZ_OUTSIDE_GUARD; cur.ys = 0; }
}

/* What a row keeps of its vertices for the cull of the cells above and below it.
 * Pixel columns/rows of a vertex: a triangle's pixel box is the min of its
 * vertices' first and the max of their last (hz_tri_box: the shifts are monotone),
 * clipped to the scissor here already (max and min distribute over it).  Columns
 * and rows travel as pairs of unsigned 16-bit halves, x low and y high, so that
 * every min / max below is one packed instruction for both axes: `c` = first pixel
 * at or beyond the vertex, `f1` = 1 + the last pixel up to it (the +1 keeps a
 * vertex left of / below the scissor, whose last pixel is -1, representable).
 * Both lie in [0, max(W,H)] - images of up to 65535 pixels a side: make_params - in
 * rows whose vertices are all inside the view volume (window position in [0,W] x
 * [0,H]), the only rows that use them. */
__device__ static inline uint32_t mr_pk_min(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ static inline uint32_t mr_pk_max(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
/* both halves of a below those of b?  (two 16-bit compares, the upper one through SDWA) */
__device__ static inline bool mr_pk_lt(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return x.x < y.x && x.y < y.y;
}
__device__ static inline uint32_t mr_from_east_u(uint32_t v) { return (uint32_t)mr_from_east((int32_t)v); }

__device__ static inline mr_rowstate_t mr_rowstate_of(const hz_wvert_t& cur, const hz_params_t& p)
{
    mr_rowstate_t now;
    now.xn = cur.xn; now.xs = cur.xs; now.ys = cur.ys; now.cmask = cur.cmask;
    /* (rows with a vertex outside the view volume do not use these: there a half may overflow into the other) */
    const uint32_t c_x  = (uint32_t)((cur.xs + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS), c_y  = (uint32_t)((cur.ys + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS);
    const uint32_t f1_x = (uint32_t)((cur.xs + HZ_SUBPIXEL_ONE) >> HZ_SUBPIXEL_BITS),     f1_y = (uint32_t)((cur.ys + HZ_SUBPIXEL_ONE) >> HZ_SUBPIXEL_BITS);
    now.c  = mr_pk_max(c_x  | (c_y  << 16), (uint32_t)p.col0);                        /* (row 0 is the scissor's first) */
    now.f1 = mr_pk_min(f1_x | (f1_y << 16), (uint32_t)p.col1 | ((uint32_t)p.H << 16));
    /* the same over this vertex and its eastern neighbour (the neighbour's value
     * comes through a DPP wave shift), and the step to that neighbour */
    now.e_c  = mr_from_east_u(now.c);
    now.e_f1 = mr_from_east_u(now.f1);
    now.h_c  = mr_pk_min(now.e_c,  now.c);
    now.h_f1 = mr_pk_max(now.e_f1, now.f1);
    now.h_dx = (int32_t)((uint32_t)mr_from_east(cur.xs) - (uint32_t)cur.xs);     /* (wraps for guard-band markers; unused then) */
    now.h_dy = (int32_t)((uint32_t)mr_from_east(cur.ys) - (uint32_t)cur.ys);
    return now;
}

/* The cull of the two triangles of cell (i, j-1) - v00 = prev, v01 = now, v10 / v11 =
 * those of the lane to the east; t0 = (v00,v11,v01), t1 = (v00,v10,v11), reference
 * horizonator-lib.c:500-506 - for rows whose vertices all lie inside the view volume
 * and the guard band.  Returns false if the row has to go the long way
 * (hz_tri_cull), which is decided for the whole wave; else keep0 / keep1.
 *
 * Reference geometry.glsl:21-27 (a triangle spanning more than 0.5 in NDC x = a
 * quarter of the image is dropped) cannot apply to a cell whose vertices are all
 * within quad_max_dx of v00 in snapped x: any two of them are then less than a
 * quarter of the image minus two pixels apart, and window x follows NDC x to
 * within a hundredth of a pixel.  A wider cell is rare (the +-180 degree seam,
 * cells next to the viewer) and sends the row the long way.  With all four
 * vertices inside the volume and the band and no discard, what is left of
 * hz_tri_cull() is the back-face test on the snapped area and the pixel box.
 * (tests/test_gpu_exactness.py checks this function against hz_tri_cull() on
 * seeded rows around every one of those borders.) */
__device__ static inline bool mr_simple_cull(const mr_rowstate_t& prev, const mr_rowstate_t& now, bool has_cell,
                                             const hz_params_t& p, bool* keep0, bool* keep1)
{
    /* steps from v00 to the cell's other vertices, in 1/256 pixel */
    const int32_t d01x = (int32_t)((uint32_t)now.xs - (uint32_t)prev.xs);               /* v01 - v00 */
    const int32_t d01y = (int32_t)((uint32_t)now.ys - (uint32_t)prev.ys);
    const int32_t d11x = (int32_t)((uint32_t)d01x + (uint32_t)now.h_dx);                /* v11 - v00 = (v01 - v00) + (v11 - v01) */
    const int32_t d11y = (int32_t)((uint32_t)d01y + (uint32_t)now.h_dy);
    const int32_t d10x = prev.h_dx, d10y = prev.h_dy;                                   /* v10 - v00 */
    const int32_t lo = hz_imin(hz_imin(d01x, d11x), d10x), hi = hz_imax(hz_imax(d01x, d11x), d10x);
    if(__any(has_cell && !(lo > -p.quad_max_dx && hi < p.quad_max_dx))) return false;
    const int64_t area0 = (int64_t)d11x*(int64_t)d01y - (int64_t)d01x*(int64_t)d11y;
    const int64_t area1 = (int64_t)d10x*(int64_t)d11y - (int64_t)d11x*(int64_t)d10y;
    /* t0 = the row's own edge v01-v11 plus v00; t1 = the lower edge v00-v10 plus v11.
     * A box holds a pixel centre iff first < 1 + last on both axes */
    const uint32_t c0 = mr_pk_min(now.h_c, prev.c),                 f0 = mr_pk_max(now.h_f1, prev.f1);
    const uint32_t c1 = mr_pk_min(now.e_c, prev.h_c), f1 = mr_pk_max(now.e_f1, prev.h_f1);
    const bool box0 = mr_pk_lt(c0, f0), box1 = mr_pk_lt(c1, f1);
    *keep0 = has_cell && area0 > 0 && box0;
    *keep1 = has_cell && area1 > 0 && box1;
    return true;
}

/* lane k holds triangle record r with npix pixel centres in its box (0 = none):
 * spread all those pixel centres over the 64 lanes (wave prefix sum + search),
 * so that every lane tests one pixel per pass whatever the mix of box sizes */
__device__ static void mr_distribute(const hz_rec_t& r, uint32_t npix, int lane,
                                     unsigned long long* fb, const hz_params_t& p)
{
    /* rounds in which every lane tests the next pixel of ITS OWN triangle: no
     * cross-lane traffic, no search, short dependency chains.  Worth it while
     * at least half the lanes still have a pixel left (boxes of similar size,
     * the common case inside one flush) */
    uint32_t done = 0;
    for(;;)
    {
        const bool more = npix > done;
        if(__popcll(__ballot(more)) < 32) break;
        if(more)
        {
            const int ry = (int)(((float)done + 0.5f) * r.inv_bw);
            const int rx = (int)done - ry*r.bw;
            hz_emit_rec<false>(fb, p, r, r.px0 + rx, r.py0 + ry);
            done++;
        }
    }

    /* what is left (a few larger boxes) is spread evenly over the lanes */
    const uint32_t rest  = npix > done ? npix - done : 0;
    const uint32_t incl  = mr_scan(rest, lane);
    const uint32_t excl  = incl - rest;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for(uint32_t base = 0; base < total; base += 64)
    {
        const uint32_t it = base + lane;
        /* owner = last lane whose exclusive prefix is <= it */
        int lo = 0;
        #pragma unroll
        for(int step=32; step>=1; step>>=1)
        {
            const uint32_t v = __shfl(excl, lo + step);
            if(v <= it) lo += step;
        }
        hz_rec_t o;
        #pragma unroll
        for(int m=0; m<3; m++)
        {
            o.e.dx[m]  = __shfl(r.e.dx[m], lo);  o.e.ndy[m] = __shfl(r.e.ndy[m], lo);
            o.e.glo[m] = __shfl(r.e.glo[m], lo); o.e.ghi[m] = __shfl(r.e.ghi[m], lo);
        }
        o.z_org = __shfl(r.z_org, lo); o.dzdx = __shfl(r.dzdx, lo); o.dzdy = __shfl(r.dzdy, lo);
        o.r_org = __shfl(r.r_org, lo); o.drdx = __shfl(r.drdx, lo); o.drdy = __shfl(r.drdy, lo);
        o.px0 = __shfl(r.px0, lo); o.py0 = __shfl(r.py0, lo); o.bw = __shfl(r.bw, lo);
        o.inv_bw = __shfl(r.inv_bw, lo);
        o.prim = __shfl(r.prim, lo);
        const uint32_t oexcl = __shfl(excl, lo), odone = __shfl(done, lo);
        if(it < total)
        {
            const uint32_t local = it - oexcl + odone;
            const int ry = (int)(((float)local + 0.5f) * o.inv_bw);
            const int rx = (int)local - ry*o.bw;
            hz_emit_rec<false>(fb, p, o, o.px0 + rx, o.py0 + ry);
        }
    }
}

/* medium triangles queued by k_march: one wave per 64 records */
__global__ __launch_bounds__(64)
void k_mid(unsigned long long* __restrict__ fb, const hz_rec_t* __restrict__ midrec,
           const unsigned int* __restrict__ counters, unsigned int midrec_capacity, hz_params_t p)
{
    const int lane = threadIdx.x;
    /* record slots [0, n): slot g holds a record if the shard it belongs to got that far (hz_types.h, HZ_QSLOT; one shard:
     * every slot below its count).  Records from a shard's first reservation that did not fit were not written
     * (their triangles were rasterised by the marching wave instead) */
    const int sl = p.qshards_log2;
    unsigned int longest = 0;
    #pragma unroll
    for(int s=0; s<HZ_QSHARDS; s++) { const unsigned int* c = hz_qshard(counters, s) + 4; longest = max(longest, min(min(c[0], ~c[1]), HZ_QSHARD_ROOM(midrec_capacity, sl))); }
    const unsigned int n = ((longest + HZ_QBLOCK-1) & ~(unsigned int)(HZ_QBLOCK-1)) << sl;
    for(unsigned int base = blockIdx.x*64u; base < n; base += gridDim.x*64u)
    {
        hz_rec_t r = {};
        uint32_t npix = 0;
        const unsigned int g = base + lane, block = g >> HZ_QBLOCK_LOG2;
        const unsigned int* c = hz_qshard(counters, (int)(block & ((1u << sl) - 1u))) + 4;
        if(g < n && (((block >> sl) << HZ_QBLOCK_LOG2) | (g & (HZ_QBLOCK-1))) < min(min(c[0], ~c[1]), HZ_QSHARD_ROOM(midrec_capacity, sl)))
        {
            r = midrec[g];
            /* queued records carry the pixel count of the box in the inv_bw
             * slot (the reciprocal is cheaper to redo than to store) */
            npix = __float_as_uint(r.inv_bw);
            r.inv_bw = 1.0f / (float)r.bw;      /* (boxes of up to big_min pixels, which a test may set to anything) */
        }
        mr_distribute(r, npix, lane, fb, p);
    }
}

/* set up and rasterise the `n` (<= 64) oldest pending triangles */
template<bool HIZ, bool SHARDS>
__device__ static void mr_flush(const mr_lds_t& L, unsigned int head, unsigned int n, int lane,
                                int jbeg, int i0,
                                unsigned long long* fb, const mr_queue_t& q, const hz_params_t& p,
                                unsigned int* dbg = nullptr)
{
    hz_rec_t r;
    uint32_t npix = 0;
    int bh = 0;
    bool live = (unsigned int)lane < n;
    const bool valid = live;
    /* Every lane goes through the whole set-up, the lanes beyond `n` as copies
     * of lane 0 (n >= 1: the callers see to that; v_readlane_b32 with the lane
     * spelled out - the compiler moves a v_readfirstlane_b32 into the branch of
     * the idle lanes, where the first active lane is one of them): all their values are defined,
     * none is used - `valid` / `live` guard everything that leaves the wave -
     * and the wave saves the ~50 instructions that gave those lanes zeros
     * (3.5 % of k_march's instructions went there). */
    const uint32_t id_own = L.ids[(head + lane) & (MR_CAP-1)];
    const uint32_t id = valid ? id_own : (uint32_t)__builtin_amdgcn_readlane((int)id_own, 0);
    const int t = id & 1, l = (id >> 1) & 63, rowoff = id >> 7;
    const int s0 = rowoff & (MR_RSLOTS-1), s1 = (rowoff+1) & (MR_RSLOTS-1);
    /* LDS row slot and lane of the three vertices, reference horizonator-lib.c:500-506 */
    const int sa = s0,               la = l;
    const int sb = t == 0 ? s1 : s0, lb = l+1;
    const int sc = s1,               lc = t == 0 ? l : l+1;
    /* position and depth now; the colour only for triangles that get drawn */
    hz_wvert_t a = mr_load_vert_pos(L, sa, la);
    hz_wvert_t b = mr_load_vert_pos(L, sb, lb);
    hz_wvert_t c = mr_load_vert_pos(L, sc, lc);
    hz_box_t box;
    hz_tri_box(&box, &a, &b, &c, p.col0, p.col1-1, 0, p.H-1);
    if(p.early_z)
    {
        /* early depth test (exact, see hz_tri_hidden): behind the ridges next to
         * the viewer almost every survivor ends here, and a flush whose triangles
         * are all hidden costs neither plane set-up nor pixel tests.  For boxes of
         * at most 4 x 2 pixel centres - nearly all of the far field's, which is
         * seen at grazing angles - and with the eight depths fetched at once (a
         * narrower box fetches pixels twice). */
        if(valid && box.px1 - box.px0 <= 3 && box.py1 - box.py0 <= 1)
        {
            /* depth = the upper 24 bits of a word's upper half.  Byte offsets in 32 bits
             * (the test is only switched on for framebuffers below 4 GB: draw_impl): one scalar base and a
             * 32-bit offset per load instead of eight 64-bit address computations */
            const char* hi = (const char*)fb + 4;
            const uint32_t rowbytes = (uint32_t)p.SW*8u;
            const uint32_t o0 = (uint32_t)box.py0*rowbytes, o1 = box.py1 > box.py0 ? o0 + rowbytes : o0;
            const int c0 = box.px0 - p.col0, cl = box.px1 - p.col0;
            /* two neighbouring pixels to a load (12 bytes from the first one's upper half: its depth, the second one's
             * lower half, the second one's depth): the box's first two columns and its last two - four loads where round 4
             * had eight, one per pixel (k_march alone 0.635 -> 0.627 ms, a render of a series -1.6 %: profiles/
             * r5_ab_march_loop.txt).  A box one column wide takes a neighbour's depth along (a larger zs: fewer triangles
             * found hidden, never one too many); the pair stays inside the row (SW >= 2: draw_impl).  (The second pair and the second
             * row only in the lanes whose box has them - fewer addresses, two branches - is slower: 0.648 against 0.628 ms.) */
            typedef uint32_t u3_t __attribute__((ext_vector_type(3)));
            const int ca = min(c0, p.SW - 2), cb = min(max(cl - 1, c0), p.SW - 2);
            const uint32_t ba = 8u*(uint32_t)ca, bb = 8u*(uint32_t)cb;
            const u3_t a0 = *(const u3_t*)(hi + (o0 + ba)), b0 = *(const u3_t*)(hi + (o0 + bb));
            const u3_t a1 = *(const u3_t*)(hi + (o1 + ba)), b1 = *(const u3_t*)(hi + (o1 + bb));
            const uint32_t zs = max(max(max(a0.x, a0.z), max(b0.x, b0.z)), max(max(a1.x, a1.z), max(b1.x, b1.z))) >> 8;
            if(hz_tri_hidden(&a, &b, &c, p.z_hide_k, zs)) live = false;
        }
        else if(HIZ && valid && p.hiz)
        {
            /* larger boxes, where the draw keeps coarse depth (hz_k_hiz.h): a box of at most 9 x 5 pixels lies in
             * 2 x 2 tiles of 8 x 4, one of at most 33 x 17 in 2 x 2 tiles of 32 x 16.  Each word is >= the depth
             * of every pixel of its tile (taken earlier in this draw; depths only decrease). */
            const int bw1 = box.px1 - box.px0, bh1 = box.py1 - box.py0;
            const bool lv1 = bw1 <= (1 << HIZ1_W_LOG2) && bh1 <= (1 << HIZ1_H_LOG2);
            if(lv1 || (bw1 <= (1 << HIZ2_W_LOG2) && bh1 <= (1 << HIZ2_H_LOG2)))
            {
                const uint32_t* t = lv1 ? p.hiz : hiz_level2(p);
                const int sx = lv1 ? HIZ1_W_LOG2 : HIZ2_W_LOG2, sy = lv1 ? HIZ1_H_LOG2 : HIZ2_H_LOG2;
                const uint32_t tw = (uint32_t)(lv1 ? hiz_w1(p.SW) : hiz_w2(p.SW));
                const uint32_t tx0 = (uint32_t)(box.px0 - p.col0) >> sx, tx1 = (uint32_t)(box.px1 - p.col0) >> sx;
                const uint32_t o0 = ((uint32_t)box.py0 >> sy)*tw, o1 = ((uint32_t)box.py1 >> sy)*tw;
                const uint32_t z00 = t[o0 + tx0], z01 = t[o0 + tx1], z10 = t[o1 + tx0], z11 = t[o1 + tx1];
                const uint32_t zs = max(max(z00, z01), max(z10, z11)) >> 8;
                if(hz_tri_hidden(&a, &b, &c, p.z_hide_k, zs)) live = false;
            }
        }
        if(dbg) { dbg[5] += (unsigned int)__popcll(__ballot(valid && !live)); }
        if(HZ_DEBUG(p) == 2) return;
        if(!__any(live))
        {
            if(dbg) { dbg[0] += 1; dbg[1] += n; }
            return;
        }
    }
    {
        a.red = __uint_as_float(L.rows[sa][3][la]);
        b.red = __uint_as_float(L.rows[sb][3][lb]);
        c.red = __uint_as_float(L.rows[sc][3][lc]);
        hz_tri_t tri;
        hz_tri_planes(&tri, &a, &b, &c);
        hz_rec_from_tri(r, tri);
        r.px0 = box.px0; r.bw = box.px1 - box.px0 + 1;
        r.py0 = box.py0; bh   = box.py1 - box.py0 + 1;
        r.prim = (uint32_t)(((size_t)(jbeg + rowoff)*(p.N-1) + (i0 + l))*2 + t);
        npix = live ? (uint32_t)r.bw*(uint32_t)bh : 0u;       /* the others own no pixel: mr_distribute never reads their record */
        /* inv_bw turns a pixel's number k inside the box into its row, floor((k + 0.5)/bw).
         * For boxes of at most 64 pixels - every box of nearly every flush - v_rcp_f32's
         * 1 ulp is plenty: the quotient is off by < 64 * 3 * 2^-24 and lies at least
         * 0.5/64 away from an integer.  (Larger boxes are rasterised here only when
         * a queue is full.) */
        r.inv_bw = __all(npix <= 64u) ? __builtin_amdgcn_rcpf((float)r.bw) : 1.0f / (float)r.bw;
    }

    /* large boxes go to k_big: one record, ceil(tiles/64) work items */
    const bool is_big = live && npix > p.big_min;
    const unsigned long long bigmask = __ballot(is_big);
    if(dbg) { dbg[0] += 1; dbg[1] += n; dbg[2] += (unsigned int)__popcll(bigmask); }
    const unsigned long long t_app0 = (dbg && HZ_DEBUG(p) == 4) ? __builtin_amdgcn_s_memtime() : 0ull;
    if(bigmask)
    {
        uint32_t chunks = 0;
        if(is_big)
        {
            chunks = hz_big_chunks(r.bw, bh);
        }
        const uint32_t incl  = mr_scan(chunks, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t nb    = (uint32_t)__popcll(bigmask);
        uint32_t rbase = 0, ibase = 0, ok = 0;
        /* (SHARDS: p.qshards_log2 = HZ_QSHARDS_LOG2, zoomed views; else 0 - as a constant: whole panoramas append with the code they
         * had before there were shards, which as a variable shift cost their marching kernel 1.5 %) */
        const int sl = SHARDS ? HZ_QSHARDS_LOG2 : 0, shard = (int)((blockIdx.x + 5u*blockIdx.y) & ((1u << sl) - 1u));
        if(lane == 0)
        {
            /* records and items in one step (the two counters of a shard are the halves of one 64-bit word), through the
             * counter of this wave's shard: atomics on one address are served one after the other, 12 ns apiece
             * (hz_types.h: HZ_QSHARDS; whole panoramas: one shard) */
            ok = hz_queue_reserve(q, shard, sl, nb, total, &rbase, &ibase) ? 1u : 0u;
        }
        rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase); ibase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ibase); ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ok);
        if(is_big)
        {
            if(ok)
            {
                const uint32_t ri = HZ_QSLOT(rbase + (uint32_t)__popcll(bigmask & ((1ull << lane) - 1ull)), shard, sl);
                const uint32_t ii = ibase + (incl - chunks);
                q.bigrec[ri].r = r; q.bigrec[ri].bh = bh;
                for(uint32_t c2=0; c2<chunks; c2++) { const uint32_t g = HZ_QSLOT(ii + c2, shard, sl); q.bigitem[g].rec = ri; q.bigitem[g].chunk = c2; }
            }
            else
            {
                /* queue full (capacities are sized for 32k-wide panoramas): slow but correct */
                for(int py = r.py0; py < r.py0 + bh; py++)
                    for(int px = r.px0; px < r.px0 + r.bw; px++)
                        hz_emit_rec<true>(fb, p, r, px, py);
            }
            npix = 0;
        }
    }

    /* medium boxes go to k_mid, which spreads them over the whole chip: left
     * here they make the waves next to the viewer the critical path */
    const bool is_mid = live && npix > p.inline_max;
    const unsigned long long midmask = __ballot(is_mid);
    if(dbg) { dbg[3] += (unsigned int)__popcll(midmask); }
    if(midmask)
    {
        /* (through the counter of this wave's shard, like the big triangles: hz_types.h, HZ_QSHARDS) */
        const int slm = SHARDS ? HZ_QSHARDS_LOG2 : 0, mshard = (int)((blockIdx.x + 5u*blockIdx.y) & ((1u << slm) - 1u));
        unsigned int* const mc = hz_qshard(q.counters, mshard) + 4;     /* [0] the shard's records, [1] ~(its first invalid one) */
        uint32_t mbase = 0;
        if(lane == 0) mbase = atomicAdd(mc, (uint32_t)__popcll(midmask));
        mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)mbase);
        if(mbase + (uint32_t)__popcll(midmask) <= HZ_QSHARD_ROOM(q.midrec_capacity, slm))
        {
            if(is_mid)
            {
                hz_rec_t m = r;
                m.inv_bw = __uint_as_float(npix);       /* see k_mid */
                q.midrec[HZ_QSLOT(mbase + (uint32_t)__popcll(midmask & ((1ull << lane) - 1ull)), mshard, slm)] = m;
                npix = 0;
            }
        }
        /* else: queue full, they stay here; the shard's slots from mbase on hold nothing
         * of this draw and k_mid must not read them */
        else if(lane == 0) atomicMax(mc + 1, ~mbase);
    }

    if(dbg && HZ_DEBUG(p) == 4)
    {
        /* HZ_MARCH_DEBUG=4 (tools/wave_timing.py): "mid" and "items" count cycles / 16 of the queue appends and of the pixel turns */
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        dbg[3] += (unsigned int)((t1 - t_app0) >> 4);
        mr_distribute(r, npix, lane, fb, p);
        dbg[4] += (unsigned int)((__builtin_amdgcn_s_memtime() - t1) >> 4);
        return;
    }
    if(dbg) { const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)mr_scan(npix, lane), 63); dbg[4] += tot; }
    mr_distribute(r, npix, lane, fb, p);
}

/* COUNTERS: the diagnostics instance (HZ_WAVE_TIMING: per-wave duration and
 * counters into p.wave_cycles).  The production instance carries none of it -
 * no scratch memory for the counters, none of their branches, fewer scalar
 * registers held. */
/* MR_WAVES_PER_EU (a build-time experiment, profiles/r3_experiments.json): ask the compiler for that many
 * marching waves per SIMD instead of the four that 105 registers allow */
#ifdef MR_WAVES_PER_EU
#define MR_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MR_WAVES_PER_EU, MR_WAVES_PER_EU)))
#elif defined(MR_VGPRS)
#define MR_OCCUPANCY __attribute__((amdgpu_num_vgpr(MR_VGPRS)))
#else
#define MR_OCCUPANCY
#endif
/* HIZ: the instance for draws that keep coarse depth (hz_k_hiz.h; zoomed views).  The test of the larger boxes is
 * ~100 instructions in each of the four places the flush is inlined and a dozen scalars held across the marching
 * loop: compiled into the one kernel it cost whole panoramas, which never use it, 2 % more instructions per render. */
/* VCACHE (round 5): the instance for draws whose viewer stands where the draw before stood.  The expensive half of the
 * transform - two atan, two square roots: 42 % of this kernel's instructions - depends on the viewer's position alone;
 * a context keeps its four numbers per vertex (p.vcache, 16 bytes each, written by k_polar_fill when a viewpoint is drawn
 * a second time) and this instance reads them instead of the elevation: a 16-byte load and the ~30 instructions of
 * hz_finish() per vertex where the cold instance spends ~155.  Same operations on the same numbers: same bits. */
template<bool COUNTERS, bool HIZ, bool VCACHE, bool SHARDS>
__global__ __launch_bounds__(64) MR_OCCUPANCY
void k_march(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb,
             mr_queue_t q, mr_zones_t zn, hz_params_t p)
{
    __shared__ mr_lds_t L;
    const unsigned long long t_start = COUNTERS ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned int dbgv[6] = {0,0,0,0,0,0};
    unsigned int* dbg = COUNTERS ? dbgv : nullptr;

    const int lane = threadIdx.x;
    /* which strip: from the launch grid (whole panoramas: every strip has work),
     * or from the draw's work list (azimuth sectors and views of less than 360
     * degrees: the host lists the strips that can reach the drawn columns, so a
     * sector launches - and pays for - its own share of the waves only) */
    int sx, seg;
    if(p.worklist)
    {
        const uint32_t item = p.worklist[blockIdx.x];
        sx = (int)(item & ((1u << MR_ITEM_SX_BITS) - 1u)); seg = (int)(item >> MR_ITEM_SX_BITS);
    }
    else { sx = (int)blockIdx.x + (p.pass == 1 ? p.near_x0 : 0); seg = (int)blockIdx.y; }
    const int i0   = sx*MR_COLS;
    const int i    = i0 + lane;
    int jbeg, jend;                                     /* vertex rows jbeg..jend, cell rows jbeg..jend-1 */
    mr_segment_rows(zn, seg, &jbeg, &jend);
    if(p.pass)
    {
        const bool near = sx >= p.near_x0 && sx <= p.near_x1 && jbeg < p.near_j1 && jend > p.near_j0;
        if((p.pass == 1) != near) return;
    }
    const bool has_vertex = i < p.N;
    const bool has_cell   = lane < MR_COLS && i < p.N-1;
    const int  ic = has_vertex ? i : p.N-1;             /* clamped: idle lanes redo the last column */

    /* azimuth-sector shard (multi-GPU): a segment that does not contain the
     * viewer is a convex patch seen from outside, so its azimuth extent is that
     * of its four corner vertices; if that lies outside this GPU's columns the
     * whole wave has nothing to draw.  (The corners are real vertices: their x
     * is computed exactly as the rasteriser computes it.) */
    if(p.cull_strips)
    {
        const int ia = i0, ib = min(i0 + MR_COLS, p.N-1);
        const bool viewer_inside = p.u.viewer_cell_i >= (float)(ia-1) && p.u.viewer_cell_i <= (float)(ib+1) &&
                                   p.u.viewer_cell_j >= (float)(jbeg-1) && p.u.viewer_cell_j <= (float)(jend+1);
        if(!viewer_inside)
        {
            /* lanes 0..3 take one corner each (the others repeat them): one
             * transform's worth of instructions for the wave instead of four */
            const hz_vertex_t v = hz_transform_en(&p.u, hz_east(&p.u, (float)((lane & 1) ? ib : ia)),
                                                  hz_north(&p.u, (float)((lane & 2) ? jend : jbeg)), 0.f);
            float xlo = 2.f, xhi = -2.f;
            #pragma unroll
            for(int c=0; c<4; c++)
            {
                const float xc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), c));
                xlo = hz_min(xlo, xc); xhi = hz_max(xhi, xc);
            }
            if(xhi - xlo <= 1.0f)       /* not across the +-180 degree seam */
            {
                const float flo = (xlo*p.halfW + p.halfW) - 2.5f, fhi = (xhi*p.halfW + p.halfW) + 1.5f;
                if(fhi < (float)p.col0 || flo > (float)p.col1) return;
            }
        }
    }

    const float e = hz_east(&p.u, (float)ic);
    /* the north offset of vertex row jbeg+lane, computed once per strip: a row
     * then takes its n with one v_readlane instead of redoing the (wave-uniform)
     * arithmetic with its IEEE division 64 lanes wide in every row */
    const float n_tab = hz_north(&p.u, (float)(jbeg + lane));
    auto north_of = [&](int rel) -> float
    {
        if(rel < 64) return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n_tab), rel));
        return hz_north(&p.u, (float)(jbeg + rel));
    };
    /* A strip whose vertex rows all lie beyond zfar (the test the row loop
     * makes per row, for the row nearest the viewer: rounding is monotonic, so
     * min over rows of fl(fl(n^2) + fl(e^2)) = fl(min fl(n^2) + fl(e^2))) would
     * walk its rows without transforming one: it leaves here.  With the API's
     * default far clip of 40 km that is 90 % of the strips of a 7x7-tile mosaic. */
    if(p.far_strips)
    {
        const int nrows = jend - jbeg;                  /* vertex rows 0..nrows */
        float nn = lane <= nrows ? n_tab*n_tab : __builtin_inff();
        if(nrows >= 64) { const float n64 = hz_north(&p.u, (float)(jbeg + 64)); nn = hz_min(nn, n64*n64); }
        #pragma unroll
        for(int m=32; m>=1; m>>=1) nn = hz_min(nn, __shfl_xor(nn, m));
        if(__all(nn + e*e > p.far_dd)) return;
    }

    /* abridged division / square-root sequences (hz_fast.h): allowed where the
     * operands are in range - the draw's uniforms (host), this strip's east
     * offsets, each row's north offset */
    const hzf_const_t fc = hzf_setup(&p.u);
    const bool fast_strip = p.fast_ok && __all(hzf_in_range(e));
    const unsigned long long fast_rows = __ballot(hzf_in_range(n_tab));
    /* the reciprocal the azimuth's arc tangent begins with (hzf_atan2_r) is that of |n| north of the viewer - one value
     * per row - and that of e elsewhere - one per lane, the same in every row: ahead of time, like n itself, in ONE
     * register (a strip lies north or south of the viewer; the few that hold the viewer's row compute it row by row) */
    const int  tab_rows   = min(jend - jbeg, 63);
    const bool rcp_south  = __all(lane > tab_rows || 0.f >= n_tab);
    const bool rcp_north  = __all(lane > tab_rows || !(0.f >= n_tab));
    const float rcp_tab   = VCACHE ? 0.f : hzf_rcp(rcp_south ? e : hz_abs(n_tab));

    /* pending-triangle ring, wave-uniform state */
    unsigned int head = 0, count = 0;
    if(lane == 0) L.nclip = 0;
    __syncthreads();
    int first_row = 0;                                  /* cell row (relative) of the oldest pending triangle */
    /* what a row keeps of itself for the cells above it (the attributes of its
     * vertices live in LDS, where mr_flush takes them from): per lane the
     * vertex's NDC x, snapped position and clip mask, its pixel columns/rows
     * (mr_vcull_t) and the same combined with the vertex one lane to the east */
    mr_rowstate_t prev = {};
    bool prev_simple = false;
    int16_t z_next = VCACHE ? (int16_t)0 : mosaic[(size_t)jbeg*p.N + ic];
    hz_polar_t q_next = {};
    if(VCACHE) q_next = p.vcache[(size_t)jbeg*p.N + ic];
    /* rows whose 64 vertices all lie safely beyond zfar (by horizontal distance
     * alone, 0.1% margin): their triangles can only be far-clipped, so a vertex
     * row is transformed only if it or a neighbouring row is not such a row.
     * With the default zfar = 40 km this is most of a large mosaic. */
    float n_cur = north_of(0);
    bool far_prev = true;
    bool far_cur  = __all(n_cur*n_cur + e*e > p.far_dd);
    for(int j = jbeg; j <= jend; j++)
    {
        const int rel = j - jbeg;
        const float z = (float)z_next;
        const hz_polar_t q_cur = q_next;
        if(j < jend)
        {
            if(VCACHE) q_next = p.vcache[(size_t)(j+1)*p.N + ic];
            else z_next = mosaic[(size_t)(j+1)*p.N + ic];
        }
        const float n_next   = (j == jend) ? 0.f : north_of(rel+1);
        const bool  far_next = (j == jend) || __all(n_next*n_next + e*e > p.far_dd);
        const bool  skip_row   = far_prev && far_cur && far_next;   /* vertex row j not needed       */
        const bool  skip_cells = far_prev && far_cur;               /* cell row j-1 entirely clipped */
        const float n = n_cur;
        n_cur = n_next; far_prev = far_cur; far_cur = far_next;
        if(skip_row) continue;

        const bool fast = fast_strip && (rel >= 64 ? hzf_in_range(n) : (int)((fast_rows >> rel) & 1ull));
        hz_vertex_t vtx;
        if(VCACHE) vtx = fast ? hzf_finish(&p.u, &fc, q_cur) : hz_finish(&p.u, q_cur);
        else if(fast)
        {
            float rcp_az;
            if(rel >= 64 || !(rcp_south || rcp_north)) rcp_az = hzf_rcp((0.f >= n) ? e : hz_abs(n));
            else if(rcp_south) rcp_az = rcp_tab;
            else rcp_az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rcp_tab), rel));
            vtx = hzf_transform_en_r(&p.u, &fc, e, n, z, rcp_az);
#ifdef HZ_EXP_TRANSFORM_TWICE
            /* (tools/march_bounds.py: what does one more transform per vertex cost the kernel?  The same picture: the second
             * result replaces the first where they differ, which is nowhere - but the compiler cannot know) */
            {
                float n2 = n; asm volatile("" : "+v"(n2));
                const hz_vertex_t v2 = hzf_transform_en_r(&p.u, &fc, e, n2, z, rcp_az);
                vtx.x = (v2.x == vtx.x) ? vtx.x : v2.y; vtx.y = (v2.y == vtx.y) ? vtx.y : v2.z;
                vtx.z = (v2.z == vtx.z) ? vtx.z : v2.x; vtx.red = (v2.red == vtx.red) ? vtx.red : v2.x;
            }
#endif
        }
        else vtx = hz_transform_en(&p.u, e, n, z);

        bool in_volume, in_guard;
        hz_wvert_t cur = mr_window(vtx, p, &in_volume, &in_guard);
        const bool cur_simple = __all(in_guard && in_volume);
        if(!cur_simple) mr_window_flags(cur, vtx, in_guard);

        /* this row replaces vertex row rel-MR_RSLOTS in LDS: triangles that
         * still need it are set up now (happens where survivors are sparse) */
        if(count && first_row <= rel - MR_RSLOTS)
        {
            __syncthreads();
            mr_flush<HIZ, SHARDS>(L, head, count, lane, jbeg, i0, fb, q, p, dbg);
            __syncthreads();
            head = (head + count) & (MR_CAP-1);
            count = 0;
        }
        mr_store_row(L, rel & (MR_RSLOTS-1), lane, cur);

        const mr_rowstate_t now = mr_rowstate_of(cur, p);

        if(j > jbeg && !skip_cells)
        {
            /* cell (i, j-1): v00 = prev, v01 = cur, v10 / v11 = those of the lane to
             * the east; triangles t0 = (v00,v11,v01), t1 = (v00,v10,v11), reference
             * horizonator-lib.c:500-506 */
            bool keep0 = false, keep1 = false;
            if(cur_simple && prev_simple && mr_simple_cull(prev, now, has_cell, p, &keep0, &keep1))
            {
            }
            else
            {
                hz_wvert_t v00 = {}, v01 = {}, v10 = {}, v11 = {};
                v00.xn = prev.xn; v00.xs = prev.xs; v00.ys = prev.ys; v00.cmask = prev.cmask;
                v01.xn = cur.xn;  v01.xs = cur.xs;  v01.ys = cur.ys;  v01.cmask = cur.cmask;
                /* (the neighbour's snapped position from the step to it: the DPP read of it stays fused into that subtraction) */
                v10.xn = mr_from_east(prev.xn);
                v10.xs = (int32_t)((uint32_t)prev.xs + (uint32_t)prev.h_dx); v10.ys = (int32_t)((uint32_t)prev.ys + (uint32_t)prev.h_dy);
                v10.cmask = (uint32_t)mr_from_east((int32_t)prev.cmask);
                v11.xn = mr_from_east(cur.xn);
                v11.xs = (int32_t)((uint32_t)cur.xs + (uint32_t)now.h_dx);   v11.ys = (int32_t)((uint32_t)cur.ys + (uint32_t)now.h_dy);
                v11.cmask = (uint32_t)mr_from_east((int32_t)cur.cmask);
                hz_box_t box;
                const int verdict0 = has_cell ? hz_tri_cull(&box, &v00, &v11, &v01, p.col0, p.col1-1, 0, p.H-1) : HZ_TRI_DROP;
                const int verdict1 = has_cell ? hz_tri_cull(&box, &v00, &v10, &v11, p.col0, p.col1-1, 0, p.H-1) : HZ_TRI_DROP;
                /* crossing the image border or the near/far sphere: k_clip */
                const uint32_t prim0 = (uint32_t)(((size_t)(j-1)*(p.N-1) + i)*2);
                mr_clip_note(L, q, verdict0 == HZ_TRI_CLIP, prim0,   lane);
                mr_clip_note(L, q, verdict1 == HZ_TRI_CLIP, prim0+1, lane);
                keep0 = verdict0 == HZ_TRI_DRAW; keep1 = verdict1 == HZ_TRI_DRAW;
            }
            #pragma unroll
            for(int t=0; t<2; t++)
            {
                const bool keep = (t == 0 ? keep0 : keep1) && HZ_DEBUG(p) != 1;
                const unsigned long long m = __ballot(keep);
                if(m)
                {
                    if(count == 0) first_row = rel-1;
                    if(keep)
                    {
                        const unsigned int at = (head + count + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))) & (MR_CAP-1);
                        L.ids[at] = ((uint32_t)(rel-1) << 7) | ((uint32_t)lane << 1) | (uint32_t)t;
                    }
                    count += (unsigned int)__popcll(m);
                    if(count >= 64)
                    {
                        __syncthreads();        /* one wave: orders the LDS writes before the reads */
                        mr_flush<HIZ, SHARDS>(L, head, 64, lane, jbeg, i0, fb, q, p, dbg);
                        head = (head + 64) & (MR_CAP-1);
                        count -= 64;
                        if(count) first_row = (int)(L.ids[head] >> 7);
                        __syncthreads();
                    }
                }
            }
        }
        prev = now; prev_simple = cur_simple;
    }
    if(count)
    {
        __syncthreads();
        mr_flush<HIZ, SHARDS>(L, head, count, lane, jbeg, i0, fb, q, p, dbg);
    }
    /* (Four inlined copies of mr_flush - this one, the one before a row is stored, two in the unrolled loop over a cell's
     * triangles - and 104 registers.  Round 5 tried ONE place at the top of the row loop instead (tools/patches/
     * r5_one_flush_copy.diff): 96 registers, scalar spills 24 -> 9, a third of the code - and with four waves per SIMD
     * the kernel takes 0.675 instead of 0.640 ms, a render of a series 0.885 instead of 0.836; with the five waves 96
     * registers allow 0.625 alone and 0.98 in a series, whose other kernels then find no registers beside the marching
     * waves; two copies: 0.670 / 0.873.  profiles/r5_ab_march_loop.txt.) */
    mr_clip_flush(L, q, lane);
    if(COUNTERS && p.wave_cycles && lane == 0)
    {
        unsigned long long* o = &p.wave_cycles[((size_t)blockIdx.y*gridDim.x + blockIdx.x)*4];     /* (grid launches) */
        o[0] = __builtin_amdgcn_s_memtime() - t_start;
        o[1] = ((unsigned long long)dbgv[0] << 32) | dbgv[1];     /* flushes, triangles set up */
        o[2] = ((unsigned long long)dbgv[2] << 32) | dbgv[3];     /* to k_big, to k_mid        */
        o[3] = ((unsigned long long)dbgv[5] << 32) | dbgv[4];     /* hidden by the early depth test, pixel centres tested here */
    }
}
