/* hz_draw.cpp - one draw of the HIP render path: glClear + glDrawElements of reference horizonator-lib.c:896-897 as
 * kernels on the context's streams - the next framebuffer, one or two rounds of k_march with the kernels that finish
 * what the marching waves queued (k_clip, k_mid, k_big, the tile kernels), coarse depth between the rounds, the vertex
 * cache.  hz_hip_draw() of include/hz_hip.h.  Plain C++ over the HIP runtime API (compiled by g++); every kernel is
 * reached through its launcher in hz_launch.h; the plan comes from hz_plan.cpp. */
#include "hz_dev.h"
#include "hz_fast.h"

/* the list of round `which` (0: first round, on nstream; 1: second or only round, on
 * `stream`) resident in d_list[which], uploaded on the stream that consumes it */
static int upload_list(hz_dev_t* d, int which, hipStream_t st, const std::vector<uint32_t>& items)
{
    hz_worklists_t& wl = *d->lists;
    const size_t n = items.size();
    if(n > wl.cap[which])
    {
        /* (rare: the first sector draw of a context, a much wider sector) kernels in flight may still read the old one */
        HZ_CHECK(hz_sync_all(d));
        (void)hipFree(wl.d_items[which]); wl.d_items[which] = NULL;
        for(int t=0; t<2; t++) { if(wl.h_items[which][t]) (void)hipHostFree(wl.h_items[which][t]); wl.h_items[which][t] = NULL; }
        wl.cap[which] = 0;
        const size_t cap = n + n/4 + 1024;
        hipError_t e = hipMalloc(&wl.d_items[which], cap*sizeof(uint32_t));
        for(int t=0; t<2 && e == hipSuccess; t++) e = hipHostMalloc((void**)&wl.h_items[which][t], cap*sizeof(uint32_t), hipHostMallocDefault);
        if(e != hipSuccess)
        {
            (void)hipFree(wl.d_items[which]); wl.d_items[which] = NULL;
            for(int t=0; t<2; t++) { if(wl.h_items[which][t]) (void)hipHostFree(wl.h_items[which][t]); wl.h_items[which][t] = NULL; }
            HZ_CHECK(e);
        }
        wl.cap[which] = cap;
    }
    /* A view that moves from draw to draw (a sequence of viewpoints or azimuths over one sector) brings a new list with
     * every draw.  The copy of a list sits on `st` behind that draw's wait for its framebuffer: with one pinned buffer
     * the host would stall here about one draw behind the device; with two in turn it waits for the copy two lists back. */
    const int t = wl.turn[which]; wl.turn[which] ^= 1;
    if(!wl.ev_copied[which][t]) HZ_CHECK(hipEventCreateWithFlags(&wl.ev_copied[which][t], hipEventDisableTiming));
    else HZ_CHECK(hipEventSynchronize(wl.ev_copied[which][t]));   /* the copy engine is done with this pinned buffer */
    if(n)
    {
        memcpy(wl.h_items[which][t], items.data(), n*sizeof(uint32_t));
        HZ_CHECK(hipMemcpyAsync(wl.d_items[which], wl.h_items[which][t], n*sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    HZ_CHECK(hipEventRecord(wl.ev_copied[which][t], st));
    wl.n[which] = (unsigned int)n;
    return 0;
}

/* One draw = (clear: see below), then one or two rounds of
 *   k_march            every strip (one round), or: round 1 the strips next to the
 *                      viewer, round 2 all the others
 *   k_clip k_mid k_big what the marching waves queued
 * In a two-round draw the second round tests its survivors against the depth
 * the first left in the framebuffer (mr_flush, early_z): everything large on
 * screen - the occluders - is there by then, and in the benchmark scene 95 % of
 * the far field's survivors stop at that test.  The split changes no result: the
 * framebuffer word is order-independent and the test only skips triangles that
 * cannot win a pixel (hz_tri_depth_floor).  Round 1 is a few waves followed by
 * k_big on its own; run alone it costs most of what round 2 saves, so it runs
 * on a stream of its own (nstream) and thereby beside round 2 of the panorama
 * BEFORE, whenever renders are queued back to back: a context cycles through
 * HZ_NFB = 3 framebuffers and queue sets, so round 1 of panorama k+1 needs
 * nothing of panorama k - nor of the conversion of panorama k-1, which with two
 * framebuffers sat between every first round and the framebuffer it waits for
 * (measured, 16000x4000: 1.25 -> 1.17 ms per render; a fourth changes nothing).
 *
 *   nstream | round1 k+1: march(near) clip big | round1 k+2 ...
 *   stream  | round2 k:   march(far, early z)  | round2 k+1 (waits ev_near)
 *   qstream |             clip mid big of k (waits ev_marched)
 *   rstream |             resolve k-1 (clears behind itself)      | resolve k
 *
 * A round's queues are empty when it starts: they are emptied together with
 * their framebuffer (hz_counters_consume; no launch for that in front of k_march).
 *
 * HZ_TWO_PASS=0/1 forces one / two rounds; otherwise contexts of at least
 * HZ_TWO_PASS_MIN_MPIX (default 6) megapixels whose far clip lies well
 * beyond the first round's strips draw in two rounds (plan_rounds). */
int hz_draw_impl(hz_dev_t* d, const hz_view_t* view);

extern "C" int hz_hip_draw(hz_dev_t* d, const hz_view_t* view)
{
    HZ_ON_DEVICE(d);
    return hz_draw_impl(d, view);
}

/* (Every command between two marching kernels on `stream` - an event to record, an event to wait for - is a packet the
 * command processor works through before it launches the second: ~50 us lay between consecutive second rounds with three
 * waits and three records in there, profiles/r3_pipelined_timeline.txt.  So: ev_free[i] stands for the queue sets of
 * framebuffer i as well - it is recorded behind a conversion or a clear, both of which wait for ev_drawn, the end of
 * every kernel of the draw -, and `stream` only tells rstream about readers of the framebuffer when there were any.)
 * The framebuffer of the previous draw goes back to "cleared" (glClear,
 * reference horizonator-lib.c:896: depth = 1.0 -> all-ones words): its
 * conversion did that already (k_resolve<true>), or a memset does it now on
 * rstream, behind the conversions of that draw and behind whatever `stream`
 * still had to read from it; this draw takes the next framebuffer. */
static int next_framebuffer(hz_dev_t* d, hz_params_t& p)
{
    const bool prof = d->profiling != 0;
    const int prev = d->fbi, next = (prev + 1) % HZ_NFB;
    if(d->stream_reads_fb)
    {
        HZ_CHECK(hipEventRecord(d->ev_readers, d->stream));
        HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_readers, 0));
        d->stream_reads_fb = 0;
    }
    HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_drawn, 0));       /* the previous draw's last kernels (qstream) */
    if(prof) HZ_CHECK(hipEventRecord(d->ev[0], d->rstream));
    if(d->fb_used[prev])
    {
        HZ_CHECK(hipMemsetAsync(d->d_fbs[prev], 0xFF, d->fb_used[prev]*sizeof(unsigned long long), d->rstream));
        HZ_CHECK(hipMemsetAsync(d->d_touched[prev], 0, (size_t)d->seg_stride*d->H, d->rstream));
        /* ... and its queue sets with it (the clearing conversions do that themselves) */
        /* (the counters of the mid and clip queues, [3..5], and the shards of the big triangles' behind HZ_CNT_LAST's copy: two pieces each) */
        for(int w=0; w<2; w++)
        {
            unsigned int* c = d->d_big_counters_s[(w ? HZ_NFB : 0) + prev];
            HZ_CHECK(hipMemsetAsync(c, 0, 6*sizeof(unsigned int), d->rstream));
            HZ_CHECK(hipMemsetAsync(c + HZ_QSHARD0, 0, (size_t)HZ_QSHARDS*HZ_QSHARD_STRIDE*sizeof(unsigned int), d->rstream));
        }
    }
    if(prof) HZ_CHECK(hipEventRecord(d->ev[1], d->rstream));
    d->fb_used[prev] = 0;
    HZ_CHECK(hipEventRecord(d->ev_free[prev], d->rstream));
    d->fbi = next; d->d_fb = d->d_fbs[next];
    d->fb_used[next] = (size_t)p.SW*p.H;
    p.touched = d->d_touched[next];
    return 0;
}

/* what the marching waves queued: clipper, medium boxes, large boxes */
static int queue_kernels(hz_dev_t* d, const mr_queue_t& q, const hz_params_t& pp, hipStream_t st, int set, bool by_tile, bool zoomed, unsigned int* report = NULL)
{
    hzk_clip(zoomed, dim3(1024), dim3(64), st, (const int16_t*)d->d_mosaic, d->d_fb, q, pp);
    HZ_CHECK(hipGetLastError());
    if(d->raster != HZ_RASTER_SCATTER && pp.inline_max < pp.big_min)      /* else nothing is ever queued for it */
    {
        hzk_mid(dim3(2048), dim3(64), st, d->d_fb, (const hz_rec_t*)q.midrec, (const unsigned int*)q.counters, q.midrec_capacity, pp);
        HZ_CHECK(hipGetLastError());
    }
    const unsigned int* tile_state = NULL;
    if(by_tile)
    {
        /* the round's large triangles by screen tile, depth in LDS (hz_k_tile.h): list, draw.  k_big follows and
         * stands down unless some tile's list overflowed. */
        tl_bins_t tb = d->tiles_s[set];
        tb.tiles_x = (pp.SW + TL_W-1)/TL_W; tb.tiles_y = (pp.H + TL_H-1)/TL_H;
        tb.list_cap = d->env.tile_list > 0 && d->env.tile_list < TL_LIST ? (unsigned int)d->env.tile_list : TL_LIST;
        tb.units_cap = (unsigned int)(TL_UNITS_PER_TILE*(size_t)((d->W + TL_W-1)/TL_W)*((d->H + TL_H-1)/TL_H));
        HZ_CHECK(hipMemsetAsync(tb.state, 0, 2*sizeof(unsigned int), st));
        HZ_CHECK(hipMemsetAsync(tb.cursor, 0, (size_t)tb.tiles_x*tb.tiles_y*sizeof(unsigned int), st));
        hzk_tile_bin(dim3(1024), dim3(256), st, (const hz_bigrec_t*)q.bigrec, (const hz_bigitem_t*)q.bigitem, (const unsigned int*)q.counters, q.bigrec_capacity, tb, pp);
        hzk_tile_raster(dim3(2048), dim3(256), st, d->d_fb, (const hz_bigrec_t*)q.bigrec, tb, pp);
        HZ_CHECK(hipGetLastError());
        tile_state = tb.state;
    }
    hzk_big(dim3(4096), dim3(256), st, d->d_fb, (const hz_bigrec_t*)q.bigrec, (const hz_bigitem_t*)q.bigitem, q.counters, q.bigrec_capacity, q.bigitem_capacity, pp, tile_state, report);
    HZ_CHECK(hipGetLastError());
    return 0;
}

/* one k_march launch: over the grid (every strip, or pass 1's columns next to the
 * viewer), or over a work list */
static int launch_march(hz_dev_t* d, hipStream_t st, const mr_queue_t& q, const mr_zones_t& zn, hz_params_t pm,
                        const uint32_t* d_list, unsigned int nlist)
{
    const int nsx = (pm.N-1 + MR_COLS-1)/MR_COLS;
    dim3 grid(pm.pass == 1 ? pm.near_x1 - pm.near_x0 + 1 : nsx, zn.total);
    pm.worklist = NULL;
    if(d_list)
    {
        if(nlist == 0) return 0;                    /* nothing of the DEM lies behind these columns */
        pm.worklist = d_list;
        grid = dim3(nlist, 1);
    }
#ifdef HZ_SELFTEST
    /* diagnostics (hz_hip_debug_wave_timing, libhorizonator_selftest.so only): the instance with per-wave counters */
    if(d->wave_timing.d_cycles && pm.pass != 1)
    {
        if((size_t)grid.x*grid.y*4 > d->wave_timing.capacity) { snprintf(g_last_error, sizeof(g_last_error), "wave timing buffer too small"); return -1; }
        pm.wave_cycles = d->wave_timing.d_cycles;
        d->wave_timing.grid_x = grid.x; d->wave_timing.grid_y = grid.y;
        hzk_march(true, true, false, grid, dim3(64), st, (const int16_t*)d->d_mosaic, d->d_fb, q, zn, pm);
    }
    else
#endif
    hzk_march(false, pm.hiz != NULL, pm.vcache != NULL, grid, dim3(64), st, (const int16_t*)d->d_mosaic, d->d_fb, q, zn, pm);
    HZ_CHECK(hipGetLastError());
    return 0;
}

/* Coarse depth of framebuffer `next` (hz_k_hiz.h): the tables of a draw with the geometry of p, allocated on first use */
static int hiz_tables(hz_dev_t* d, int next, const hz_params_t& p, hz_hiz_t* hz)
{
    if(d->hiz_unavailable) return 1;
    if(!d->d_hiz[HZ_NFB-1])
        for(int i=0; i<HZ_NFB; i++)
            if(!d->d_hiz[i] && hipMalloc(&d->d_hiz[i], hiz_words(d->W, d->H)*sizeof(uint32_t)) != hipSuccess)
            {
                /* no memory for them: this and every later draw of the context does without (same bytes, more fragments) */
                (void)hipGetLastError();
                d->d_hiz[i] = NULL;
                d->hiz_unavailable = 1;
                return 1;
            }
    hz->w1 = (int)hiz_w1(p.SW); hz->w2 = (int)hiz_w2(p.SW);
    hz->l1 = d->d_hiz[next];
    hz->l2 = hz->l1 + (size_t)hz->w1*hiz_h1(p.H);
    return 0;
}
/* one sweep over the framebuffer on `st`: every tile of both levels is written (nothing to reset between draws) */
static int hiz_sweep(hz_dev_t* d, hipStream_t st, int next, const hz_params_t& p, const hz_hiz_t& hz)
{
    const unsigned int nunits = (unsigned int)d->seg_stride*(unsigned int)((p.H + HIZ_UNIT_ROWS-1)/HIZ_UNIT_ROWS);
    hzk_hiz(dim3((nunits + 3)/4), dim3(256), st, (const unsigned long long*)d->d_fbs[next], (const unsigned char*)d->d_touched[next], d->seg_stride, p.SW, p.H, hz, nunits);
    HZ_CHECK(hipGetLastError());
    return 0;
}

/* ---- the vertex cache ------------------------------------------------------------------------------------------------
 * The expensive half of the vertex transform (reference vertex.glsl:133-134, 154, 156: two atan, two square roots - 42 % of
 * the marching kernel's instructions, and that kernel is bound by instruction issue at a tenth of the chip's memory bandwidth)
 * depends on where the viewer stands and on nothing else.  A caller that turns, zooms or changes the depth extents
 * (horizonator_pan_zoom, horizonator_set_zextents - the reference's interactive use, and every render of a series from one
 * viewpoint) redraws the same vertices from the same place: the second draw from a viewpoint writes that half of every
 * vertex into HBM (k_polar_fill: 16 bytes per vertex, 1.13 GB for the 7x7-tile mosaic - 288 GB are there to be used), and
 * the draws after it read it back (k_march<.., VCACHE>) instead of computing it.  The FIRST draw from a viewpoint is what it
 * always was - a viewer that moves with every draw (BASELINE configs[3]) never pays for a fill it would not use.
 * Same operations on the same numbers in the same order: the bytes drawn do not depend on it (tests/test_gpu_sequences.py
 * interleaves cached and cold draws with moves).  Dropped by horizonator_move (a new key), by a new mosaic, with the
 * context; off where memory is short or hz_options_t::vertex_cache says so. */
static bool same_viewpoint(const hz_xform_t& a, const hz_xform_t& b)
{
    return a.viewer_cell_i == b.viewer_cell_i && a.viewer_cell_j == b.viewer_cell_j && a.viewer_z == b.viewer_z &&
           a.cos_viewer_lat == b.cos_viewer_lat && a.deg_per_cell == b.deg_per_cell;
}
static int vertex_cache(hz_dev_t* d, hz_params_t& p)
{
    p.vcache = NULL;
    if(!d->env.vertex_cache || d->vc.unavailable) return 0;
    if(!d->vc.state || !same_viewpoint(d->vc.key, p.u))
    {
        d->vc.key = p.u; d->vc.state = 1;           /* seen once: this draw computes everything, as always */
        return 0;
    }
    /* (a call that delivers into host memory draws its panorama in sectors, hz_hostpath.cpp: the later sectors are the SAME
     * draw from this viewpoint, not the second - a viewer that moves between such calls never pays for a fill) */
    if(d->vc.state == 1 && d->vc.same_draw) return 0;
    if(d->vc.state == 1)
    {
        /* the second draw from here: fill.  On the first round's stream, behind everything that may still read the cache of
         * the viewpoint before (first rounds: that stream itself; second rounds: ev_marched), in front of both rounds of this draw. */
        if(!d->vc.d_polar)
        {
            const size_t bytes = (size_t)d->N*d->N*sizeof(hz_polar_t);
            size_t free_b = 0, total_b = 0;
            if(hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2*bytes + ((size_t)4 << 30) ||
               hipMalloc(&d->vc.d_polar, bytes) != hipSuccess)
            {
                (void)hipGetLastError();
                d->vc.d_polar = NULL; d->vc.unavailable = 1;        /* memory is short: every draw computes everything */
                return 0;
            }
            HZ_CHECK(hipEventCreateWithFlags(&d->vc.ev_filled, hipEventDisableTiming));
        }
        HZ_CHECK(hipStreamWaitEvent(d->nstream, d->ev_marched, 0));
        hzk_polar_fill(dim3((unsigned)((d->N + 255)/256), 1024), dim3(256), d->nstream, (const int16_t*)d->d_mosaic, d->vc.d_polar, d->N, p.u);
        HZ_CHECK(hipGetLastError());
        HZ_CHECK(hipEventRecord(d->vc.ev_filled, d->nstream));
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->vc.ev_filled, 0));
        d->vc.state = 2;
    }
    p.vcache = d->vc.d_polar;
    return 0;
}

int hz_draw_impl(hz_dev_t* d, const hz_view_t* view)
{
    hz_params_t p = hz_make_params(d, view);
    const bool prof = d->profiling != 0;
    d->last_view = *view; d->have_view = 1; d->fb_consumed = 0;
    /* what the second rounds of the draws before had to queue (adapt): those of this view whose copy has arrived */
    if(!d->adapt.have_view || memcmp(view, &d->adapt.view, sizeof(*view)) != 0 || d->adapt.col0 != d->col0 || d->adapt.col1 != d->col1)
    {
        d->adapt.view = *view; d->adapt.col0 = d->col0; d->adapt.col1 = d->col1; d->adapt.have_view = 1;
        d->adapt.serial++; d->adapt.items_short = 0; d->adapt.tried_long = 0; d->adapt.long_reach = 0;
    }
    for(int k=0; k<HZ_NFB; k++)
        if(d->adapt.pending[k] && hipEventQuery(d->adapt.ev[k]) == hipSuccess)
        {
            d->adapt.pending[k] = 0;
            const unsigned int items = d->adapt.h_counts[k][1];
            d->adapt.seen_reach = d->adapt.reach_of[k]; d->adapt.seen_records = d->adapt.h_counts[k][0]; d->adapt.seen_items = items;
            if(d->adapt.serial_of[k] != d->adapt.serial) continue;                  /* (about another view) */
            if(!d->adapt.long_of[k])
            {
                d->adapt.items_short = items;
                /* (500 K work items at 16000 x 4000 - the views that gain had 750 K and more, the others 320 K and fewer -, in
                 * proportion to the image's pixels elsewhere: a triangle's pixels, hence what counts as large, go with them) */
                const double hi = d->env.adapt_hi >= 0 ? (double)d->env.adapt_hi : 500000.0*((double)(d->col1 - d->col0)*(double)d->H/64.0e6);
                if(!d->adapt.tried_long && (double)items > hi) d->adapt.long_reach = 1;
            }
            else
            {
                d->adapt.tried_long = 1;
                if(!(d->adapt.items_short && (unsigned long long)items*10ull < (unsigned long long)d->adapt.items_short*7ull)) d->adapt.long_reach = 0;
            }
        }
    (void)hipGetLastError();                /* (hipErrorNotReady from a query is not an error) */
    if(next_framebuffer(d, p) != 0) return -1;
    const int next = d->fbi;
    if(d->raster != HZ_RASTER_SCATTER && vertex_cache(d, p) != 0) return -1;

    const mr_queue_t q = hz_queue_set(d, next);        /* one-round draw, or second round */
    /* large triangles by screen tile instead of by k_big's atomics (HZ_TILES; hz_k_tile.h): in every round (1), or in first
     * and middle rounds - where the large triangles lie hills behind hills and k_big is bound by its atomics - always (2)
     * or when the view is zoomed (the default; decided below).  The tiles merge into the framebuffer with atomic minima:
     * no ownership, nothing else has to wait. */
    bool by_tile_first = d->env.tiles > 0 && d->raster != HZ_RASTER_SCATTER;
    bool near_beside_far = false;                   /* the second round did not wait for the first */
    bool waited_near = false;                       /* ... or it did */
    bool use_hiz = false;                           /* the second round keeps coarse depth (hz_k_hiz.h) */
    bool zoomed_view = false;                       /* a cell at the first round's reach is still HZ_HIZ_MIN_PX pixels wide */

    if(d->raster == HZ_RASTER_SCATTER)
    {
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_free[next], 0));  /* (the framebuffer is free: so are its queue sets) */
        if(prof) { HZ_CHECK(hipEventRecord(d->ev[7], d->stream)); HZ_CHECK(hipEventRecord(d->ev[6], d->stream)); HZ_CHECK(hipEventRecord(d->ev[9], d->stream)); }
        dim3 grid((p.N-1 + SC_CX-1)/SC_CX, (p.N-1 + SC_CY-1)/SC_CY);
        hzk_scatter(grid, dim3(SC_THREADS), d->stream, (const int16_t*)d->d_mosaic, d->d_fb, q, p);
        HZ_CHECK(hipGetLastError());
    }
    else
    {
        const bool two_pass = hz_plan_rounds(d, view, p) == 2;
        const mr_zones_t zn = hz_make_zones(p, two_pass);
        /* sectors and views of less than the full circle: only the strips behind the drawn columns */
        double a0 = 0, a1 = 0;
        const bool listed = d->env.worklists && hz_azimuths_of_columns(p, &a0, &a1) && zn.total < (1 << (32 - MR_ITEM_SX_BITS))
                            && (p.N-1 + MR_COLS-1)/MR_COLS <= (1 << MR_ITEM_SX_BITS);
        p.cull_strips = (p.col0 > 0 || p.col1 < p.W || listed) ? 1 : 0;
        bool fresh_lists = false;
        if(listed)
        {
            hz_listkey_t key;
            memset(&key, 0, sizeof(key));
            key.view = *view; key.col0 = p.col0; key.col1 = p.col1; key.two_pass = two_pass ? 1 : 0;
            key.near_x0 = p.near_x0; key.near_x1 = p.near_x1; key.near_j0 = p.near_j0; key.near_j1 = p.near_j1; key.far_rows = zn.rows[0] | (zn.rows[1] << 8);
            /* the entry that holds this key's lists, or the one not used for the longest time */
            hz_worklists_t* hit = NULL; hz_worklists_t* lru = &d->list_cache[0];
            for(int c=0; c<HZ_LIST_CACHE; c++)
            {
                hz_worklists_t* e = &d->list_cache[c];
                if(e->valid && memcmp(&key, &e->key, sizeof(key)) == 0) { hit = e; break; }
                if((!e->valid && lru->valid) || (e->valid == lru->valid && e->last_used < lru->last_used)) lru = e;
            }
            fresh_lists = hit == NULL;
            d->lists = hit ? hit : lru;
            d->lists->last_used = ++d->lists_clock;
            if(fresh_lists) { d->lists->valid = 0; d->lists->key = key; }
        }
        if(two_pass)
        {
            /* (a sector that draws in two rounds keeps boxes of up to 64 pixels in the marching
             * waves like a whole image does; k_mid was the remedy for one-round sector draws.
             * Gathering rank, 2 / 4 / 8 sectors: 1.19 -> 1.03, 0.82 -> 0.75, 0.69 -> 0.64 ms) */
            p.inline_max = HZ_INLINE_MAX_PIX;
            /* round 1, on its own stream */
            const mr_queue_t qn = hz_queue_set(d, HZ_NFB + next);
            HZ_CHECK(hipStreamWaitEvent(d->nstream, d->ev_free[next], 0));
            {
                /* zoomed: a cell at the first round's reach is still HZ_HIZ_MIN_PX pixels wide.  Such a view's waves append to
                 * the queue of big triangles with nearly every flush: through sixteen counters instead of one (hz_types.h,
                 * HZ_QSHARDS: the seven views of profiles/r5_zoomed_views.txt 8.8 -> 7.4 ms in sum, the slowest 1.69 -> 1.49) */
                const float ppr = p.halfW * p.u.az_ndc_per_rad;
                const float reach = 0.5f*(float)(p.near_j1 - p.near_j0);
                zoomed_view = reach > 0.f && ppr/reach >= HZ_HIZ_MIN_PX;
                /* ... and so do the waves of a draw whose far clip is close (the API's default 40 km: every wave is next to the viewer):
                 * a render of a series at 40 km 0.593 -> 0.571 ms.  Sectors of a whole panorama: no difference (0.166 ms a strip of an
                 * eighth either way), whole panoramas: 1 % slower - one counter (profiles/r5_ab_queue_shards.txt) */
                const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
                p.qshards_log2 = (zoomed_view || cells_to_zfar <= 0.25f*ppr) ? HZ_QSHARDS_LOG2 : 0;
            }
            hz_params_t p1 = p;
            p1.pass = 1; p1.early_z = 0;
            if(fresh_lists)
            {
                /* (both rounds' lists in one walk over the segments: hz_plan.cpp) */
                hz_list_rounds(p1, zn, a0, a1, false, d->list_scratch, *d->list_scratch2);
                if(upload_list(d, 0, d->nstream, *d->list_scratch) != 0) return -1;
            }
            if(prof) HZ_CHECK(hipEventRecord(d->ev[7], d->nstream));
            if(launch_march(d, d->nstream, qn, zn, p1, listed ? d->lists->d_items[0] : NULL, d->lists->n[0]) != 0) return -1;
            {
                /* (zoomed further than coarse depth asks for: a cell at the first round's reach still HZ_TILES_MIN_PX = 35 pixels wide - a 45
                 * degree view of 16000 columns: 40; a 90 degree view, 26, is better off with k_big: 0.92 against 1.08 ms) */
                const float ppr = p.halfW * p.u.az_ndc_per_rad;
                const float reach = 0.5f*(float)(p.near_j1 - p.near_j0);
                if(d->env.tiles < 0 && d->raster != HZ_RASTER_SCATTER && reach > 0.f && ppr/reach >= HZ_TILES_MIN_PX) by_tile_first = true;
                if(by_tile_first && hz_tile_bins(d, HZ_NFB + next) != 0) by_tile_first = false;       /* (no memory for the bins: k_big) */
            }
            if(queue_kernels(d, qn, p1, d->nstream, HZ_NFB + next, by_tile_first, zoomed_view) != 0) return -1;
            if(prof) HZ_CHECK(hipEventRecord(d->ev[6], d->nstream));
            /* The second round waits for the first - unless the chip is idle: a draw that
             * finds the marching kernel of the draw before it finished (a single render, or
             * the first of a series) starts its second round at once, beside its first.  The
             * early depth test then sees fewer occluders and skips less; what it skips is
             * hidden whenever it looks (depths only decrease), so the bytes are the same. */
            const bool busy = hipEventQuery(d->ev_marched) != hipSuccess;
            (void)hipGetLastError();            /* (hipErrorNotReady from the query is not an error) */
            /* Coarse depth (hz_k_hiz.h) for the second round's boxes beyond the 4 x 2 pixels its early depth test
             * reads itself: the tables take in the first round's picture, on its stream, and the second round waits
             * for them.  Zoomed views always (where the first round ends a cell is still ppr/reach pixels wide: 20
             * in a whole panorama, 53 in a 45 degree view of 16000 columns - the reach is capped - and few boxes are
             * small: 2.3 -> 1.0 ms); the others when the second round has to wait for the first anyway, i.e. in a
             * series of renders: the headline 0.98 -> 0.965 ms per render at K = 40, the rough DEM 1.095 -> 1.045,
             * 8000 x 2000 0.283 -> 0.273, the summit +3 % - while a single render keeps its second round beside its
             * first (with tables it would have to wait: 1.26 -> 1.40 ms).  One sweep: more of them beside the second
             * round, one in front of k_big (which tests its chunks of rows against the same tables), one between a
             * nearer and a farther band of the second round - each was measured, none paid. */
            const bool early_z = p.SW >= 2;   /* (the early depth test addresses the framebuffer with 32-bit byte offsets - every framebuffer is below 4 GB, hz_hip_create - and reads pixels in pairs) */
            hz_hiz_t hz = {};
            {
                const bool zoomed = zoomed_view;
                /* (azimuth sectors, round 4 - with k_big's chunk test against the tables: a half gains 8 %, a quarter 6 (strips back to
                 * back 0.273 -> 0.259, 0.250 -> 0.234 ms), an eighth's widest sector 11 and its narrowest loses 4; beside the 8-row
                 * segments narrow sectors get (mr_make_zones) an eighth loses 5: from a sixth of the image on.  profiles/r4_sector_rules.txt) */
                /* (the far clip: at 80 km the tables gain 3 %, at 150 km 4 % (round 4); with the API's 40 km - a second round of a few
                 * thousand waves, all next to the viewer - they changed nothing then (0.602 / 0.607 ms without / with) and cost a
                 * sweep in the first round's chain now: 0.544 / 0.561 ms with them, 0.520 / 0.533 without (round 6, two alternating
                 * sweeps, gpurun_out/r6v): not where even the farthest cell is four pixels wide, unless the view is zoomed) */
                const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
                const bool close_clip = cells_to_zfar <= 0.25f*(p.halfW * p.u.az_ndc_per_rad);
                use_hiz = early_z && (d->env.coarse_depth >= 0 ? d->env.coarse_depth != 0 : (zoomed || (busy && 6*p.SW >= p.W && !close_clip)));
                if(use_hiz && hiz_tables(d, next, p, &hz) != 0) { use_hiz = false; hz = hz_hiz_t{}; }
                /* (The sweep on a stream of its own, so that the next panorama's first round need not queue behind it: tried
                 * in round 4 - with HIP's four hardware queues a fifth stream shares one, nothing changes; with eight
                 * queues the first round's k_big gets the chip earlier and the second round's marching kernel pays for
                 * it: 0.91 -> 1.00 ms per render.  It stays here.) */
                if(use_hiz && hiz_sweep(d, d->nstream, next, p, hz) != 0) return -1;
            }
            HZ_CHECK(hipEventRecord(d->ev_near, d->nstream));
            if(use_hiz || busy)
            {
                HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_near, 0));
                waited_near = true;             /* (the first round itself waited for the framebuffer: no second wait for that below) */
            }
            else
                near_beside_far = true;         /* "drawn" then has to wait for both rounds: see qstream below */
            /* (the early depth test addresses the framebuffer with 32-bit byte offsets) */
            p.pass = 2; p.early_z = early_z ? 1 : 0;
            /* A far clip so close that even the farthest cell is four pixels wide (the API's default 40 km at 16000 columns:
             * 432 cells away, 6 px) leaves the second round a few thousand waves, all of them next to the viewer and all with
             * medium-sized triangles: the kernel is then as long as its longest wave (0.24 ms alone, 0.58 beside the next first
             * round).  Boxes beyond 32 pixels go to k_mid there, which spreads them over the chip: 0.585 -> 0.562 ms per
             * render at 40 km (three alternating triples); where the far field is most of the work the marching waves keep
             * up to 64 pixels (the headline: 0.815 against 0.822 with 32; a far clip of 80 km, 864 cells: 0.612 / 0.617; 150 km:
             * 0.636 / 0.661). */
            {
                const float ppr = p.halfW * p.u.az_ndc_per_rad;
                const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
                if(cells_to_zfar <= 0.25f*ppr) p.inline_max = 32;
                /* (zoomed views keep 64 as well: with 32 the seven views of profiles/r5_zoomed_views.txt take 10.0 instead of 8.8 ms
                 * in sum, with 16 10.8 - their medium triangles are most of their pixels, and k_mid draws them no faster) */
            }
            p.hiz = hz.l1;
            /* ... and its waves read a framebuffer word before the atomic and leave the atomic out where the fragment
             * cannot win (a stale, larger value only costs the atomic) - where the framebuffer is larger than the
             * 256 MB of the chip's last-level cache.  Behind the first round's occluders most fragments of the second
             * lose; an atomic that has to go out to HBM then costs more than the read it saves is worth - 16000x4000
             * (512 MB): 1.06 -> 1.00 ms per render, the rough DEM 1.26 -> 1.17, a 45 degree view 3.9 -> 2.9.  With
             * the framebuffer resident in that cache the atomics are cheap and the read's latency is all that is
             * left: 8000x2000 (128 MB) 0.32 -> 0.38, a quarter-sector of 16000x4000 0.28 -> 0.31; and in a first
             * round or a one-round draw most fragments win (2000x500: 0.144 -> 0.205).  profiles/r3_experiments.json,
             * profiles/r3_scenes.json; k_big's own look before its atomics loses everywhere (HZ_PRETEST=1). */
            p.pretest_march = d->env.pretest_march >= 0 ? d->env.pretest_march
                                                        : ((unsigned long long)p.SW*(unsigned long long)p.H*8ull > (256ull << 20) ? 1 : 0);
        }
        else if(prof) { HZ_CHECK(hipEventRecord(d->ev[7], d->stream)); HZ_CHECK(hipEventRecord(d->ev[6], d->stream)); }
        if(!waited_near) HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_free[next], 0));
        if(fresh_lists)
        {
            if(!two_pass) hz_list_rounds(p, zn, a0, a1, true, NULL, *d->list_scratch2);
            if(upload_list(d, 1, d->stream, *d->list_scratch2) != 0) return -1;
            d->lists->valid = 1;
        }
        if(prof) HZ_CHECK(hipEventRecord(d->ev[9], d->stream));
        if(launch_march(d, d->stream, q, zn, p, listed ? d->lists->d_items[1] : NULL, d->lists->n[1]) != 0) return -1;
    }
    if(prof) HZ_CHECK(hipEventRecord(d->ev[2], d->stream));
    /* the kernels that finish the draw run on qstream, so that the next draw's
     * marching kernel can start beside them */
    HZ_CHECK(hipEventRecord(d->ev_marched, d->stream));
    HZ_CHECK(hipStreamWaitEvent(d->qstream, d->ev_marched, 0));
    /* ev_drawn (below) tells the conversion that the draw is complete: that is the end of the
     * second round's queue kernels only as long as the second round itself waited for the first */
    if(near_beside_far) HZ_CHECK(hipStreamWaitEvent(d->qstream, d->ev_near, 0));
    if(prof) HZ_CHECK(hipEventRecord(d->ev[8], d->qstream));
    /* (adapt: k_big itself leaves the round's queue counters in pinned host memory - no copy in front of ev_drawn) */
    unsigned int* report = NULL;
    if(d->env.adapt == 1 && d->env.near_cells < 0 && p.pass == 2 && d->adapt.h_counts[next] && !d->adapt.pending[next])
    {
        const float ppr = p.halfW * p.u.az_ndc_per_rad;
        if(ppr/(float)HZ_NEAR_CELLS_MAX >= HZ_HIZ_MIN_PX) report = d->adapt.h_counts[next];        /* (a view whose reach has the choice) */
    }
    /* (the second round's large triangles by screen tile too: round 4 measured +5..140 % over the suite's scenes; round 5 on the
     * views whose second round queues the most - summit 1.69 -> 2.09 ms, the 10 degree view 1.54 -> 3.71, south 1.27 -> 1.36: no) */
    if(queue_kernels(d, q, p, d->qstream, next, false, zoomed_view, report) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[3], d->qstream));
    if(report)
    {
        HZ_CHECK(hipEventRecord(d->adapt.ev[next], d->qstream));
        d->adapt.pending[next] = 1; d->adapt.reach_of[next] = (p.near_j1 - p.near_j0)/2;
        d->adapt.long_of[next] = d->adapt.reach_of[next] > HZ_NEAR_CELLS_WIDE ? 1 : 0; d->adapt.serial_of[next] = d->adapt.serial;
    }
    d->last_qshards_log2 = p.qshards_log2;
    d->last_plan[0] = p.pass == 2 ? 2 : 1; d->last_plan[1] = use_hiz ? 1 : 0;
    d->last_plan[2] = p.pass == 2 ? (p.near_j1 - p.near_j0)/2 : 0; d->last_plan[3] = p.cull_strips ? 1 : 0;
    d->last_plan[4] = p.vcache ? 1 : 0;
    HZ_CHECK(hipEventRecord(d->ev_drawn, d->qstream));
    d->have_times = prof ? 1 : 0;
    return 0;
}

/* a reader of the framebuffer finds it consumed (cleared by the conversion
 * that ran before): the draw is repeated - same view, same bytes */
int hz_fb_refill(hz_dev_t* d)
{
    if(!d->fb_consumed) return 0;
    if(!d->have_view) { snprintf(g_last_error, sizeof(g_last_error), "nothing has been drawn yet"); return -1; }
    const hz_view_t v = d->last_view;
    return hz_draw_impl(d, &v);
}

/* the conversion just queued on rstream left the framebuffer of the last draw
 * all ones */
int hz_fb_mark_consumed(hz_dev_t* d)
{
    d->fb_consumed = 1;
    d->fb_used[d->fbi] = 0;
    HZ_CHECK(hipEventRecord(d->ev_free[d->fbi], d->rstream));
    return 0;
}
