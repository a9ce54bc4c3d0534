/* hz_draw.cpp - the host side of the render path: contexts, streams and events, the plan of a draw (rounds, zones, work
 * lists), conversions into DEVICE memory, strips for the multi-GPU gather, pick and the annotator passes - the C-ABI of
 * include/hz_hip.h except the calls that deliver into host memory (hz_hostpath.cpp).  Plain C++ over the HIP runtime API
 * (compiled by g++); every kernel is reached through its launcher in hz_launch.h.
 *
 * HBM layout
 *   mosaic  int16 [N][N], row j = constant latitude (south first), i fastest
 *   fb      uint64 [H][SW]  GL row order (row 0 = bottom), SW = sector width
 *           word = z24<<40 | primitive<<8 | red8, cleared to all ones
 */
#include "hz_dev.h"
#include "hz_fast.h"

#include <time.h>

thread_local char hz_g_last_error[512];
extern "C" const char* hz_hip_last_error(void) { return g_last_error; }

/* ------------------------------------------------------------------------ */
/* host side of the C-ABI                                                    */

/* The tunables of a context (include/hz_hip.h: hz_options_t, hz_hip_set_options).  Every one of them changes how a
 * picture is made, none what is in it.  The environment is a debugging override read HERE and nowhere else, once,
 * when a context is created: HZ_<NAME IN CAPITALS>=value for each field of the struct. */
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static hz_options_t default_options(void)
{
    hz_options_t o;
    o.serial         = 0;
    o.rounds         = 0;
    o.near_cells     = -1;
    o.coarse_depth   = -1;
    o.tiles          = -1;
    o.tile_list      = 0;
    o.adapt          = 1;
    o.adapt_hi       = -1;
    o.pretest_march  = -1;
    o.worklists      = 1;
    o.fast_math      = 1;
    o.resolve_clears = 1;
    o.queue_capacity = 0;
    o.host_dense     = 0;
    o.host_sectors   = 0;
    o.host_times     = 0;
    o.vertex_cache   = 1;
    return o;
}
hz_options_t hz_options_from_env(void)
{
    hz_options_t o = default_options();
    o.serial         = env_int("HZ_SERIAL", o.serial) != 0;
    if(getenv("HZ_TWO_PASS")) o.rounds = env_int("HZ_TWO_PASS", 0) != 0 ? 2 : 1;
    o.near_cells     = env_int("HZ_NEAR_CELLS", o.near_cells);
    if(getenv("HZ_HIZ")) o.coarse_depth = env_int("HZ_HIZ", 0) != 0;
    o.tiles          = env_int("HZ_TILES", o.tiles);
    o.tile_list      = env_int("HZ_TILE_LIST", o.tile_list);
    o.adapt          = env_int("HZ_ADAPT", o.adapt);
    o.adapt_hi       = env_int("HZ_ADAPT_HI", o.adapt_hi);
    if(getenv("HZ_PRETEST_MARCH")) o.pretest_march = env_int("HZ_PRETEST_MARCH", 0) != 0;
    o.worklists      = env_int("HZ_NO_WORKLIST", 0) == 0;
    o.fast_math      = env_int("HZ_NO_FAST_MATH", 0) == 0;
    o.resolve_clears = env_int("HZ_RESOLVE_CLEARS", o.resolve_clears) != 0;
    o.queue_capacity = env_int("HZ_QUEUE_CAPACITY", o.queue_capacity);
    o.host_dense     = env_int("HZ_HOST_DENSE", o.host_dense) != 0;
    o.host_sectors   = env_int("HZ_HOST_SECTORS", o.host_sectors);
    o.host_times     = env_int("HZ_HOST_TIMES", o.host_times) != 0;
    o.vertex_cache   = env_int("HZ_VERTEX_CACHE", o.vertex_cache) != 0;
    return o;
}
#ifdef HZ_EXPERIMENTS                   /* (switches that draw wrong pictures exist in builds with -DHZ_EXPERIMENTS only: tools/experiments.py) */
static hz_experiments_t experiments_from_env(void)
{
    hz_experiments_t e = { env_int("HZ_MARCH_DEBUG", 0), env_int("HZ_EXP_FB_MARCH", 0), env_int("HZ_EXP_FB_BIG", 0) };
    return e;
}
#endif
/* constants that used to be switches (each was swept: DESIGN.md section 4, docs/history/) */
#define HZ_NEAR_PX            20.0f     /* the first round takes the strips whose cells are wider than this many pixels */
#define HZ_TWO_ROUNDS_MIN_PIX 6.0e6     /* two rounds from this many pixels on */
#define HZ_TILES_MIN_PX       35.0f     /* from this width of a cell at the first round's reach on, that round's large triangles go by screen tile */
#define HZ_HIZ_MIN_PX         25.0f     /* "zoomed" = a cell at the first round's reach is at least this wide */


extern "C" int hz_hip_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* everything queued on any of the context's streams is done */
hipError_t hz_sync_all(hz_dev_t* d)
{
    hipError_t rc = hipSuccess;
    hipStream_t all[4] = { d->stream, d->nstream, d->qstream, d->rstream };
    for(int k=0; k<4; k++)
        if(all[k]) { const hipError_t e = hipStreamSynchronize(all[k]); if(e != hipSuccess) rc = e; }
    return rc;
}

extern "C" void hz_hip_destroy(hz_dev_t* d)
{
    if(!d) return;
    hz_device_guard device_guard_(d->device);
    (void)hz_sync_all(d);          /* nothing of this context is still running when its memory goes */
    (void)hipFree(d->d_mosaic);
    for(int i=0; i<HZ_NFB; i++) { (void)hipFree(d->d_fbs[i]); (void)hipFree(d->d_touched[i]); }
    for(int i=0; i<2*HZ_NFB; i++)
    {
        (void)hipFree(d->d_bigrec_s[i]);
        (void)hipFree(d->d_bigitem_s[i]);
        (void)hipFree(d->d_midrec_s[i]);
        (void)hipFree(d->d_clip_s[i]);
        (void)hipFree(d->d_big_counters_s[i]);
        (void)hipFree(d->tiles_s[i].cursor); (void)hipFree(d->tiles_s[i].pairs); (void)hipFree(d->tiles_s[i].state); (void)hipFree(d->tiles_s[i].busy);
    }
    if(d->ev_marched) (void)hipEventDestroy(d->ev_marched);
    if(d->ev_near)    (void)hipEventDestroy(d->ev_near);
    for(int i=0; i<HZ_NFB; i++) (void)hipFree(d->d_hiz[i]);
    for(int c=0; c<HZ_LIST_CACHE; c++)
        for(int k=0; k<HZ_NLISTS; k++)
        {
            hz_worklists_t& wl = d->list_cache[c];
            (void)hipFree(wl.d_items[k]);
            for(int t=0; t<2; t++)
            {
                if(wl.h_items[k][t])   (void)hipHostFree(wl.h_items[k][t]);
                if(wl.ev_copied[k][t]) (void)hipEventDestroy(wl.ev_copied[k][t]);
            }
        }
    delete d->list_scratch;
    (void)hipFree(d->d_texels);
    (void)hipFree(d->d_tanel);
    free(d->h_tanel);
    hz_hostpath_destroy(d);
    (void)hipFree(d->vc.d_polar);
    if(d->vc.ev_filled) (void)hipEventDestroy(d->vc.ev_filled);
    for(int k=0; k<10; k++) if(d->ev[k]) (void)hipEventDestroy(d->ev[k]);
    for(int k=0; k<HZ_NFB; k++) { if(d->adapt.ev[k]) (void)hipEventDestroy(d->adapt.ev[k]); if(d->adapt.h_counts[k]) (void)hipHostFree(d->adapt.h_counts[k]); }
    if(d->ev_drawn)   (void)hipEventDestroy(d->ev_drawn);
    for(int i=0; i<HZ_NFB; i++) if(d->ev_free[i]) (void)hipEventDestroy(d->ev_free[i]);
    if(d->ev_readers) (void)hipEventDestroy(d->ev_readers);
    if(d->ev_tanel)   (void)hipEventDestroy(d->ev_tanel);
    if(d->rstream && d->rstream != d->stream) (void)hipStreamDestroy(d->rstream);
    if(d->qstream && d->qstream != d->stream) (void)hipStreamDestroy(d->qstream);
    if(d->nstream && d->nstream != d->stream) (void)hipStreamDestroy(d->nstream);

    if(d->stream) (void)hipStreamDestroy(d->stream);
    free(d);
}

static mr_queue_t queue_set(const hz_dev_t* d, int k);

/* the tile bins of queue set `set` (hz_k_tile.h): 8 KB of list per 64 x 64 pixel tile of the image - 129 MB per set at
 * 16000 x 4000 -, allocated when a round first draws by tile.  Returns 0, or 1 if there is no memory for them (not tried
 * again: the rounds stay with k_big). */
static int tile_bins(hz_dev_t* d, int set)
{
    tl_bins_t& tb = d->tiles_s[set];
    if(tb.cursor) return 0;
    if(d->tiles_unavailable) return 1;
    const size_t ntiles = (size_t)((d->W + TL_W-1)/TL_W)*((d->H + TL_H-1)/TL_H);
    hipError_t e = hipMalloc(&tb.pairs, ntiles*TL_LIST*sizeof(unsigned int));
    if(e == hipSuccess) e = hipMalloc(&tb.state, 2*sizeof(unsigned int));
    if(e == hipSuccess) e = hipMalloc(&tb.busy, TL_UNITS_PER_TILE*ntiles*sizeof(unsigned int));
    if(e == hipSuccess) e = hipMalloc(&tb.cursor, ntiles*sizeof(unsigned int));
    if(e != hipSuccess)
    {
        (void)hipGetLastError();
        (void)hipFree(tb.pairs); (void)hipFree(tb.state); (void)hipFree(tb.busy); (void)hipFree(tb.cursor);
        tb.pairs = tb.state = tb.busy = tb.cursor = NULL;
        d->tiles_unavailable = 1;
        return 1;
    }
    return 0;
}

/* HZ_INIT_TIMES=1: what a context's set-up is made of, on stderr (tools/init_times.py) */
struct hz_stopwatch
{
    bool on; timespec t0;
    explicit hz_stopwatch(const char* var) : on(getenv(var) && atoi(getenv(var)) != 0) { clock_gettime(CLOCK_MONOTONIC, &t0); }
    void lap(const char* what)
    {
        if(!on) return;
        timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "hz_hip init: %-34s %8.2f ms\n", what, 1e3*(double)(t1.tv_sec - t0.tv_sec) + 1e-6*(double)(t1.tv_nsec - t0.tv_nsec));
        t0 = t1;
    }
};

static int create_impl(hz_dev_t* d)
{
    hz_stopwatch sw("HZ_INIT_TIMES");
    HZ_ON_DEVICE(d);
    sw.lap("first HIP call (runtime, device)");
    d->env = hz_options_from_env();
#ifdef HZ_EXPERIMENTS
    d->exp = experiments_from_env();
#endif
    d->list_scratch = new std::vector<uint32_t>();
    d->lists = &d->list_cache[0];
    HZ_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    if(d->env.serial) d->rstream = d->stream;
    else HZ_CHECK(hipStreamCreateWithFlags(&d->rstream, hipStreamNonBlocking));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_drawn,   hipEventDisableTiming));
    for(int i=0; i<HZ_NFB; i++) HZ_CHECK(hipEventCreateWithFlags(&d->ev_free[i], hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_readers, hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_tanel,   hipEventDisableTiming));
    HZ_CHECK(hipMalloc(&d->d_mosaic, (size_t)d->N*d->N*sizeof(int16_t)));
    sw.lap("streams, events, mosaic");
    d->seg_stride = (d->W + HZ_SEG-1) / HZ_SEG;
    for(int i=0; i<HZ_NFB; i++)
    {
        /* glClear (reference horizonator-lib.c:896): depth = 1.0 -> all-ones words */
        HZ_CHECK(hipMalloc(&d->d_fbs[i], (size_t)d->W*d->H*sizeof(unsigned long long)));
        HZ_CHECK(hipMemsetAsync(d->d_fbs[i], 0xFF, (size_t)d->W*d->H*sizeof(unsigned long long), d->rstream));
        HZ_CHECK(hipMalloc(&d->d_touched[i], (size_t)d->seg_stride*d->H));
        HZ_CHECK(hipMemsetAsync(d->d_touched[i], 0, (size_t)d->seg_stride*d->H, d->rstream));
        HZ_CHECK(hipEventRecord(d->ev_free[i], d->rstream));
        d->fb_used[i] = 0;
    }
    d->fbi = HZ_NFB-1; d->d_fb = d->d_fbs[HZ_NFB-1];
    sw.lap("framebuffers");
    /* queues of triangles too large for the marching wave (k_scatter: for the in-block
     * pass).  The benchmark panorama (16000x4000) produces ~0.3 M records and ~0.4 M work
     * items, a 45 degree view of the same size 1.5 M records (every triangle covers 64
     * times the pixels).  A full queue is correct but slow - the producer then rasterises
     * on the spot, one lane per triangle: the zoomed view took 48 ms instead of 4 with
     * queues of a million records - so the sizes follow the image generously, one record
     * per 16 pixels (HBM is not what this path is short of): 4 M records = 0.4 GB per set
     * for 64 Mpix, 32 K for the smallest contexts.  A first round only sees the triangles
     * of the strips next to the viewer - at most 2*(2r+2)*(2r+126) for a reach of r
     * cells - and never queues medium boxes. */
    {
        const size_t per16 = (size_t)d->W*d->H/16;
        unsigned int rec = per16 > (1u<<24) ? (1u<<24) : per16 < (1u<<15) ? (1u<<15) : (unsigned int)per16;
        d->bigrec_capacity  = rec;
        d->bigitem_capacity = 2*rec;
        d->midrec_capacity  = rec;
        d->clip_capacity    = rec;
        const size_t r = (size_t)(d->env.near_cells > HZ_NEAR_CELLS_MAX ? d->env.near_cells : HZ_NEAR_CELLS_MAX);
        const size_t near_tris = 2*(2*r + 2)*(2*r + 2*MR_COLS);
        d->near_bigrec_capacity  = near_tris < rec ? (unsigned int)near_tris : rec;
        d->near_bigitem_capacity = 2*rec;
        d->near_clip_capacity    = d->near_bigrec_capacity;
        if(d->env.queue_capacity > 0)
            d->bigrec_capacity = d->bigitem_capacity = d->midrec_capacity = d->clip_capacity =
            d->near_bigrec_capacity = d->near_bigitem_capacity = d->near_clip_capacity = (unsigned int)d->env.queue_capacity;
    }
    if(d->env.serial) d->qstream = d->nstream = d->stream;
    else
    {
        HZ_CHECK(hipStreamCreateWithFlags(&d->qstream, hipStreamNonBlocking));
        HZ_CHECK(hipStreamCreateWithFlags(&d->nstream, hipStreamNonBlocking));
    }
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_marched, hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_near,    hipEventDisableTiming));
    for(int i=0; i<2*HZ_NFB; i++)
    {
        const mr_queue_t q = queue_set(d, i);       /* (for the capacities of set i) */
        HZ_CHECK(hipMalloc(&d->d_bigrec_s[i],  (size_t)q.bigrec_capacity*sizeof(hz_bigrec_t)));
        HZ_CHECK(hipMalloc(&d->d_bigitem_s[i], (size_t)q.bigitem_capacity*sizeof(hz_bigitem_t)));
        if(q.midrec_capacity) HZ_CHECK(hipMalloc(&d->d_midrec_s[i], (size_t)q.midrec_capacity*sizeof(hz_rec_t)));
        HZ_CHECK(hipMalloc(&d->d_clip_s[i],    (size_t)q.clip_capacity*sizeof(uint32_t)));
        HZ_CHECK(hipMalloc(&d->d_big_counters_s[i], HZ_NCOUNTERS*sizeof(unsigned int)));
        HZ_CHECK(hipMemset(d->d_big_counters_s[i], 0, HZ_NCOUNTERS*sizeof(unsigned int)));
        /* (the tile bins of the rounds that use them whatever the view - HZ_TILES=1: all, 2: the first rounds' queue sets;
         * by default they are made when a zoomed view first asks for them: tile_bins()) */
        if(d->env.tiles > 0 && i >= HZ_NFB && tile_bins(d, i) != 0) return -1;
    }
    sw.lap("queue sets");
    HZ_CHECK(hipEventRecord(d->ev_drawn, d->qstream));
    HZ_CHECK(hipMalloc(&d->d_tanel, (size_t)d->H*sizeof(float)));
    d->h_tanel = (float*)malloc((size_t)d->H*sizeof(float));
    d->tanel_resident = 0;
    for(int k=0; k<10; k++) HZ_CHECK(hipEventCreate(&d->ev[k]));
    for(int k=0; k<HZ_NFB; k++)
    {
        HZ_CHECK(hipHostMalloc((void**)&d->adapt.h_counts[k], 6*sizeof(unsigned int), hipHostMallocDefault));
        /* (release to system scope: k_big's report lies in pinned HOST memory, and the host reads it when it finds this event complete) */
        HZ_CHECK(hipEventCreateWithFlags(&d->adapt.ev[k], hipEventDisableTiming | hipEventReleaseToSystem));
    }
    sw.lap("the rest");
    return 0;
}

extern "C" hz_dev_t* hz_hip_create(int device, int N, int width, int height)
{
    if(N < 2 || width <= 0 || height <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_create: bad sizes N=%d W=%d H=%d", N, width, height);
        return NULL;
    }
    /* (framebuffer words are addressed with 32-bit byte offsets: hz_fb_min) */
    if((unsigned long long)width*(unsigned long long)height >= (1ull << 29))
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_create: images of up to 2^29 pixels (%d x %d asked for)", width, height);
        fprintf(stderr, "hz_hip: %s\n", g_last_error);
        return NULL;
    }
    hz_dev_t* d = (hz_dev_t*)calloc(1, sizeof(*d));
    if(!d) return NULL;
    d->device = device; d->N = N; d->W = width; d->H = height;
    d->col0 = 0; d->col1 = width;
    d->raster = HZ_RASTER_AUTO;
    if(create_impl(d) != 0) { hz_hip_destroy(d); return NULL; }
    return d;
}

extern "C" int hz_hip_upload_mosaic(hz_dev_t* d, const int16_t* mosaic)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hz_sync_all(d));      /* draws in flight (first rounds run on a stream of their own) still read the old one */
    HZ_CHECK(hipMemcpyAsync(d->d_mosaic, mosaic, (size_t)d->N*d->N*sizeof(int16_t), hipMemcpyHostToDevice, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    d->adapt.have_view = 0;         /* (what the draws of the old terrain had to queue says nothing about the new one) */
    d->vc.state = 0;                /* ... and the vertex cache held the old terrain's heights */
    return 0;
}

extern "C" int hz_hip_download_mosaic(hz_dev_t* d, int16_t* mosaic)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hipMemcpyAsync(mosaic, d->d_mosaic, (size_t)d->N*d->N*sizeof(int16_t), hipMemcpyDeviceToHost, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}


extern "C" int hz_hip_set_sector(hz_dev_t* d, int col0, int col1)
{
    if(col0 < 0 || col1 > d->W || col0 >= col1)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_set_sector: bad sector [%d,%d) of %d", col0, col1, d->W);
        return -1;
    }
    d->col0 = col0; d->col1 = col1;
    return 0;
}

extern "C" int hz_hip_set_raster(hz_dev_t* d, int which)
{
    if(which < HZ_RASTER_AUTO || which > HZ_RASTER_MARCH) return -1;
    d->raster = which;
    return 0;
}

/* texture path: uploads the mosaic of map tiles (texels_bgr: [tex_h][tex_w][3]
 * bytes, B,G,R, row 0 = southern edge) and switches textured resolves on;
 * texels_bgr == NULL with a texture resident only replaces the parameters
 * (they change with every move of the viewer); params == NULL switches the
 * path off again */
extern "C" int hz_hip_set_texture(hz_dev_t* d, const hz_texparams_t* params, const unsigned char* texels_bgr)
{
    HZ_ON_DEVICE(d);
    if(params == NULL) { d->tex_on = 0; return 0; }
    if(params->tex_w <= 0 || params->tex_h <= 0 || params->ntiles_x <= 0 || params->ntiles_y <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_set_texture: empty texture");
        return -1;
    }
    if(texels_bgr != NULL)
    {
        const size_t n = (size_t)params->tex_w*params->tex_h;
        uint32_t* packed = (uint32_t*)malloc(n*sizeof(uint32_t));
        if(!packed) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_set_texture: out of memory"); return -1; }
        for(size_t k=0; k<n; k++)
            packed[k] = (uint32_t)texels_bgr[3*k] | ((uint32_t)texels_bgr[3*k+1] << 8) | ((uint32_t)texels_bgr[3*k+2] << 16);
        HZ_CHECK(hipStreamSynchronize(d->stream));
        HZ_CHECK(hipStreamSynchronize(d->rstream));
        (void)hipFree(d->d_texels); d->d_texels = NULL;
        hipError_t e = hipMalloc(&d->d_texels, n*sizeof(uint32_t));
        if(e == hipSuccess) e = hipMemcpy(d->d_texels, packed, n*sizeof(uint32_t), hipMemcpyHostToDevice);
        free(packed);
        HZ_CHECK(e);
    }
    else if(d->d_texels == NULL || params->tex_w != d->tex.tex_w || params->tex_h != d->tex.tex_h)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_set_texture: no texture of that size is resident");
        return -1;
    }
    d->tex = *params;
    d->tex_on = 1;
    return 0;
}

extern "C" int hz_hip_set_profiling(hz_dev_t* d, int on) { d->profiling = on; return 0; }

extern "C" int hz_hip_get_options(hz_dev_t* d, hz_options_t* o)
{
    if(!d || !o) return -1;
    *o = d->env;
    return 0;
}
extern "C" int hz_hip_set_options(hz_dev_t* d, const hz_options_t* o)
{
    if(!d || !o) return -1;
    HZ_ON_DEVICE(d);
    HZ_CHECK(hz_sync_all(d));
    const int serial = d->env.serial, queue_capacity = d->env.queue_capacity;       /* (streams and queues exist already) */
    d->env = *o;
    d->env.serial = serial; d->env.queue_capacity = queue_capacity;
    for(int c=0; c<HZ_LIST_CACHE; c++) d->list_cache[c].valid = 0;
    d->adapt.have_view = 0;
    return 0;
}
extern "C" void* hz_hip_stream(hz_dev_t* d) { return (void*)d->rstream; }

extern "C" int hz_hip_wait_outputs(hz_dev_t* d, void* stream)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hipEventRecord(d->ev_tanel, d->rstream));          /* a spare untimed event */
    HZ_CHECK(hipStreamWaitEvent((hipStream_t)stream, d->ev_tanel, 0));
    return 0;
}

extern "C" int hz_hip_wait_for(hz_dev_t* d, void* stream)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hipEventRecord(d->ev_tanel, (hipStream_t)stream));
    HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_tanel, 0));
    return 0;
}

/* segment zones of k_march for this view: a cell `r` rows away from the viewer
 * is about ppr/r pixels wide (ppr = pixels per radian of azimuth) */
mr_zones_t hz_make_zones(const hz_params_t& p, bool near_first)
{
    const float ppr = p.halfW * p.u.az_ndc_per_rad;
    const int   ncr = p.N-1;                                /* cell rows */
    const float vj  = p.u.viewer_cell_j;
    const int r2  = (int)(ppr/16.f) + 1;                    /* cells wider than ~16 px: 2-row segments */
    const int r4  = (int)(ppr/4.f) + 1;                     /* ~4 px: 4-row segments                   */
    const int r16 = (int)(ppr/1.f) + 1;                     /* ~1 px: 16-row segments                  */
    auto clampi = [&](float x) { int v = (int)floorf(x); if(v < 0) v = 0; if(v > ncr) v = ncr; return v; };
    mr_zones_t z;
    z.row0[0] = 0;
    z.row0[1] = clampi(vj - (float)r16);
    z.row0[2] = clampi(vj - (float)r4);
    z.row0[3] = clampi(vj - (float)r2);
    z.row0[4] = clampi(vj + (float)r2 + 1.f);
    z.row0[5] = clampi(vj + (float)r4 + 1.f);
    z.row0[6] = clampi(vj + (float)r16 + 1.f);
    z.row0[7] = ncr;
    /* a narrow azimuth sector keeps only a fraction of the waves alive: shorter
     * segments far from the viewer then restore the parallelism (at the price
     * of one extra vertex row per segment) */
    /* (32 rows at most since round 5: the far zones are dispatched last in draws with the early depth test, and with 64 rows
     * their waves - 100 us each - were the kernel's tail: whole panorama, k_march alone 0.623 -> 0.613 ms, a render of a
     * series 0.844 -> 0.837, two alternating runs; 16 rows: 0.616 / 0.842) */
    int far_rows = 64*p.SW/p.W;
    if(far_rows < 16) far_rows = 16;
    if(far_rows > 32) far_rows = 32;
    /* ... and nearer in (cells of 1 to 4 pixels) a narrow sector's kernel was as long as its longest waves: 16 rows of
     * 63 cells with a visible triangle in nearly every lane and a flush per row take 100-170 us (tools/wave_timing.py,
     * HZ_WT_SECTOR=8,0), the whole sector's waves 92 us of the chip - the kernel took 174.  Sectors of less than a sixth
     * of the image cut that zone into 8-row segments: an eighth's strips back to back 0.198 -> 0.169 ms (the widest),
     * 0.156 -> 0.153 (the narrowest); a quarter's and the whole image's waves are many enough to hide their longest
     * (0.273 -> 0.275; profiles/r4_sector_rules.txt). */
    const int z16 = 6*p.SW < p.W ? 8 : 16;          /* (a whole panorama with 8: the kernel +3 %; with 32: -1 % alone, the same in a series - profiles/r5_ab_march_loop.txt (8)) */
    /* A far clip so close that even the farthest cell is four pixels wide (the API's default 40 km at 16000 columns) leaves a
     * second round fewer waves than the chip has slots for, and the kernel is as long as the longest of them (tools/wave_timing.py,
     * HZ_WT_ZFAR=40000: 3.8 K waves, 106 us of work per slot, the longest wave 379 us): two rows to a wave there instead of four -
     * the kernel 0.141 -> 0.105 ms, a render that is waited for 0.896 -> 0.861, a render of a series 0.589 -> 0.583 (its first
     * round is what a series at 40 km waits for); the whole panorama at 600 km with 2: 0.829 -> 0.850. */
    const float cells_to_zfar = sqrtf(p.far_dd)/(p.u.deg_per_cell*111194.9f);
    const int z4 = cells_to_zfar <= 0.25f*ppr ? 2 : 4;
    int rows[MR_NZONES] = { far_rows, z16, z4, 2, z4, z16, far_rows };
    /* Small draws (round 6).  The chip runs 4096 marching waves at a time; a whole image over a small mosaic has too few of them
     * for their lengths to average out - BASELINE's configs[1] (3x3 tiles, 8000 x 2000): 20.6 K waves, 125 us of work per slot,
     * a kernel of 216 us that ends on the far zones' 32-row waves; configs[0] (2000 x 500): 2 K waves, the kernel as long as
     * its longest.  Fewer than 32 K waves: the far zones in 16 rows, the next in 8, two rows to a wave next to the viewer -
     * configs[0] 0.106 -> 0.069 ms per render of a series, 3x3 tiles at 4000 x 1000 0.200 -> 0.188, configs[1] 0.257 -> 0.249
     * (two alternating sweeps over seven settings: profiles/r6_small_images.txt).  Sectors keep their own rules above. */
    if(p.SW == p.W)
    {
        long waves = 0;
        for(int k=0; k<MR_NZONES; k++) waves += (z.row0[k+1] - z.row0[k] + rows[k]-1)/rows[k];
        waves *= (p.N-1 + MR_COLS-1)/MR_COLS;
        if(waves < 32768) { rows[0] = rows[6] = rows[0] < 16 ? rows[0] : 16; rows[1] = rows[5] = rows[1] < 8 ? rows[1] : 8; rows[2] = rows[4] = 2; }
    }
    {
        /* HZ_ZONE_ROWS=far,z16,z4: an experiment's override */
        static const char* e = getenv("HZ_ZONE_ROWS");
        int a = 0, b = 0, c = 0;
        if(e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { rows[0] = rows[6] = a; rows[1] = rows[5] = b; rows[2] = rows[4] = c; }
    }
    /* segment numbers (= blockIdx.y = dispatch order) are handed out to the
     * zones with the longest segments first: the long far-field waves start
     * early and the kernel ends on short ones.  With the early depth test
     * (second round of a two-round draw) the order is from the viewer's row
     * outwards instead, so that the ridges in between are in the framebuffer
     * before the far field is tested against it. */
    const int order_far_first [MR_NZONES] = { 0, 6, 1, 5, 2, 4, 3 };
    const int order_near_first[MR_NZONES] = { 3, 2, 4, 1, 5, 0, 6 };
    z.near_first = near_first ? 1 : 0;
    const int* order = near_first ? order_near_first : order_far_first;
    int seg = 0;
    for(int o=0; o<MR_NZONES; o++)
    {
        const int k = order[o];
        z.rows[k] = rows[k];
        z.seg0[k] = seg;
        const int n = z.row0[k+1] - z.row0[k];
        z.nseg[k] = (n + rows[k]-1)/rows[k];
        seg += z.nseg[k];
    }
    z.total = seg;
    return z;
}

hz_params_t hz_make_params(const hz_dev_t* d, const hz_view_t* v)
{
    hz_params_t p;
    memset(&p, 0, sizeof(p));
    p.u.viewer_cell_i  = v->viewer_cell_i;
    p.u.viewer_cell_j  = v->viewer_cell_j;
    p.u.viewer_z       = v->viewer_z;
    p.u.cos_viewer_lat = v->cos_viewer_lat;
    p.u.deg_per_cell   = v->deg_per_cell;
    p.u.aspect         = v->aspect;
    p.u.znear          = v->znear;
    p.u.zfar           = v->zfar;
    p.u.znear_color    = v->znear_color;
    p.u.zfar_color     = v->zfar_color;
    hz_frame_from_az(v->az_deg0, v->az_deg1, &p.u.az_center, &p.u.az_ndc_per_rad);
    p.halfW = (float)d->W * 0.5f;
    p.halfH = (float)d->H * 0.5f;
    p.N = d->N; p.W = d->W; p.H = d->H;
    p.col0 = d->col0; p.col1 = d->col1; p.SW = d->col1 - d->col0;
    p.touched = d->d_touched[d->fbi]; p.seg_stride = d->seg_stride;
    /* Whole panorama on one GPU: the marching waves keep everything up to 64
     * pixels (cheapest in total).  One azimuth sector of several: the waves next
     * to the viewer become the critical path, so medium boxes are handed to
     * k_mid, which spreads them over the chip (measured: 8 sectors 0.97 -> 0.58 ms). */
    p.inline_max = (p.SW == p.W) ? HZ_INLINE_MAX_PIX : 16;
    p.far_dd = (v->zfar*1.001f)*(v->zfar*1.001f);
    {
        /* the mosaic's corners are its most distant vertices */
        const float e0 = hz_abs(hz_east(&p.u, 0.f)),  e1 = hz_abs(hz_east(&p.u, (float)(p.N-1)));
        const float n0 = hz_abs(hz_north(&p.u, 0.f)), n1 = hz_abs(hz_north(&p.u, (float)(p.N-1)));
        const float em = e0 > e1 ? e0 : e1, nm = n0 > n1 ? n0 : n1;
        p.far_strips = !(nm*nm + em*em <= p.far_dd);
    }
    p.big_min    = HZ_INLINE_MAX_PIX;
    p.z_guard = 1.0f/500.0f + (float)(d->W > d->H ? d->W : d->H) * (1.0f/4194304.0f);
    p.z_hide_k = 1.03f * p.z_guard * 16777215.f;
    /* (0 = no cell is ever culled the short way: tiny images, and sides that do not fit the packed 16-bit pixel boxes) */
    p.quad_max_dx = d->W >= 64 && d->W <= 65535 && d->H <= 65535 ? 256*(d->W/16 - 1) : 0;
#ifdef HZ_EXPERIMENTS
    p.exp_fb[HZ_WHO_MARCH] = d->exp.exp_fb_march; p.exp_fb[HZ_WHO_BIG] = d->exp.exp_fb_big;
    p.debug   = d->exp.march_debug;
#endif
    p.pretest_march = 0;                /* (the second round of a two-round draw may switch it on: draw_impl) */
    p.fast_ok = hzf_draw_ok(&p.u) && d->env.fast_math;
    return p;
}

/* ---- which strips can reach the drawn columns -----------------------------------
 * A draw that does not cover the full circle - one GPU's azimuth sector of a
 * panorama, or a view of less than 360 degrees - needs only the strips of the
 * DEM that lie in the wedge of azimuths behind its columns.  Launching every
 * strip and letting the others leave (k_march's corner test) costs a sector the
 * whole grid's launch plus a vertex transform per wave: 0.3 ms of a 0.4 ms
 * sector at 8 sectors.  So the host lists, per draw, the (segment, strip column)
 * pairs worth launching: per segment - a band of rows, i.e. of north offsets -
 * the east extent of wedge x band, in double precision with margins (4 pixels of
 * azimuth, a cell in every direction).  The list only has to be a superset: the
 * corner test stays in the kernel and decides with the rasteriser's own arithmetic. */

/* east extent [lo,hi] of { t*(sin a, cos a) : t >= 0, a in [a0,a1] } intersected with
 * the band n_lo <= n <= n_hi; a1 - a0 <= pi (convex).  false: empty. */
static bool wedge_band_extent(double a0, double a1, double n_lo, double n_hi, double* lo, double* hi)
{
    const double inf = 1e300;
    double e_lo = inf, e_hi = -inf;
    auto add = [&](double e) { if(e < e_lo) e_lo = e; if(e > e_hi) e_hi = e; };
    if(n_lo <= 0.0 && 0.0 <= n_hi) add(0.0);                   /* the apex */
    const double rays[2] = { a0, a1 };
    for(int r=0; r<2; r++)
    {
        const double s = sin(rays[r]), c = cos(rays[r]);
        if(fabs(c) < 1e-12)
        {
            if(n_lo <= 0.0 && 0.0 <= n_hi) add(s > 0 ? inf : -inf);
            continue;
        }
        const double bounds[2] = { n_lo, n_hi };
        for(int b=0; b<2; b++)
        {
            const double t = bounds[b]/c;
            if(t >= 0.0) add(t*s);
        }
    }
    if(e_lo > e_hi) return false;
    /* unbounded towards east / west: the wedge contains that direction */
    auto contains = [&](double dir) { double x = fmod(dir - a0, 2.0*M_PI); if(x < 0) x += 2.0*M_PI; return x <= a1 - a0; };
    if(contains( 0.5*M_PI)) e_hi =  inf;
    if(contains(-0.5*M_PI)) e_lo = -inf;
    *lo = e_lo; *hi = e_hi;
    return true;
}

/* strip columns [x0,x1] of the band of cell rows jbeg..jend that can reach the
 * azimuths [a0,a1] (radians, a1 - a0 < 2 pi); false: none */
static bool strips_behind_columns(const hz_params_t& p, double a0, double a1, int jbeg, int jend, int nsx, int* x0, int* x1)
{
    const double m_per_cell_n = (double)HZ_REARTH_PI * (double)p.u.deg_per_cell / 180.0;
    const double m_per_cell_e = m_per_cell_n * (double)p.u.cos_viewer_lat;
    const double n_lo = ((double)(jbeg-1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
    const double n_hi = ((double)(jend+1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
    double lo = 0, hi = 0;
    bool any = false;
    const int parts = (a1 - a0 > M_PI) ? 2 : 1;               /* a wedge of more than 180 degrees: two convex halves */
    for(int k=0; k<parts; k++)
    {
        const double b0 = a0 + (a1 - a0)*k/parts, b1 = a0 + (a1 - a0)*(k+1)/parts;
        double l, h;
        if(!wedge_band_extent(b0, b1, n_lo, n_hi, &l, &h)) continue;
        if(!any) { lo = l; hi = h; any = true; }
        else { if(l < lo) lo = l; if(h > hi) hi = h; }
    }
    if(!any) return false;
    /* cells, then strip columns; a strip reaches MR_COLS cells east of its first column */
    double i_lo = lo/m_per_cell_e + (double)p.u.viewer_cell_i - 2.0, i_hi = hi/m_per_cell_e + (double)p.u.viewer_cell_i + 2.0;
    if(!(i_lo > -1e9)) i_lo = -1e9;
    if(!(i_hi <  1e9)) i_hi =  1e9;
    /* strip sx holds the vertex columns sx*MR_COLS .. sx*MR_COLS + MR_COLS: it meets [i_lo, i_hi] iff sx*MR_COLS <= i_hi and
     * sx*MR_COLS + MR_COLS >= i_lo  (round 6: the western end was floor(i_lo/MR_COLS) - 1, one strip too many in every band) */
    int a = (int)ceil(i_lo/(double)MR_COLS) - 1, b = (int)floor(i_hi/(double)MR_COLS);
    if(a < 0) a = 0;
    if(b > nsx-1) b = nsx-1;
    if(a > b) return false;
    *x0 = a; *x1 = b;
    return true;
}

/* the azimuths behind image columns [col0,col1) +- 4 pixels; false: (nearly) the full circle */
bool hz_azimuths_of_columns(const hz_params_t& p, double* a0, double* a1)
{
    const double k = (double)p.u.az_ndc_per_rad, c = (double)p.u.az_center, hw = (double)p.halfW;
    const double lo = c + (((double)p.col0 - 4.0)/hw - 1.0)/k, hi = c + (((double)p.col1 + 4.0)/hw - 1.0)/k;
    if(!(hi - lo < 2.0*M_PI - 1e-3) || !(hi > lo)) return false;
    *a0 = lo; *a1 = hi;
    return true;
}

/* the (segment, strip column) items of one k_march launch (p.pass says which
 * round's) into `out`, in dispatch order: segments as mr_make_zones numbered
 * them, strip columns west to east */
void hz_list_items(const hz_params_t& p, const mr_zones_t& zn, double a0, double a1, std::vector<uint32_t>& out, bool every_strip)
{
    const int nsx = (p.N-1 + MR_COLS-1)/MR_COLS;
    out.clear();
    /* Round 6: the band's east extent above is the test along the patch's own axes; a patch (a strip's cells in a band of
     * rows: a rectangle in east and north, the viewer outside it) can lie inside that extent and still beside the wedge -
     * in the corner between a ray and the band's edge.  What separates a rectangle from a convex wedge besides its own axes
     * are the wedge's two rays: a patch with all four corners on the outer side of one of them is not listed (two cross
     * products per corner; the wedge in two halves where it is wider than 180 degrees, the patch two cells larger east and
     * west, one north and south).  The eight sectors of the benchmark panorama: 1.03 of the grid's waves listed in sum. */
    const double m_per_cell_n = (double)HZ_REARTH_PI * (double)p.u.deg_per_cell / 180.0;
    const double m_per_cell_e = m_per_cell_n * (double)p.u.cos_viewer_lat;
    const int parts = (a1 - a0 > M_PI) ? 2 : 1;
    double ray[3][2];                               /* (sin, cos) of the parts' edges */
    for(int k=0; k<=parts; k++) { const double a = a0 + (a1 - a0)*k/parts; ray[k][0] = sin(a); ray[k][1] = cos(a); }
    for(int seg=0; seg<zn.total; seg++)
    {
        int jbeg, jend;
        mr_segment_rows(zn, seg, &jbeg, &jend);
        int x0 = 0, x1 = nsx-1;
        if(!every_strip && !strips_behind_columns(p, a0, a1, jbeg, jend, nsx, &x0, &x1)) continue;
        const bool near_rows = jbeg < p.near_j1 && jend > p.near_j0;
        const double n_lo = ((double)(jbeg-1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
        const double n_hi = ((double)(jend+1) - (double)p.u.viewer_cell_j) * m_per_cell_n;
        for(int sx=x0; sx<=x1; sx++)
        {
            if(!every_strip)
            {
                const double e_lo = ((double)(sx*MR_COLS - 2) - (double)p.u.viewer_cell_i) * m_per_cell_e;
                const double e_hi = ((double)(sx*MR_COLS + MR_COLS + 2) - (double)p.u.viewer_cell_i) * m_per_cell_e;
                const bool holds_viewer = e_lo <= 0.0 && 0.0 <= e_hi && n_lo <= 0.0 && 0.0 <= n_hi;
                bool reaches = holds_viewer;
                for(int k=0; k<parts && !reaches; k++)
                {
                    /* cross(ray, corner) = sin*n - cos*e: negative = clockwise of the ray.  Inside the part: clockwise of its
                     * first ray (or on it) and counter-clockwise of its second */
                    const double c[4][2] = { { e_lo, n_lo }, { e_hi, n_lo }, { e_lo, n_hi }, { e_hi, n_hi } };
                    bool before_first = true, beyond_second = true;
                    for(int q=0; q<4; q++)
                    {
                        if(!(ray[k][0]*c[q][1] - ray[k][1]*c[q][0] > 0.0))   before_first = false;
                        if(!(ray[k+1][0]*c[q][1] - ray[k+1][1]*c[q][0] < 0.0)) beyond_second = false;
                    }
                    reaches = !before_first && !beyond_second;
                }
                if(!reaches) continue;
            }
            if(p.pass)                          /* (as k_march decides it) */
            {
                const bool near = near_rows && sx >= p.near_x0 && sx <= p.near_x1;
                if((p.pass == 1) != near) continue;
            }
            out.push_back(MR_ITEM(seg, sx));
        }
    }
}

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

static mr_queue_t queue_set(const hz_dev_t* d, int k)
{
    const bool first_round = k >= HZ_NFB;
    mr_queue_t q = { d->d_bigrec_s[k], d->d_bigitem_s[k], d->d_midrec_s[k], d->d_clip_s[k], d->d_big_counters_s[k],
                     first_round ? d->near_bigrec_capacity  : d->bigrec_capacity,
                     first_round ? d->near_bigitem_capacity : d->bigitem_capacity,
                     first_round ? 0u : d->midrec_capacity,
                     first_round ? d->near_clip_capacity : d->clip_capacity };
    return q;
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

/* the draw's plan: one round or two, and which strips are "next to the viewer" */
int hz_plan_rounds(const hz_dev_t* d, const hz_view_t* view, hz_params_t& p)
{
    const int nsx = (p.N-1 + MR_COLS-1)/MR_COLS;
    /* The first round's reach: the cells that are wider than ~20 pixels on screen - a cell r rows
     * from the viewer is about ppr/r pixels wide (ppr = pixels per radian of azimuth), so r = ppr/20:
     * 127 cells for a 16000-wide panorama (where 32..256 were timed: hz_k_march.h), 64 for 8000, 260
     * for 32768, at most HZ_NEAR_CELLS_WIDE - and HZ_NEAR_CELLS_MAX for views zoomed far enough (see there).
     * profiles/r3_scenes.json holds the sweep over the scenes of tools/scenes.py.
     * (A middle round between the two - the ring out to 640 cells with the early test against the first round's
     * tables - was built and measured in round 4: two of seven zoomed views gained, five paid its fixed cost, 10.7 ->
     * 11.0 ms in sum; removed in round 5, profiles/r4_middle_round.txt.) */
    int near_cells = d->env.near_cells;
    if(near_cells < 0)
    {
        const float ppr = p.halfW * p.u.az_ndc_per_rad;
        near_cells = (int)(ppr / HZ_NEAR_PX + 0.5f);
        if(near_cells < 16) near_cells = 16;
        if(near_cells > HZ_NEAR_CELLS_WIDE) near_cells = HZ_NEAR_CELLS_WIDE;
        /* (zoomed even at the long reach: there, if the draws before say so - hz_k_march.h, adapt) */
        if(ppr/(float)HZ_NEAR_CELLS_MAX >= HZ_HIZ_MIN_PX && (d->env.adapt == 2 || (d->env.adapt == 1 && d->adapt.long_reach))) near_cells = HZ_NEAR_CELLS_MAX;
    }
    p.near_x0 = (int)floorf((p.u.viewer_cell_i - (float)near_cells)/(float)MR_COLS);
    p.near_x1 = (int)floorf((p.u.viewer_cell_i + (float)near_cells)/(float)MR_COLS);
    if(p.near_x0 < 0) p.near_x0 = 0;
    if(p.near_x1 > nsx-1) p.near_x1 = nsx-1;
    p.near_j0 = (int)floorf(p.u.viewer_cell_j - (float)near_cells);
    p.near_j1 = (int)ceilf (p.u.viewer_cell_j + (float)near_cells);
    /* Two rounds pay where there is terrain behind the first round's strips to be hidden by them
     * and enough pixels for the second round's early depth test to save work; a small image is
     * faster in one round (three kernel launches less).  Measured over the scenes of tools/scenes.py
     * (profiles/r3_scenes.json, ms per render one round / two rounds): 2000x500 0.143 / 0.167,
     * 4000x1000 0.224 / 0.225, 8000x2000 0.353 / 0.316 (a batch of viewpoints of that size 0.512 /
     * 0.465), 16000x4000 1.33 / 1.12, with the API's 40 km far clip 0.752 / 0.725, 32768x8192 12.5 /
     * 11.1 - so: from 6 Mpix on, and a far clip at least three reaches of the first round away.
     * Azimuth sectors decide by the size of the whole image: their renders overlap just the same. */
    const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
    const bool want_two = d->env.rounds > 0 ? d->env.rounds == 2
                             : ((double)p.W*(double)p.H >= HZ_TWO_ROUNDS_MIN_PIX && cells_to_zfar >= 3.0f*(float)near_cells);
    return want_two && near_cells > 0 && p.near_x1 >= p.near_x0 ? 2 : 1;
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
    hz_stopwatch sw("HZ_DRAW_TIMES");
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
    sw.lap("draw: params, adapt");
    if(next_framebuffer(d, p) != 0) return -1;
    sw.lap("draw: next framebuffer");
    const int next = d->fbi;
    if(d->raster != HZ_RASTER_SCATTER && vertex_cache(d, p) != 0) return -1;

    const mr_queue_t q = queue_set(d, next);        /* one-round draw, or second round */
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
            const mr_queue_t qn = queue_set(d, HZ_NFB + next);
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
                hz_list_items(p1, zn, a0, a1, *d->list_scratch);
                if(upload_list(d, 0, d->nstream, *d->list_scratch) != 0) return -1;
            }
            sw.lap("draw: first list");
            if(prof) HZ_CHECK(hipEventRecord(d->ev[7], d->nstream));
            if(launch_march(d, d->nstream, qn, zn, p1, listed ? d->lists->d_items[0] : NULL, d->lists->n[0]) != 0) return -1;
            sw.lap("draw: first march launched");
            {
                /* (zoomed further than coarse depth asks for: a cell at the first round's reach still HZ_TILES_MIN_PX = 35 pixels wide - a 45
                 * degree view of 16000 columns: 40; a 90 degree view, 26, is better off with k_big: 0.92 against 1.08 ms) */
                const float ppr = p.halfW * p.u.az_ndc_per_rad;
                const float reach = 0.5f*(float)(p.near_j1 - p.near_j0);
                if(d->env.tiles < 0 && d->raster != HZ_RASTER_SCATTER && reach > 0.f && ppr/reach >= HZ_TILES_MIN_PX) by_tile_first = true;
                if(by_tile_first && tile_bins(d, HZ_NFB + next) != 0) by_tile_first = false;       /* (no memory for the bins: k_big) */
            }
            if(queue_kernels(d, qn, p1, d->nstream, HZ_NFB + next, by_tile_first, zoomed_view) != 0) return -1;
            sw.lap("draw: first queue kernels");
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
                /* (whatever the far clip: with the API's 40 km the tables change nothing - 0.602 / 0.607 ms without / with,
                 * three alternating pairs -, at 80 km they gain 3 %, at 150 km 4 %: round 4) */
                use_hiz = early_z && (d->env.coarse_depth >= 0 ? d->env.coarse_depth != 0 : (zoomed || (busy && 6*p.SW >= p.W)));
                if(use_hiz && hiz_tables(d, next, p, &hz) != 0) { use_hiz = false; hz = hz_hiz_t{}; }
                /* (The sweep on a stream of its own, so that the next panorama's first round need not queue behind it: tried
                 * in round 4 - with HIP's four hardware queues a fifth stream shares one, nothing changes; with eight
                 * queues the first round's k_big gets the chip earlier and the second round's marching kernel pays for
                 * it: 0.91 -> 1.00 ms per render.  It stays here.) */
                if(use_hiz && hiz_sweep(d, d->nstream, next, p, hz) != 0) return -1;
            }
            sw.lap("draw: coarse depth");
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
            hz_list_items(p, zn, a0, a1, *d->list_scratch);
            if(upload_list(d, 1, d->stream, *d->list_scratch) != 0) return -1;
            d->lists->valid = 1;
        }
        sw.lap("draw: second list");
        if(prof) HZ_CHECK(hipEventRecord(d->ev[9], d->stream));
        if(launch_march(d, d->stream, q, zn, p, listed ? d->lists->d_items[1] : NULL, d->lists->n[1]) != 0) return -1;
        sw.lap("draw: second march launched");
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
    sw.lap("draw: the rest");
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

/* The per-row tan(elevation) table only changes with the azimuth extents: a
 * table equal to the resident one is not sent again (a host->device copy from
 * pageable memory would otherwise stall the host on the stream once per render) */
int hz_upload_tanel(hz_dev_t* d, const float* tanel)
{
    if(!tanel) { snprintf(g_last_error, sizeof(g_last_error), "a tanel table is required"); return -1; }
    const size_t bytes = (size_t)d->H*sizeof(float);
    if(d->tanel_resident && memcmp(d->h_tanel, tanel, bytes) == 0) return 0;
    /* a different table (the azimuth extents changed): nothing queued on either
     * stream may still read the old one, and both streams must see the new one */
    HZ_CHECK(hz_sync_all(d));
    memcpy(d->h_tanel, tanel, bytes);
    HZ_CHECK(hipMemcpy(d->d_tanel, d->h_tanel, bytes, hipMemcpyHostToDevice));
    d->tanel_resident = 1;
    return 0;
}

/* conversions of the last draw run on rstream, behind that draw */
int hz_rstream_after_draw(hz_dev_t* d)
{
    HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_drawn, 0));
    return 0;
}

/* the conversion of the last draw into DEVICE buffers; nbands > 1 (wide path only): in that many bands of
 * rows, top first, ev_band[k] recorded on rstream behind band k - copy_out lets the first band's bytes leave
 * for the host while the others are still being converted */
int hz_resolve_impl(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                        unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int nbands, hipEvent_t* ev_band, int* band_rows)
{
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(ranges)
    {
        if(!tanel)
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve: ranges requested without a tanel table");
            return -1;
        }
        if(hz_upload_tanel(d, tanel) != 0) return -1;
    }
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    /* the textured resolve reads the framebuffer after this kernel: no fused clear then */
    const bool clears = d->env.resolve_clears && !(d->tex_on && bgr);
    unsigned int* const qa = d->d_big_counters_s[d->fbi], * const qb = d->d_big_counters_s[HZ_NFB + d->fbi];    /* emptied with the framebuffer */
    const bool wide = (SW % 4) == 0 && (((uintptr_t)bgr | (uintptr_t)ranges | (uintptr_t)index | (uintptr_t)z24 | (uintptr_t)d->d_fb) & 15u) == 0;
    if(!wide || (d->tex_on && bgr) || nbands < 1) nbands = 1;
    if(nbands > d->H) nbands = d->H;
    if(band_rows) *band_rows = (d->H + nbands-1)/nbands;
    if(wide)
    {
        const int rows = (d->H + nbands-1)/nbands;
        for(int k=0; k<nbands; k++)
        {
            const int yo0 = k*rows, yo1 = (k+1)*rows < d->H ? (k+1)*rows : d->H;
            const dim3 grid((unsigned)((SW/4 + 255)/256), (unsigned)(yo1 - yo0 < 2048 ? yo1 - yo0 : 2048));
            if(clears)
                hzk_resolve4(true, grid, dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride, k == 0 ? qa : (unsigned int*)NULL, qb, yo0, yo1, 1);
            else
                hzk_resolve4(false, grid, dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride, (unsigned int*)NULL, (unsigned int*)NULL, yo0, yo1, 1);
            HZ_CHECK(hipGetLastError());
            if(ev_band) HZ_CHECK(hipEventRecord(ev_band[k], d->rstream));
        }
    }
    else
    {
        if(clears)
            hzk_resolve(true, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, qa, qb);
        else
            hzk_resolve(false, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, qa, qb);
        HZ_CHECK(hipGetLastError());
    }
    if(clears && hz_fb_mark_consumed(d) != 0) return -1;
    if(d->tex_on && bgr)
    {
        /* reference fragment.glsl:17-22 instead of :15-16 for the terrain pixels */
        const hz_params_t p = hz_make_params(d, view);
        size_t nchunks = (npix + TX_CHUNK-1)/TX_CHUNK;
        if(nchunks > 256*64) nchunks = 256*64;
        hzk_shade_tex(dim3((unsigned)nchunks), dim3(64), d->rstream, (const unsigned long long*)d->d_fb, (const int16_t*)d->d_mosaic, (const uint32_t*)d->d_texels, d->tex, bgr, p);
        HZ_CHECK(hipGetLastError());
    }
    if(!wide && ev_band) HZ_CHECK(hipEventRecord(ev_band[0], d->rstream));
    else if(wide && nbands == 1 && ev_band && d->tex_on && bgr) HZ_CHECK(hipEventRecord(ev_band[0], d->rstream));   /* (behind the shading kernel) */
    if(prof)
    {
        HZ_CHECK(hipEventRecord(d->ev[5], d->rstream));
        d->have_times = 2;
    }
    return nbands;
}

extern "C" int hz_hip_resolve(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                              unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    return hz_resolve_impl(d, view, tanel, bgr, ranges, index, z24, 1, NULL, NULL) < 0 ? -1 : 0;
}

/* the draw's result as one word per pixel, z24<<8 | red8, top row first:
 * what a rank sends to the gathering rank (d_packed: DEVICE, [H][sector width]) */
extern "C" int hz_hip_pack(hz_dev_t* d, uint32_t* d_packed)
{
    HZ_ON_DEVICE(d);
    if(d->tex_on)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_pack: packed strips carry the shade only, not a textured colour");
        return -1;
    }
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    if(d->env.resolve_clears)
    {
        hzk_pack(true, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, d_packed, SW, d->H, d->d_big_counters_s[d->fbi], d->d_big_counters_s[HZ_NFB + d->fbi]);
        HZ_CHECK(hipGetLastError());
        if(hz_fb_mark_consumed(d) != 0) return -1;
    }
    else
        hzk_pack(false, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, d_packed, SW, d->H, (unsigned int*)NULL, (unsigned int*)NULL);
    HZ_CHECK(hipGetLastError());
    if(prof) { HZ_CHECK(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
    return 0;
}

/* The readback conversion on packed words, wherever they were drawn: columns
 * [0,ncols) of d_packed[H][stride] become columns [out_col0, out_col0+ncols) of
 * the FULL-width outputs d_bgr[H][W][3] / d_ranges[H][W] (either may be NULL). */
extern "C" int hz_hip_resolve_packed(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     const uint32_t* d_packed, int stride, int ncols, int out_col0,
                                     unsigned char* d_bgr, float* d_ranges)
{
    HZ_ON_DEVICE(d);
    if(ncols <= 0 || stride < ncols || out_col0 < 0 || out_col0 + ncols > d->W)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_packed: columns [%d,%d) do not fit a %d-wide image",
                 out_col0, out_col0 + ncols, d->W);
        return -1;
    }
    if(d_ranges && hz_upload_tanel(d, tanel) != 0) return -1;
    const size_t npix = (size_t)ncols*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    hzk_resolve_packed(dim3((unsigned)nblocks), dim3(256), d->rstream, d_packed, stride, ncols, (const float*)d->d_tanel, d_bgr, d_ranges, d->W, out_col0, d->H, view->znear, view->zfar);
    HZ_CHECK(hipGetLastError());
    return 0;
}

/* the draw's result as a sparse strip (see k_pack_sparse): d_out must hold
 * 1 + H + H*mask_stride + H*(sector width) words; the first word ends up as the
 * number of terrain pixels T, and only the first 1 + H + H*mask_stride + T words
 * carry information.  mask_stride >= ceil(sector width / 32). */
extern "C" int hz_hip_pack_sparse(hz_dev_t* d, uint32_t* d_out, int mask_stride)
{
    HZ_ON_DEVICE(d);
    const int SW = d->col1 - d->col0;
    if(d->tex_on || mask_stride < (SW + 31)/32 || SW > SP_MAXIT*256)
    {
        snprintf(g_last_error, sizeof(g_last_error), d->tex_on ? "hz_hip_pack_sparse: strips carry the shade only, not a textured colour"
                                                    : SW > SP_MAXIT*256 ? "hz_hip_pack_sparse: sectors of up to 65536 columns"
                                                               : "hz_hip_pack_sparse: mask stride too small");
        return -1;
    }
    const bool prof = d->profiling != 0;
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    HZ_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint32_t), d->rstream));
    if(d->env.resolve_clears)
    {
        hzk_pack_sparse(true, dim3((unsigned)((d->H + SP_WAVES-1)/SP_WAVES)), dim3(64*SP_WAVES), d->rstream, d->d_fb, d_out, SW, d->H, mask_stride, d->d_touched[d->fbi], d->seg_stride, d->d_big_counters_s[d->fbi], d->d_big_counters_s[HZ_NFB + d->fbi]);
        HZ_CHECK(hipGetLastError());
        if(hz_fb_mark_consumed(d) != 0) return -1;
    }
    else
        hzk_pack_sparse(false, dim3((unsigned)((d->H + SP_WAVES-1)/SP_WAVES)), dim3(64*SP_WAVES), d->rstream, d->d_fb, d_out, SW, d->H, mask_stride, d->d_touched[d->fbi], d->seg_stride, (unsigned int*)NULL, (unsigned int*)NULL);
    HZ_CHECK(hipGetLastError());
    if(prof) { HZ_CHECK(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
    return 0;
}

extern "C" int hz_hip_resolve_sparse_strips(hz_dev_t* d, const hz_view_t* view, const float* tanel, int nstrips,
                                            const uint32_t* const* d_in, int mask_stride, const int* ncols, const int* out_col0,
                                            unsigned char* d_bgr, float* d_ranges)
{
    HZ_ON_DEVICE(d);
    if(nstrips < 0 || (nstrips > 0 && (!d_in || !ncols || !out_col0)))
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse_strips: bad arguments");
        return -1;
    }
    for(int k=0; k<nstrips; k++)
        if(ncols[k] < 0 || (ncols[k] > 0 && (mask_stride < (ncols[k] + 31)/32 || out_col0[k] < 0 || out_col0[k] + ncols[k] > d->W || !d_in[k])))
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse_strips: columns [%d,%d) of strip %d do not fit a %d-wide image",
                     out_col0[k], out_col0[k] + ncols[k], k, d->W);
            return -1;
        }
    if(d_ranges && hz_upload_tanel(d, tanel) != 0) return -1;
    for(int k0=0; k0<nstrips; k0+=HZ_MAX_STRIPS)
    {
        hz_strips_t st;
        memset(&st, 0, sizeof(st));
        const int n = nstrips - k0 < HZ_MAX_STRIPS ? nstrips - k0 : HZ_MAX_STRIPS;
        for(int k=0; k<n; k++) { st.in[k] = d_in[k0+k]; st.ncols[k] = ncols[k0+k]; st.col0[k] = out_col0[k0+k]; }
        hzk_resolve_sparse(dim3((unsigned)d->H, (unsigned)n), dim3(256), d->rstream, st, mask_stride, (const float*)d->d_tanel, d_bgr, d_ranges, d->W, d->H, view->znear, view->zfar);
        HZ_CHECK(hipGetLastError());
    }
    return 0;
}

extern "C" int hz_hip_resolve_sparse(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     const uint32_t* d_in, int mask_stride, int ncols, int out_col0,
                                     unsigned char* d_bgr, float* d_ranges)
{
    if(ncols <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse: columns [%d,%d) do not fit a %d-wide image",
                 out_col0, out_col0 + ncols, d->W);
        return -1;
    }
    return hz_hip_resolve_sparse_strips(d, view, tanel, 1, &d_in, mask_stride, &ncols, &out_col0, d_bgr, d_ranges);
}


extern "C" int hz_hip_read_depth(hz_dev_t* d, int x, int y, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(x < d->col0 || x >= d->col1 || y < 0 || y >= d->H)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_read_depth: (%d,%d) outside the drawn sector", x, y);
        return -1;
    }
    const int SW = d->col1 - d->col0;
    unsigned long long key = 0;
    HZ_CHECK(hipMemcpyAsync(&key, &d->d_fb[(size_t)(d->H-1-y)*SW + (x - d->col0)], sizeof(key),
                            hipMemcpyDeviceToHost, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    *z24 = (uint32_t)(key >> 40);
    return 0;
}


extern "C" int hz_hip_link_cells(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                 const float* sin_az, const float* cos_az, const double* cos_el,
                                 double viewer_lat, double cos_viewer_lat, double viewer_lon,
                                 int cell_w, int cell_h, int nx, int ny, float* lat, float* lon)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(d->col0 != 0 || d->col1 != d->W || cell_w <= 0 || cell_h <= 0 || nx <= 0 || ny <= 0 || !sin_az || !cos_az || !cos_el)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells: needs a full-width context, positive sizes and the three tables");
        return -1;
    }
    if(hz_upload_tanel(d, tanel) != 0) return -1;
    const size_t n = (size_t)nx*ny;
    /* one allocation: lat, lon, then the tables */
    const size_t bytes = 2*n*sizeof(float) + 2*(size_t)nx*sizeof(float) + (size_t)ny*sizeof(double) + 16;
    unsigned char* buf = NULL;
    HZ_CHECK(hipMalloc(&buf, bytes));
    double* d_cos_el = (double*)buf;
    float* d_lat = (float*)(d_cos_el + ny), * d_lon = d_lat + n, * d_sin = d_lon + n, * d_cos = d_sin + nx;
    int rc = 0;
    if(hipMemcpyAsync(d_cos_el, cos_el, (size_t)ny*sizeof(double), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(d_sin, sin_az, (size_t)nx*sizeof(float), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(d_cos, cos_az, (size_t)nx*sizeof(float), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0)
    {
        hzk_link_cells(dim3((unsigned)((n + 255)/256)), dim3(256), d->stream, (const unsigned long long*)d->d_fb, (const float*)d->d_tanel, (const float*)d_sin, (const float*)d_cos, (const double*)d_cos_el, d_lat, d_lon, d->W, d->H, cell_w, cell_h, nx, ny, view->znear, view->zfar, viewer_lat, cos_viewer_lat, viewer_lon);
        if(hipGetLastError() != hipSuccess) rc = -1;
    }
    if(rc == 0 && hipMemcpyAsync(lat, d_lat, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(lon, d_lon, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(buf);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells failed");
    return rc;
}

extern "C" int hz_hip_poi_visibility(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     int cut_off_bottom_px, const hz_poi_proj_t* proj, int npois,
                                     unsigned char* visible, float* label_x, float* label_y)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(d->col0 != 0 || d->col1 != d->W || npois < 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility: needs a full-width context");
        return -1;
    }
    if(npois == 0) return 0;
    if(hz_upload_tanel(d, tanel) != 0) return -1;
    hz_poi_proj_t* d_proj = NULL; unsigned char* d_vis = NULL; float *d_x = NULL, *d_y = NULL;
    HZ_CHECK(hipMalloc(&d_proj, (size_t)npois*sizeof(hz_poi_proj_t)));
    HZ_CHECK(hipMalloc(&d_vis, (size_t)npois));
    HZ_CHECK(hipMalloc(&d_x, (size_t)npois*sizeof(float)));
    HZ_CHECK(hipMalloc(&d_y, (size_t)npois*sizeof(float)));
    int rc = 0;
    if(hipMemcpyAsync(d_proj, proj, (size_t)npois*sizeof(hz_poi_proj_t), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0)
    {
        hzk_poi(dim3((unsigned)((npois + 255)/256)), dim3(256), d->stream, (const unsigned long long*)d->d_fb, (const float*)d->d_tanel, (const hz_poi_proj_t*)d_proj, npois, d_vis, d_x, d_y, d->W, d->H, d->H - cut_off_bottom_px, view->znear, view->zfar);
        if(hipGetLastError() != hipSuccess) rc = -1;
    }
    if(rc == 0 && hipMemcpyAsync(visible, d_vis, (size_t)npois, hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_x, d_x, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_y, d_y, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(d_proj); (void)hipFree(d_vis); (void)hipFree(d_x); (void)hipFree(d_y);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility failed");
    return rc;
}

/* what the last draw was (bench.py records it beside every timing, tests assert on it) - out[0] rounds (1 / 2),
 * [1] its second round kept coarse depth (hz_k_hiz.h), [2] the first round's reach in cells (0: one round), [3] only
 * the strips behind the drawn columns were launched (sectors, views of less than the full circle) */
extern "C" int hz_hip_last_queue_counts(hz_dev_t* d, unsigned int* out)
{
    if(!d || !out) return -1;
    out[0] = (unsigned int)d->adapt.seen_reach; out[1] = d->adapt.seen_records; out[2] = d->adapt.seen_items; out[3] = (unsigned int)d->adapt.long_reach;
    return 0;
}

extern "C" int hz_hip_last_plan(hz_dev_t* d, int* out)
{
    if(!d || !out) return -1;
    for(int k=0; k<5; k++) out[k] = d->last_plan[k];
    return 0;
}


extern "C" int hz_hip_sync(hz_dev_t* d)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hz_sync_all(d));
    return 0;
}

extern "C" int hz_hip_last_times(hz_dev_t* d, hz_times_t* t)
{
    memset(t, 0, sizeof(*t));
    if(!d->have_times) return -1;
    HZ_ON_DEVICE(d);
    HZ_CHECK(hz_sync_all(d));
    /* clear_ms is the clear this draw queued: that of the OTHER framebuffer, which runs on
     * rstream beside the draw.  total_ms is the sum of the stages, not a latency. */
    HZ_CHECK(hipEventElapsedTime(&t->clear_ms,  d->ev[0], d->ev[1]));
    HZ_CHECK(hipEventElapsedTime(&t->near_ms,   d->ev[7], d->ev[6]));
    HZ_CHECK(hipEventElapsedTime(&t->raster_ms, d->ev[9], d->ev[2]));
    HZ_CHECK(hipEventElapsedTime(&t->big_ms,    d->ev[8], d->ev[3]));
    if(d->have_times == 2)
        HZ_CHECK(hipEventElapsedTime(&t->resolve_ms, d->ev[4], d->ev[5]));
    t->total_ms = t->clear_ms + t->near_ms + t->raster_ms + t->big_ms + t->resolve_ms;
    return 0;
}

