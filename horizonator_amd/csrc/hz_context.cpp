/* hz_context.cpp - a context of the HIP render path (hz_dev, hz_dev.h): options, streams and events, the memory a draw
 * needs (mosaic, framebuffers, queues), the DEM and texture uploads, what callers ask about the last draw - the part of the
 * C-ABI of include/hz_hip.h that is about a context rather than about a picture.  Plain C++ over the HIP runtime API
 * (compiled by g++).  The plan of a draw: hz_plan.cpp; the draw: hz_draw.cpp; conversions and readers of a finished
 * draw: hz_convert.cpp; results into host memory: hz_hostpath.cpp; the DEM's tiles: hz_ingest.cpp.
 *
 * HBM layout
 *   mosaic  int16 [N][N], row j = constant latitude (south first), i fastest
 *   fb      uint64 [H][SW]  GL row order (row 0 = bottom), SW = sector width
 *           word = z24<<40 | primitive<<8 | red8, cleared to all ones
 */
#include "hz_dev.h"

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
    delete d->list_scratch; delete d->list_scratch2;
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

/* the tile bins of queue set `set` (hz_k_tile.h): 8 KB of list per 64 x 64 pixel tile of the image - 129 MB per set at
 * 16000 x 4000 -, allocated when a round first draws by tile.  Returns 0, or 1 if there is no memory for them (not tried
 * again: the rounds stay with k_big). */
int hz_tile_bins(hz_dev_t* d, int set)
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

/* the queues of framebuffer set k (first rounds' sets: k >= HZ_NFB) */
mr_queue_t hz_queue_set(const hz_dev_t* d, int k)
{
    const bool first_round = k >= HZ_NFB;
    mr_queue_t q = { d->d_bigrec_s[k], d->d_bigitem_s[k], d->d_midrec_s[k], d->d_clip_s[k], d->d_big_counters_s[k],
                     first_round ? d->near_bigrec_capacity  : d->bigrec_capacity,
                     first_round ? d->near_bigitem_capacity : d->bigitem_capacity,
                     first_round ? 0u : d->midrec_capacity,
                     first_round ? d->near_clip_capacity : d->clip_capacity };
    return q;
}


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
    d->list_scratch2 = new std::vector<uint32_t>();
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
        const mr_queue_t q = hz_queue_set(d, i);       /* (for the capacities of set i) */
        HZ_CHECK(hipMalloc(&d->d_bigrec_s[i],  (size_t)q.bigrec_capacity*sizeof(hz_bigrec_t)));
        HZ_CHECK(hipMalloc(&d->d_bigitem_s[i], (size_t)q.bigitem_capacity*sizeof(hz_bigitem_t)));
        if(q.midrec_capacity) HZ_CHECK(hipMalloc(&d->d_midrec_s[i], (size_t)q.midrec_capacity*sizeof(hz_rec_t)));
        HZ_CHECK(hipMalloc(&d->d_clip_s[i],    (size_t)q.clip_capacity*sizeof(uint32_t)));
        HZ_CHECK(hipMalloc(&d->d_big_counters_s[i], HZ_NCOUNTERS*sizeof(unsigned int)));
        HZ_CHECK(hipMemset(d->d_big_counters_s[i], 0, HZ_NCOUNTERS*sizeof(unsigned int)));
        /* (the tile bins of the rounds that use them whatever the view - HZ_TILES=1: all, 2: the first rounds' queue sets;
         * by default they are made when a zoomed view first asks for them: tile_bins()) */
        if(d->env.tiles > 0 && i >= HZ_NFB && hz_tile_bins(d, i) != 0) return -1;
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
