/* hz_dev.h - the context (hz_dev) and what the host translation units of the HIP side share: hz_context.cpp (contexts,
 * streams, memory, options), hz_plan.cpp (the plan of a draw: zones, rounds, work lists - host arithmetic only), hz_draw.cpp
 * (a draw), hz_convert.cpp (conversions into device memory, strips, pick and the annotator passes), hz_hostpath.cpp (results
 * into the caller's host memory), hz_ingest.cpp (the DEM's tiles) and the diagnostics of libhorizonator_selftest.so.
 * Host code only - plain C++ over the HIP runtime API; the kernels are reached through hz_launch.h. */
#pragma once

#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>

#include "hz_hip.h"
#include "hz_types.h"
#include "hz_launch.h"

/* ------------------------------------------------------------------------ */
/* errors                                                                    */

extern thread_local char hz_g_last_error[512];    /* per thread: contexts may live on different threads (defined in hz_context.cpp) */
#define g_last_error hz_g_last_error

#define HZ_CHECK(call)                                                        \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if(e_ != hipSuccess)                                                  \
        {                                                                     \
            snprintf(g_last_error, sizeof(g_last_error), "%s:%d %s -> %s",    \
                     __FILE__, __LINE__, #call, hipGetErrorString(e_));       \
            fprintf(stderr, "hz_hip: %s\n", g_last_error);                    \
            return -1;                                                        \
        }                                                                     \
    } while(0)

/* every entry point works on its context's device and leaves the caller's
 * current device as it found it (a torch process has its own idea of it) */
struct hz_device_guard
{
    int prev, dev; bool ok;
    explicit hz_device_guard(int device) : prev(-1), dev(device), ok(true)
    {
        if(hipGetDevice(&prev) != hipSuccess) prev = -1;
        if(prev != dev)
        {
            const hipError_t e = hipSetDevice(dev);
            if(e != hipSuccess)
            {
                ok = false;
                snprintf(g_last_error, sizeof(g_last_error), "hipSetDevice(%d) -> %s", dev, hipGetErrorString(e));
                fprintf(stderr, "hz_hip: %s\n", g_last_error);
            }
        }
    }
    ~hz_device_guard() { if(ok && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
#define HZ_ON_DEVICE(d) hz_device_guard device_guard_((d)->device); if(!device_guard_.ok) return -1

#ifdef HZ_EXPERIMENTS                   /* (switches that draw wrong pictures exist in builds with -DHZ_EXPERIMENTS only: tools/experiments.py) */
struct hz_experiments_t { int march_debug, exp_fb_march, exp_fb_big; };
#endif

/* what decides a draw's work lists */
struct hz_listkey_t
{
    hz_view_t view; int col0, col1, two_pass, near_x0, near_x1, near_j0, near_j1, far_rows;
};
/* the work lists of sector draws (see strips_behind_columns): [0] first round, [1] second or only round */
#define HZ_NLISTS 2
struct hz_worklists_t
{
    uint32_t*    d_items[HZ_NLISTS];
    uint32_t*    h_items[HZ_NLISTS][2]; /* pinned; two per list, taken in turn: the host only waits for the copy two lists back */
    size_t       cap[HZ_NLISTS];
    unsigned int n[HZ_NLISTS];
    hipEvent_t   ev_copied[HZ_NLISTS][2];
    int          turn[HZ_NLISTS];
    int          valid;                 /* the resident lists are those of `key` */
    unsigned int last_used;             /* (hz_dev::lists_clock when a draw last took this entry) */
    hz_listkey_t key;
};
/* A context keeps the lists of the last HZ_LIST_CACHE (view, sector) pairs it drew: a call that delivers into host memory
 * draws its panorama in several sectors (hz_hostpath.cpp) and a rank of a multi-GPU job may be handed another sector -
 * repeated draws of the same few sectors then find their lists resident instead of building and uploading them again. */
#define HZ_LIST_CACHE 8

struct hz_dev
{
    int device;
    hz_options_t env;                   /* (the options; "env" from the time when the environment was the only way to set them) */
#ifdef HZ_EXPERIMENTS
    hz_experiments_t exp;
#endif
    int N, W, H;
    int col0, col1;
    int raster;
    int profiling;
    hz_worklists_t list_cache[HZ_LIST_CACHE];
    hz_worklists_t* lists;              /* the entry of the current draw (never NULL) */
    unsigned int lists_clock;
    std::vector<uint32_t>* list_scratch;
    std::vector<uint32_t>* list_scratch2;       /* (the second round's items, built in the same walk as the first's) */
    /* diagnostics (hz_hip_debug_wave_timing): where the next draw's marching waves leave their counters */
    struct { unsigned long long* d_cycles; size_t capacity; unsigned int grid_x, grid_y; } wave_timing;

    /* Streams and HZ_NFB framebuffers.  A draw (stream, nstream, qstream) fills
     * one framebuffer; the readback conversion of that draw (rstream) reads it
     * and clears it behind itself; the NEXT draw goes into the next framebuffer
     * at once, while rstream is still converting the first.  Back-to-back
     * renders thereby overlap the bandwidth-bound conversion of panorama k with
     * the instruction-bound rasterisation of panorama k+1 (see draw_impl).
     *   ev_drawn        stream:  the last draw is complete
     *   ev_free[i]      rstream: framebuffer i is all ones again
     *   ev_readers      stream:  everything queued on `stream` before the current draw
     *                            (readers of the previous framebuffer among it) is done */
    hipStream_t stream, rstream;
    hipEvent_t  ev_drawn, ev_free[HZ_NFB], ev_readers, ev_tanel;
    int16_t*            d_mosaic;
    unsigned long long* d_fbs[HZ_NFB];  /* W*H words each (a sector uses a prefix)                    */
    size_t              fb_used[HZ_NFB];    /* words of d_fbs[i] that may differ from all ones            */
    int                 fbi;            /* framebuffer of the last draw                                */
    unsigned long long* d_fb;           /* = d_fbs[fbi]                                               */
    unsigned char*      d_touched[HZ_NFB];  /* hz_params_t::touched of each framebuffer: seg_stride*H bytes */
    int                 seg_stride;         /* ceil(W / HZ_SEG)                                            */
    /* the queues between the marching kernel and the kernels that finish a draw
     * (clipped, medium, large triangles): one set per framebuffer, so
     * that those kernels of panorama k (qstream) run beside k_march of k+1.
     * A two-round draw (see hz_hip_draw) has as many sets again for its first
     * round, which runs on a stream of its own (nstream) beside the second round
     * of the panorama before. */
    hipStream_t         qstream, nstream;
    hipEvent_t          ev_marched, ev_near;
    /* coarse depth of each framebuffer (hz_k_hiz.h), allocated by the first draw that wants it */
    uint32_t*           d_hiz[HZ_NFB];
    int                 hiz_unavailable;        /* their allocation failed once: not tried again with every draw */
    int                 tiles_unavailable;      /* ... the tile bins' (tile_bins) */
    /* The first round's reach of a zoomed view follows what the second round had to draw (plan_rounds, draw_impl): how
     * far the first round has to reach for the ridge that hides most of the view to be in its picture depends on the
     * view, and what a reach was worth shows in what the second round still had to queue for k_big.  The second round's
     * queue counters are copied to the host behind its queue kernels; a later draw of the SAME view that finds the copy
     * complete looks at them: many work items behind a short first round (HZ_ADAPT_HI) - the next draws try the long
     * reach; if that leaves fewer than 70 % of them, they stay with it, else they go back for good.  A new view starts
     * short.  The bytes do not depend on the reach. */
    struct
    {
        unsigned int* h_counts[HZ_NFB];     /* pinned, 6 words each: the second round's queue counters */
        hipEvent_t    ev[HZ_NFB];
        int           pending[HZ_NFB], long_of[HZ_NFB], reach_of[HZ_NFB];
        unsigned int  serial_of[HZ_NFB];
        hz_view_t     view; int col0, col1, have_view;     /* the view the observations are about */
        unsigned int  serial;               /* ... its number */
        unsigned int  items_short;          /* what its second round queued behind a short first round (0: not seen yet) */
        int           tried_long;
        int           long_reach;           /* the choice for its next draw */
        int           seen_reach; unsigned int seen_records, seen_items;   /* the last observation (hz_hip_last_queue_counts) */
    } adapt;
    /* the vertex cache (hz_draw.cpp: vertex_cache): the view-independent half of every vertex's transform for the viewpoint `key` */
    struct { hz_polar_t* d_polar; hz_xform_t key; int state /* 0 nothing, 1 the viewpoint has been drawn once, 2 filled */; int unavailable; hipEvent_t ev_filled;
             int same_draw;     /* this draw is another sector of the call that made the draw before (hz_hostpath.cpp): not another draw from the viewpoint */ } vc;
    int                 last_qshards_log2;      /* hz_params_t::qshards_log2 of the last draw (diagnostics: hz_hip_debug_bigqueue) */
    int                 last_plan[5];           /* the last draw (hz_hip_last_plan): rounds, coarse depth kept, the first round's reach in cells, launched from a work list */
    int                 stream_reads_fb;       /* a reader of the framebuffer (pick, annotator passes) was queued on `stream` since the last draw */
    hz_bigrec_t*        d_bigrec_s[2*HZ_NFB];          /* [0..NFB) one-round draws and second rounds, [NFB..2 NFB) first rounds */
    hz_bigitem_t*       d_bigitem_s[2*HZ_NFB];
    hz_rec_t*           d_midrec_s[2*HZ_NFB];
    uint32_t*           d_clip_s[2*HZ_NFB];
    unsigned int*       d_big_counters_s[2*HZ_NFB];    /* HZ_NCOUNTERS each, see mr_queue_t */
    unsigned int        bigrec_capacity, bigitem_capacity, midrec_capacity, clip_capacity;
    unsigned int        near_bigrec_capacity, near_bigitem_capacity, near_clip_capacity;    /* first rounds' sets: no medium queue */
    tl_bins_t           tiles_s[2*HZ_NFB];             /* the tile bins of each queue set (hz_k_tile.h) */
    /* the last draw: a conversion that clears the framebuffer behind itself
     * (k_resolve<true>) consumes it; whoever wants to read it after that gets it
     * drawn again first (fb_refill) */
    hz_view_t           last_view;
    int                 have_view;
    int                 fb_consumed;
    float*              d_tanel;
    float*              h_tanel;        /* the table d_tanel holds (or is about to, in stream order) */
    int                 tanel_resident;

    /* results into caller-owned host memory (hz_hostpath.cpp): staging ring, copy streams, the streams of blobs of the
     * calls in flight; made on first use */
    struct hz_hoststate* host;

    /* texture path: the mosaic of map tiles, one uint32 B|G<<8|R<<16 per texel */
    uint32_t*      d_texels;
    hz_texparams_t tex;
    int            tex_on;

    hipEvent_t ev[10];
    int        have_times;
    hz_times_t times;
};

/* ---- constants, diagnostics ----------------------------------------------------------------------------------------- */
/* constants that used to be switches (each was swept: DESIGN.md section 4, docs/history/) */
#define HZ_NEAR_PX            20.0f     /* the first round takes the strips whose cells are wider than this many pixels */
#define HZ_TWO_ROUNDS_MIN_PIX 6.0e6     /* two rounds from this many pixels on */
#define HZ_TILES_MIN_PX       35.0f     /* from this width of a cell at the first round's reach on, that round's large triangles go by screen tile */
#define HZ_HIZ_MIN_PX         25.0f     /* "zoomed" = a cell at the first round's reach is at least this wide */



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


/* ---- hz_context.cpp, hz_plan.cpp, hz_draw.cpp, hz_convert.cpp, for the other host translation units ---------------------------------------------------------- */
hz_options_t hz_options_from_env(void);
mr_queue_t   hz_queue_set(const hz_dev_t* d, int k);
int          hz_tile_bins(hz_dev_t* d, int set);             /* the tile bins of a queue set, made on first use; 1: no memory for them */
hipError_t   hz_sync_all(hz_dev_t* d);                          /* everything queued on any of the context's streams is done */
hz_params_t  hz_make_params(const hz_dev_t* d, const hz_view_t* v);
int          hz_plan_rounds(const hz_dev_t* d, const hz_view_t* view, hz_params_t& p);
mr_zones_t   hz_make_zones(const hz_params_t& p, bool near_first);
bool         hz_azimuths_of_columns(const hz_params_t& p, double* a0, double* a1);
void         hz_list_items(const hz_params_t& p, const mr_zones_t& zn, double a0, double a1, std::vector<uint32_t>& out, bool every_strip = false);
void         hz_list_rounds(const hz_params_t& p, const mr_zones_t& zn, double a0, double a1, bool one_round,
                            std::vector<uint32_t>* first, std::vector<uint32_t>& second, bool every_strip = false);
int          hz_draw_impl(hz_dev_t* d, const hz_view_t* view);
int          hz_fb_refill(hz_dev_t* d);                         /* a reader finds the framebuffer consumed: the draw is repeated */
int          hz_fb_mark_consumed(hz_dev_t* d);                  /* the conversion just queued on rstream left the framebuffer all ones */
int          hz_upload_tanel(hz_dev_t* d, const float* tanel);
int          hz_rstream_after_draw(hz_dev_t* d);
int          hz_resolve_impl(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                             unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int nbands, hipEvent_t* ev_band, int* band_rows);
/* ---- hz_hostpath.cpp ---------------------------------------------------------------------------------------------- */
void         hz_hostpath_destroy(hz_dev_t* d);                  /* staging ring, copy streams, the stream of blobs */
int          hz_gpu_numa_node(const hz_dev_t* d);               /* the NUMA node the context's GPU hangs off (Linux sysfs), -1: unknown */
void         hz_hostpath_landing(hz_dev_t* d, unsigned char** pinned, size_t* bytes);   /* pinned memory nothing is using now (no panorama in flight), or 0 bytes */
