/* hz_hostpath.cpp - results into the caller's HOST memory: horizonator_render_offscreen() as the reference's callers
 * see it (reference horizonator-lib.c:911-1051: one call, BGR image and ranges in the caller's buffers when it returns).
 * Plain C++ over the HIP runtime API (compiled by g++); kernels through hz_launch.h.
 *
 * What lies between the framebuffer and the caller's buffers is PCIe (a 16000 x 4000 panorama is 448 MB of results;
 * the link moves ~55 GB/s), so a call is organised around the link:
 *
 *   - Only the terrain pixels travel, 4 bytes each (k_pack_host: blobs of z24<<8 | red8 with a mask, hz_scatter.h);
 *     a pool of host threads (hz_pool.h) makes BGR bytes, depth and range of each blob as it arrives (hz_scatter.c:
 *     the readback conversion, reference :1013-1025, in the host's vector unit - bit for bit what k_resolve4
 *     computes).  103 MB instead of 448.
 *   - Each panorama in flight has a pinned LANDING area of its own, and the copy engine moves a sector's stream there
 *     in a few copies (4, 8, then 16 MB) as soon as the host knows the stream's length - which the device tells it
 *     (k_tell, hz_k_tell.h: a few words into pinned memory behind each k_pack_host, polled) instead of being asked.
 *     Round 5 moved the streams through one ring of chunks shared by everything: a copy could only be issued when its
 *     slot had been scattered, and a panorama's copies only from its own hz_hip_host_end().  Now whoever waits - for a
 *     sector, for a chunk - issues whatever copy has become possible meanwhile, the NEXT panorama's included: the copy
 *     engine goes from one panorama's last blob to the next one's first without the host in between.
 *     (Why the copy engine and not a kernel: a kernel storing the stream into host memory reaches the link's 55 GB/s
 *     too, but the kernels beside it then take 1.5 to 100 times their time; beside the copy engine, 1.00 -
 *     tools/pcie_beside.hip, profiles/r6_pcie_beside.txt.)
 *   - A single call draws and ships its panorama in azimuth SECTORS (hz_options_t::host_sectors; whole images of
 *     12 Mpix and more: 2, from 32 Mpix: 4): sector s+1 is drawn while the blobs of sector s cross the link - the draw
 *     is hidden behind the transfer except for the first sector's.  A sector's pixels are bit-identical to the same
 *     pixels of a whole draw (hz_hip_set_sector), so the bytes the caller gets do not depend on the number.
 *   - hz_hip_host_begin() / hz_hip_host_end() split a call in two: a caller that begins panorama k+1 before it ends
 *     panorama k (two sets of buffers) has the device draw k+1 while k crosses the link, and k+1's blobs follow k's
 *     without a gap.
 *   - 62 % of the benchmark image is sky - BGR (255,0,0), range -1 (reference horizonator-lib.c:185, :1016).  The pool
 *     writes it with streaming stores: the rows [0, y_pre) beforehand (as much as there is time for until the first
 *     blobs arrive); below y_pre a blob writes the sky pixels of its own tile (hz_blob_scatter_mode: every byte
 *     once), and the tiles without a blob - k_pack_host leaves a bitmap of the tiles it sent, k_tell hands it over -
 *     are filled as soon as their sector is known, while the host waits for chunks.
 *
 * The dense path at the end of the file (every pixel travels: 448 MB, through a ring of pinned chunks) serves textured colour, images taller than 65535 rows and hz_options_t::host_dense. */
#include "hz_dev.h"
#include "hz_pool.h"

#include <chrono>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>

/* ------------------------------------------------------------------------ */
/* the state of a context's host path                                        */

#define HZ_HOST_MAX_SECTORS 8
#define HZ_HOST_JOBS        2           /* panoramas between hz_hip_host_begin() and hz_hip_host_end() */

/* the control words of a panorama in flight, in HBM (d_ctl) and mirrored in pinned host memory (h_ctl):
 *   [4*s .. 4*s+3]          sector s: k_pack_host's cursor words / k_tell's info words
 *   [HZ_CTL_PRESENT + ..]   the sectors' tile bitmaps (sector s's: pres0[s], npres[s] words) */
#define HZ_CTL_PRESENT (4*HZ_HOST_MAX_SECTORS)

/* Pinned host memory by the hundred megabytes.  hipHostMalloc pins page by page of 4 KB: 277 MB - the landing area of a
 * 16000 x 4000 panorama - take it 45-47 ms, 1.1 GB 180 ms, most of what horizonator_init() costs a warm process.  Anonymous
 * memory in 2 MB pages (transparent huge pages, where the system grants them on request) that the driver is then asked to
 * pin - hipHostRegister, which faults the pages in itself - takes 11.5 / 46 ms, and the copy engine moves the same 55 GB/s
 * into it (tools/pinned_alloc.hip, profiles/r6_pinned_alloc.txt).  hipHostMalloc places its pages on the NUMA node next to
 * the GPU; here they land where the thread that faults them in runs, so that thread visits that node for the duration.
 * Whatever fails on the way - no mmap, no registration - ends in hipHostMalloc. */
struct hz_pinned_t { void* map; size_t map_bytes; bool registered; };
static void* pinned_alloc(hz_pinned_t* m, size_t bytes, int numa_node)
{
    m->map = NULL; m->map_bytes = 0; m->registered = false;
    const char* how = getenv("HZ_PINNED");              /* "malloc": hipHostMalloc, as before round 6's last day */
    if(!(how && strcmp(how, "malloc") == 0) && bytes >= ((size_t)8 << 20))
    {
        const size_t huge = (size_t)2 << 20;
        cpu_set_t before, node_cpus;
        const bool moved = numa_node >= 0 && hz_copy_pool::cpus_of_node(numa_node, &node_cpus) &&
                           sched_getaffinity(0, sizeof(before), &before) == 0 && sched_setaffinity(0, sizeof(node_cpus), &node_cpus) == 0;
        void* map = mmap(NULL, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        void* p = NULL;
        if(map != MAP_FAILED)
        {
            p = (void*)(((uintptr_t)map + huge-1) & ~(uintptr_t)(huge-1));
            (void)madvise(p, bytes, MADV_HUGEPAGE);
            if(hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess)
            {
                (void)hipGetLastError();
                munmap(map, bytes + huge);
                p = NULL;
            }
            else { m->map = map; m->map_bytes = bytes + huge; m->registered = true; }
        }
        if(moved) (void)sched_setaffinity(0, sizeof(before), &before);
        if(p) return p;
    }
    void* p = NULL;
    if(hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    return p;
}
static void pinned_free(hz_pinned_t* m, void* p)
{
    if(!p) return;
    if(m->registered) { (void)hipHostUnregister(p); munmap(m->map, m->map_bytes); }
    else (void)hipHostFree(p);
    m->map = NULL; m->map_bytes = 0; m->registered = false;
}

struct hz_hostjob
{
    bool      active;
    bool      clears;                   /* its conversions cleared the framebuffers behind themselves */
    hz_view_t view;
    uint32_t  flags;                    /* HZ_BLOB_*: what its blobs carry */
    unsigned int epoch;                 /* what k_tell writes into this job's info words */
    int       nsec;
    int       col[HZ_HOST_MAX_SECTORS+1];   /* image columns: sector s = [col[s], col[s+1]) */
    int       out_col0, out_w;          /* the caller's buffers are [H][out_w] and start at image column out_col0 */
    size_t    off[HZ_HOST_MAX_SECTORS];     /* where sector s's stream starts in d_hs / h_land (words; a multiple of the chunk size) */
    size_t    cap[HZ_HOST_MAX_SECTORS];     /* ... and the room it has */
    size_t    chunk0[HZ_HOST_MAX_SECTORS], pres0[HZ_HOST_MAX_SECTORS], npres[HZ_HOST_MAX_SECTORS];
    hz_copy_pool::scatter_t sc;
    std::vector<std::atomic<int>>* band_left;
    std::vector<float>* tanel;          /* the job's own copy: the scatter tasks read it */
    hz_copy_pool::batch_t filled;       /* the sky tasks queued by begin */
    /* device side */
    uint32_t*     d_hs;                 /* the streams of blobs of the job's sectors */
    size_t        hs_capacity;          /* words */
    unsigned int* d_ctl;
    size_t        ctl_capacity;         /* words (of d_ctl and of h_ctl) */
    /* pinned host memory */
    uint32_t*     h_land;               /* where the copy engine puts the streams: the same offsets as in d_hs */
    size_t        land_capacity;        /* words */
    hz_pinned_t   land_mem;             /* ... and how that memory was had (pinned_alloc) */
    unsigned int* h_ctl;
    hipEvent_t    ev_told;              /* rstream: every k_tell of this job has run */
    /* the transfer as far as the host has driven it (advance()) */
    int           known;                /* sectors whose info words have arrived - and whose copies have been issued */
    size_t        nwords[HZ_HOST_MAX_SECTORS], nblobs[HZ_HOST_MAX_SECTORS], nchunks[HZ_HOST_MAX_SECTORS];
    bool          overflowed;
    std::vector<hipEvent_t>* ev_copy;   /* [chunk number]: behind the copy that begins with that chunk */
    std::vector<size_t>*     ev_of;     /* [chunk number]: the chunk number whose event says this one has landed */
    double        t_known[HZ_HOST_MAX_SECTORS];
    std::chrono::steady_clock::time_point t_begin;
    double t_sky_queued, t_queued[HZ_HOST_MAX_SECTORS];     /* host_times: ms since t_begin when the sky tasks / sector s's work had been queued */
};

struct hz_hoststate
{
    hipStream_t    cstream[HZ_COPY_STREAMS];    /* the copies, taken in turn */
    hz_hostjob     job[HZ_HOST_JOBS];
    int            next_begin, next_end;    /* jobs are ended in the order they were begun */
    unsigned int   epoch;
    size_t         ncopies;
    /* the dense path: a ring of pinned chunks (made when that path is first taken) and internal output buffers */
    unsigned char* h_stage[HZ_STAGE_SLOTS];     /* slot k of ONE pinned allocation (h_stage[0]) */
    hz_pinned_t    stage_mem;
    hipEvent_t     ev_stage[HZ_STAGE_SLOTS];
    hipEvent_t     ev_band[HZ_HOST_BANDS];
    unsigned char* d_bgr;
    float*         d_ranges;
    int32_t*       d_index;
    uint32_t*      d_z24;
};

static int make_hoststate(hz_hoststate* h)
{
    /* The copies get streams of the highest priority: HIP deals its streams onto a handful of hardware queues, per
     * priority level, and a queue is worked through in order - on a queue shared with one of the context's draw streams
     * the copies of sector 0 sat behind the marching kernels of sectors 1 to 3 (round 5: first chunk 0.75 ms after its
     * sector was known, profiles/r5_host_inclusive.txt). */
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for(int k=0; k<HZ_COPY_STREAMS; k++) HZ_CHECK(hipStreamCreateWithPriority(&h->cstream[k], hipStreamNonBlocking, prio_hi));
    for(int j=0; j<HZ_HOST_JOBS; j++)
    {
        hz_hostjob& jb = h->job[j];
        jb.band_left = new std::vector<std::atomic<int>>();
        jb.tanel = new std::vector<float>();
        jb.ev_copy = new std::vector<hipEvent_t>();
        jb.ev_of = new std::vector<size_t>();
        HZ_CHECK(hipEventCreateWithFlags(&jb.ev_told, hipEventDisableTiming));
    }
    return 0;
}

/* the NUMA node the context's GPU hangs off (Linux sysfs through its PCI address), -1: unknown */
int hz_gpu_numa_node(const hz_dev_t* d)
{
    char bdf[64] = "";
    if(hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), d->device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for(char* c = bdf; *c; c++) if(*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160]; snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r"); if(!f) return -1;
    int node = -1;
    if(fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

static int ensure_host(hz_dev_t* d)
{
    if(d->host) return 0;
    (void)copy_pool(hz_gpu_numa_node(d));      /* (the process's pool is made by its first context: near that context's GPU) */
    hz_hoststate* h = new hz_hoststate();
    memset((void*)h, 0, sizeof(*h));
    d->host = h;
    if(make_hoststate(h) != 0) { hz_hostpath_destroy(d); return -1; }     /* (half made: nothing of it stays) */
    return 0;
}

/* the dense path's ring and band events */
static int ensure_ring(hz_dev_t* d)
{
    hz_hoststate* h = d->host;
    if(h->h_stage[0] && h->ev_stage[HZ_STAGE_SLOTS-1]) return 0;
    if(!h->h_stage[0])
    {
        unsigned char* ring = (unsigned char*)pinned_alloc(&h->stage_mem, (size_t)HZ_STAGE_SLOTS*HZ_STAGE_BYTES, hz_gpu_numa_node(d));
        if(!ring) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip: no pinned memory for the ring of staging chunks"); return -1; }
        for(int k=0; k<HZ_STAGE_SLOTS; k++) h->h_stage[k] = ring + (size_t)k*HZ_STAGE_BYTES;      /* (the context's from here on: freed with it) */
    }
    for(int k=0; k<HZ_HOST_BANDS; k++)  if(!h->ev_band[k])  HZ_CHECK(hipEventCreateWithFlags(&h->ev_band[k], hipEventDisableTiming));
    for(int k=0; k<HZ_STAGE_SLOTS; k++) if(!h->ev_stage[k]) HZ_CHECK(hipEventCreateWithFlags(&h->ev_stage[k], hipEventDisableTiming));
    return 0;
}

static int host_end(hz_dev_t* d);

/* the landing area of the first job as pinned scratch memory for whoever needs some while no panorama is in flight
 * (hz_ingest.cpp stages the DEM's tiles through it) */
void hz_hostpath_landing(hz_dev_t* d, unsigned char** pinned, size_t* bytes)
{
    *pinned = NULL; *bytes = 0;
    hz_hoststate* h = d->host;
    if(!h || h->next_begin != h->next_end || !h->job[0].h_land) return;
    *pinned = (unsigned char*)h->job[0].h_land; *bytes = h->job[0].land_capacity*sizeof(uint32_t);
}

void hz_hostpath_destroy(hz_dev_t* d)
{
    hz_hoststate* h = d->host;
    if(!h) return;
    /* panoramas begun and never ended: the pool's tasks name their buffers and counters, copies write their landing */
    while(h->next_end != h->next_begin) (void)host_end(d);
    for(int k=0; k<HZ_COPY_STREAMS; k++) if(h->cstream[k]) (void)hipStreamSynchronize(h->cstream[k]);
    pinned_free(&h->stage_mem, h->h_stage[0]);
    for(int k=0; k<HZ_STAGE_SLOTS; k++) if(h->ev_stage[k]) (void)hipEventDestroy(h->ev_stage[k]);
    for(int k=0; k<HZ_HOST_BANDS; k++)   if(h->ev_band[k]) (void)hipEventDestroy(h->ev_band[k]);
    for(int k=0; k<HZ_COPY_STREAMS; k++) if(h->cstream[k]) (void)hipStreamDestroy(h->cstream[k]);
    for(int j=0; j<HZ_HOST_JOBS; j++)
    {
        hz_hostjob& jb = h->job[j];
        (void)hipFree(jb.d_hs); (void)hipFree(jb.d_ctl);
        pinned_free(&jb.land_mem, jb.h_land);
        if(jb.h_ctl)  (void)hipHostFree(jb.h_ctl);
        if(jb.ev_told) (void)hipEventDestroy(jb.ev_told);
        if(jb.ev_copy) for(hipEvent_t e : *jb.ev_copy) (void)hipEventDestroy(e);
        delete jb.band_left; delete jb.tanel; delete jb.ev_copy; delete jb.ev_of;
    }
    (void)hipFree(h->d_bgr); (void)hipFree(h->d_ranges); (void)hipFree(h->d_index); (void)hipFree(h->d_z24);
    delete h;
    d->host = NULL;
}

/* ------------------------------------------------------------------------ */
/* without the sky: begin (queue the draws, conversions and shipments, start the sky) */

/* words a sector's stream may need: every pixel terrain - its words, a byte of shade where that is what travels -,
 * per blob header + masks + padding, and per chunk one blob's worth of skipped room; a multiple of the chunk size */
static size_t hs_words_needed(int SW, int H, uint32_t flags)
{
    const size_t npix = (size_t)SW*H;
    const size_t wpp = ((flags & HZ_BLOB_PACKED) ? 1 : 0) + ((flags & HZ_BLOB_INDEX) ? 1 : 0);
    const size_t tiles = (size_t)((SW + HZ_BLOB_COLS-1)/HZ_BLOB_COLS)*(size_t)((H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS);
    size_t words = npix*wpp + ((flags & HZ_BLOB_RED) ? npix/4 + tiles : 0) + tiles*(HZ_BLOB_HDR + HZ_BLOB_ROWS*(HZ_BLOB_COLS/32) + 4);
    const size_t blob_max = HZ_BLOB_HDR + HZ_BLOB_ROWS*(HZ_BLOB_COLS/32) + (size_t)HZ_BLOB_ROWS*HZ_BLOB_COLS*(wpp + 1) + 4;
    const size_t chunk = HZ_STAGE_BYTES/4;
    words += (words/chunk + 2)*blob_max;
    return (words + chunk-1)/chunk*chunk;
}

/* how many sectors a call draws and ships its panorama in */
static int sectors_for(const hz_dev_t* d, const hz_view_t* view, bool draws, bool another_in_flight)
{
    if(!draws || d->col0 != 0 || d->col1 != d->W) return 1;        /* (a context that is itself one sector of a panorama; a conversion of a draw already made) */
    int n = d->env.host_sectors;
    if(n <= 0)
    {
        /* (a series - another panorama is crossing the link now - keeps the sectors: this one's first blobs are ready for the copy
         * engine half a millisecond after the call, not a whole draw later, and a draw beside the copy engine costs what it costs
         * alone.  Drawn whole - one draw of 0.84 ms instead of four sectors' 1.4 - a series took 2.95-3.1 ms per panorama
         * against 3.0-3.6: nothing, round 6) */
        (void)another_in_flight;
        const double npix = (double)d->W*(double)d->H;
        n = npix >= 32.0e6 ? 4 : npix >= 12.0e6 ? 2 : 1;
        if(n > 1)
        {
            /* zoomed views stay whole: what their draws cost is the first round's large triangles, which every sector
             * they reach into would set up again */
            hz_params_t p = hz_make_params(d, view);
            (void)hz_plan_rounds(d, view, p);
            const float ppr = p.halfW * p.u.az_ndc_per_rad, reach = 0.5f*(float)(p.near_j1 - p.near_j0);
            if(reach > 0.f && ppr/reach >= 25.0f) n = 1;
        }
    }
    if(n > HZ_HOST_MAX_SECTORS) n = HZ_HOST_MAX_SECTORS;
    while(n > 1 && d->W/n < 256) n--;
    return n < 1 ? 1 : n;
}

/* the layout of a job: sectors, their streams, control words.  Returns the words of stream the job needs; *ctl_words: of control */
static size_t lay_out(hz_hostjob& jb, const hz_dev_t* d, int nsec, uint32_t flags, size_t* ctl_words)
{
    const size_t chunk = HZ_STAGE_BYTES/4;
    jb.nsec = nsec;
    jb.out_col0 = d->col0; jb.out_w = d->col1 - d->col0;
    for(int s=0; s<=nsec; s++) jb.col[s] = s == nsec ? d->col1 : d->col0 + (int)((long long)jb.out_w*s/nsec) / 64 * 64;
    size_t need = 0, nchunks = 0, npres = 0;
    for(int s=0; s<nsec; s++)
    {
        const int sw = jb.col[s+1] - jb.col[s];
        jb.cap[s] = hs_words_needed(sw, d->H, flags);
        jb.off[s] = need; need += jb.cap[s];
        jb.chunk0[s] = nchunks; nchunks += jb.cap[s]/chunk;
        jb.npres[s] = ((size_t)((sw + HZ_BLOB_COLS-1)/HZ_BLOB_COLS)*(size_t)((d->H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS) + 31)/32;
        jb.pres0[s] = npres; npres += jb.npres[s];
    }
    for(int s=0; s<nsec; s++) jb.pres0[s] += HZ_CTL_PRESENT;
    *ctl_words = HZ_CTL_PRESENT + npres;
    jb.ev_of->assign(nchunks, 0);
    return need;
}

/* the job's memory: streams in HBM, the landing and both sets of control words; -2: none to be had (the dense path) */
static int job_memory(hz_dev_t* d, hz_hostjob& jb, size_t need, size_t ctl_words)
{
    if(need > jb.hs_capacity || need > jb.land_capacity || ctl_words > jb.ctl_capacity)
        HZ_CHECK(hz_sync_all(d));       /* (nothing of this job is in flight - it is not active -, but frees serialise with the device anyway) */
    if(need > jb.hs_capacity)
    {
        (void)hipFree(jb.d_hs); jb.d_hs = NULL; jb.hs_capacity = 0;
        if(hipMalloc(&jb.d_hs, need*sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return -2; }
        jb.hs_capacity = need;
    }
    if(need > jb.land_capacity)
    {
        pinned_free(&jb.land_mem, jb.h_land);
        jb.h_land = NULL; jb.land_capacity = 0;
        jb.h_land = (uint32_t*)pinned_alloc(&jb.land_mem, need*sizeof(uint32_t), hz_gpu_numa_node(d));
        if(!jb.h_land) return -2;
        jb.land_capacity = need;
    }
    if(ctl_words > jb.ctl_capacity)
    {
        (void)hipFree(jb.d_ctl); jb.d_ctl = NULL;
        if(jb.h_ctl) (void)hipHostFree(jb.h_ctl);
        jb.h_ctl = NULL; jb.ctl_capacity = 0;
        const size_t cap = ctl_words + ctl_words/4 + 256;
        if(hipMalloc(&jb.d_ctl, cap*sizeof(unsigned int)) != hipSuccess) { (void)hipGetLastError(); return -2; }
        if(hipHostMalloc((void**)&jb.h_ctl, cap*sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return -2; }
        memset(jb.h_ctl, 0, cap*sizeof(unsigned int));      /* (no epoch is 0) */
        jb.ctl_capacity = cap;
    }
    return 0;
}

static uint32_t blob_flags(const void* bgr, const void* ranges, const void* index, const void* z24)
{
    return ((ranges || z24) ? HZ_BLOB_PACKED : bgr ? HZ_BLOB_RED : 0u) | (index ? HZ_BLOB_INDEX : 0u);
}

/* the device's share of a panorama into host memory: the draws (sector by sector), the conversions into streams of blobs,
 * the words that tell the host about them - queued on the context's streams.  0, or -1 with the error text set. */
static int queue_device_side(hz_dev_t* d, hz_hostjob& jb, const hz_view_t* view, bool draws, size_t ctl_words)
{
    const uint32_t flags = jb.flags;
    const int H = d->H;
    const bool prof = d->profiling != 0;
    const int user_col0 = d->col0, user_col1 = d->col1;
    int rc = 0;
    hipError_t err = hipSuccess;
    const char* what = "";
    #define HZ_TRY(call) do { if(err == hipSuccess && rc == 0) { err = (call); if(err != hipSuccess) what = #call; } } while(0)
    HZ_TRY(hipMemsetAsync(jb.d_ctl, 0, ctl_words*sizeof(unsigned int), d->rstream));
    jb.clears = d->env.resolve_clears != 0;
    for(int s=0; s<jb.nsec && rc == 0 && err == hipSuccess; s++)
    {
        if(draws)
        {
            d->col0 = jb.col[s]; d->col1 = jb.col[s+1];
            d->vc.same_draw = s > 0;            /* (the sectors of a call are ONE draw from its viewpoint: hz_draw.cpp, vertex_cache) */
            const int drawn = hz_draw_impl(d, view);
            d->vc.same_draw = 0;
            if(drawn != 0) { rc = -1; break; }
        }
        else if(hz_fb_refill(d) != 0) { rc = -1; break; }
        if(hz_rstream_after_draw(d) != 0) { rc = -1; break; }
        const int SW = d->col1 - d->col0;
        if(prof && s == jb.nsec-1) HZ_TRY(hipEventRecord(d->ev[4], d->rstream));
        hz_hostpack_t hp = { jb.d_hs + jb.off[s], jb.d_ctl + 4*s, (unsigned int)jb.cap[s], (unsigned int)(HZ_STAGE_BYTES/4), flags, jb.d_ctl + jb.pres0[s] };
        const dim3 grid((unsigned)((SW + HZ_BLOB_COLS-1)/HZ_BLOB_COLS), (unsigned)((H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS));
        unsigned int* const qa = d->d_big_counters_s[d->fbi], * const qb = d->d_big_counters_s[HZ_NFB + d->fbi];
        if(err == hipSuccess)
        {
            hzk_pack_host(jb.clears, grid, dim3(64*HZ_BLOB_ROWS), d->rstream, d->d_fb, hp, SW, H, d->col0 - jb.out_col0,
                          d->d_touched[d->fbi], d->seg_stride, jb.clears ? qa : (unsigned int*)NULL, jb.clears ? qb : (unsigned int*)NULL);
            HZ_TRY(hipGetLastError());
        }
        if(err == hipSuccess && jb.clears && hz_fb_mark_consumed(d) != 0) rc = -1;
        if(prof && s == jb.nsec-1) { HZ_TRY(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
        /* ... and the host is told: the stream's length, the tiles it holds blobs for */
        if(err == hipSuccess && rc == 0)
        {
            hz_tell_t tl;
            tl.cursor = jb.d_ctl + 4*s; tl.present = jb.d_ctl + jb.pres0[s];
            tl.h_info = jb.h_ctl + 4*s; tl.h_present = jb.h_ctl + jb.pres0[s];
            tl.capacity = (unsigned int)jb.cap[s]; tl.npresent = (unsigned int)jb.npres[s]; tl.epoch = jb.epoch;
            hzk_tell(d->rstream, tl);
            HZ_TRY(hipGetLastError());
        }
        jb.t_queued[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count();
    }
    HZ_TRY(hipEventRecord(jb.ev_told, d->rstream));
    #undef HZ_TRY
    d->col0 = user_col0; d->col1 = user_col1;
    /* (a reader of the framebuffer after this call - pick, the annotator passes - wants the whole view: the last sector's
     * framebuffer is not it) */
    if(jb.nsec > 1) d->fb_consumed = 1;
    if(err != hipSuccess)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_begin: %s -> %s", what, hipGetErrorString(err));
        fprintf(stderr, "hz_hip: %s\n", g_last_error);
        rc = -1;
    }
    return rc;
}

/* Queues everything the device has to do for one panorama into host memory and starts the sky.  draws: the panorama is
 * drawn here, sector by sector (else: the conversion of the draw already queued, or made again if it was consumed).
 * Returns the job's number, -1 on an error, -2 if this panorama has to take the dense path (no room for the stream). */
static int host_begin(hz_dev_t* d, const hz_view_t* view, const float* tanel, bool draws,
                      unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    hz_hoststate* h = d->host;
    if(h->next_begin == h->next_end) h->next_begin = h->next_end = 0;      /* (nothing in flight: the first set of memory again, not the other one) */
    hz_hostjob& jb = h->job[h->next_begin % HZ_HOST_JOBS];
    if(jb.active) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_begin: %d panoramas are in flight already: end one first", HZ_HOST_JOBS); return -1; }
    const int H = d->H;
    const uint32_t flags = blob_flags(bgr, ranges, index, z24);
    if(ranges && !tanel) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: ranges requested without a tanel table"); return -1; }
    const bool another = h->next_begin != h->next_end;
    size_t ctl_words = 0;
    const size_t need = lay_out(jb, d, sectors_for(d, view, draws, another), flags, &ctl_words);
    for(int s=0; s<jb.nsec; s++) if(jb.cap[s] >= ((size_t)1 << 32)) return -2;      /* (a stream is addressed in 32 bits) */
    {
        const int rc = job_memory(d, jb, need, ctl_words);
        if(rc != 0) return rc;
    }
    jb.view = *view; jb.flags = flags;
    if(++h->epoch == 0) h->epoch = 1;
    jb.epoch = h->epoch;
    jb.known = 0; jb.overflowed = false;
    while(jb.ev_copy->size() < jb.ev_of->size())
    {
        hipEvent_t e = NULL;
        HZ_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        jb.ev_copy->push_back(e);
    }
    jb.t_begin = std::chrono::steady_clock::now();
    if(ranges) jb.tanel->assign(tanel, tanel + H); else jb.tanel->clear();

    /* the sky, while the device draws: every requested buffer, sector by sector, in pieces of ~2 MB.  (The box's cores
     * fill 448 MB in 1.2-1.5 ms with 12-24 threads' streaming stores - tools/hostfill_bench.c.) */
    hz_copy_pool* pool = copy_pool();
    hz_copy_pool::scatter_t& sc = jb.sc;
    sc.dst.W = jb.out_w; sc.dst.H = H; sc.dst.bgr = bgr; sc.dst.ranges = ranges; sc.dst.index = index; sc.dst.z24 = z24;
    sc.dst.tanel = ranges ? jb.tanel->data() : NULL; sc.dst.znear = view->znear; sc.dst.zfar = view->zfar;
    sc.bad.store(0);
    struct { unsigned char* p; size_t px_bytes; int sky; } bufs[4];
    int nbuf = 0;
    if(bgr)    bufs[nbuf++] = { bgr, 3, HZ_SKY_BGR };
    if(ranges) bufs[nbuf++] = { (unsigned char*)ranges, 4, HZ_SKY_RANGES };
    if(index)  bufs[nbuf++] = { (unsigned char*)index, 4, HZ_SKY_INDEX };
    if(z24)    bufs[nbuf++] = { (unsigned char*)z24, 4, HZ_SKY_Z24 };
    /* (fresh pages: transparent huge pages where the system offers them on request - 224 faults instead of 110 000) */
    for(int k=0; k<nbuf; k++)
    {
        const uintptr_t huge = (uintptr_t)2 << 20, lo = ((uintptr_t)bufs[k].p + huge-1) & ~(huge-1), hi = ((uintptr_t)bufs[k].p + (size_t)jb.out_w*H*bufs[k].px_bytes) & ~(huge-1);
        if(hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
    }
    /* How much of the image gets its sky beforehand.  Filling (448 MB in ~1.05 ms: tools/hostfill_bench.c) and scattering get
     * in each other's way when they run at the same time - side by side they take 4 ms where one after the other they take
     * 2.4 (tools/scatter_bench.c, profiles/r5_host_microbenchmarks.txt) - so only as much of the image is filled beforehand as
     * there is time for until the first sector's blobs arrive; below that row a blob writes the sky pixels of its own tile
     * (hz_blob_scatter_mode: every byte once), and the tiles without a blob are filled as soon as their sector's bitmap
     * has arrived.  All of it for a call in one sector (the draw takes longer than the fill), the upper 60 % otherwise -
     * sky for the most part - and 30 % when another panorama is in flight (its blobs are arriving now).  Round 6 swept
     * both again, three alternating runs each (profiles/r6_host_path.txt): 30 / 45 / 60 / 75 / 100 % for a call 3.35 / 3.33 /
     * 3.16 / 3.31 / 3.28 ms, 0 / 30 / 45 / 60 / 100 % for a series 2.73 / 2.56 / 2.64 / 2.75 / 2.58 ms per panorama - the
     * box's run-to-run spread is as large.  HZ_HOST_PREFILL / HZ_HOST_PREFILL_SERIES = percent override. */
    {
        int percent = jb.nsec <= 1 ? 100 : 60;
        if(another) percent = 30;
        const char* e = getenv(another ? "HZ_HOST_PREFILL_SERIES" : "HZ_HOST_PREFILL");
        if(e && atoi(e) >= 0 && atoi(e) <= 100) percent = atoi(e);
        sc.y_pre = (int)((long long)H*percent/100) / HZ_BLOB_ROWS * HZ_BLOB_ROWS;
        if(percent >= 100) sc.y_pre = (H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS*HZ_BLOB_ROWS;
    }
    int widest = 1;
    for(int s=0; s<jb.nsec; s++) if(jb.col[s+1] - jb.col[s] > widest) widest = jb.col[s+1] - jb.col[s];
    sc.band_rows = (int)(((size_t)2 << 20)/((size_t)widest*4) + 1);
    if(sc.band_rows < HZ_BLOB_ROWS) sc.band_rows = HZ_BLOB_ROWS;
    sc.band_rows = (sc.band_rows + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS*HZ_BLOB_ROWS;
    const int pre_rows = sc.y_pre < H ? sc.y_pre : H;
    sc.nbands = (pre_rows + sc.band_rows-1)/sc.band_rows;
    std::vector<std::atomic<int>>(static_cast<size_t>(jb.nsec)*sc.nbands).swap(*jb.band_left);
    sc.band_left = jb.band_left->data();
    jb.filled.pending = 0;
    std::vector<hz_copy_pool::task_t> tasks;
    for(int s=0; s<jb.nsec; s++)
        for(int b=0; b<sc.nbands; b++)
        {
            const int y0 = b*sc.band_rows, y1 = y0 + sc.band_rows < pre_rows ? y0 + sc.band_rows : pre_rows;
            sc.band_left[(size_t)s*sc.nbands + b].store(nbuf);
            for(int k=0; k<nbuf; k++)
            {
                hz_copy_pool::task_t t = {};
                t.kind = hz_copy_pool::FILL; t.dst = bufs[k].p; t.sky = bufs[k].sky; t.left = &sc.band_left[(size_t)s*sc.nbands + b];
                const size_t x0 = (size_t)(jb.col[s] - jb.out_col0), w = (size_t)(jb.col[s+1] - jb.col[s]);
                if(jb.nsec == 1) { t.lo = (size_t)y0*jb.out_w*bufs[k].px_bytes; t.n = (size_t)(y1 - y0)*jb.out_w*bufs[k].px_bytes; t.rows = 1; t.pitch = 0; }
                else { t.lo = ((size_t)y0*jb.out_w + x0)*bufs[k].px_bytes; t.n = w*bufs[k].px_bytes; t.rows = y1 - y0; t.pitch = (size_t)jb.out_w*bufs[k].px_bytes; }
                tasks.push_back(t);
            }
        }
    pool->push_tasks(&jb.filled, tasks);
    jb.t_sky_queued = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count();
    jb.active = true;
    h->next_begin++;
    /* from here on the pool's tasks name the job and the caller's buffers: whatever fails below, hz_hip_host_end() (or
     * the caller of this function, on -1) has to wait for them */

    int rc = queue_device_side(d, jb, view, draws, ctl_words);
    if(rc != 0)
    {
        (void)hipStreamSynchronize(d->rstream);     /* (k_tell's already queued write this job's control words) */
        pool->wait(&jb.filled);
        jb.active = false;
        h->next_begin--;
        return -1;
    }
    return (h->next_begin - 1) % HZ_HOST_JOBS;
}

/* ------------------------------------------------------------------------ */
/* ... end: the blobs into their places as their chunks land                 */

/* Whatever of panorama jb's transfer has become possible: the sectors whose info words have arrived (in order) are
 * learned, and the copies of their streams issued - the first copy of a sector one chunk (its blobs can be scattered
 * when it has landed), the second two, then four; a copy costs the engine ~20 us beyond its bytes (4 MB copies ran at
 * 46 GB/s, 16 MB ones at 55: tools/zero_copy.hip); the last chunks of the last sector one by one again (what arrives
 * last is scattered with nothing left to hide behind).  Returns -1 on a HIP error. */
static int advance(hz_dev_t* d, hz_hostjob& jb)
{
    hz_hoststate* h = d->host;
    const size_t chunk_words = HZ_STAGE_BYTES/4;
    while(jb.known < jb.nsec)
    {
        const int s = jb.known;
        const unsigned int* info = jb.h_ctl + 4*s;
        if(__atomic_load_n(info + 3, __ATOMIC_ACQUIRE) != jb.epoch) break;
        jb.t_known[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count();
        jb.nwords[s] = info[0]; jb.nblobs[s] = info[1];
        if(info[2] || jb.nwords[s] > jb.cap[s]) { jb.overflowed = true; jb.nwords[s] = 0; jb.nblobs[s] = 0; }
        jb.nchunks[s] = (jb.nwords[s] + chunk_words-1)/chunk_words;
        for(size_t c = 0, run = 0; c < jb.nchunks[s]; run++)
        {
            size_t g = run == 0 ? 1 : run == 1 ? 2 : 4;
            if(s == jb.nsec-1 && jb.nchunks[s] - c <= 3) g = 1;
            if(g > jb.nchunks[s] - c) g = jb.nchunks[s] - c;
            const size_t w0 = c*chunk_words, w1 = (c + g)*chunk_words < jb.nwords[s] ? (c + g)*chunk_words : jb.nwords[s];
            hipStream_t cs = h->cstream[h->ncopies++ % HZ_COPY_STREAMS];
            HZ_CHECK(hipMemcpyAsync(jb.h_land + jb.off[s] + w0, jb.d_hs + jb.off[s] + w0, (w1 - w0)*sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
            HZ_CHECK(hipEventRecord((*jb.ev_copy)[jb.chunk0[s] + c], cs));
            for(size_t i=0; i<g; i++) (*jb.ev_of)[jb.chunk0[s] + c + i] = jb.chunk0[s] + c;
            c += g;
        }
        jb.known++;
    }
    return 0;
}

static int host_end(hz_dev_t* d)
{
    hz_hoststate* h = d->host;
    hz_hostjob& jb = h->job[h->next_end % HZ_HOST_JOBS];
    if(!jb.active) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: no panorama is in flight"); return -1; }
    /* the panorama begun after this one, if there is one: its copies are issued from here as they become possible */
    hz_hostjob* next = h->next_begin - h->next_end > 1 ? &h->job[(h->next_end + 1) % HZ_HOST_JOBS] : NULL;
    hz_copy_pool* pool = copy_pool();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count(); };
    const double t_enter = since();
    double t_first = 0, t_arrived = 0, t_waited = 0;

    const int H = d->H, y_pre = jb.sc.y_pre, ty0 = y_pre/HZ_BLOB_ROWS, nty = (H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS;
    const size_t chunk_words = HZ_STAGE_BYTES/4;
    std::vector<std::vector<size_t>> offs(jb.nsec);     /* where the blobs of sector s start, chunk after chunk */
    hz_copy_pool::batch_t placed = { 0 };               /* the job's scatter tasks and late sky */
    struct { unsigned char* p; size_t px_bytes; int sky; } bufs[4];
    int nbuf = 0;
    if(jb.sc.dst.bgr)    bufs[nbuf++] = { jb.sc.dst.bgr, 3, HZ_SKY_BGR };
    if(jb.sc.dst.ranges) bufs[nbuf++] = { (unsigned char*)jb.sc.dst.ranges, 4, HZ_SKY_RANGES };
    if(jb.sc.dst.index)  bufs[nbuf++] = { (unsigned char*)jb.sc.dst.index, 4, HZ_SKY_INDEX };
    if(jb.sc.dst.z24)    bufs[nbuf++] = { (unsigned char*)jb.sc.dst.z24, 4, HZ_SKY_Z24 };
    std::vector<hz_copy_pool::task_t> tasks;
    /* the sky of sector s's tiles without a blob (rows from y_pre down; `present`: k_pack_host's bitmap of the tiles it
     * sent, NULL: none): one task per run of such tiles in a column of tiles */
    auto fill_absent = [&](int s, const unsigned int* present)
    {
        if(ty0 >= nty) return;
        const int sw = jb.col[s+1] - jb.col[s], ntx = (sw + HZ_BLOB_COLS-1)/HZ_BLOB_COLS;
        auto sent = [&](int tx, int ty) { const size_t t = (size_t)ty*ntx + tx; return present && ((present[t >> 5] >> (t & 31)) & 1u); };
        for(int tx=0; tx<ntx; tx++)
        {
            const size_t x0 = (size_t)(jb.col[s] - jb.out_col0) + (size_t)tx*HZ_BLOB_COLS;
            const size_t w = (size_t)(sw - tx*HZ_BLOB_COLS < HZ_BLOB_COLS ? sw - tx*HZ_BLOB_COLS : HZ_BLOB_COLS);
            for(int ty=ty0; ty<nty; )
            {
                if(sent(tx, ty)) { ty++; continue; }
                int t1 = ty + 1;
                while(t1 < nty && t1 - ty < 64 && !sent(tx, t1)) t1++;
                const int y0 = ty*HZ_BLOB_ROWS, y1 = t1*HZ_BLOB_ROWS < H ? t1*HZ_BLOB_ROWS : H;
                for(int k=0; k<nbuf; k++)
                {
                    hz_copy_pool::task_t t = {};
                    t.kind = hz_copy_pool::FILL; t.dst = bufs[k].p; t.sky = bufs[k].sky; t.left = NULL;
                    t.lo = ((size_t)y0*jb.out_w + x0)*bufs[k].px_bytes; t.n = w*bufs[k].px_bytes; t.rows = y1 - y0; t.pitch = (size_t)jb.out_w*bufs[k].px_bytes;
                    tasks.push_back(t);
                }
                ty = t1;
            }
        }
        pool->push_tasks(&placed, tasks);       /* (behind the blobs already queued: they are what the call waits for) */
    };
    int rc = 0, s_done = 0;
    size_t total_words = 0, total_blobs = 0, nchunks_all = 0, ncopies = 0;
    /* one turn of every wait below: copies that have become possible are issued - this panorama's later sectors, and, once
     * all of this one's are on their way, the next panorama's */
    auto turn = [&]() -> int
    {
        if(advance(d, jb) != 0) return -1;
        if(next && jb.known == jb.nsec && advance(d, *next) != 0) return -1;
        _mm_pause();
        return 0;
    };
    for(int s=0; s<jb.nsec && rc == 0; s++, s_done++)
    {
        for(unsigned int spins = 1; jb.known <= s && rc == 0; spins++)
        {
            if(turn() != 0) { rc = -1; break; }
            /* (the words come from k_tell: should its stream have run dry without them - a launch that failed, a device
             * error - say so instead of spinning for ever) */
            if((spins & 0xFFFu) == 0 && jb.known <= s)
            {
                const hipError_t q = hipEventQuery(jb.ev_told);
                (void)hipGetLastError();
                if(q != hipErrorNotReady && (advance(d, jb) != 0 || jb.known <= s))
                {
                    snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: sector %d never reported%s%s", s, q != hipSuccess ? ": " : "", q != hipSuccess ? hipGetErrorString(q) : "");
                    rc = -1;
                }
            }
        }
        if(rc != 0) break;
        if(jb.overflowed) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: the stream of blobs overflowed (%zu words)", jb.cap[s]); rc = -1; break; }
        const size_t nwords = jb.nwords[s], nblobs = jb.nblobs[s];
        total_words += nwords; total_blobs += nblobs;
        fill_absent(s, jb.h_ctl + jb.pres0[s]);
        offs[s].resize(nblobs + 1);
        size_t noffs = 0, first = 0;
        const uint32_t* land = jb.h_land + jb.off[s];
        for(size_t w0 = 0, c = 0; w0 < nwords && rc == 0; w0 += chunk_words, c++, nchunks_all++)
        {
            const size_t nw = w0 + chunk_words < nwords ? chunk_words : nwords - w0;
            const double t_w0 = since();
            const size_t carrier = (*jb.ev_of)[jb.chunk0[s] + c];
            if(carrier == jb.chunk0[s] + c) ncopies++;
            for(;;)
            {
                const hipError_t q = hipEventQuery((*jb.ev_copy)[carrier]);
                if(q == hipSuccess) break;
                (void)hipGetLastError();
                if(q != hipErrorNotReady || turn() != 0)
                {
                    if(q != hipErrorNotReady) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: chunk %zu of sector %d: %s", c, s, hipGetErrorString(q));
                    rc = -1; break;
                }
            }
            if(rc != 0) break;
            t_arrived = since(); t_waited += t_arrived - t_w0;
            if(nchunks_all == 0) t_first = t_arrived;
            const uint32_t* chunk = land + w0;
            size_t* const o = offs[s].data() + noffs;
            const size_t room = offs[s].size() - noffs;
            const size_t nb = hz_blob_walk(chunk, nw, first, o, room, &first);
            if(nb == (size_t)-1 || nb > room) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: chunk %zu of sector %d is not a sequence of blobs", c, s); rc = -1; break; }
            noffs += nb;
            /* tasks of ~256 KB of blobs */
            for(size_t b0=0; b0<nb; )
            {
                size_t b1 = b0 + 1;
                while(b1 < nb && o[b1] - o[b0] < 65536) b1++;
                hz_copy_pool::task_t t = {};
                t.kind = hz_copy_pool::SCATTER; t.sc = &jb.sc; t.sector = s; t.chunk = chunk; t.offs = o + b0; t.nblobs = b1 - b0;
                tasks.push_back(t);
                b0 = b1;
            }
            pool->push_tasks(&placed, tasks);
        }
    }
    /* whatever ended the loop early: the sectors not reached keep the caller's buffers defined (sky), and nothing of the
     * job may still be on its way when its memory is handed to the next */
    if(rc != 0)
    {
        for(int s=s_done; s<jb.nsec; s++) fill_absent(s, NULL);
        (void)hipEventSynchronize(jb.ev_told);
        (void)advance(d, jb);                   /* (whatever it still learns is issued, and waited for here) */
        for(int k=0; k<HZ_COPY_STREAMS; k++) (void)hipStreamSynchronize(h->cstream[k]);
        (void)hipGetLastError();
    }
    /* the last blobs are being placed: the next panorama's copies keep being issued meanwhile */
    {
        std::unique_lock<std::mutex> lk(pool->m);
        while(placed.pending != 0)
        {
            lk.unlock();
            if(next && rc == 0 && advance(d, *next) != 0) rc = -1;
            lk.lock();
            if(placed.pending != 0) pool->cv_done.wait_for(lk, std::chrono::microseconds(next ? 20 : 1000));
        }
    }
    const double t_scattered = since();
    pool->wait(&jb.filled);
    if(d->env.host_times)
    {
        fprintf(stderr, "hz_hip host path: %.1f MB of blobs (%zu) in %d sector(s), %zu copies, for %.1f MB of results; ms since the call began: sky tasks queued %.2f, sectors queued",
                4e-6*(double)total_words, total_blobs, jb.nsec, ncopies,
                1e-6*(double)jb.out_w*d->H*((jb.sc.dst.bgr ? 3 : 0) + (jb.sc.dst.ranges ? 4 : 0) + (jb.sc.dst.index ? 4 : 0) + (jb.sc.dst.z24 ? 4 : 0)), jb.t_sky_queued);
        for(int s=0; s<jb.nsec; s++) fprintf(stderr, " %.2f", jb.t_queued[s]);
        fprintf(stderr, ", end() entered %.2f, sectors known", t_enter);
        for(int s=0; s<jb.known; s++) fprintf(stderr, " %.2f", jb.t_known[s]);
        fprintf(stderr, ", first chunk here %.2f, last chunk here %.2f (%.2f spent waiting for chunks), blobs in place %.2f, sky and everything %.2f\n",
                t_first, t_arrived, t_waited, t_scattered, since());
    }
    jb.active = false;
    h->next_end++;
    if(rc != 0) fprintf(stderr, "hz_hip: %s\n", g_last_error);
    if(rc == 0 && jb.sc.bad.load()) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: a blob does not describe pixels of this image"); rc = -1; }
    return rc;
}

/* The pool, this context's streams and the memory of one panorama of the given outputs, ahead of the first call (the
 * library's horizonator_init does this: the reference's CLI makes ONE call per process, standalone.c:433-460).
 * warm_view (may be NULL): the device's share of such a call is run once and thrown away - the first launch of a kernel,
 * the first use of a stream or of the copy engine, the coarse-depth tables and work lists a draw allocates when it first
 * wants them cost the first call 15-40 ms otherwise (profiles/r6_host_path.txt). */
extern "C" int hz_hip_host_prepare(hz_dev_t* d, int want_bgr, int want_ranges, int want_index, int want_z24, const hz_view_t* warm_view)
{
    HZ_ON_DEVICE(d);
    if(ensure_host(d) != 0) return -1;
    hz_hoststate* h = d->host;
    if(h->next_begin != h->next_end || d->env.host_dense || d->H > 65535) return 0;
    h->next_begin = h->next_end = 0;
    hz_hostjob& jb = h->job[0];
    const uint32_t flags = blob_flags(want_bgr ? d : NULL, want_ranges ? d : NULL, want_index ? d : NULL, want_z24 ? d : NULL);
    if(!flags || d->col0 != 0 || d->col1 != d->W) return 0;
    /* (the layout with the most sectors the automatic rule may choose: it needs the most room) */
    const double npix = (double)d->W*(double)d->H;
    int nsec = d->env.host_sectors > 0 ? d->env.host_sectors : npix >= 32.0e6 ? 4 : npix >= 12.0e6 ? 2 : 1;
    if(nsec > HZ_HOST_MAX_SECTORS) nsec = HZ_HOST_MAX_SECTORS;
    while(nsec > 1 && d->W/nsec < 256) nsec--;
    size_t ctl_words = 0;
    const size_t need = lay_out(jb, d, nsec, flags, &ctl_words);
    for(int s=0; s<jb.nsec; s++) if(jb.cap[s] >= ((size_t)1 << 32)) return 0;
    if(job_memory(d, jb, need, ctl_words) != 0) return 0;      /* (no memory: the call itself will find out, and take the dense path) */
    if(!warm_view) return 0;
    jb.view = *warm_view; jb.flags = flags;
    if(++h->epoch == 0) h->epoch = 1;
    jb.epoch = h->epoch;
    jb.t_begin = std::chrono::steady_clock::now();
    int rc = queue_device_side(d, jb, warm_view, true, ctl_words);
    (void)hipEventSynchronize(jb.ev_told);
    /* the copy engine, once on each stream, with a copy large enough to go the way a sector's stream goes (the runtime moves
     * a few KB by other means) */
    {
        const size_t warm_words = jb.land_capacity < ((size_t)1 << 20) ? jb.land_capacity/HZ_COPY_STREAMS : ((size_t)1 << 19);
        for(int k=0; k<HZ_COPY_STREAMS && rc == 0 && warm_words; k++)
            if(hipMemcpyAsync(jb.h_land + warm_words*k, jb.d_hs + warm_words*k, warm_words*sizeof(uint32_t), hipMemcpyDeviceToHost, h->cstream[k]) != hipSuccess) rc = -1;
    }
    for(int k=0; k<HZ_COPY_STREAMS; k++) (void)hipStreamSynchronize(h->cstream[k]);
    if(hz_sync_all(d) != hipSuccess) rc = -1;
    (void)hipGetLastError();
    d->have_view = 0;                           /* (nothing the caller asked for has been drawn) */
    d->vc.state = 0;
    d->adapt.have_view = 0;
    return rc;
}

/* ------------------------------------------------------------------------ */
/* the dense path: every pixel travels                                       */
/*
 * hipMemcpy into pageable memory moves 448 MB at ~11 GB/s (40 ms, twenty times the render); here
 *   - the conversion runs in HZ_HOST_BANDS bands of rows, and the bytes of a band leave as soon as that band is converted;
 *   - the copy engines (two streams in turn) write chunks into the ring of pinned staging buffers at the link's rate;
 *   - the pool moves each finished chunk on into the caller's (pageable) buffer while the next chunks are in flight - and,
 *     before the first chunk has arrived, has the kernel map the caller's pages (MADV_POPULATE_WRITE): arrays fresh from
 *     the allocator - what the reference's Python wrapper hands over on every call, horizonator-pywrap.c:234-250 - otherwise
 *     fault in page by page under the copies. */
static int ensure_out_buffers(hz_dev_t* d, bool bgr, bool ranges, bool index, bool z24)
{
    hz_hoststate* h = d->host;
    const size_t npix = (size_t)d->W*d->H;
    if(bgr    && !h->d_bgr)    HZ_CHECK(hipMalloc(&h->d_bgr,    npix*3));
    if(ranges && !h->d_ranges) HZ_CHECK(hipMalloc(&h->d_ranges, npix*sizeof(float)));
    if(index  && !h->d_index)  HZ_CHECK(hipMalloc(&h->d_index,  npix*sizeof(int32_t)));
    if(z24    && !h->d_z24)    HZ_CHECK(hipMalloc(&h->d_z24,    npix*sizeof(uint32_t)));
    return 0;
}

/* The device buffers of the conversion just queued -> the caller's host buffers.  The conversion ran in
 * `nbands` bands of `band_rows` rows (ev_band[k] behind band k); buffer b has row_bytes[b] bytes per row:
 * the chunks go band by band, every buffer's rows of a band before the next band's. */
static int copy_out(hz_dev_t* d, int nbuf, unsigned char* const* dst, const unsigned char* const* src, const size_t* row_bytes,
                    int rows_total, int nbands, int band_rows, hz_copy_pool* pool)
{
    hz_hoststate* h = d->host;
    std::lock_guard<std::mutex> one(pool->busy);
    struct chunk_t { unsigned char* dst; const unsigned char* src; size_t n; int band; };
    std::vector<chunk_t> chunks;
    for(int k=0; k<nbands; k++)
    {
        const int y0 = k*band_rows, y1 = (k+1)*band_rows < rows_total ? (k+1)*band_rows : rows_total;
        for(int b=0; b<nbuf; b++)
        {
            const size_t lo = (size_t)y0*row_bytes[b], hi = (size_t)y1*row_bytes[b];
            for(size_t off=lo; off<hi; off+=HZ_STAGE_BYTES)
                chunks.push_back({ dst[b] + off, src[b] + off, hi - off < HZ_STAGE_BYTES ? hi - off : HZ_STAGE_BYTES, k });
        }
    }
    const size_t nc = chunks.size();
    size_t issued = 0;
    int band_seen[HZ_COPY_STREAMS];
    for(int k=0; k<HZ_COPY_STREAMS; k++) band_seen[k] = -1;
    /* the host threads' copies are queued as the chunks arrive and waited for together at the end; a staging
     * slot is reused only after the copy out of it has been waited for */
    std::vector<hz_copy_pool::batch_t> done(nc);
    for(size_t k=0; k<nc; k++) done[k].pending = 0;
    /* (a failing HIP call ends the issuing, not the function: the pool's tasks name `done` and the caller's
     * buffers, so every batch already pushed is waited for before either goes away) */
    hipError_t err = hipSuccess;
    const char* what = "";
    #define HZ_TRY(call) do { if(err == hipSuccess) { err = (call); if(err != hipSuccess) what = #call; } } while(0)
    for(size_t k=0; k<nc && err == hipSuccess; k++)
    {
        for(; issued < nc && issued < k + HZ_STAGE_SLOTS - 2 && err == hipSuccess; issued++)
        {
            const int slot = (int)(issued % HZ_STAGE_SLOTS);
            if(issued >= HZ_STAGE_SLOTS) pool->wait(&done[issued - HZ_STAGE_SLOTS]);    /* the slot's previous chunk has left it */
            hipStream_t cs = h->cstream[issued % HZ_COPY_STREAMS];
            if(band_seen[issued % HZ_COPY_STREAMS] < chunks[issued].band)
            {
                HZ_TRY(hipStreamWaitEvent(cs, h->ev_band[chunks[issued].band], 0));
                band_seen[issued % HZ_COPY_STREAMS] = chunks[issued].band;
            }
            HZ_TRY(hipMemcpyAsync(h->h_stage[slot], chunks[issued].src, chunks[issued].n, hipMemcpyDeviceToHost, cs));
            HZ_TRY(hipEventRecord(h->ev_stage[slot], cs));
        }
        const int slot = (int)(k % HZ_STAGE_SLOTS);
        HZ_TRY(hipEventSynchronize(h->ev_stage[slot]));
        if(err == hipSuccess) pool->push(&done[k], chunks[k].dst, h->h_stage[slot], chunks[k].n, 65536);
    }
    #undef HZ_TRY
    for(size_t k=0; k<nc; k++) pool->wait(&done[k]);
    if(err != hipSuccess)
    {
        for(int k=0; k<HZ_COPY_STREAMS; k++) (void)hipStreamSynchronize(h->cstream[k]);     /* copies in flight write the staging ring */
        snprintf(g_last_error, sizeof(g_last_error), "copy_out: %s -> %s", what, hipGetErrorString(err));
        fprintf(stderr, "hz_hip: %s\n", g_last_error);
        return -1;
    }
    return 0;
}

static int resolve_to_host_dense(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                 unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    hz_hoststate* h = d->host;
    if(ensure_ring(d) != 0 || ensure_out_buffers(d, bgr != NULL, ranges != NULL, index != NULL, z24 != NULL) != 0) return -1;
    const int SW = d->col1 - d->col0;
    const size_t npix = (size_t)SW*d->H;
    unsigned char* dst[4]; const unsigned char* src[4]; size_t row_bytes[4];
    int nbuf = 0;
    if(bgr)    { dst[nbuf] = bgr;                    src[nbuf] = h->d_bgr;                            row_bytes[nbuf++] = (size_t)SW*3; }
    if(ranges) { dst[nbuf] = (unsigned char*)ranges; src[nbuf] = (const unsigned char*)h->d_ranges;   row_bytes[nbuf++] = (size_t)SW*sizeof(float); }
    if(index)  { dst[nbuf] = (unsigned char*)index;  src[nbuf] = (const unsigned char*)h->d_index;    row_bytes[nbuf++] = (size_t)SW*sizeof(int32_t); }
    if(z24)    { dst[nbuf] = (unsigned char*)z24;    src[nbuf] = (const unsigned char*)h->d_z24;      row_bytes[nbuf++] = (size_t)SW*sizeof(uint32_t); }
    /* the draw is in flight (asynchronous): while it runs, the pool maps the caller's pages */
    hz_copy_pool* pool = copy_pool();
    hz_copy_pool::batch_t mapped = { 0 };
    for(int b=0; b<nbuf; b++) pool->push(&mapped, dst[b], NULL, row_bytes[b]*d->H, (size_t)4 << 20);
    int band_rows = d->H;
    /* (small images: one band - an event and a launch per band are not free) */
    const int want_bands = npix*7 >= ((size_t)64 << 20) ? HZ_HOST_BANDS : 1;
    const int nbands = hz_resolve_impl(d, view, tanel, bgr ? h->d_bgr : NULL, ranges ? h->d_ranges : NULL,
                                       index ? h->d_index : NULL, z24 ? h->d_z24 : NULL, want_bands, h->ev_band, &band_rows);
    int rc = nbands < 0 ? -1 : 0;
    if(rc == 0) rc = copy_out(d, nbuf, dst, src, row_bytes, d->H, nbands, band_rows, pool);
    pool->wait(&mapped);                                /* (its tasks name the caller's buffers: none may outlive this call) */
    return rc;
}

/* ------------------------------------------------------------------------ */
/* the C-ABI (include/hz_hip.h)                                              */

/* without the sky, unless: a textured colour (three bytes per terrain pixel that are not the shade), an image too
 * large for a blob's 16-bit row field, nothing asked for, or the option says so */
static bool sparse_ok(const hz_dev_t* d, const void* bgr, const void* ranges, const void* index, const void* z24)
{
    return !d->env.host_dense && !(d->tex_on && bgr) && (bgr || ranges || index || z24) && d->H <= 65535 && d->col1 - d->col0 >= 1;
}

static int to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel, bool draws,
                   unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    if(ensure_host(d) != 0) return -1;
    if(d->host->next_begin != d->host->next_end)
    {
        snprintf(g_last_error, sizeof(g_last_error), "a panorama begun with hz_hip_host_begin() is still in flight: end it first");
        return -1;
    }
    if(sparse_ok(d, bgr, ranges, index, z24))
    {
        const int j = host_begin(d, view, tanel, draws, bgr, ranges, index, z24);
        if(j == -1) return -1;
        if(j >= 0) return host_end(d);
        /* (-2: no room for the stream - the dense path) */
    }
    if(draws && hz_draw_impl(d, view) != 0) return -1;
    return resolve_to_host_dense(d, view, tanel, bgr, ranges, index, z24);
}

extern "C" int hz_hip_resolve_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                      unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    return to_host(d, view, tanel, false, bgr, ranges, index, z24);
}

extern "C" int hz_hip_render_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    return to_host(d, view, tanel, true, bgr, ranges, index, z24);
}

extern "C" int hz_hip_host_begin(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                 unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    if(ensure_host(d) != 0) return -1;
    if(!sparse_ok(d, bgr, ranges, index, z24))
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_begin: textured colour, images of more than 65535 rows and host_dense contexts deliver with hz_hip_render_to_host() only");
        return -1;
    }
    const int j = host_begin(d, view, tanel, true, bgr, ranges, index, z24);
    if(j == -2) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_begin: no device memory for the stream of blobs");
    return j < 0 ? -1 : 0;
}

extern "C" int hz_hip_host_end(hz_dev_t* d)
{
    HZ_ON_DEVICE(d);
    if(!d->host) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: no panorama is in flight"); return -1; }
    return host_end(d);
}
