/* hz_hostpath.cpp - results into the caller's HOST memory: horizonator_render_offscreen() as the reference's callers
 * see it (reference horizonator-lib.c:911-1051: one call, BGR image and ranges in the caller's buffers when it returns).
 * Plain C++ over the HIP runtime API (compiled by g++); kernels through hz_launch.h.
 *
 * What lies between the framebuffer and the caller's buffers is PCIe (a 16000 x 4000 panorama is 448 MB of results;
 * the link moves ~50 GB/s in copies of 4 MB, ~56 in copies of 16), so a call is organised around the link:
 *
 *   - Only the terrain pixels travel, 4 bytes each (k_pack_host: blobs of z24<<8 | red8 with a mask, hz_scatter.h);
 *     a pool of host threads makes BGR bytes, depth and range of each blob as it arrives (hz_scatter.c: the readback
 *     conversion, reference :1013-1025, in the host's vector unit - bit for bit what k_resolve4 computes).  103 MB
 *     instead of 448.
 *   - The panorama is drawn and shipped in azimuth SECTORS (hz_options_t::host_sectors; whole images of 12 Mpix and
 *     more: 2, from 32 Mpix: 4): sector s+1 is drawn while the blobs of sector s cross the link - the draw is hidden
 *     behind the transfer except for the first sector's.  A sector's pixels are bit-identical to the same pixels of
 *     a whole draw (hz_hip_set_sector), so the bytes the caller gets do not depend on the number of sectors.
 *   - The stream of a sector travels through a ring of HZ_STAGE_SLOTS pinned chunks of HZ_STAGE_BYTES; one copy moves
 *     up to four consecutive chunks (a copy costs the engine ~22 us beyond its bytes), a sector's first copy one, the
 *     panorama's last ones too (they are scattered with nothing left to hide behind).  Two copy streams in turn, of
 *     the HIGHEST priority: HIP deals streams onto a few hardware queues, and on a queue shared with a draw stream the
 *     first sector's copies waited behind the marching kernels of the sectors queued after it.
 *   - 62 % of the benchmark image is sky - BGR (255,0,0), range -1 (reference horizonator-lib.c:185, :1016).  The pool
 *     writes it with streaming stores, but only the rows [0, y_pre) beforehand: as much as there is time for until
 *     the first blobs arrive (one sector: everything, behind the draw; four: the upper 30 %).  Below y_pre a blob
 *     writes the sky pixels of its own tile (hz_blob_scatter_mode), and the tiles without a blob are filled when
 *     their sector has been walked (fill_absent).  Filling and scattering side by side cost the host 3.8-4.3 ms where
 *     one after the other they cost 2.4 (profiles/r5_host_microbenchmarks.txt): every byte is written once.
 *   - hz_hip_host_begin() / hz_hip_host_end() split a call in two: a caller that begins panorama k+1 before it ends
 *     panorama k (two sets of buffers) has the device draw k+1 while k crosses the link, and pays the link only.
 *
 * Round 4 (one sector, 5 bytes per pixel, draw and transfer strictly in series): 4.3 ms per call; this version:
 * 3.2-3.5 ms, its timeline in profiles/r5_host_inclusive.txt (HZ_HOST_TIMES=1 prints one per call).  Tried and dropped:
 * k_pack_host storing straight into pinned host memory (38 GB/s from its compacted stores, and the host reads that
 * memory slowly: 3.9-6.0 ms).
 *
 * The dense path at the end of the file (every pixel travels: 448 MB) serves textured colour, images taller than
 * 65535 rows and hz_options_t::host_dense. */
#include "hz_dev.h"

#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

/* ------------------------------------------------------------------------ */
/* the pool of host threads                                                  */

struct hz_copy_pool
{
    struct batch_t { int pending; };
    /* what the blobs of a panorama are scattered into (hz_scatter.c), and how far the sky is: the buffers are filled
     * sector by sector, band of rows by band of rows; band_left[sector*nbands + b] = fill tasks of that piece not yet
     * finished */
    struct scatter_t
    {
        hz_scatter_dst_t dst;
        int y_pre;                      /* rows [0, y_pre) get the sky beforehand (band by band); a blob below writes the sky pixels of its tile itself */
        int band_rows, nbands;
        std::atomic<int>* band_left;
        std::atomic<int> bad;
    };
    enum { COPY = 0, MAP, FILL, SCATTER };
    struct task_t
    {
        int kind;
        unsigned char* dst; const unsigned char* src; size_t n;     /* COPY: dst[0..n) = src[0..n); MAP: the pages of dst[0..n) */
        /* FILL: `rows` runs of n bytes, the first at byte lo of dst, `pitch` bytes apart; which constants (HZ_SKY_*); the piece's counter */
        size_t lo, pitch; int rows, sky; std::atomic<int>* left;
        scatter_t* sc; int sector; const uint32_t* chunk; const size_t* offs; size_t nblobs;     /* SCATTER: blobs chunk + offs[0..nblobs) of `sector` */
        batch_t* batch;
    };
    std::mutex m, busy;                 /* busy: one call's transfer at a time (contexts on several threads share the pool and nothing else) */
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    /* two queues: blobs (and copies) before sky - a caller with two panoramas in flight has the sky of the second queued
     * while the blobs of the first arrive, and those are what its hz_hip_host_end() waits for */
    std::deque<task_t> q_hi, q_lo;
    bool stop = false;
    std::atomic<bool> populate_works{true};     /* does this kernel know MADV_POPULATE_WRITE?  Probed once, on a page of our own */

    explicit hz_copy_pool(int n)
    {
        /* (EINVAL on a private anonymous page = the flag is unknown to this kernel; any later failure is about
         * the caller's buffer - a pinned or device mapping, an unmapped range - and only skips that buffer) */
        void* probe = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if(probe != MAP_FAILED)
        {
            if(madvise(probe, 4096, MADV_POPULATE_WRITE) != 0) populate_works = false;
            munmap(probe, 4096);
        }
        /* HZ_COPY_NODE=here (an experiment of round 5): the pool's threads stay on the NUMA node of the thread that made the
         * pool - the caller's buffers were most likely first touched there */
        cpu_set_t node_cpus; bool pin = false;
        const char* where = getenv("HZ_COPY_NODE");
        if(where && strcmp(where, "here") == 0)
        {
            const int cpu = sched_getcpu();
            for(int node=0; node<16 && !pin; node++)
            {
                char path[96]; snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
                FILE* f = fopen(path, "r"); if(!f) break;
                char buf[4096]; if(!fgets(buf, sizeof(buf), f)) { fclose(f); continue; } fclose(f);
                CPU_ZERO(&node_cpus); bool mine = false;
                for(char* tok = strtok(buf, ",\n"); tok; tok = strtok(NULL, ",\n"))
                {
                    int a, b;
                    if(sscanf(tok, "%d-%d", &a, &b) != 2) { if(sscanf(tok, "%d", &a) != 1) continue; b = a; }
                    for(int c=a; c<=b; c++) { CPU_SET(c, &node_cpus); if(c == cpu) mine = true; }
                }
                pin = mine;
            }
        }
        for(int k=0; k<n; k++)
        {
            threads.emplace_back([this] { run(); });
            if(pin) (void)pthread_setaffinity_np(threads.back().native_handle(), sizeof(node_cpus), &node_cpus);
        }
    }
    ~hz_copy_pool()
    {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_work.notify_all();
        for(auto& t : threads) t.join();
    }
    void map_pages(unsigned char* p, size_t n)
    {
        const uintptr_t page = 4096, lo = ((uintptr_t)p + page-1) & ~(page-1), hi = ((uintptr_t)p + n) & ~(page-1);
        if(hi <= lo) return;
        if(populate_works) { (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE); return; }     /* (a failure: the copies fault the pages in themselves) */
        /* an older kernel: a write that changes nothing, one per page (atomic: a copy into the same page may be running) */
        for(uintptr_t a = lo; a < hi; a += page) (void)__atomic_fetch_add((unsigned char*)a, 0, __ATOMIC_RELAXED);
    }
    void execute(const task_t& t)
    {
        switch(t.kind)
        {
        case COPY: memcpy(t.dst, t.src, t.n); break;
        case MAP:  map_pages(t.dst, t.n); break;
        case FILL:
            for(int r=0; r<t.rows; r++) hz_sky_fill(t.dst, t.lo + (size_t)r*t.pitch, t.lo + (size_t)r*t.pitch + t.n, t.sky);
            if(t.left) t.left->fetch_sub(1, std::memory_order_release);
            break;
        case SCATTER:
            for(size_t k=0; k<t.nblobs; k++)
            {
                const uint32_t* blob = t.chunk + t.offs[k];
                /* The terrain goes on top of the sky, which has to be there first: a blob waits for the piece(s) of its
                 * sector that hold its rows.  Sky tasks queue behind blobs (q_lo), so the ones this blob waits for may
                 * not have been taken by any thread yet: the waiting thread takes sky tasks itself. */
                const int yo = (int)(blob[0] & 0xFFFFu);
                const bool prefilled = yo < t.sc->y_pre;
                if(prefilled)
                    for(int b = yo/t.sc->band_rows; b <= (yo + HZ_BLOB_ROWS-1)/t.sc->band_rows && b < t.sc->nbands; b++)
                        while(t.sc->band_left[(size_t)t.sector*t.sc->nbands + b].load(std::memory_order_acquire) > 0)
                            if(!run_one_low()) std::this_thread::yield();
                if(hz_blob_scatter_mode(blob, &t.sc->dst, prefilled ? 0 : 1) != 0) t.sc->bad.store(1);
            }
            break;
        }
    }
    void finished(const task_t& t)      /* m held */
    {
        if(--t.batch->pending == 0) cv_done.notify_all();
    }
    /* a thread that waits for sky takes one sky task; false: none queued (others are working on them) */
    bool run_one_low()
    {
        task_t t;
        {
            std::lock_guard<std::mutex> lk(m);
            if(q_lo.empty()) return false;
            t = q_lo.front(); q_lo.pop_front();
        }
        execute(t);
        std::lock_guard<std::mutex> lk(m);
        finished(t);
        return true;
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for(;;)
        {
            cv_work.wait(lk, [this] { return stop || !q_hi.empty() || !q_lo.empty(); });
            if(stop) return;
            std::deque<task_t>& q = !q_hi.empty() ? q_hi : q_lo;
            const task_t t = q.front(); q.pop_front();
            lk.unlock();
            execute(t);
            lk.lock();
            finished(t);
        }
    }
    /* the tasks of one job: [dst, dst+n) in parts of at least `grain` bytes, at most one per thread */
    void push(batch_t* b, unsigned char* d, const unsigned char* s, size_t n, size_t grain)
    {
        size_t nparts = threads.size(); if(nparts > n/grain + 1) nparts = n/grain + 1;
        std::lock_guard<std::mutex> lk(m);
        for(size_t k=0; k<nparts; k++)
        {
            const size_t lo = n*k/nparts, hi = n*(k+1)/nparts;
            task_t t = {};
            t.kind = s ? COPY : MAP; t.dst = d + lo; t.src = s ? s + lo : NULL; t.n = hi - lo; t.batch = b;
            (s ? q_hi : q_lo).push_back(t);
            b->pending++;
        }
        cv_work.notify_all();
    }
    /* several tasks of one batch at once (one trip through the lock) */
    void push_tasks(batch_t* b, std::vector<task_t>& ts)
    {
        if(ts.empty()) return;
        {
            std::lock_guard<std::mutex> lk(m);
            for(task_t& t : ts) { t.batch = b; (t.kind == FILL || t.kind == MAP ? q_lo : q_hi).push_back(t); }
            b->pending += (int)ts.size();
        }
        cv_work.notify_all();
        ts.clear();
    }
    /* ... sky tasks that are not ahead of anything: into the queue that is served first */
    void push_tasks_hi(batch_t* b, std::vector<task_t>& ts)
    {
        if(ts.empty()) return;
        {
            std::lock_guard<std::mutex> lk(m);
            for(task_t& t : ts) { t.batch = b; q_hi.push_back(t); }
            b->pending += (int)ts.size();
        }
        cv_work.notify_all();
        ts.clear();
    }
    void wait(batch_t* b)
    {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [b] { return b->pending == 0; });
    }
};

static hz_copy_pool* copy_pool()
{
    /* one pool per process, created on first use, never torn down (its threads
     * sleep on a condition variable) */
    static hz_copy_pool* pool = nullptr;
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if(!pool)
    {
        /* 24: a 16000x4000 panorama into kept buffers takes 5.6 / 4.9 / 4.2 ms with 8 / 12 / 24 threads on the 2 x 64-core
         * host of an 8-GPU node (round 4, profiles/r4_host_inclusive.txt); at most an eighth of the machine's hardware
         * threads, so that eight processes, one per GPU, do not get in each other's way.  HZ_COPY_THREADS: the one switch
         * that belongs to the process, not to a context. */
        const unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 32 ? (int)(hw/8 < 24 ? hw/8 : 24) : 4;
        const char* e = getenv("HZ_COPY_THREADS");
        if(e && atoi(e) > 0) n = atoi(e);
        if(hw && (unsigned)n > hw) n = (int)hw;
        pool = new hz_copy_pool(n);
    }
    return pool;
}

/* ------------------------------------------------------------------------ */
/* the state of a context's host path                                        */

#define HZ_HOST_MAX_SECTORS 8
#define HZ_HOST_JOBS        2           /* panoramas between hz_hip_host_begin() and hz_hip_host_end() */

struct hz_hostjob
{
    bool      active;
    bool      clears;                   /* its conversions cleared the framebuffers behind themselves */
    hz_view_t view;
    uint32_t  flags;                    /* HZ_BLOB_*: what its blobs carry */
    int       nsec;
    int       col[HZ_HOST_MAX_SECTORS+1];   /* image columns: sector s = [col[s], col[s+1]) */
    int       out_col0, out_w;          /* the caller's buffers are [H][out_w] and start at image column out_col0 */
    size_t    off[HZ_HOST_MAX_SECTORS];     /* where sector s's stream starts in d_hs (words; a multiple of the chunk size) */
    size_t    cap[HZ_HOST_MAX_SECTORS];     /* ... and the room it has */
    hz_copy_pool::scatter_t sc;
    std::vector<std::atomic<int>>* band_left;
    std::vector<float>* tanel;          /* the job's own copy: the scatter tasks read it */
    hz_copy_pool::batch_t filled;
    /* device side */
    uint32_t*     d_hs;                 /* the streams of blobs of the job's sectors */
    size_t        hs_capacity;          /* words */
    unsigned int* d_cursor;             /* 4 words per sector: [0] words in use, [1] blobs, [2] nonzero: a blob did not fit */
    unsigned int* h_cursor;             /* the same in pinned memory */
    hipEvent_t    ev_known[HZ_HOST_MAX_SECTORS];    /* sector s's cursor words have reached h_cursor */
    std::chrono::steady_clock::time_point t_begin;
    double t_sky_queued, t_queued[HZ_HOST_MAX_SECTORS];     /* host_times: ms since t_begin when the sky tasks / sector s's work had been queued */
};

struct hz_hoststate
{
    hipStream_t    cstream[HZ_COPY_STREAMS];
    unsigned char* h_stage[HZ_STAGE_SLOTS];     /* slot k of ONE pinned allocation (h_stage[0]): a copy may span consecutive slots */
    hipEvent_t     ev_stage[HZ_STAGE_SLOTS];
    hipEvent_t     ev_band[HZ_HOST_BANDS];
    hz_hostjob     job[HZ_HOST_JOBS];
    int            next_begin, next_end;    /* jobs are ended in the order they were begun */
    /* internal output buffers of the dense path */
    unsigned char* d_bgr;
    float*         d_ranges;
    int32_t*       d_index;
    uint32_t*      d_z24;
};

static int ensure_host(hz_dev_t* d)
{
    if(d->host) return 0;
    hz_hoststate* h = new hz_hoststate();
    memset((void*)h, 0, sizeof(*h));
    d->host = h;
    /* The copies get streams of the highest priority: HIP deals its streams onto a handful of hardware queues, per
     * priority level, and a queue is worked through in order - on a queue shared with one of the context's draw streams the
     * copies of sector 0 sat behind the marching kernels of sectors 1 to 3, which had been queued before the copies could be
     * (round 5: first chunk 0.75 ms after its sector was known, profiles/r5_host_inclusive.txt). */
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for(int k=0; k<HZ_COPY_STREAMS; k++) HZ_CHECK(hipStreamCreateWithPriority(&h->cstream[k], hipStreamNonBlocking, prio_hi));
    for(int k=0; k<HZ_HOST_BANDS; k++)   HZ_CHECK(hipEventCreateWithFlags(&h->ev_band[k], hipEventDisableTiming));
    HZ_CHECK(hipHostMalloc((void**)&h->h_stage[0], (size_t)HZ_STAGE_SLOTS*HZ_STAGE_BYTES, hipHostMallocDefault));
    for(int k=0; k<HZ_STAGE_SLOTS; k++)
    {
        h->h_stage[k] = h->h_stage[0] + (size_t)k*HZ_STAGE_BYTES;
        HZ_CHECK(hipEventCreateWithFlags(&h->ev_stage[k], hipEventDisableTiming));
    }
    for(int j=0; j<HZ_HOST_JOBS; j++)
    {
        hz_hostjob& jb = h->job[j];
        jb.band_left = new std::vector<std::atomic<int>>();
        jb.tanel = new std::vector<float>();
        HZ_CHECK(hipMalloc(&jb.d_cursor, 4*HZ_HOST_MAX_SECTORS*sizeof(unsigned int)));
        HZ_CHECK(hipHostMalloc((void**)&jb.h_cursor, 4*HZ_HOST_MAX_SECTORS*sizeof(unsigned int), hipHostMallocDefault));
        for(int s=0; s<HZ_HOST_MAX_SECTORS; s++) HZ_CHECK(hipEventCreateWithFlags(&jb.ev_known[s], hipEventDisableTiming));
    }
    return 0;
}

void hz_hostpath_destroy(hz_dev_t* d)
{
    hz_hoststate* h = d->host;
    if(!h) return;
    for(int k=0; k<HZ_COPY_STREAMS; k++) if(h->cstream[k]) (void)hipStreamSynchronize(h->cstream[k]);
    if(h->h_stage[0]) (void)hipHostFree(h->h_stage[0]);
    for(int k=0; k<HZ_STAGE_SLOTS; k++) if(h->ev_stage[k]) (void)hipEventDestroy(h->ev_stage[k]);
    for(int k=0; k<HZ_HOST_BANDS; k++)   if(h->ev_band[k]) (void)hipEventDestroy(h->ev_band[k]);
    for(int k=0; k<HZ_COPY_STREAMS; k++) if(h->cstream[k]) (void)hipStreamDestroy(h->cstream[k]);
    for(int j=0; j<HZ_HOST_JOBS; j++)
    {
        hz_hostjob& jb = h->job[j];
        (void)hipFree(jb.d_hs); (void)hipFree(jb.d_cursor);
        if(jb.h_cursor) (void)hipHostFree(jb.h_cursor);
        for(int s=0; s<HZ_HOST_MAX_SECTORS; s++) if(jb.ev_known[s]) (void)hipEventDestroy(jb.ev_known[s]);
        delete jb.band_left; delete jb.tanel;
    }
    (void)hipFree(h->d_bgr); (void)hipFree(h->d_ranges); (void)hipFree(h->d_index); (void)hipFree(h->d_z24);
    delete h;
    d->host = NULL;
}

/* ------------------------------------------------------------------------ */
/* without the sky: begin (queue the draws and conversions, start the sky)   */

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
static int sectors_for(const hz_dev_t* d, const hz_view_t* view, bool draws)
{
    if(!draws || d->col0 != 0 || d->col1 != d->W) return 1;        /* (a context that is itself one sector of a panorama; a conversion of a draw already made) */
    int n = d->env.host_sectors;
    if(n <= 0)
    {
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

/* Queues everything the device has to do for one panorama into host memory and starts the sky.  draws: the panorama is
 * drawn here, sector by sector (else: the conversion of the draw already queued, or made again if it was consumed).
 * Returns the job's number, -1 on an error, -2 if this panorama has to take the dense path (no room for the stream). */
static int host_begin(hz_dev_t* d, const hz_view_t* view, const float* tanel, bool draws,
                      unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    hz_hoststate* h = d->host;
    hz_hostjob& jb = h->job[h->next_begin % HZ_HOST_JOBS];
    if(jb.active) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_begin: %d panoramas are in flight already: end one first", HZ_HOST_JOBS); return -1; }
    const int H = d->H;
    const uint32_t flags = ((ranges || z24) ? HZ_BLOB_PACKED : bgr ? HZ_BLOB_RED : 0u) | (index ? HZ_BLOB_INDEX : 0u);
    if(ranges && !tanel) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: ranges requested without a tanel table"); return -1; }
    jb.nsec = sectors_for(d, view, draws);
    jb.out_col0 = d->col0; jb.out_w = d->col1 - d->col0;
    for(int s=0; s<=jb.nsec; s++) jb.col[s] = s == jb.nsec ? d->col1 : d->col0 + (int)((long long)jb.out_w*s/jb.nsec) / 64 * 64;
    size_t need = 0;
    for(int s=0; s<jb.nsec; s++)
    {
        jb.cap[s] = hs_words_needed(jb.col[s+1] - jb.col[s], H, flags);
        if(jb.cap[s] >= ((size_t)1 << 32)) return -2;              /* (a stream is addressed in 32 bits) */
        jb.off[s] = need; need += jb.cap[s];
    }
    if(need > jb.hs_capacity)
    {
        HZ_CHECK(hz_sync_all(d));
        (void)hipFree(jb.d_hs); jb.d_hs = NULL; jb.hs_capacity = 0;
        if(hipMalloc(&jb.d_hs, need*sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return -2; }
        jb.hs_capacity = need;
    }
    jb.view = *view; jb.flags = flags;
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
    /* How much of the image gets its sky beforehand.  Filling (448 MB in ~1.05 ms with this pool: tools/hostfill_bench.c) and
     * scattering get in each other's way when they run at the same time - side by side they take 4 ms where one after the
     * other they take 2.4 (tools/scatter_bench.c, profiles/r5_host_microbenchmarks.txt) - so only as much of the image is
     * filled beforehand as there is time for until the first sector's blobs arrive: all of it for a call in one sector (the
     * draw takes longer than the fill), the upper 60 % for 2 sectors, 30 % for more - sky for the most part -, and
     * nothing when another panorama is in flight (its blobs are arriving now).  Below that row a blob writes the sky
     * pixels of its own tile (hz_blob_scatter_mode: every byte once), and the tiles that turn out to have no blob are filled
     * when their sector has been walked.  HZ_HOST_PREFILL=percent overrides. */
    {
        int percent = jb.nsec <= 1 ? 100 : jb.nsec == 2 ? 60 : 30;     /* (4 sectors of 16000 x 4000: 3.36 ms with 45 %, 3.25 with 60, 3.15 with 30) */
        if(h->next_begin != h->next_end) percent = 0;
        const char* e = getenv("HZ_HOST_PREFILL");
        if(e && atoi(e) >= 0 && atoi(e) <= 100) percent = atoi(e);
        sc.y_pre = (int)((long long)H*percent/100) / HZ_BLOB_ROWS * HZ_BLOB_ROWS;
        if(percent >= 100) sc.y_pre = (H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS*HZ_BLOB_ROWS;
    }
    const int widest = jb.col[1] - jb.col[0];
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

    const bool prof = d->profiling != 0;
    const int user_col0 = d->col0, user_col1 = d->col1;
    int rc = 0;
    hipError_t err = hipSuccess;
    const char* what = "";
    #define HZ_TRY(call) do { if(err == hipSuccess && rc == 0) { err = (call); if(err != hipSuccess) what = #call; } } while(0)
    HZ_TRY(hipMemsetAsync(jb.d_cursor, 0, 4*HZ_HOST_MAX_SECTORS*sizeof(unsigned int), d->rstream));
    jb.clears = d->env.resolve_clears != 0;
    for(int s=0; s<jb.nsec && rc == 0 && err == hipSuccess; s++)
    {
        if(draws)
        {
            d->col0 = jb.col[s]; d->col1 = jb.col[s+1];
            if(hz_draw_impl(d, view) != 0) { rc = -1; break; }
        }
        else if(hz_fb_refill(d) != 0) { rc = -1; break; }
        if(hz_rstream_after_draw(d) != 0) { rc = -1; break; }
        const int SW = d->col1 - d->col0;
        if(prof && s == jb.nsec-1) HZ_TRY(hipEventRecord(d->ev[4], d->rstream));
        hz_hostpack_t hp = { jb.d_hs + jb.off[s], jb.d_cursor + 4*s, (unsigned int)jb.cap[s], (unsigned int)(HZ_STAGE_BYTES/4), flags };
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
        HZ_TRY(hipMemcpyAsync(jb.h_cursor + 4*s, jb.d_cursor + 4*s, 4*sizeof(unsigned int), hipMemcpyDeviceToHost, d->rstream));
        HZ_TRY(hipEventRecord(jb.ev_known[s], d->rstream));
        jb.t_queued[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count();
    }
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
    if(rc != 0)
    {
        pool->wait(&jb.filled);
        jb.active = false;
        h->next_begin--;
        return -1;
    }
    return (h->next_begin - 1) % HZ_HOST_JOBS;
}

/* ------------------------------------------------------------------------ */
/* ... end: the streams through the staging ring, the blobs into their places */

static int host_end(hz_dev_t* d)
{
    hz_hoststate* h = d->host;
    hz_hostjob& jb = h->job[h->next_end % HZ_HOST_JOBS];
    if(!jb.active) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: no panorama is in flight"); return -1; }
    hz_copy_pool* pool = copy_pool();
    std::lock_guard<std::mutex> one(pool->busy);
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - jb.t_begin).count(); };
    const double t_enter = since();
    double t_known[HZ_HOST_MAX_SECTORS] = { 0 }, t_first = 0, t_arrived = 0, t_waited = 0;

    struct chunk_t { int sector; size_t w0, nw; int ev; };          /* ev: the staging slot whose event says the chunk has arrived (the first slot of its copy) */
    size_t ncopies = 0;
    int run_of[HZ_HOST_MAX_SECTORS] = { 0 };                        /* copies issued for sector s so far */
    std::deque<chunk_t> chunks;                         /* (grows as the sectors' lengths become known) */
    std::deque<hz_copy_pool::batch_t> done;             /* one per chunk: its scatter tasks (references stay valid as it grows) */
    std::vector<std::vector<size_t>> offs(jb.nsec);     /* where the blobs of sector s start, chunk after chunk */
    size_t noffs[HZ_HOST_MAX_SECTORS] = { 0 }, first[HZ_HOST_MAX_SECTORS] = { 0 };
    size_t total_words = 0, total_blobs = 0;
    /* below y_pre: which tiles of sector s (4 rows x <= 2048 columns, as k_pack_host cuts them) have sent a blob; the others
     * get their sky when the sector's last chunk has been walked */
    const int H = d->H, y_pre = jb.sc.y_pre, ty0 = y_pre/HZ_BLOB_ROWS, nty = (H + HZ_BLOB_ROWS-1)/HZ_BLOB_ROWS;
    std::vector<std::vector<unsigned char>> seen(jb.nsec);
    size_t chunks_of[HZ_HOST_MAX_SECTORS] = { 0 }, walked_of[HZ_HOST_MAX_SECTORS] = { 0 };
    bool absent_done[HZ_HOST_MAX_SECTORS] = { false };
    hz_copy_pool::batch_t late_sky = { 0 };
    struct { unsigned char* p; size_t px_bytes; int sky; } bufs[4];
    int nbuf = 0;
    if(jb.sc.dst.bgr)    bufs[nbuf++] = { jb.sc.dst.bgr, 3, HZ_SKY_BGR };
    if(jb.sc.dst.ranges) bufs[nbuf++] = { (unsigned char*)jb.sc.dst.ranges, 4, HZ_SKY_RANGES };
    if(jb.sc.dst.index)  bufs[nbuf++] = { (unsigned char*)jb.sc.dst.index, 4, HZ_SKY_INDEX };
    if(jb.sc.dst.z24)    bufs[nbuf++] = { (unsigned char*)jb.sc.dst.z24, 4, HZ_SKY_Z24 };
    /* the sky of sector s's tiles without a blob (rows from y_pre down): one task per run of such tiles in a column of tiles */
    auto fill_absent = [&](int s)
    {
        if(absent_done[s] || ty0 >= nty) { absent_done[s] = true; return; }
        absent_done[s] = true;
        const int sw = jb.col[s+1] - jb.col[s], ntx = (sw + HZ_BLOB_COLS-1)/HZ_BLOB_COLS;
        std::vector<hz_copy_pool::task_t> ts;
        for(int tx=0; tx<ntx; tx++)
        {
            const size_t x0 = (size_t)(jb.col[s] - jb.out_col0) + (size_t)tx*HZ_BLOB_COLS;
            const size_t w = (size_t)(sw - tx*HZ_BLOB_COLS < HZ_BLOB_COLS ? sw - tx*HZ_BLOB_COLS : HZ_BLOB_COLS);
            for(int ty=ty0; ty<nty; )
            {
                if(!seen[s].empty() && seen[s][(size_t)(ty - ty0)*ntx + tx]) { ty++; continue; }
                int t1 = ty + 1;
                while(t1 < nty && t1 - ty < 64 && (seen[s].empty() || !seen[s][(size_t)(t1 - ty0)*ntx + tx])) t1++;
                const int y0 = ty*HZ_BLOB_ROWS, y1 = t1*HZ_BLOB_ROWS < H ? t1*HZ_BLOB_ROWS : H;
                for(int k=0; k<nbuf; k++)
                {
                    hz_copy_pool::task_t t = {};
                    t.kind = hz_copy_pool::FILL; t.dst = bufs[k].p; t.sky = bufs[k].sky; t.left = NULL;
                    t.lo = ((size_t)y0*jb.out_w + x0)*bufs[k].px_bytes; t.n = w*bufs[k].px_bytes; t.rows = y1 - y0; t.pitch = (size_t)jb.out_w*bufs[k].px_bytes;
                    ts.push_back(t);
                }
                ty = t1;
            }
        }
        pool->push_tasks_hi(&late_sky, ts);
    };
    const size_t chunk_words = HZ_STAGE_BYTES/4;
    int rc = 0, known = 0;
    hipError_t err = hipSuccess;
    const char* what = "";
    #define HZ_TRY(call) do { if(err == hipSuccess) { err = (call); if(err != hipSuccess) what = #call; } } while(0)
    std::vector<hz_copy_pool::task_t> tasks;
    size_t issued = 0, k = 0;
    /* one more sector's length: wait == false only looks */
    auto learn = [&](bool wait) -> bool
    {
        if(known >= jb.nsec || err != hipSuccess || rc != 0) return false;
        if(wait) HZ_TRY(hipEventSynchronize(jb.ev_known[known]));
        else
        {
            const hipError_t q = hipEventQuery(jb.ev_known[known]);
            if(q == hipErrorNotReady) { (void)hipGetLastError(); return false; }
            if(q != hipSuccess) { err = q; what = "hipEventQuery(ev_known)"; }
        }
        if(err != hipSuccess) return false;
        const int s = known++;
        t_known[s] = since();
        const unsigned int* c = jb.h_cursor + 4*s;
        if(c[2]) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: the stream of blobs overflowed (%zu words)", jb.cap[s]); rc = -1; return false; }
        offs[s].resize((size_t)c[1] + 1);
        total_words += c[0]; total_blobs += c[1];
        if(ty0 < nty) seen[s].assign((size_t)(nty - ty0)*(size_t)((jb.col[s+1] - jb.col[s] + HZ_BLOB_COLS-1)/HZ_BLOB_COLS), 0);
        chunks_of[s] = ((size_t)c[0] + chunk_words-1)/chunk_words;
        if(chunks_of[s] == 0) fill_absent(s);              /* (a sector without any terrain) */
        for(size_t w0 = 0; w0 < c[0]; w0 += chunk_words)
        {
            chunks.push_back({ s, w0, w0 + chunk_words < c[0] ? chunk_words : c[0] - w0, 0 });
            done.push_back({ 0 });
        }
        return true;
    };
    for(;;)
    {
        /* whatever has become known; if there is nothing else to do, wait for the next sector */
        while(learn(false)) {}
        if(k == chunks.size()) { if(known == jb.nsec || !learn(true)) break; }
        if(err != hipSuccess || rc != 0) break;
        /* keep the copy engine up to HZ_STAGE_SLOTS - 4 chunks ahead of the chunk the host threads get next.  One copy moves
         * up to four consecutive chunks of a sector (consecutive staging slots: the ring is one allocation): a copy costs the
         * engine ~22 us on top of its bytes - 4 MB copies ran at 46 GB/s where 16 MB ones reach 57 (tools/zero_copy.hip) - ,
         * but what a copy holds can only be scattered when all of it has arrived: a sector's first copy is one chunk, its
         * second two, then four. */
        while(issued < chunks.size() && issued < k + HZ_STAGE_SLOTS - 4 && err == hipSuccess)
        {
            const int slot = (int)(issued % HZ_STAGE_SLOTS), sector = chunks[issued].sector;
            size_t g = run_of[sector] == 0 ? 1 : run_of[sector] == 1 ? 2 : 4;
            /* ... and the last chunks of the last sector one by one again: what arrives last is scattered with nothing left to hide behind */
            if(known == jb.nsec && sector == jb.nsec-1 && chunks.size() - issued <= 3) g = 1;
            if(g > (size_t)(HZ_STAGE_SLOTS - slot)) g = HZ_STAGE_SLOTS - slot;                      /* (no copy wraps round the ring) */
            while(g > 1 && (issued + g > chunks.size() || issued + g > k + HZ_STAGE_SLOTS - 4 || chunks[issued + g-1].sector != sector)) g--;
            size_t nw = 0;
            for(size_t i=0; i<g; i++)
            {
                if(issued + i >= HZ_STAGE_SLOTS) pool->wait(&done[issued + i - HZ_STAGE_SLOTS]);     /* the slot's previous chunk has been scattered */
                chunks[issued + i].ev = slot;
                nw += chunks[issued + i].nw;
            }
            hipStream_t cs = h->cstream[ncopies % HZ_COPY_STREAMS];
            const chunk_t& c = chunks[issued];
            HZ_TRY(hipMemcpyAsync(h->h_stage[slot], jb.d_hs + jb.off[sector] + c.w0, nw*sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
            HZ_TRY(hipEventRecord(h->ev_stage[slot], cs));
            issued += g; ncopies++; run_of[sector]++;
        }
        if(err != hipSuccess || k >= issued) continue;
        const int slot = (int)(k % HZ_STAGE_SLOTS);
        const double t_w0 = since();
        const chunk_t c = chunks[k];
        HZ_TRY(hipEventSynchronize(h->ev_stage[c.ev]));
        if(err != hipSuccess) break;
        t_waited += since() - t_w0;
        if(k == 0) t_first = since();
        t_arrived = since();
        const uint32_t* chunk = (const uint32_t*)h->h_stage[slot];
        size_t* const o = offs[c.sector].data() + noffs[c.sector];
        const size_t room = offs[c.sector].size() - noffs[c.sector];
        const size_t nb = hz_blob_walk(chunk, c.nw, first[c.sector], o, room, &first[c.sector]);
        if(nb == (size_t)-1 || nb > room) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: chunk %zu of the stream is not a sequence of blobs", k); rc = -1; break; }
        noffs[c.sector] += nb;
        if(!seen[c.sector].empty())
        {
            const int ntx = (jb.col[c.sector+1] - jb.col[c.sector] + HZ_BLOB_COLS-1)/HZ_BLOB_COLS;
            for(size_t b=0; b<nb; b++)
            {
                const uint32_t* blob = chunk + o[b];
                const int yo = (int)(blob[0] & 0xFFFFu), tx = ((int)blob[1] - (jb.col[c.sector] - jb.out_col0))/HZ_BLOB_COLS;
                if(yo >= y_pre && yo/HZ_BLOB_ROWS < nty && tx >= 0 && tx < ntx) seen[c.sector][(size_t)(yo/HZ_BLOB_ROWS - ty0)*ntx + tx] = 1;
            }
        }
        if(++walked_of[c.sector] == chunks_of[c.sector]) fill_absent(c.sector);
        /* tasks of ~256 KB of blobs */
        for(size_t b0=0; b0<nb; )
        {
            size_t b1 = b0 + 1;
            while(b1 < nb && o[b1] - o[b0] < 65536) b1++;
            hz_copy_pool::task_t t = {};
            t.kind = hz_copy_pool::SCATTER; t.sc = &jb.sc; t.sector = c.sector; t.chunk = chunk; t.offs = o + b0; t.nblobs = b1 - b0;
            tasks.push_back(t);
            b0 = b1;
        }
        pool->push_tasks(&done[k], tasks);
        k++;
    }
    #undef HZ_TRY
    /* (whatever ended the loop early - an error: every sector still gets its sky, the tasks below name this frame's variables) */
    if(err == hipSuccess && rc == 0) for(int s=0; s<jb.nsec; s++) if(s < known) fill_absent(s);
    for(size_t i=0; i<done.size(); i++) pool->wait(&done[i]);
    const double t_scattered = since();
    pool->wait(&late_sky);
    pool->wait(&jb.filled);
    if(d->env.host_times)
    {
        fprintf(stderr, "hz_hip host path: %.1f MB of blobs (%zu) in %d sector(s), %zu copies, for %.1f MB of results; ms since the call began: sky tasks queued %.2f, sectors queued",
                4e-6*(double)total_words, total_blobs, jb.nsec, ncopies,
                1e-6*(double)jb.out_w*d->H*((jb.sc.dst.bgr ? 3 : 0) + (jb.sc.dst.ranges ? 4 : 0) + (jb.sc.dst.index ? 4 : 0) + (jb.sc.dst.z24 ? 4 : 0)), jb.t_sky_queued);
        for(int s=0; s<jb.nsec; s++) fprintf(stderr, " %.2f", jb.t_queued[s]);
        fprintf(stderr, ", end() entered %.2f, sectors known", t_enter);
        for(int s=0; s<jb.nsec; s++) fprintf(stderr, " %.2f", t_known[s]);
        fprintf(stderr, ", first chunk here %.2f, last chunk here %.2f (%.2f spent waiting for chunks), blobs in place %.2f, sky and everything %.2f\n",
                t_first, t_arrived, t_waited, t_scattered, since());
    }
    jb.active = false;
    h->next_end++;
    if(err != hipSuccess)
    {
        for(int i=0; i<HZ_COPY_STREAMS; i++) (void)hipStreamSynchronize(h->cstream[i]);
        (void)hipStreamSynchronize(d->rstream);
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_host_end: %s -> %s", what, hipGetErrorString(err));
        fprintf(stderr, "hz_hip: %s\n", g_last_error);
        return -1;
    }
    if(rc == 0 && jb.sc.bad.load()) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_to_host: a blob does not describe pixels of this image"); rc = -1; }
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
    if(ensure_out_buffers(d, bgr != NULL, ranges != NULL, index != NULL, z24 != NULL) != 0) return -1;
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
