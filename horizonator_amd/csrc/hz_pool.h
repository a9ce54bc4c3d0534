/* hz_pool.h - the pool of host threads that writes results into the caller's buffers (hz_hostpath.cpp): sky constants
 * (hz_sky_fill), blobs of terrain pixels (hz_blob_scatter_mode: the readback conversion of reference
 * horizonator-lib.c:1013-1025 in the host's vector unit), plain copies and page mapping for the dense path.  One pool per
 * process, made on first use or by hz_hip_host_prepare(); its threads sleep on a condition variable. */
#pragma once

#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "hz_scatter.h"

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

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

    /* the CPUs of NUMA node `node` (Linux sysfs); false: no such node */
    static bool cpus_of_node(int node, cpu_set_t* cpus)
    {
        char path[96]; snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r"); if(!f) return false;
        char buf[4096]; const bool got = fgets(buf, sizeof(buf), f) != NULL; fclose(f);
        if(!got) return false;
        CPU_ZERO(cpus); int n = 0;
        for(char* tok = strtok(buf, ",\n"); tok; tok = strtok(NULL, ",\n"))
        {
            int a, b;
            if(sscanf(tok, "%d-%d", &a, &b) != 2) { if(sscanf(tok, "%d", &a) != 1) continue; b = a; }
            for(int c=a; c<=b && c<CPU_SETSIZE; c++) { CPU_SET(c, cpus); n++; }
        }
        return n > 0;
    }
    /* n threads.  gpu_node: the NUMA node the context's GPU hangs off (-1: unknown).  The threads stay on ONE node - where
     * they write the caller's pages (first touched by them, if the buffers are fresh), read the landing areas and meet in
     * this pool's lock: with the threads wherever the scheduler put them on the box's two sockets a call took 3.7 ms,
     * with the process on either node 3.3-3.4, a series with two in flight 3.7 against 3.1-3.3 (profiles/
     * r6_host_path.txt).  HZ_COPY_NODE = gpu (default) | here (the node of the thread that makes the pool) | any | a number. */
    hz_copy_pool(int n, int gpu_node)
    {
        /* (EINVAL on a private anonymous page = the flag is unknown to this kernel; any later failure is about
         * the caller's buffer - a pinned or device mapping, an unmapped range - and only skips that buffer) */
        void* probe = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if(probe != MAP_FAILED)
        {
            if(madvise(probe, 4096, MADV_POPULATE_WRITE) != 0) populate_works = false;
            munmap(probe, 4096);
        }
        cpu_set_t node_cpus; bool pin = false;
        const char* where = getenv("HZ_COPY_NODE");
        if(!where) where = "gpu";
        if(strcmp(where, "here") == 0)
        {
            const int cpu = sched_getcpu();
            for(int node=0; node<64 && !pin; node++)
            {
                if(!cpus_of_node(node, &node_cpus)) break;
                pin = cpu >= 0 && cpu < CPU_SETSIZE && CPU_ISSET(cpu, &node_cpus);
            }
        }
        else if(strcmp(where, "gpu") == 0) pin = gpu_node >= 0 && cpus_of_node(gpu_node, &node_cpus);
        else if(where[0] >= '0' && where[0] <= '9') pin = cpus_of_node(atoi(where), &node_cpus);
        /* (a node with fewer CPUs than threads - a container's slice of the machine: the threads stay where they are) */
        if(pin && CPU_COUNT(&node_cpus) < n) pin = false;
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

/* gpu_node: see the constructor (used when this call makes the pool) */
static inline hz_copy_pool* copy_pool(int gpu_node = -1)
{
    /* one pool per process, created on first use, never torn down (its threads
     * sleep on a condition variable) */
    static hz_copy_pool* pool = nullptr;
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if(!pool)
    {
        /* an eighth of the machine's hardware threads - eight processes, one per GPU, then do not get in each other's way -,
         * at most 32: a 16000x4000 panorama into kept buffers took 5.6 / 4.9 / 4.2 ms with 8 / 12 / 24 threads in round 4, and
         * on the 2 x 64-core host of round 6 (three alternating runs each, threads on the GPU's NUMA node: a call / a series
         * with two in flight) 3.6 / 3.0 ms with 24, 3.4 / 2.8 with 32; 48 and more lose again (profiles/r6_host_path.txt).
         * HZ_COPY_THREADS: the one switch that belongs to the process, not to a context. */
        const unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 32 ? (int)(hw/8 < 32 ? hw/8 : 32) : 4;
        const char* e = getenv("HZ_COPY_THREADS");
        if(e && atoi(e) > 0) n = atoi(e);
        if(hw && (unsigned)n > hw) n = (int)hw;
        pool = new hz_copy_pool(n, gpu_node);
    }
    return pool;
}

