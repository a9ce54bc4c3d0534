/* hz_k_common.h - part of hz_kernels.hip (included there, in this order; one translation unit):
 * kernel parameters, the records that travel between kernels, device helpers, k_clip. */
#pragma once

/* ------------------------------------------------------------------------ */
/* kernel parameters                                                         */


/* the one place fragments enter the framebuffer.
 * Builds with -DHZ_EXPERIMENTS only (tools/experiments.py; never the library that ships): p.exp_fb
 * (HZ_EXP_FB_MARCH / HZ_EXP_FB_BIG, experiments with WRONG pictures - profiles/r3_experiments.json):
 * 1 = the fragment is dropped here, 2 = a plain store instead of the atomic minimum: what the atomics
 * cost, i.e. what a rasteriser that owned its pixels could gain; p.debug (HZ_MARCH_DEBUG): timing splits */
#ifdef HZ_EXPERIMENTS
#define HZ_DEBUG(p) ((p).debug)
#else
#define HZ_DEBUG(p) 0
#endif
template<int WHO = HZ_WHO_OTHER>
__device__ static inline void hz_fb_min(unsigned long long* fb, const hz_params_t& p, int px, int py, unsigned long long key)
{
    const int x = px - p.col0;
#ifdef HZ_EXPERIMENTS
    if(WHO != HZ_WHO_OTHER && p.exp_fb[WHO == HZ_WHO_OTHER ? 0 : WHO])
    {
        if(p.exp_fb[WHO] == 2) { p.touched[(size_t)py*p.seg_stride + (x >> HZ_SEG_LOG2)] = 1; fb[(size_t)py*p.SW + x] = key; }
        return;
    }
#endif
    /* A framebuffer is smaller than 4 GB (hz_hip_create: images of up to 2^29 pixels; the reference's own limit,
     * GL_MAX_RENDERBUFFER_SIZE squared on llvmpipe, is 2^28): byte offsets in 32 bits, so that each of the two
     * accesses is one multiply-add and a shift on top of a scalar base - the 64-bit form is a 64-bit multiply-add,
     * two 64-bit shift-adds and a sign extension per address, for every fragment of the draw. */
    const uint32_t ot = (uint32_t)py*(uint32_t)p.seg_stride + ((uint32_t)x >> HZ_SEG_LOG2);
    const uint32_t ob = ((uint32_t)py*(uint32_t)p.SW + (uint32_t)x) << 3;
    *(p.touched + ot) = 1;
    atomicMin((unsigned long long*)((char*)fb + ob), key);
}
/* the framebuffer word of pixel (px, py), for the reads in front of an atomic */
__device__ static inline const unsigned long long* hz_fb_word(const unsigned long long* fb, const hz_params_t& p, int px, int py)
{
    return (const unsigned long long*)((const char*)fb + (((uint32_t)py*(uint32_t)p.SW + (uint32_t)(px - p.col0)) << 3));
}


/* triangles that cross a plane of the view volume: their ids go to k_clip.
 * One atomic per wave.  Ids that do not fit are not stored, but still counted:
 * counters[4] > capacity makes k_clip redo the job without the queue. */
__device__ static inline void hz_queue_clip(const mr_queue_t& q, bool want, uint32_t prim, int lane)
{
    const unsigned long long m = __ballot(want);
    if(!m) return;
    uint32_t base = 0;
    if(lane == (int)__builtin_ctzll(m)) base = atomicAdd(&q.counters[4], (uint32_t)__popcll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if(!want) return;
    const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if(at < q.clip_capacity) q.clip[at] = prim;
}

/* An empty queue set is all zeros ([2] and [5] hold the COMPLEMENT of the first
 * invalid index, raised with atomicMax).  The queue sets of a framebuffer are
 * emptied together with it: by the conversion that clears behind itself (one
 * thread, hz_counters_consume) or by the memset that clears it otherwise - no
 * launch of its own in front of every marching kernel (on that stream it was
 * 40-90 us between consecutive panoramas), and no arrival counter in k_big
 * (4096 atomics on one address: + 40 us per launch, measured). */
__device__ static inline unsigned int* hz_qshard(unsigned int* counters, int s) { return counters + HZ_QSHARD0 + s*HZ_QSHARD_STRIDE; }
__device__ static inline const unsigned int* hz_qshard(const unsigned int* counters, int s) { return counters + HZ_QSHARD0 + s*HZ_QSHARD_STRIDE; }
/* the items shard s holds (those at and beyond its first overflow were drawn by their producer) */
__device__ static inline unsigned int hz_queue_nitems_of(const unsigned int* counters, int s)
{
    const unsigned int* c = hz_qshard(counters, s);
    return min(c[1], ~c[2]);
}
/* is item slot g in use?  (g = HZ_QSLOT(l, s, sl) taken apart) */
__device__ static inline bool hz_queue_item_valid(const unsigned int* counters, unsigned int g, int sl)
{
    const unsigned int block = g >> HZ_QBLOCK_LOG2;
    const unsigned int l = ((block >> sl) << HZ_QBLOCK_LOG2) | (g & (HZ_QBLOCK-1));
    return l < hz_queue_nitems_of(counters, (int)(block & ((1u << sl) - 1u)));
}
/* the item slots a consumer has to look at: [0, that) */
__device__ static inline unsigned int hz_queue_span(const unsigned int* counters, int sl)
{
    unsigned int m = 0;
    #pragma unroll
    for(int s=0; s<HZ_QSHARDS; s++) m = max(m, hz_queue_nitems_of(counters, s));        /* (shards a draw does not use stay at zero) */
    return ((m + HZ_QBLOCK-1) & ~(unsigned int)(HZ_QBLOCK-1)) << sl;
}
__device__ static inline void hz_queue_totals(const unsigned int* counters, unsigned int* records, unsigned int* items)
{
    unsigned int r = 0, n = 0;
    #pragma unroll
    for(int s=0; s<HZ_QSHARDS; s++) { r += hz_qshard(counters, s)[0]; n += hz_qshard(counters, s)[1]; }
    *records = r; *items = n;
}
/* room for nrec records and nitems items from shard `shard`: the shard's own numbers of the first record and of the first
 * item (their slots: HZ_QSLOT(number, shard, sl)); false: they do not fit - the caller draws them itself, and the shard's
 * items from here on are not valid */
__device__ static inline bool hz_queue_reserve(const mr_queue_t& q, int shard, int sl, uint32_t nrec, uint32_t nitems, uint32_t* rec0, uint32_t* item0)
{
    unsigned int* c = hz_qshard(q.counters, shard);
    const unsigned long long both = atomicAdd((unsigned long long*)c, (unsigned long long)nrec | ((unsigned long long)nitems << 32));
    const uint32_t rl = (uint32_t)both, il = (uint32_t)(both >> 32);
    *rec0 = rl; *item0 = il;
    /* (a shard's room: whole blocks, so that its last slot lies inside the array) */
    if((unsigned long long)rl + nrec <= HZ_QSHARD_ROOM(q.bigrec_capacity, sl) && (unsigned long long)il + nitems <= HZ_QSHARD_ROOM(q.bigitem_capacity, sl)) return true;
    atomicMax(c + 2, ~il);
    return false;
}
__device__ static inline void hz_counters_consume(unsigned int* a, unsigned int* b)
{
    /* (one thread of a conversion runs this: loops that stay loops - unrolled they cost k_resolve4 ten registers, and with them
     * its second wave beside the marching kernel's) */
    #pragma unroll 1
    for(int w=0; w<2; w++)
    {
        unsigned int* c = w ? b : a;
        unsigned int records = 0, items = 0, mids = 0;
        #pragma unroll 1
        for(int s=0; s<HZ_QSHARDS; s++)
        {
            unsigned int* q = hz_qshard(c, s);
            records += q[0]; items += q[1]; mids += q[4];
            q[0] = 0u; q[1] = 0u; q[2] = 0u; q[4] = 0u; q[5] = 0u;
        }
        c[HZ_CNT_LAST + 0] = records; c[HZ_CNT_LAST + 1] = items; c[HZ_CNT_LAST + 2] = 0u;
        c[HZ_CNT_LAST + 3] = mids; c[HZ_CNT_LAST + 4] = c[4]; c[HZ_CNT_LAST + 5] = 0u;
        c[4] = 0u;
    }
}
/* k_march: boxes up to p.inline_max pixels are rasterised by the marching wave;
 * larger ones up to HZ_INLINE_MAX_PIX go to k_mid, the rest to k_big */
/* k_big walks a triangle's box in chunks of pixel rows, one wave per chunk
 * (lane = row for the row's span of covered pixels, then lane = pixel): 64 rows,
 * fewer for wide boxes so that a chunk holds at most ~8192 box pixels (the
 * triangles next to the viewer reach thousands of pixels in width; a wave that
 * had 64 such rows to itself would set the kernel's duration).  Producer
 * (queueing) and consumer (k_big) derive the chunking from the box alone. */
__device__ static inline int hz_big_rows_log2(int bw)
{
    return bw <= 128 ? 6 : bw <= 256 ? 5 : bw <= 512 ? 4 : bw <= 1024 ? 3 : bw <= 2048 ? 2 : bw <= 4096 ? 1 : 0;
}
__device__ static inline uint32_t hz_big_chunks(int bw, int bh)
{
    const int rl = hz_big_rows_log2(bw);
    return ((uint32_t)bh + (1u << rl) - 1u) >> rl;
}

/* ------------------------------------------------------------------------ */
/* device helpers                                                            */

/* Inclusive prefix sum / running maximum over the 64 lanes, every lane active.  DPP: within each row of 16 lanes by
 * row_shr 1, 2, 4, 8 (a lane without a source takes `old` = 0, the identity of both), then lane 15 of rows 0 and 2
 * into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31) - six VALU operations with the
 * neighbour read fused in, where __shfl_up() is six ds_bpermute through the LDS crossbar, each with its address
 * arithmetic and a select (24 against 4 cycles apiece: profiles/valu_issue.json). */
#define HZ_DPP(v, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rows), 0xF, false))
__device__ static inline uint32_t mr_scan(uint32_t v, int lane)
{
    (void)lane;
    v += HZ_DPP(v, 0x111, 0xF);         /* row_shr:1 */
    v += HZ_DPP(v, 0x112, 0xF);         /* row_shr:2 */
    v += HZ_DPP(v, 0x114, 0xF);         /* row_shr:4 */
    v += HZ_DPP(v, 0x118, 0xF);         /* row_shr:8 */
    v += HZ_DPP(v, 0x142, 0xA);         /* row_bcast:15 -> rows 1, 3 */
    v += HZ_DPP(v, 0x143, 0xC);         /* row_bcast:31 -> rows 2, 3 */
    return v;
}
__device__ static inline uint32_t mr_scan_max(uint32_t v)
{
    #define HZ_MAXU(a, b) ((a) > (b) ? (a) : (b))
    uint32_t t;
    t = HZ_DPP(v, 0x111, 0xF); v = HZ_MAXU(v, t);
    t = HZ_DPP(v, 0x112, 0xF); v = HZ_MAXU(v, t);
    t = HZ_DPP(v, 0x114, 0xF); v = HZ_MAXU(v, t);
    t = HZ_DPP(v, 0x118, 0xF); v = HZ_MAXU(v, t);
    t = HZ_DPP(v, 0x142, 0xA); v = HZ_MAXU(v, t);
    t = HZ_DPP(v, 0x143, 0xC); v = HZ_MAXU(v, t);
    #undef HZ_MAXU
    return v;
}


/* record <- set-up triangle (planes + coverage); the box and the id are the caller's */
__device__ static inline void hz_rec_from_tri(hz_rec_t& r, const hz_tri_t& t)
{
    hz_edges_of(&r.e, &t);
    r.z_org = t.z_org; r.dzdx = t.dzdx; r.dzdy = t.dzdy;
    r.r_org = t.r_org; r.drdx = t.drdx; r.drdy = t.drdy;
}
/* the planes of a record as a hz_tri_t for hz_tri_fragment() (which reads nothing else) */
__device__ static inline void hz_planes_from_rec(hz_tri_t& t, const hz_rec_t& r)
{
    #pragma unroll
    for(int m=0; m<3; m++) { t.xs[m] = 0; t.ys[m] = 0; }
    t.z_org = r.z_org; t.dzdx = r.dzdx; t.dzdy = r.dzdy;
    t.r_org = r.r_org; t.drdx = r.drdx; t.drdy = r.drdy;
}
/* one pixel centre of a record's triangle: coverage, depth, colour, framebuffer */
template<bool PRETEST>
__device__ static inline void hz_emit_rec(unsigned long long* fb, const hz_params_t& p, const hz_rec_t& r, int px, int py)
{
    if(!hz_edges_cover(&r.e, px, py)) return;
    hz_tri_t t;
    hz_planes_from_rec(t, r);
    uint32_t zi, r8;
    if(!hz_tri_fragment(&t, px, py, &zi, &r8)) return;
    const unsigned long long key = hz_pack(zi, r.prim, r8);
    /* p.pretest_march (second rounds into framebuffers beyond the last-level cache: draw_impl): the marching waves
     * read the word first and leave the atomic out where the fragment cannot win */
    if(PRETEST || p.pretest_march)
    {
        if(key < __hip_atomic_load(hz_fb_word(fb, p, px, py), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            hz_fb_min<HZ_WHO_MARCH>(fb, p, px, py, key);
    }
    else
        hz_fb_min<HZ_WHO_MARCH>(fb, p, px, py, key);
}

/* PRETEST: read the word first and skip the atomic when the fragment cannot
 * win (a stale, larger value only costs the atomic).  It saves atomics but puts
 * a dependent HBM round trip into the loop that calls it; callers that walk
 * many pixels per lane in sequence do better without. */
template<bool PRETEST>
__device__ static inline void hz_emit_t(unsigned long long* fb, const hz_params_t& p,
                                        const hz_tri_t& t, uint32_t prim, int px, int py)
{
    if(!hz_tri_covers(&t, px, py)) return;
    uint32_t zi, r8;
    if(!hz_tri_fragment(&t, px, py, &zi, &r8)) return;
    const unsigned long long key = hz_pack(zi, prim, r8);
    if(PRETEST)
    {
        if(key < __hip_atomic_load(hz_fb_word(fb, p, px, py), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            hz_fb_min(fb, p, px, py, key);
    }
    else
        hz_fb_min(fb, p, px, py, key);
}
__device__ static inline void hz_emit(unsigned long long* fb, const hz_params_t& p,
                                      const hz_tri_t& t, uint32_t prim, int px, int py)
{
    hz_emit_t<true>(fb, p, t, prim, px, py);
}

__device__ static inline hz_wvert_t hz_vertex_at(const hz_params_t& p, const int16_t* mosaic, int i, int j)
{
    const float z = (float)mosaic[(size_t)j*p.N + i];
    return hz_to_window(hz_transform(&p.u, (float)i, (float)j, z), p.halfW, p.halfH);
}

/* clip one triangle of the grid (by id) and hand its pieces on: to the k_big
 * queue, or - `inline_ok` and no room - straight into the framebuffer */
/* (jobs, njobs: the work items of pieces with more than HZ_CLIP_JOB_MIN chunks are left to the caller - 3 words per piece:
 * first item, record, chunks - instead of being written here, one after the other by this thread: k_clip<true>) */
#define HZ_CLIP_JOB_MIN 32
__device__ static void hz_clip_and_draw(const int16_t* mosaic, unsigned long long* fb, const mr_queue_t& q,
                                        const hz_params_t& p, uint32_t prim, bool inline_ok,
                                        hz_cvert_t* bufa, hz_cvert_t* bufb, uint32_t (*jobs)[3] = NULL, int* njobs = NULL)
{
    const uint32_t cell = prim >> 1;
    const int t = prim & 1;
    const int j = cell / (uint32_t)(p.N-1);
    const int i = cell - (uint32_t)j*(uint32_t)(p.N-1);
    /* reference horizonator-lib.c:500-506 */
    const int ib = i+1,           jb = t == 0 ? j+1 : j;
    const int ic = t == 0 ? i : i+1, jc = j+1;
    const hz_cvert_t a = hz_cvert(hz_transform(&p.u, (float)i,  (float)j,  (float)mosaic[(size_t)j *p.N + i ]), p.halfW, p.halfH);
    const hz_cvert_t b = hz_cvert(hz_transform(&p.u, (float)ib, (float)jb, (float)mosaic[(size_t)jb*p.N + ib]), p.halfW, p.halfH);
    const hz_cvert_t c = hz_cvert(hz_transform(&p.u, (float)ic, (float)jc, (float)mosaic[(size_t)jc*p.N + ic]), p.halfW, p.halfH);

    hz_cvert_t* poly;
    const int n = hz_clip_triangle(bufa, bufb, &poly, &a, &b, &c, p.halfW, p.halfH);
    /* fan that keeps vertex 0 last (GL provoking-vertex convention).  First
     * pass: which pieces draw anything, and how much queue they need - so that
     * the whole triangle reserves its records and work items with two atomics
     * (one round trip each) instead of two per piece: k_clip runs a handful of
     * threads and its time is the length of this dependency chain. */
    uint32_t npieces = 0, nchunks = 0;
    for(int k=2; k<n; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        npieces++;
        nchunks += hz_big_chunks(box.px1 - box.px0 + 1, box.py1 - box.py0 + 1);
    }
    if(npieces == 0) return;
    uint32_t ri = 0, ii = 0;            /* the shard's own numbers of the next record / item of this reservation */
    const int sl = p.qshards_log2, shard = (int)(blockIdx.x & ((1u << sl) - 1u));
    const bool queued = hz_queue_reserve(q, shard, sl, npieces, nchunks, &ri, &ii);
    if(!queued && !inline_ok) return;
    for(int k=2; k<n; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        hz_tri_t tri;
        hz_tri_planes(&tri, &va, &vb, &vc);
        if(!queued)
        {
            for(int py = box.py0; py <= box.py1; py++)
                for(int px = box.px0; px <= box.px1; px++)
                    hz_emit(fb, p, tri, prim, px, py);
            continue;
        }
        hz_bigrec_t br;
        hz_rec_from_tri(br.r, tri);
        br.r.px0 = box.px0; br.r.py0 = box.py0; br.r.bw = box.px1 - box.px0 + 1;
        br.r.inv_bw = 1.0f / (float)br.r.bw;
        br.r.prim = prim;
        br.bh = box.py1 - box.py0 + 1;
        const uint32_t chunks = hz_big_chunks(br.r.bw, br.bh);
        const uint32_t rslot = HZ_QSLOT(ri, shard, sl);
        q.bigrec[rslot] = br;
        if(jobs && chunks > HZ_CLIP_JOB_MIN) { uint32_t* job = jobs[*njobs]; job[0] = ii; job[1] = rslot; job[2] = chunks | ((uint32_t)shard << 24); (*njobs)++; }     /* (chunks: at most 65535, an image's rows) */
        else for(uint32_t c2=0; c2<chunks; c2++) { const uint32_t g = HZ_QSLOT(ii + c2, shard, sl); q.bigitem[g].rec = rslot; q.bigitem[g].chunk = c2; }
        ri++; ii += chunks;
    }
}

/* one thread per queued triangle id.  The clipper's two polygon buffers are
 * indexed dynamically, which would put them into scratch memory: a handful of
 * threads, each a chain of dependent scratch round trips, was 40 us of every
 * draw.  They live in LDS instead (one 64-thread block per CU is plenty here).
 *
 * If the id queue overflowed (never with the default capacity) the ids that
 * did not fit are lost: the kernel then finds every triangle that needs the
 * clipper again, one thread per cell, and clips it on the spot.  Slow, correct. */
/* Resources matter more than speed here: the kernel of the first round runs
 * beside the marching kernel of the panorama before, whose waves fill every
 * SIMD's registers; a workgroup that wants 60 KB of contiguous LDS and half a
 * SIMD's registers (what this kernel took with 64 clipping lanes per block)
 * waited ~0.7 ms to be placed.  HZ_CLIP_LANES lanes of a block clip, the
 * others leave at once. */
#define HZ_CLIP_LANES 4
/* WAVE_ITEMS (round 5; zoomed views: draw_impl): the work items of a clipped triangle's large pieces are written by the whole
 * wave.  A triangle next to the viewer of a zoomed view is clipped into pieces thousands of rows high and as wide as the
 * image - one row to a chunk, so thousands of work items apiece -, and the one thread that clipped it wrote them one after
 * the other: the kernel took 90-150 us for 2000 triangles whose average wave needed 9; with the wave writing them 154 -> 21 us,
 * the 45 degree view towards the east 1.31 -> 1.17 ms (profiles/r4_ab_kclip_items.txt).  The 60 lanes that do not clip cost
 * nothing while they wait - a wave's registers are allocated whatever its lanes do.  A whole panorama's clipped triangles
 * have no such pieces, and its draws keep the instance they were tuned with: the same machine code as before, to the byte
 * (round 4 measured 0.8 % on the headline with the one kernel serving both - its first round's k_big then starts 17 us
 * earlier beside the second round of the panorama before). */
template<bool WAVE_ITEMS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void k_clip(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb, mr_queue_t q, hz_params_t p)
{
    __shared__ hz_cvert_t polygon[HZ_CLIP_LANES][2][HZ_MAX_CLIPPED+1];
    __shared__ uint32_t s_job[WAVE_ITEMS ? HZ_CLIP_LANES : 1][WAVE_ITEMS ? HZ_MAX_CLIPPED : 1][3];
    __shared__ int      s_njobs[HZ_CLIP_LANES];
    const bool clipper = threadIdx.x < HZ_CLIP_LANES;
    if(!WAVE_ITEMS && !clipper) return;
    const int  cl = clipper ? (int)threadIdx.x : 0;
    hz_cvert_t* poly0 = polygon[cl][0];
    hz_cvert_t* poly1 = polygon[cl][1];
    const unsigned int n = q.counters[4];
    const unsigned int me = blockIdx.x*HZ_CLIP_LANES + threadIdx.x, stride = gridDim.x*HZ_CLIP_LANES;
    if(n <= q.clip_capacity)
    {
        if(!WAVE_ITEMS)
        {
            for(unsigned int k = me; k < n; k += stride)
                hz_clip_and_draw(mosaic, fb, q, p, q.clip[k], true, poly0, poly1);
            return;
        }
        for(unsigned int k0 = blockIdx.x*HZ_CLIP_LANES; k0 < n; k0 += stride)        /* (the same trips for every lane) */
        {
            int njobs = 0;
            if(clipper && k0 + cl < n) hz_clip_and_draw(mosaic, fb, q, p, q.clip[k0 + cl], true, poly0, poly1, s_job[cl], &njobs);
            /* (most trips have none - pieces of a few chunks are written by their thread as before: no LDS, no barrier) */
            if(__ballot(njobs > 0) == 0ull) continue;
            if(clipper) s_njobs[cl] = njobs;
            __syncthreads();
            for(int c=0; c<HZ_CLIP_LANES; c++)
                for(int jb=0; jb<s_njobs[c]; jb++)
                {
                    const uint32_t ii = s_job[c][jb][0], rslot = s_job[c][jb][1], chunks = s_job[c][jb][2] & 0xFFFFFFu, shard = s_job[c][jb][2] >> 24;
                    for(uint32_t c2 = threadIdx.x; c2 < chunks; c2 += 64u) { const uint32_t g = HZ_QSLOT(ii + c2, shard, p.qshards_log2); q.bigitem[g].rec = rslot; q.bigitem[g].chunk = c2; }
                }
            __syncthreads();
        }
        return;
    }
    if(!clipper) return;
    const size_t ncells = (size_t)(p.N-1)*(p.N-1);
    for(size_t cell = me; cell < ncells; cell += stride)
    {
        const int j = (int)(cell / (size_t)(p.N-1)), i = (int)(cell - (size_t)j*(p.N-1));
        const hz_wvert_t v00 = hz_vertex_at(p, mosaic, i, j),   v10 = hz_vertex_at(p, mosaic, i+1, j);
        const hz_wvert_t v01 = hz_vertex_at(p, mosaic, i, j+1), v11 = hz_vertex_at(p, mosaic, i+1, j+1);
        hz_box_t box;
        if(hz_tri_cull(&box, &v00, &v11, &v01, p.col0, p.col1-1, 0, p.H-1) == HZ_TRI_CLIP)
            hz_clip_and_draw(mosaic, fb, q, p, (uint32_t)(cell*2),     true, poly0, poly1);
        if(hz_tri_cull(&box, &v00, &v10, &v11, p.col0, p.col1-1, 0, p.H-1) == HZ_TRI_CLIP)
            hz_clip_and_draw(mosaic, fb, q, p, (uint32_t)(cell*2 + 1), true, poly0, poly1);
    }
}
