/* hz_kernels.hip - the DEM -> panorama render path as HIP kernels for gfx950,
 * and the C-ABI (include/hz_hip.h) through which the C host drives them.
 *
 * What runs here is what the reference hands to OpenGL:
 *   vertex.glsl:111-162     per-vertex transform            -> hz_transform()
 *   horizonator-lib.c:487-512 index buffer (2 tris per cell) -> implicit from (i,j,t)
 *   geometry.glsl:21-27     wide/seam triangle discard      -> hz_tri_cull()
 *   fixed function          clip, cull, raster, depth test  -> hz_raster.h
 *   fragment.glsl:15-16     colour = (red,0,0)              -> packed red8
 *   fragment.glsl:17-22     0.7*texture + 0.3*shade         -> k_shade_tex (hz_tex.h)
 *   horizonator-lib.c:936-1048 readback, flip, depth->range -> k_resolve
 *
 * HBM layout
 *   mosaic  int16 [N][N], row j = constant latitude (south first), i fastest
 *   fb      uint64 [H][SW]  GL row order (row 0 = bottom), SW = sector width
 *           word = z24<<40 | primitive<<8 | red8, cleared to all ones
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "hz_hip.h"
#include "hz_raster.h"
#include "hz_fast.h"
#include "hz_tex.h"

/* ------------------------------------------------------------------------ */
/* errors                                                                    */

static thread_local char g_last_error[512];       /* per thread: contexts may live on different threads */
extern "C" const char* hz_hip_last_error(void) { return g_last_error; }

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

/* ------------------------------------------------------------------------ */
/* kernel parameters                                                         */

struct hz_params_t
{
    hz_xform_t u;
    float halfW, halfH;
    int   N;                /* samples per mosaic axis                  */
    int   W, H;             /* full image size                          */
    int   col0, col1;       /* sector [col0,col1)                       */
    int   SW;               /* col1-col0, row stride of fb              */
    unsigned long long* wave_cycles;   /* diagnostics: per-wave duration of k_march, or NULL */
    unsigned int inline_max;           /* k_march: boxes up to this many pixels are rasterised by the marching wave */
    unsigned int big_min;              /* k_march: boxes above this many pixels go to k_big (tiles), between: k_mid  */
    float far_dd;                      /* k_march: squared horizontal distance beyond which a vertex is surely past zfar */
    int   far_strips;                  /* some vertex of the mosaic lies beyond that: whole strips may (k_march asks) */
    /* two-pass draw (see hz_hip_draw): which strips a k_march launch takes, and
     * whether it tests its survivors against the depth already in the framebuffer */
    int   pass;                        /* 0 every strip, 1 only the strips next to the viewer, 2 all the others */
    int   near_x0, near_x1;            /* strip columns [x0,x1] and                                             */
    int   near_j0, near_j1;            /* cell rows [j0,j1) that make up "next to the viewer"                   */
    int   early_z;                     /* mr_flush: skip triangles whose box is already covered by nearer depth */
    float z_guard;                     /* hz_tri_depth_floor(): 1/500 + max(W,H)*2^-22                          */
    float z_hide_k;                    /* hz_tri_hidden(): 1.03 * z_guard * (2^24-1)                            */
    int   fast_ok;                     /* hzf_draw_ok(): the uniforms allow the abridged division/sqrt sequences */
    int   quad_max_dx;                 /* k_march: 256*(W/16 - 1): see the cull of whole cells                  */
    int   debug;                       /* HZ_MARCH_DEBUG (timing splits, wrong pictures): 1 survivors are dropped,
                                        * 2 survivors are dropped after the early depth test */
    /* one byte per HZ_SEG consecutive pixels of a framebuffer row (row stride
     * seg_stride): nonzero once anything was drawn there.  Every write to the
     * framebuffer sets it (hz_fb_min); the conversion skips reading - and
     * clearing - segments nothing touched: the sky, 62 % of the benchmark's
     * pixels.  A stale nonzero byte only costs the read. */
    unsigned char* touched;
    int   seg_stride;
};
#define HZ_SEG_LOG2 8
#define HZ_SEG      (1 << HZ_SEG_LOG2)

/* the one place fragments enter the framebuffer */
__device__ static inline void hz_fb_min(unsigned long long* fb, const hz_params_t& p, int px, int py, unsigned long long key)
{
    const int x = px - p.col0;
    p.touched[(size_t)py*p.seg_stride + (x >> HZ_SEG_LOG2)] = 1;
    atomicMin(&fb[(size_t)py*p.SW + x], key);
}

/* a set-up triangle as it travels between phases: through LDS inside
 * k_scatter (stride 23 dwords = odd, conflict-free), through HBM to k_mid and
 * k_big.  Coverage as hz_edges_t: what the pixel loops need, ready made. */
struct hz_rec_t
{
    hz_edges_t e;
    float    z_org, dzdx, dzdy, r_org, drdx, drdy;
    int32_t  px0, py0, bw;
    float    inv_bw;
    uint32_t prim;
};
struct hz_bigrec_t { hz_rec_t r; int32_t bh; };

/* work item of the large-triangle pass: 64 tiles of one triangle */
struct hz_bigitem_t { uint32_t rec; uint32_t chunk; };

/* the HBM queues between the kernels of one draw */
struct mr_queue_t
{
    hz_bigrec_t*  bigrec;           /* set-up triangles for k_big                                */
    hz_bigitem_t* bigitem;          /* ... and their work items                                  */
    hz_rec_t*     midrec;           /* set-up triangles for k_mid                                */
    uint32_t*     clip;             /* ids of triangles that have to go through the clipper      */
    unsigned int* counters;         /* [0] big records [1] big items [2] first invalid big item
                                     * [3] mid records [4] clip ids [5] first invalid mid record */
    unsigned int  bigrec_capacity, bigitem_capacity, midrec_capacity, clip_capacity;
};

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

#define HZ_NCOUNTERS 6
#ifndef HZ_NFB
#define HZ_NFB 3                        /* framebuffers (and queue sets per round) a context cycles through */
#endif
#define HZ_STAGE_SLOTS 4                /* pinned staging chunks in flight between device and caller memory */
#define HZ_STAGE_BYTES ((size_t)32 << 20)
#define HZ_INLINE_MAX_PIX  64       /* k_scatter: boxes up to this many pixel centres are rasterised in the block */
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
    if(PRETEST)
    {
        if(key < __hip_atomic_load(&fb[(size_t)py*p.SW + (px - p.col0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            hz_fb_min(fb, p, px, py, key);
    }
    else
        hz_fb_min(fb, p, px, py, key);
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
        if(key < __hip_atomic_load(&fb[(size_t)py*p.SW + (px - p.col0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
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
__device__ static void hz_clip_and_draw(const int16_t* mosaic, unsigned long long* fb, const mr_queue_t& q,
                                        const hz_params_t& p, uint32_t prim, bool inline_ok,
                                        hz_cvert_t* bufa, hz_cvert_t* bufb)
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
    bool queued = false;
    uint32_t ri = atomicAdd(&q.counters[0], npieces), ii = 0;
    if(ri + npieces <= q.bigrec_capacity)
    {
        ii = atomicAdd(&q.counters[1], nchunks);
        if(ii + nchunks <= q.bigitem_capacity) queued = true;
        else atomicMin(&q.counters[2], ii);
    }
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
        q.bigrec[ri] = br;
        for(uint32_t c2=0; c2<chunks; c2++) { q.bigitem[ii+c2].rec = ri; q.bigitem[ii+c2].chunk = c2; }
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void k_clip(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb, mr_queue_t q, hz_params_t p)
{
    __shared__ hz_cvert_t polygon[HZ_CLIP_LANES][2][HZ_MAX_CLIPPED+1];
    if(threadIdx.x >= HZ_CLIP_LANES) return;
    hz_cvert_t* poly0 = polygon[threadIdx.x][0];
    hz_cvert_t* poly1 = polygon[threadIdx.x][1];
    const unsigned int n = q.counters[4];
    const unsigned int me = blockIdx.x*HZ_CLIP_LANES + threadIdx.x, stride = gridDim.x*HZ_CLIP_LANES;
    if(n <= q.clip_capacity)
    {
        for(unsigned int k = me; k < n; k += stride)
            hz_clip_and_draw(mosaic, fb, q, p, q.clip[k], true, poly0, poly1);
        return;
    }
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

#define SC_CX 64
#define SC_CY 4
#define SC_VX (SC_CX+1)
#define SC_VY (SC_CY+1)
#define SC_THREADS (SC_CX*SC_CY)
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
                unsigned int ri = atomicAdd(&big_counters[0], 1u), ii = 0;
                bool queued = false;
                if(ri < bigrec_capacity)
                {
                    ii = atomicAdd(&big_counters[1], chunks);
                    if(ii + chunks <= bigitem_capacity) queued = true;
                    else atomicMin(&big_counters[2], ii);      /* items from here on are not valid */
                }
                if(queued)
                {
                    bigrec[ri].r = r; bigrec[ri].bh = bh;
                    for(unsigned int c2=0; c2<chunks; c2++) { bigitem[ii+c2].rec = ri; bigitem[ii+c2].chunk = c2; }
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

/* inclusive prefix sum over the 64 lanes */
__device__ static inline uint32_t mr_scan(uint32_t v, int lane)
{
    #pragma unroll
    for(int d=1; d<64; d<<=1)
    {
        const uint32_t up = __shfl_up(v, d);
        if(lane >= d) v += up;
    }
    return v;
}

/* floor(n / d) for d > 0: a double-precision estimate (r = 1/d to full double
 * accuracy, computed by the caller once per edge), then the remainder decides -
 * exactly.  |n| < 2^55, d < 2^31; results beyond +-2^30 come back clamped (the
 * caller only compares them with pixel columns). */
__device__ static inline int32_t hz_floor_div(int64_t n, int32_t d, double r)
{
    double qd = __builtin_floor((double)n * r);
    qd = qd < -1073741824.0 ? -1073741824.0 : (qd > 1073741824.0 ? 1073741824.0 : qd);
    int32_t q = (int32_t)qd;
    int64_t rem = n - (int64_t)q*(int64_t)d;
    /* the estimate is off by one at most (two steps each way for good measure) */
    if(rem < 0)  { q--; rem += d; }
    if(rem < 0)  { q--; rem += d; }
    if(rem >= d) { q++; rem -= d; }
    if(rem >= d) { q++; rem -= d; }
    return q;
}

/* large triangles: one wave per work item = 64 pixel rows of a queued triangle.
 * Lane = row: the covered pixel centres of a row are a span [x0, x1] - each
 * edge function is linear in px, so each edge bounds the span from one side, at
 * a column that an integer division gives exactly (the ownership of zeros
 * included).  Then lane = pixel: the spans of the 64 rows are laid end to end
 * (wave prefix sum) and every lane takes one covered pixel per pass, whatever
 * the shape of the triangle - the long thin slivers next to the viewer cover a
 * quarter of their boxes. */
__global__ __launch_bounds__(256)
void k_big(unsigned long long* __restrict__ fb,
           const hz_bigrec_t* __restrict__ bigrec, const hz_bigitem_t* __restrict__ bigitem,
           const unsigned int* __restrict__ big_counters,
           unsigned int bigrec_capacity, unsigned int bigitem_capacity, hz_params_t p)
{
    /* items at and beyond the first overflow were rasterised inline by their producer */
    const unsigned int nitems = min(big_counters[1], big_counters[2]);
    (void)bigrec_capacity; (void)bigitem_capacity;
    const int lane = threadIdx.x & 63;
    const unsigned int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x*(blockDim.x/64) + (threadIdx.x >> 6));
    const unsigned int nwaves = gridDim.x*(blockDim.x/64);
    /* item and record come through the scalar cache (wave-uniform addresses);
     * the next item's are requested before the current one is rasterised, so
     * their latency hides behind the pixel work */
    hz_bigitem_t item_next = {};
    hz_bigrec_t  rec_next  = {};
    if(wave_global < nitems) { item_next = bigitem[wave_global]; rec_next = bigrec[item_next.rec]; }
    for(unsigned int it = wave_global; it < nitems; it += nwaves)
    {
        const hz_bigitem_t item = item_next;
        const hz_bigrec_t  br   = rec_next;
        if(it + nwaves < nitems) { item_next = bigitem[it + nwaves]; rec_next = bigrec[item_next.rec]; }
        hz_tri_t tri;
        hz_planes_from_rec(tri, br.r);
        const int px0 = br.r.px0, py0 = br.r.py0, bw = br.r.bw, bh = br.bh;
        const uint32_t prim = br.r.prim;

        /* lane = row */
        const int rows_log2 = hz_big_rows_log2(bw);
        const int row_first = py0 + ((int)item.chunk << rows_log2);
        const int row = row_first + lane;
        int32_t x0 = px0, x1 = px0 + bw - 1;
        bool any = lane < (1 << rows_log2) && row < py0 + bh;
        #pragma unroll
        for(int m=0; m<3; m++)
        {
            /* edge m covers px in this row iff g + dx*row - dy*px >= 0 (hz_edges_t), g and
             * the deltas wave-uniform: a bound on px from one side, by an exact division */
            const int32_t dx = br.r.e.dx[m], dy = -br.r.e.ndy[m];
            const int64_t n8 = hz_edges_g(&br.r.e, m) + (int64_t)dx*(int64_t)row;
            if(dy > 0)
            {
                /* dy*px <= n8  <=>  px <= floor(n8 / dy) */
                const int32_t q = hz_floor_div(n8, dy, 1.0/(double)dy);
                x1 = x1 < q ? x1 : q;
            }
            else if(dy < 0)
            {
                /* |dy|*px >= -n8  <=>  px >= ceil(-n8 / |dy|) = -floor(n8 / |dy|) */
                const int32_t q = hz_floor_div(n8, -dy, 1.0/(double)(-dy));
                x0 = x0 > -q ? x0 : -q;
            }
            else if(n8 < 0) any = false;                /* a horizontal edge: the whole row is on one side */
        }
        const uint32_t count = (any && x1 >= x0) ? (uint32_t)(x1 - x0 + 1) : 0u;

        /* lane = pixel */
        const uint32_t incl  = mr_scan(count, lane);
        const uint32_t excl  = incl - count;
        const uint32_t total = __shfl(incl, 63);
        for(uint32_t base = 0; base < total; base += 64)
        {
            const uint32_t k = base + lane;
            /* the row that holds pixel k: last lane whose exclusive prefix is <= k */
            int own = 0;
            #pragma unroll
            for(int step=32; step>=1; step>>=1)
            {
                const uint32_t v = __shfl(excl, own + step);
                if(v <= k) own += step;
            }
            const int px = __shfl(x0, own) + (int)(k - __shfl(excl, own));
            const int py = row_first + own;
            if(k < total)
            {
                uint32_t zi, r8;
                if(hz_tri_fragment(&tri, px, py, &zi, &r8))
                    hz_fb_min(fb, p, px, py, hz_pack(zi, prim, r8));
            }
        }
    }
}

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

#define MR_COLS   63
#define MR_CAP    128               /* pending-triangle ids, ring (power of two)    */
#define MR_RSLOTS 4                 /* vertex rows kept in LDS (power of two)       */
#define MR_FIELDS 6                 /* wx wy zw red xs ys                            */
#define MR_NEAR_CELLS 64             /* round 1 of a draw: strips within this many cells of the viewer */

/* LDS of one wave: the last MR_RSLOTS vertex rows (structure of arrays: one
 * conflict-free 256-byte store per field and row) and a ring of ids of the
 * triangles waiting for set-up.  id = (cell row - first row of the segment)<<7
 * | lane<<1 | t.  Only 6 + 2 LDS stores per row of 126 triangles. */
struct mr_lds_t
{
    uint32_t rows[MR_RSLOTS][MR_FIELDS][64];
    uint32_t ids[MR_CAP];
};

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

/* The strips are cut into segments of rows; the segment length depends on the
 * distance (in rows) from the viewer's row so that every wave gets a comparable
 * amount of pixel work: zones south->north with 64, 16, 4, 2, 4, 16, 64 rows
 * per segment.  Built on the host per draw (mr_make_zones).  Measured: with
 * uniform 64-row segments the waves next to the viewer run 10-50x longer than
 * the median and set the kernel time. */
#define MR_NZONES 7
struct mr_zones_t
{
    int row0[MR_NZONES+1];          /* first cell row of each zone; row0[MR_NZONES] = N-1 */
    int rows[MR_NZONES];            /* cell rows per segment                              */
    int seg0[MR_NZONES];            /* number of the zone's first segment                 */
    int nseg[MR_NZONES];            /* segments in the zone                               */
    int total;                      /* all segments = gridDim.y                           */
    int near_first;                 /* dispatch order: segments nearest to the viewer's row first */
};

/* value held by the lane one to the east (lane+1): DPP wave shift, one VALU
 * move instead of an LDS-crossbar permute (gfx9 family: wave_shl:1).  Lane 63
 * gets an unspecified value; it has no cell. */
__device__ static inline int32_t mr_from_east(int32_t v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
__device__ static inline float mr_from_east(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));
}



__device__ static inline int32_t hz_imin(int32_t a, int32_t b) { return a < b ? a : b; }
__device__ static inline int32_t hz_imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* what k_march keeps of a vertex row for the cells between it and the next */
struct mr_rowstate_t
{
    float    xn;                        /* NDC x (discard rule)                              */
    int32_t  xs, ys;                    /* snapped position                                  */
    uint32_t cmask;                     /* clip mask (0 in rows that are wholly inside)      */
    int32_t  c_x, f_x, c_y, f_y;        /* first / last pixel column and row at or beyond / up to the vertex, clipped to the scissor */
    int32_t  h_c_x, h_f_x, h_c_y, h_f_y;        /* the same, over the vertex and its eastern neighbour */
    int32_t  h_dx, h_dy;                /* eastern neighbour's snapped position minus this vertex's */
};


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
    const uint32_t total = __shfl(incl, 63);
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
    /* records from the first reservation that did not fit were not written
     * (their triangles were rasterised by the marching wave instead) */
    const unsigned int n = min(min(counters[3], counters[5]), midrec_capacity);
    for(unsigned int base = blockIdx.x*64u; base < n; base += gridDim.x*64u)
    {
        hz_rec_t r = {};
        uint32_t npix = 0;
        if(base + lane < n)
        {
            r = midrec[base + lane];
            /* queued records carry the pixel count of the box in the inv_bw
             * slot (the reciprocal is cheaper to redo than to store) */
            npix = __float_as_uint(r.inv_bw);
            r.inv_bw = 1.0f / (float)r.bw;
        }
        mr_distribute(r, npix, lane, fb, p);
    }
}

/* set up and rasterise the `n` (<= 64) oldest pending triangles */
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
    hz_wvert_t a = {}, b = {}, c = {};
    hz_box_t box = {};
    int t = 0, l = 0, rowoff = 0;
    int sa = 0, sb = 0, sc = 0, la = 0, lb = 0, lc = 0;      /* LDS row slot and lane of the three vertices */
    if(valid)
    {
        const uint32_t id = L.ids[(head + lane) & (MR_CAP-1)];
        t = id & 1; l = (id >> 1) & 63; rowoff = id >> 7;
        const int s0 = rowoff & (MR_RSLOTS-1), s1 = (rowoff+1) & (MR_RSLOTS-1);
        /* reference horizonator-lib.c:500-506 */
        sa = s0;               la = l;
        sb = t == 0 ? s1 : s0; lb = l+1;
        sc = s1;               lc = t == 0 ? l : l+1;
        /* position and depth now; the colour only for triangles that get drawn */
        a = mr_load_vert_pos(L, sa, la);
        b = mr_load_vert_pos(L, sb, lb);
        c = mr_load_vert_pos(L, sc, lc);
        hz_tri_box(&box, &a, &b, &c, p.col0, p.col1-1, 0, p.H-1);
    }
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
            const uint32_t* fbw = (const uint32_t*)fb;              /* depth = the upper 24 bits of the upper word */
            const uint32_t* row0 = fbw + 2*((size_t)box.py0*p.SW) + 1;
            const uint32_t* row1 = fbw + 2*((size_t)box.py1*p.SW) + 1;
            const int c0 = box.px0 - p.col0, cl = box.px1 - p.col0;
            const int c1 = min(c0+1, cl), c2 = min(c0+2, cl);
            uint32_t z[8];
            z[0] = row0[2*c0]; z[1] = row0[2*c1]; z[2] = row0[2*c2]; z[3] = row0[2*cl];
            z[4] = row1[2*c0]; z[5] = row1[2*c1]; z[6] = row1[2*c2]; z[7] = row1[2*cl];
            const uint32_t zs = max(max(max(z[0], z[1]), max(z[2], z[3])), max(max(z[4], z[5]), max(z[6], z[7]))) >> 8;
            if(hz_tri_hidden(&a, &b, &c, p.z_hide_k, zs)) live = false;
        }
        if(dbg) { dbg[5] += (unsigned int)__popcll(__ballot(valid && !live)); }
        if(p.debug == 2) return;
        if(!__any(live))
        {
            if(dbg) { dbg[0] += 1; dbg[1] += n; }
            return;
        }
    }
    if(live)
    {
        a.red = __uint_as_float(L.rows[sa][3][la]);
        b.red = __uint_as_float(L.rows[sb][3][lb]);
        c.red = __uint_as_float(L.rows[sc][3][lc]);
    }
    if(live)
    {
        hz_tri_t tri;
        hz_tri_planes(&tri, &a, &b, &c);
        hz_rec_from_tri(r, tri);
        r.px0 = box.px0; r.bw = box.px1 - box.px0 + 1;
        r.py0 = box.py0; bh   = box.py1 - box.py0 + 1;
        r.inv_bw = 1.0f / (float)r.bw;
        r.prim = (uint32_t)(((size_t)(jbeg + rowoff)*(p.N-1) + (i0 + l))*2 + t);
        npix = (uint32_t)r.bw*(uint32_t)bh;
    }
    else
    {
        #pragma unroll
        for(int m=0; m<3; m++) { r.e.dx[m] = 0; r.e.ndy[m] = 0; r.e.glo[m] = 0; r.e.ghi[m] = 0; }
        r.z_org = r.dzdx = r.dzdy = r.r_org = r.drdx = r.drdy = 0.f;
        r.px0 = r.py0 = 0; r.bw = 1; r.inv_bw = 1.f; r.prim = 0;
    }

    /* large boxes go to k_big: one record, ceil(tiles/64) work items */
    const bool is_big = live && npix > p.big_min;
    const unsigned long long bigmask = __ballot(is_big);
    if(dbg) { dbg[0] += 1; dbg[1] += n; dbg[2] += (unsigned int)__popcll(bigmask); }
    if(bigmask)
    {
        uint32_t chunks = 0;
        if(is_big)
        {
            chunks = hz_big_chunks(r.bw, bh);
        }
        const uint32_t incl  = mr_scan(chunks, lane);
        const uint32_t total = __shfl(incl, 63);
        const uint32_t nb    = (uint32_t)__popcll(bigmask);
        uint32_t rbase = 0, ibase = 0, ok = 0;
        if(lane == 0)
        {
            rbase = atomicAdd(&q.counters[0], nb);
            if(rbase + nb <= q.bigrec_capacity)
            {
                ibase = atomicAdd(&q.counters[1], total);
                if(ibase + total <= q.bigitem_capacity) ok = 1;
                else atomicMin(&q.counters[2], ibase);          /* items from here on are not valid */
            }
        }
        rbase = __shfl(rbase, 0); ibase = __shfl(ibase, 0); ok = __shfl(ok, 0);
        if(is_big)
        {
            if(ok)
            {
                const uint32_t ri = rbase + (uint32_t)__popcll(bigmask & ((1ull << lane) - 1ull));
                const uint32_t ii = ibase + incl - chunks;
                q.bigrec[ri].r = r; q.bigrec[ri].bh = bh;
                for(uint32_t c2=0; c2<chunks; c2++) { q.bigitem[ii+c2].rec = ri; q.bigitem[ii+c2].chunk = c2; }
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
        uint32_t mbase = 0;
        if(lane == 0) mbase = atomicAdd(&q.counters[3], (uint32_t)__popcll(midmask));
        mbase = __shfl(mbase, 0);
        if(mbase + (uint32_t)__popcll(midmask) <= q.midrec_capacity)
        {
            if(is_mid)
            {
                hz_rec_t m = r;
                m.inv_bw = __uint_as_float(npix);       /* see k_mid */
                q.midrec[mbase + (uint32_t)__popcll(midmask & ((1ull << lane) - 1ull))] = m;
                npix = 0;
            }
        }
        /* else: queue full, they stay here; the slots from mbase on hold nothing
         * of this draw and k_mid must not read them */
        else if(lane == 0) atomicMin(&q.counters[5], mbase);
    }

    if(dbg) { const uint32_t tot = __shfl(mr_scan(npix, lane), 63); dbg[4] += tot; }
    mr_distribute(r, npix, lane, fb, p);
}

__global__ __launch_bounds__(64)
void k_march(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb,
             mr_queue_t q, mr_zones_t zn, hz_params_t p)
{
    __shared__ mr_lds_t L;
    const unsigned long long t_start = p.wave_cycles ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned int dbgv[6] = {0,0,0,0,0,0};
    unsigned int* dbg = p.wave_cycles ? dbgv : nullptr;

    const int lane = threadIdx.x;
    const int sx   = (int)blockIdx.x + (p.pass == 1 ? p.near_x0 : 0);     /* strip column */
    const int i0   = sx*MR_COLS;
    const int i    = i0 + lane;
    int zone = 0;
    #pragma unroll
    for(int z=1; z<MR_NZONES; z++)
        if((int)blockIdx.y >= zn.seg0[z] && (int)blockIdx.y < zn.seg0[z] + zn.nseg[z]) zone = z;
    int sseg = (int)blockIdx.y - zn.seg0[zone];
    if(zn.near_first && zone < MR_NZONES/2) sseg = zn.nseg[zone]-1 - sseg;      /* south of the viewer: northernmost first */
    const int jbeg = zn.row0[zone] + sseg*zn.rows[zone];
    const int jend = min(jbeg + zn.rows[zone], zn.row0[zone+1]);   /* vertex rows jbeg..jend, cell rows jbeg..jend-1 */
    if(p.pass)
    {
        const bool near = sx >= p.near_x0 && sx <= p.near_x1 && jbeg < p.near_j1 && jend > p.near_j0;
        if(near != (p.pass == 1)) return;
    }
    const bool has_vertex = i < p.N;
    const bool has_cell   = lane < MR_COLS && i < p.N-1;
    const int  ic = has_vertex ? i : p.N-1;             /* clamped: idle lanes redo the last column */

    /* azimuth-sector shard (multi-GPU): a segment that does not contain the
     * viewer is a convex patch seen from outside, so its azimuth extent is that
     * of its four corner vertices; if that lies outside this GPU's columns the
     * whole wave has nothing to draw.  (The corners are real vertices: their x
     * is computed exactly as the rasteriser computes it.) */
    if(p.col0 > 0 || p.col1 < p.W)
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

    /* pending-triangle ring, wave-uniform state */
    unsigned int head = 0, count = 0;
    int first_row = 0;                                  /* cell row (relative) of the oldest pending triangle */
    /* what a row keeps of itself for the cells above it (the attributes of its
     * vertices live in LDS, where mr_flush takes them from): per lane the
     * vertex's NDC x, snapped position and clip mask, its pixel columns/rows
     * (mr_vcull_t) and the same combined with the vertex one lane to the east */
    mr_rowstate_t prev = {};
    bool prev_simple = false;
    int16_t z_next = mosaic[(size_t)jbeg*p.N + ic];
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
        if(j < jend) z_next = mosaic[(size_t)(j+1)*p.N + ic];
        const float n_next   = (j == jend) ? 0.f : north_of(rel+1);
        const bool  far_next = (j == jend) || __all(n_next*n_next + e*e > p.far_dd);
        const bool  skip_row   = far_prev && far_cur && far_next;   /* vertex row j not needed       */
        const bool  skip_cells = far_prev && far_cur;               /* cell row j-1 entirely clipped */
        const float n = n_cur;
        n_cur = n_next; far_prev = far_cur; far_cur = far_next;
        if(skip_row) continue;

        const bool fast = fast_strip && (rel >= 64 ? hzf_in_range(n) : (int)((fast_rows >> rel) & 1ull));
        const hz_vertex_t vtx = fast ? hzf_transform_en(&p.u, &fc, e, n, z) : hz_transform_en(&p.u, e, n, z);

        /* window position as hz_to_window() computes it.  Two facts about the
         * whole row are established on the way, which decide how its cells are
         * culled: every vertex inside the view volume (clip mask 0: with
         * xn + 1 < 0 <=> xn < -1 for every float, "inside" is |x|,|y|,|z| <= 1)
         * and every vertex inside the guard band. */
        hz_wvert_t cur;
        cur.xn  = vtx.x;
        cur.wx  = vtx.x*p.halfW + p.halfW;
        cur.wy  = vtx.y*p.halfH + p.halfH;
        cur.zw  = vtx.z*0.5f + 0.5f;
        cur.red = vtx.red;
        const float fxw = cur.wx - 0.5f, fyw = cur.wy - 0.5f;
        const bool  in_guard  = hz_abs(fxw) <= HZ_GUARD_PX && hz_abs(fyw) <= HZ_GUARD_PX;     /* a NaN is outside */
        /* fmaxf skips a NaN, as the six comparisons of hz_clip_mask() do (all false) */
        const bool  in_volume = __builtin_fmaxf(__builtin_fmaxf(hz_abs(vtx.x), hz_abs(vtx.y)), hz_abs(vtx.z)) <= 1.0f;
        const bool  cur_simple = __all(in_guard && in_volume);
        cur.xs = (int32_t)hz_roundeven(fxw*256.f);
        cur.ys = (int32_t)hz_roundeven(fyw*256.f);
        cur.cmask = 0;
        if(!cur_simple)
        {
            cur.cmask = hz_clip_mask(vtx.x, vtx.y, vtx.z);
            if(!in_guard) { cur.xs = HZ_OUTSIDE_GUARD; cur.ys = 0; }
        }

        /* this row replaces vertex row rel-MR_RSLOTS in LDS: triangles that
         * still need it are set up now (happens where survivors are sparse) */
        if(count && first_row <= rel - MR_RSLOTS)
        {
            __syncthreads();
            mr_flush(L, head, count, lane, jbeg, i0, fb, q, p, dbg);
            __syncthreads();
            head = (head + count) & (MR_CAP-1);
            count = 0;
        }
        mr_store_row(L, rel & (MR_RSLOTS-1), lane, cur);

        mr_rowstate_t now;
        now.xn = cur.xn; now.xs = cur.xs; now.ys = cur.ys; now.cmask = cur.cmask;
        /* pixel columns/rows of the vertex: a triangle's pixel box is the min of
         * its vertices' first and the max of their last (hz_tri_box: the shifts
         * are monotone), clipped to the scissor here already (max and min
         * distribute over it) */
        now.c_x = hz_imax((cur.xs + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS, p.col0);
        now.f_x = hz_imin(cur.xs >> HZ_SUBPIXEL_BITS, p.col1-1);
        now.c_y = hz_imax((cur.ys + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS, 0);
        now.f_y = hz_imin(cur.ys >> HZ_SUBPIXEL_BITS, p.H-1);
        /* the same over this vertex and its eastern neighbour (one DPP-fused
         * instruction each: the neighbour's value never lands in a register of
         * its own), and the step to that neighbour */
        now.h_c_x  = hz_imin(mr_from_east(now.c_x), now.c_x);
        now.h_f_x  = hz_imax(mr_from_east(now.f_x), now.f_x);
        now.h_c_y  = hz_imin(mr_from_east(now.c_y), now.c_y);
        now.h_f_y  = hz_imax(mr_from_east(now.f_y), now.f_y);
        now.h_dx   = (int32_t)((uint32_t)mr_from_east(cur.xs) - (uint32_t)cur.xs);     /* (wraps for guard-band markers; unused then) */
        now.h_dy   = (int32_t)((uint32_t)mr_from_east(cur.ys) - (uint32_t)cur.ys);

        if(j > jbeg && !skip_cells)
        {
            /* cell (i, j-1): v00 = prev, v01 = cur, v10 / v11 = those of the lane to
             * the east; triangles t0 = (v00,v11,v01), t1 = (v00,v10,v11), reference
             * horizonator-lib.c:500-506 */
            bool keep0 = false, keep1 = false;
            bool simple = cur_simple && prev_simple;
            /* steps from v00 to the cell's other vertices, in 1/256 pixel */
            const int32_t d01x = (int32_t)((uint32_t)cur.xs - (uint32_t)prev.xs);               /* v01 - v00 */
            const int32_t d01y = (int32_t)((uint32_t)cur.ys - (uint32_t)prev.ys);
            const int32_t d11x = (int32_t)((uint32_t)d01x + (uint32_t)now.h_dx);                /* v11 - v00 = (v01 - v00) + (v11 - v01) */
            const int32_t d11y = (int32_t)((uint32_t)d01y + (uint32_t)now.h_dy);
            const int32_t d10x = prev.h_dx, d10y = prev.h_dy;                                   /* v10 - v00 */
            if(simple)
            {
                /* reference geometry.glsl:21-27 (a triangle spanning more than 0.5 in
                 * NDC x = a quarter of the image is dropped) cannot apply to a cell
                 * whose vertices are all within quad_max_dx of v00 in snapped x: any
                 * two of them are then less than a quarter of the image minus two
                 * pixels apart, and window x follows NDC x to within a hundredth
                 * of a pixel.  A wider cell is rare (the +-180 degree seam, cells
                 * next to the viewer) and sends the row the long way. */
                const int32_t lo = hz_imin(hz_imin(d01x, d11x), d10x), hi = hz_imax(hz_imax(d01x, d11x), d10x);
                if(__any(has_cell && !(lo > -p.quad_max_dx && hi < p.quad_max_dx))) simple = false;
            }
            if(simple)
            {
                /* all four vertices inside the view volume and the guard band, no
                 * discard: what is left of hz_tri_cull() is the back-face test on
                 * the snapped area and the pixel box */
                const int64_t area0 = (int64_t)d11x*(int64_t)d01y - (int64_t)d01x*(int64_t)d11y;
                const int64_t area1 = (int64_t)d10x*(int64_t)d11y - (int64_t)d11x*(int64_t)d10y;
                /* t0 = the row's own edge v01-v11 plus v00; t1 = the lower edge v00-v10 plus v11 */
                const int32_t px0_0 = hz_imin(now.h_c_x, prev.c_x), px1_0 = hz_imax(now.h_f_x, prev.f_x);
                const int32_t py0_0 = hz_imin(now.h_c_y, prev.c_y), py1_0 = hz_imax(now.h_f_y, prev.f_y);
                const int32_t px0_1 = hz_imin(mr_from_east(now.c_x), prev.h_c_x), px1_1 = hz_imax(mr_from_east(now.f_x), prev.h_f_x);
                const int32_t py0_1 = hz_imin(mr_from_east(now.c_y), prev.h_c_y), py1_1 = hz_imax(mr_from_east(now.f_y), prev.h_f_y);
                keep0 = has_cell && area0 > 0 && px0_0 <= px1_0 && py0_0 <= py1_0;
                keep1 = has_cell && area1 > 0 && px0_1 <= px1_1 && py0_1 <= py1_1;
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
                hz_queue_clip(q, verdict0 == HZ_TRI_CLIP, prim0,   lane);
                hz_queue_clip(q, verdict1 == HZ_TRI_CLIP, prim0+1, lane);
                keep0 = verdict0 == HZ_TRI_DRAW; keep1 = verdict1 == HZ_TRI_DRAW;
            }
            #pragma unroll
            for(int t=0; t<2; t++)
            {
                const bool keep = (t == 0 ? keep0 : keep1) && p.debug != 1;
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
                        mr_flush(L, head, 64, lane, jbeg, i0, fb, q, p, dbg);
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
        mr_flush(L, head, count, lane, jbeg, i0, fb, q, p, dbg);
    }
    if(p.wave_cycles && lane == 0)
    {
        unsigned long long* o = &p.wave_cycles[((size_t)blockIdx.y*gridDim.x + blockIdx.x)*4];
        o[0] = __builtin_amdgcn_s_memtime() - t_start;
        o[1] = ((unsigned long long)dbgv[0] << 32) | dbgv[1];     /* flushes, triangles set up */
        o[2] = ((unsigned long long)dbgv[2] << 32) | dbgv[3];     /* to k_big, to k_mid        */
        o[3] = ((unsigned long long)dbgv[5] << 32) | dbgv[4];     /* hidden by the early depth test, pixel centres tested here */
    }
}

/* ------------------------------------------------------------------------ */
/* resolve: framebuffer words -> BGR8, range, primitive id, z24; flips rows  */

/* CLEAR: the kernel is the last reader of this draw: it leaves the framebuffer
 * as glClear would (reference horizonator-lib.c:896), storing all ones behind
 * itself where a triangle had written - the words of the sky (62 % of the
 * benchmark image) are all ones already and are not written again */
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_resolve(unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
               unsigned char* __restrict__ bgr, float* __restrict__ ranges,
               int32_t* __restrict__ index, uint32_t* __restrict__ z24,
               int SW, int H, float znear, float zfar)
{
    const size_t npix = (size_t)SW*H;
    for(size_t o = (size_t)blockIdx.x*blockDim.x + threadIdx.x; o < npix; o += (size_t)gridDim.x*blockDim.x)
    {
        const int yo  = (int)(o / SW);          /* output row, 0 = top            */
        const int x   = (int)(o - (size_t)yo*SW);
        const int row = H-1 - yo;               /* GL row, reference horizonator-lib.c:949-958 */
        const unsigned long long key = fb[(size_t)row*SW + x];
        if(CLEAR && key != HZ_FB_CLEAR) fb[(size_t)row*SW + x] = HZ_FB_CLEAR;
        const uint32_t zi = (uint32_t)(key >> 40);
        const bool sky = (zi == HZ_Z24_MAX);
        if(bgr)
        {
            /* reference horizonator-lib.c:185 clear colour (0,0,1) -> B=255;
             * reference fragment.glsl:15-16 terrain = (red,0,0) -> R */
            bgr[o*3+0] = sky ? 255 : 0;
            bgr[o*3+1] = 0;
            bgr[o*3+2] = sky ? 0 : (unsigned char)(key & 0xFF);
        }
        if(index) index[o] = sky ? -1 : (int32_t)(uint32_t)((key >> 8) & 0xFFFFFFFFull);
        if(z24)   z24[o]   = zi;
        if(ranges)
        {
            /* reference horizonator-lib.c:1013-1025 */
            float r = -1.0f;
            if(!sky)
            {
                const float depth = (float)((double)zi * (1.0/16777215.0));
                const float len   = depth * (zfar-znear) + znear;
                const float zt    = tanel[row] * len;
                r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);  /* = hypotf */
            }
            ranges[o] = r;
        }
    }
}

/* the same for sector widths that are a multiple of 4 and 16-byte aligned
 * buffers (the normal case): a thread takes four neighbouring pixels of one
 * row - two 16-byte loads, one store per output - and the row/column come from
 * the launch grid instead of a 64-bit division per pixel */
__device__ static inline float hz_range_from_z24(uint32_t zi, float tan_row, float znear, float zfar)
{
    /* reference horizonator-lib.c:1013-1025 */
    const float depth = (float)((double)zi * (1.0/16777215.0));
    const float len   = depth * (zfar-znear) + znear;
    const float zt    = tan_row * len;
    return (float)sqrt((double)len*(double)len + (double)zt*(double)zt);  /* = hypotf */
}

template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_resolve4(unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
                unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                int32_t* __restrict__ index, uint32_t* __restrict__ z24,
                int SW, int H, float znear, float zfar,
                unsigned char* __restrict__ touched, int seg_stride)
{
    /* a wave = 64 lanes x 4 pixels = one HZ_SEG-pixel segment of a row */
    static_assert(HZ_SEG == 256, "k_resolve4: one wave converts one segment");
    const int x = (int)(blockIdx.x*blockDim.x + threadIdx.x)*4;
    if(x >= SW) return;
    for(int yo = blockIdx.y; yo < H; yo += gridDim.y)
    {
        const int row = H-1 - yo;               /* GL row, reference horizonator-lib.c:949-958 */
        ulonglong2* src = (ulonglong2*)(fb + (size_t)row*SW + x);
        unsigned char* flag = touched + (size_t)row*seg_stride + (x >> HZ_SEG_LOG2);
        ulonglong2 k01 = { HZ_FB_CLEAR, HZ_FB_CLEAR }, k23 = k01;
        if(*flag)                               /* (the same byte for the whole wave) */
        {
            k01 = src[0]; k23 = src[1];
            if(CLEAR)
            {
                const ulonglong2 ones = { HZ_FB_CLEAR, HZ_FB_CLEAR };
                if((k01.x & k01.y) != HZ_FB_CLEAR) src[0] = ones;
                if((k23.x & k23.y) != HZ_FB_CLEAR) src[1] = ones;
                if((x & (HZ_SEG-1)) == 0) *flag = 0;
            }
        }
        const unsigned long long key[4] = { k01.x, k01.y, k23.x, k23.y };
        uint32_t zi[4], pix[4];
        #pragma unroll
        for(int k=0; k<4; k++)
        {
            zi[k] = (uint32_t)(key[k] >> 40);
            /* reference horizonator-lib.c:185 clear colour (0,0,1) -> B=255; fragment.glsl:15-16 terrain = (red,0,0) -> R;
             * the three bytes B,G,R as the low 24 bits */
            pix[k] = zi[k] == HZ_Z24_MAX ? 0x0000FFu : (((uint32_t)key[k] & 0xFFu) << 16);
        }
        const size_t o = (size_t)yo*SW + x;
        if(bgr)
        {
            uint3 w;
            w.x = pix[0] | (pix[1] << 24);
            w.y = (pix[1] >> 8) | (pix[2] << 16);
            w.z = (pix[2] >> 16) | (pix[3] << 8);
            *(uint3*)(bgr + o*3) = w;
        }
        if(index)
        {
            int4 w;
            w.x = zi[0] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[0] >> 8);
            w.y = zi[1] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[1] >> 8);
            w.z = zi[2] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[2] >> 8);
            w.w = zi[3] == HZ_Z24_MAX ? -1 : (int32_t)(uint32_t)(key[3] >> 8);
            *(int4*)(index + o) = w;
        }
        if(z24) { uint4 w = { zi[0], zi[1], zi[2], zi[3] }; *(uint4*)(z24 + o) = w; }
        if(ranges)
        {
            const float tr = tanel[row];
            float4 w;
            w.x = zi[0] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[0], tr, znear, zfar);
            w.y = zi[1] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[1], tr, znear, zfar);
            w.z = zi[2] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[2], tr, znear, zfar);
            w.w = zi[3] == HZ_Z24_MAX ? -1.0f : hz_range_from_z24(zi[3], tr, znear, zfar);
            *(float4*)(ranges + o) = w;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* packed strips for the multi-GPU gather                                    */
/*
 * A finished strip as BGR8 + float32 range is 7 bytes per pixel, and with N
 * GPUs (N-1)/N of the panorama has to reach the gathering rank through its
 * xGMI links: at N = 2 that is 224 MB over ONE link per panorama, more time
 * than the render itself.  Everything the readback conversion needs is the
 * 24-bit depth and the 8-bit shade, so a rank ships z24<<8 | red8 (4 bytes per
 * pixel, top row first) and the gathering rank runs the conversion
 * (reference horizonator-lib.c:936-1048) on what arrives: same bytes out.
 */
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_pack(unsigned long long* __restrict__ fb, uint32_t* __restrict__ packed, int SW, int H)
{
    const size_t npix = (size_t)SW*H;
    for(size_t o = (size_t)blockIdx.x*blockDim.x + threadIdx.x; o < npix; o += (size_t)gridDim.x*blockDim.x)
    {
        const int yo = (int)(o / SW), x = (int)(o - (size_t)yo*SW);
        const unsigned long long key = fb[(size_t)(H-1 - yo)*SW + x];
        if(CLEAR && key != HZ_FB_CLEAR) fb[(size_t)(H-1 - yo)*SW + x] = HZ_FB_CLEAR;
        packed[o] = ((uint32_t)(key >> 40) << 8) | (uint32_t)(key & 0xFF);
    }
}

/* packed[H][stride] (columns 0..ncols-1 used) -> columns out_col0.. of the
 * full-width outputs bgr[H][out_W][3], ranges[H][out_W]; rows top first */
__global__ __launch_bounds__(256)
void k_resolve_packed(const uint32_t* __restrict__ packed, int stride, int ncols,
                      const float* __restrict__ tanel,
                      unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                      int out_W, int out_col0, int H, float znear, float zfar)
{
    const size_t npix = (size_t)ncols*H;
    for(size_t k = (size_t)blockIdx.x*blockDim.x + threadIdx.x; k < npix; k += (size_t)gridDim.x*blockDim.x)
    {
        const int yo = (int)(k / ncols), x = (int)(k - (size_t)yo*ncols);
        const uint32_t w  = packed[(size_t)yo*stride + x];
        const uint32_t zi = w >> 8;
        const bool sky = (zi == HZ_Z24_MAX);
        const size_t o = (size_t)yo*out_W + out_col0 + x;
        if(bgr)
        {
            bgr[o*3+0] = sky ? 255 : 0;
            bgr[o*3+1] = 0;
            bgr[o*3+2] = sky ? 0 : (unsigned char)(w & 0xFF);
        }
        if(ranges)
        {
            /* reference horizonator-lib.c:1013-1025, as k_resolve */
            float r = -1.0f;
            if(!sky)
            {
                const float depth = (float)((double)zi * (1.0/16777215.0));
                const float len   = depth * (zfar-znear) + znear;
                const float zt    = tanel[H-1 - yo] * len;
                r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);
            }
            ranges[o] = r;
        }
    }
}

/* Sparse strips: most of a panorama is sky (62 % of the benchmark image), and a
 * sky pixel carries no information.  A strip as a stream of uint32:
 *   [0]                    number of terrain pixels T
 *   [1 .. 1+H)             row_base[yo]: where row yo's words start in the data
 *   [1+H .. HDR)           terrain mask, mask_stride words per row, bit c%32 of word c/32
 *   [HDR .. HDR+T)         z24<<8 | red8 of the terrain pixels, row by row, left to right
 * with HDR = 1 + H + H*mask_stride, rows top first.  Rows may be laid out in
 * any order in the data (row_base says where): one block per row, one atomic
 * per row for its base.  The buffer must hold HDR + H*SW words; [0] must be 0
 * on entry. */
template<bool CLEAR>
__global__ __launch_bounds__(256)
void k_pack_sparse(unsigned long long* __restrict__ fb, uint32_t* __restrict__ out,
                   int SW, int H, int mask_stride)
{
    __shared__ uint32_t wave_count[4];
    __shared__ uint32_t row_base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HDR = 1 + (size_t)H + (size_t)H*mask_stride;
    for(int yo = blockIdx.x; yo < H; yo += gridDim.x)
    {
        unsigned long long* row = fb + (size_t)(H-1 - yo)*SW;
        uint32_t* mask = out + 1 + H + (size_t)yo*mask_stride;
        /* pass 1: mask and count */
        uint32_t mine = 0;
        for(int c0 = 0; c0 < SW; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            const bool terrain = c < SW && (uint32_t)(row[c] >> 40) != HZ_Z24_MAX;
            const unsigned long long b = __ballot(terrain);
            if(lane == 0  && c0 + wave*64      < SW) mask[(c0 >> 5) + wave*2]     = (uint32_t)b;
            if(lane == 32 && c0 + wave*64 + 32 < SW) mask[(c0 >> 5) + wave*2 + 1] = (uint32_t)(b >> 32);
            mine += (uint32_t)__popcll(b);                  /* the same in every lane of the wave */
        }
        if(lane == 0) wave_count[wave] = mine;
        __syncthreads();
        if(threadIdx.x == 0)
        {
            const uint32_t total = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
            const uint32_t base = atomicAdd(&out[0], total);
            out[1 + yo] = base;
            row_base_s = base;
        }
        __syncthreads();
        /* pass 2: the words (the row is in L2 now) */
        uint32_t run = row_base_s;
        for(int c0 = 0; c0 < SW; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            unsigned long long key = 0;
            bool terrain = false;
            if(c < SW)
            {
                key = row[c];
                terrain = (uint32_t)(key >> 40) != HZ_Z24_MAX;
                if(CLEAR && key != HZ_FB_CLEAR) row[c] = HZ_FB_CLEAR;
            }
            const unsigned long long b = __ballot(terrain);
            __syncthreads();
            if(lane == 0) wave_count[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0;
            for(int w=0; w<wave; w++) before += wave_count[w];
            if(terrain)
                out[HDR + run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] =
                    ((uint32_t)(key >> 40) << 8) | (uint32_t)(key & 0xFF);
            run += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        }
        __syncthreads();
    }
}

/* the readback conversion on a sparse strip: columns [0,ncols) of the strip go
 * to columns out_col0.. of the full-width outputs */
__global__ __launch_bounds__(256)
void k_resolve_sparse(const uint32_t* __restrict__ in, int mask_stride, int ncols,
                      const float* __restrict__ tanel,
                      unsigned char* __restrict__ bgr, float* __restrict__ ranges,
                      int out_W, int out_col0, int H, float znear, float zfar)
{
    __shared__ uint32_t wave_count[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HDR = 1 + (size_t)H + (size_t)H*mask_stride;
    for(int yo = blockIdx.x; yo < H; yo += gridDim.x)
    {
        const uint32_t* mask = in + 1 + H + (size_t)yo*mask_stride;
        uint32_t run = in[1 + yo];
        const float tan_row = tanel[H-1 - yo];
        for(int c0 = 0; c0 < ncols; c0 += 256)
        {
            const int c = c0 + threadIdx.x;
            const bool terrain = c < ncols && ((mask[c >> 5] >> (c & 31)) & 1u);
            const unsigned long long b = __ballot(terrain);
            __syncthreads();
            if(lane == 0) wave_count[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0;
            for(int w=0; w<wave; w++) before += wave_count[w];
            if(c < ncols)
            {
                const size_t o = (size_t)yo*out_W + out_col0 + c;
                uint32_t w = 0;
                if(terrain) w = in[HDR + run + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))];
                if(bgr)
                {
                    bgr[o*3+0] = terrain ? 0 : 255;
                    bgr[o*3+1] = 0;
                    bgr[o*3+2] = terrain ? (unsigned char)(w & 0xFF) : 0;
                }
                if(ranges)
                {
                    float r = -1.0f;
                    if(terrain)
                    {
                        const float depth = (float)((double)(w >> 8) * (1.0/16777215.0));
                        const float len   = depth * (zfar-znear) + znear;
                        const float zt    = tan_row * len;
                        r = (float)sqrt((double)len*(double)len + (double)zt*(double)zt);
                    }
                    ranges[o] = r;
                }
            }
            run += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------------ */
/* textured resolve ("next" row N4): deferred shading                         */
/*
 * The rasteriser kernels do not know about the texture: the framebuffer word
 * says which triangle won each pixel, and that is all the reference's fragment
 * stage needs beyond the triangle itself.  So for every terrain pixel this
 * kernel builds the winning triangle again (three vertices through the same
 * transform, plus their texture coordinates), sets up the planes of shade, s
 * and t with hz_tri_planes() arithmetic, evaluates them at the pixel, samples
 * the texture and blends (hz_tex.h).  A triangle the clipper cut is clipped
 * again, and the piece that covers the pixel with the stored depth supplies the
 * planes.  This path is not the benchmark's.
 */
__device__ __noinline__ static bool hz_shade_clipped(const hz_cvert_t& a, const hz_cvert_t& b, const hz_cvert_t& c,
                                                     const hz_params_t& p, int px, int py, uint32_t zi,
                                                     hz_texplanes_t* planes)
{
    hz_cvert_t bufa[HZ_MAX_CLIPPED+1], bufb[HZ_MAX_CLIPPED+1], *poly;
    const int n = hz_clip_triangle(bufa, bufb, &poly, &a, &b, &c, p.halfW, p.halfH);
    for(int k=2; k<n; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        if(px < box.px0 || px > box.px1 || py < box.py0 || py > box.py1) continue;
        hz_tri_t tri;
        hz_tri_planes(&tri, &va, &vb, &vc);
        if(!hz_tri_covers(&tri, px, py)) continue;
        uint32_t z2, r8;
        if(!hz_tri_fragment(&tri, px, py, &z2, &r8) || z2 != zi) continue;
        hz_tri_planes_tex(planes, &poly[k-1], &poly[k], &poly[0]);
        return true;
    }
    return false;
}

/* the three vertices of grid triangle `prim` with their texture coordinates */
__device__ static inline void hz_prim_cverts(const int16_t* __restrict__ mosaic, const hz_texparams_t& tp,
                                             const hz_params_t& p, uint32_t prim,
                                             hz_cvert_t* a, hz_cvert_t* b, hz_cvert_t* c)
{
    const uint32_t cell = prim >> 1;
    const int t = prim & 1;
    const int j = cell / (uint32_t)(p.N-1);
    const int i = cell - (uint32_t)j*(uint32_t)(p.N-1);
    /* reference horizonator-lib.c:500-506 */
    const int ib = i+1,              jb = t == 0 ? j+1 : j;
    const int ic = t == 0 ? i : i+1, jc = j+1;
    *a = hz_cvert(hz_transform(&p.u, (float)i,  (float)j,  (float)mosaic[(size_t)j *p.N + i ]), p.halfW, p.halfH);
    *b = hz_cvert(hz_transform(&p.u, (float)ib, (float)jb, (float)mosaic[(size_t)jb*p.N + ib]), p.halfW, p.halfH);
    *c = hz_cvert(hz_transform(&p.u, (float)ic, (float)jc, (float)mosaic[(size_t)jc*p.N + ic]), p.halfW, p.halfH);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)i,  (float)j,  &a->s, &a->t);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)ib, (float)jb, &b->s, &b->t);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)ic, (float)jc, &c->s, &c->t);
}

/* One wave shades TX_CHUNK consecutive output pixels at a time.  Neighbouring
 * pixels mostly belong to the same triangle, and the expensive part - building
 * the triangle again and setting up its planes - depends on the triangle only:
 *   A  the chunk's pixels are cut into runs of equal primitive id (ballot + popcount)
 *   B  lane = run: vertices, planes of shade/s/t into LDS (a triangle the clipper
 *      cut is flagged: its planes depend on the piece that covers the pixel)
 *   C  lane = pixel: planes of its run from LDS, evaluate, sample, blend, store
 * Next to the viewer a chunk holds a handful of runs; at the skyline every pixel
 * is its own run and the scheme falls back to one set-up per pixel. */
#define TX_SUB   4                      /* sub-spans of 64 pixels per chunk */
#define TX_CHUNK (64*TX_SUB)
#define TX_MAXCLIP 4                    /* clipped triangles per chunk whose pieces are kept in LDS */
#define TX_PIECES  (HZ_MAX_CLIPPED-2)
struct tx_lds_t
{
    uint32_t run_prim[TX_CHUNK];
    float    planes[9][TX_CHUNK];       /* r_org drdx drdy s_org dsdx dsdy t_org dtdx dtdy;
                                         * r_org = NaN: clipped, drdx then holds the slot below (-1: none) */
    /* The few triangles next to the viewer that cross the image border cover a
     * large share of the picture (9 triangles, 18% of the terrain pixels in the
     * benchmark scene): their clipped pieces are set up once per chunk.  Per
     * piece: snapped vertices (6), depth plane (3), the nine texture planes. */
    int32_t  npieces[TX_MAXCLIP];
    uint32_t piece[TX_MAXCLIP][TX_PIECES][18];
};

/* B, for a triangle the clipper cuts: all its pieces into LDS slot `slot` */
__device__ __noinline__ static void tx_store_pieces(tx_lds_t& L, int slot, const hz_cvert_t& a, const hz_cvert_t& b,
                                                    const hz_cvert_t& c, const hz_params_t& p)
{
    hz_cvert_t bufa[HZ_MAX_CLIPPED+1], bufb[HZ_MAX_CLIPPED+1], *poly;
    const int n = hz_clip_triangle(bufa, bufb, &poly, &a, &b, &c, p.halfW, p.halfH);
    int count = 0;
    for(int k=2; k<n && count < TX_PIECES; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        hz_tri_t tri;
        hz_tri_planes(&tri, &va, &vb, &vc);
        hz_texplanes_t pl;
        hz_tri_planes_tex(&pl, &poly[k-1], &poly[k], &poly[0]);
        uint32_t* o = L.piece[slot][count++];
        #pragma unroll
        for(int m=0; m<3; m++) { o[m] = (uint32_t)tri.xs[m]; o[3+m] = (uint32_t)tri.ys[m]; }
        o[6] = __float_as_uint(tri.z_org); o[7] = __float_as_uint(tri.dzdx); o[8] = __float_as_uint(tri.dzdy);
        o[9]  = __float_as_uint(pl.r_org); o[10] = __float_as_uint(pl.drdx); o[11] = __float_as_uint(pl.drdy);
        o[12] = __float_as_uint(pl.s_org); o[13] = __float_as_uint(pl.dsdx); o[14] = __float_as_uint(pl.dsdy);
        o[15] = __float_as_uint(pl.t_org); o[16] = __float_as_uint(pl.dtdx); o[17] = __float_as_uint(pl.dtdy);
    }
    L.npieces[slot] = count;
}
/* C, for a pixel of such a triangle: the piece that covers it with the stored depth */
__device__ static inline bool tx_find_piece(const tx_lds_t& L, int slot, int px, int py, uint32_t zi, hz_texplanes_t* pl)
{
    const int n = L.npieces[slot];
    for(int k=0; k<n; k++)
    {
        const uint32_t* o = L.piece[slot][k];
        hz_tri_t tri;
        #pragma unroll
        for(int m=0; m<3; m++) { tri.xs[m] = (int32_t)o[m]; tri.ys[m] = (int32_t)o[3+m]; }
        tri.z_org = __uint_as_float(o[6]); tri.dzdx = __uint_as_float(o[7]); tri.dzdy = __uint_as_float(o[8]);
        tri.r_org = tri.drdx = tri.drdy = 0.f;
        if(!hz_tri_covers(&tri, px, py)) continue;
        uint32_t z2, r8;
        if(!hz_tri_fragment(&tri, px, py, &z2, &r8) || z2 != zi) continue;
        pl->r_org = __uint_as_float(o[9]);  pl->drdx = __uint_as_float(o[10]); pl->drdy = __uint_as_float(o[11]);
        pl->s_org = __uint_as_float(o[12]); pl->dsdx = __uint_as_float(o[13]); pl->dsdy = __uint_as_float(o[14]);
        pl->t_org = __uint_as_float(o[15]); pl->dtdx = __uint_as_float(o[16]); pl->dtdy = __uint_as_float(o[17]);
        return true;
    }
    return false;
}

__global__ __launch_bounds__(64)
void k_shade_tex(const unsigned long long* __restrict__ fb, const int16_t* __restrict__ mosaic,
                 const uint32_t* __restrict__ texels, hz_texparams_t tp,
                 unsigned char* __restrict__ bgr, hz_params_t p)
{
    __shared__ tx_lds_t L;
    const int lane = threadIdx.x;
    const size_t npix = (size_t)p.SW*p.H;
    const size_t nchunks = (npix + TX_CHUNK-1)/TX_CHUNK;
    for(size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
    {
        /* A: runs */
        uint32_t zi_k[TX_SUB], run_k[TX_SUB];
        uint32_t nruns = 0, prev_last = 0xFFFFFFFFu;
        /* output position of this lane's pixel in sub-span 0 (one 64-bit division per
         * chunk), then stepped by 64 pixels */
        const size_t o0 = chunk*TX_CHUNK + lane;
        const int yo0 = (int)(o0 / p.SW), x0 = (int)(o0 - (size_t)yo0*p.SW);
        int yo = yo0, x = x0;
        #pragma unroll
        for(int k=0; k<TX_SUB; k++)
        {
            const size_t o = o0 + (size_t)k*64;
            uint32_t prim = 0xFFFFFFFFu, zi = HZ_Z24_MAX;
            if(o < npix)
            {
                const unsigned long long key = fb[(size_t)(p.H-1 - yo)*p.SW + x];
                zi = (uint32_t)(key >> 40);
                if(zi != HZ_Z24_MAX) prim = (uint32_t)((key >> 8) & 0xFFFFFFFFull);
            }
            uint32_t left = __shfl_up(prim, 1);
            if(lane == 0) left = prev_last;
            prev_last = __shfl(prim, 63);
            const bool is_start = prim != 0xFFFFFFFFu && prim != left;
            const unsigned long long starts = __ballot(is_start);
            const uint32_t upto = (uint32_t)__popcll(starts & ((2ull << lane) - 1ull));   /* starts at lanes <= lane */
            run_k[k] = nruns + upto - 1u;           /* a sky pixel gets a meaningless index it never uses */
            zi_k[k]  = zi;
            if(is_start) L.run_prim[run_k[k]] = prim;
            nruns += (uint32_t)__popcll(starts);
            x += 64;
            while(x >= p.SW) { x -= p.SW; yo++; }
        }
        if(nruns == 0) continue;                    /* sky only */
        __syncthreads();

        /* B: planes per run */
        int clip_slots = 0;
        for(uint32_t r0 = 0; r0 < nruns; r0 += 64)
        {
            const uint32_t r = r0 + lane;
            hz_cvert_t a = {}, b = {}, c = {};
            bool clipped = false;
            if(r < nruns)
            {
                hz_prim_cverts(mosaic, tp, p, L.run_prim[r], &a, &b, &c);
                clipped = (hz_clip_mask(a.xn, a.yn, a.zn) | hz_clip_mask(b.xn, b.yn, b.zn) | hz_clip_mask(c.xn, c.yn, c.zn)) != 0;
            }
            const unsigned long long cm = __ballot(clipped);
            hz_texplanes_t pl = {};
            if(clipped)
            {
                /* one LDS slot per clipped run, while they last (the same triangle may
                 * start several runs of a chunk: row after row) */
                const int slot = clip_slots + (int)__popcll(cm & ((1ull << lane) - 1ull));
                pl.r_org = __uint_as_float(0x7FC00000u);
                pl.drdx  = (float)(slot < TX_MAXCLIP ? slot : -1);
                if(slot < TX_MAXCLIP) tx_store_pieces(L, slot, a, b, c, p);
            }
            else if(r < nruns)
                hz_tri_planes_tex(&pl, &a, &b, &c);
            clip_slots += (int)__popcll(cm);
            if(r >= nruns) continue;
            L.planes[0][r] = pl.r_org; L.planes[1][r] = pl.drdx; L.planes[2][r] = pl.drdy;
            L.planes[3][r] = pl.s_org; L.planes[4][r] = pl.dsdx; L.planes[5][r] = pl.dsdy;
            L.planes[6][r] = pl.t_org; L.planes[7][r] = pl.dtdx; L.planes[8][r] = pl.dtdy;
        }
        __syncthreads();

        /* C: pixels */
        yo = yo0; x = x0;
        #pragma unroll
        for(int k=0; k<TX_SUB; k++)
        {
            const size_t o = o0 + (size_t)k*64;
            const int py = p.H-1 - yo, px = x + p.col0;
            x += 64;
            while(x >= p.SW) { x -= p.SW; yo++; }
            if(o >= npix || zi_k[k] == HZ_Z24_MAX) continue;       /* sky: k_resolve wrote the clear colour */
            const uint32_t r = run_k[k];
            hz_texplanes_t pl;
            pl.r_org = L.planes[0][r]; pl.drdx = L.planes[1][r]; pl.drdy = L.planes[2][r];
            pl.s_org = L.planes[3][r]; pl.dsdx = L.planes[4][r]; pl.dsdy = L.planes[5][r];
            pl.t_org = L.planes[6][r]; pl.dtdx = L.planes[7][r]; pl.dtdy = L.planes[8][r];
            if(!(pl.r_org == pl.r_org))
            {
                /* the clipper cut this triangle: the piece that covers this pixel with the stored depth */
                const int slot = (int)pl.drdx;
                if(slot < 0 || !tx_find_piece(L, slot, px, py, zi_k[k], &pl))
                {
                    hz_cvert_t a, b, c;
                    hz_prim_cverts(mosaic, tp, p, L.run_prim[r], &a, &b, &c);
                    if(!hz_shade_clipped(a, b, c, p, px, py, zi_k[k], &pl)) continue;   /* cannot happen; keeps the untextured colour */
                }
            }
            const float shade = hz_plane_at(pl.r_org, pl.drdx, pl.drdy, px, py);
            const float s     = hz_plane_at(pl.s_org, pl.dsdx, pl.dsdy, px, py);
            const float tt    = hz_plane_at(pl.t_org, pl.dtdx, pl.dtdy, px, py);
            const uint32_t col = hz_fragment_textured(hz_tex_sample(texels, tp.tex_w, tp.tex_h, s, tt), shade);
            bgr[o*3+0] = (unsigned char)(col & 255u);
            bgr[o*3+1] = (unsigned char)((col >> 8) & 255u);
            bgr[o*3+2] = (unsigned char)((col >> 16) & 255u);
        }
        __syncthreads();                            /* the next chunk reuses the LDS tables */
    }
}

/* ------------------------------------------------------------------------ */
/* host side of the C-ABI                                                    */

struct hz_dev
{
    int device;
    int N, W, H;
    int col0, col1;
    int raster;
    int profiling;
    int serial;                         /* HZ_SERIAL: the four streams are one */

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
    hipEvent_t          ev_marched, ev_qfree[HZ_NFB], ev_near, ev_nqfree[HZ_NFB];
    hz_bigrec_t*        d_bigrec_s[2*HZ_NFB];          /* [0..NFB) one-round draws and second rounds, [NFB..2 NFB) first rounds */
    hz_bigitem_t*       d_bigitem_s[2*HZ_NFB];
    hz_rec_t*           d_midrec_s[2*HZ_NFB];
    uint32_t*           d_clip_s[2*HZ_NFB];
    unsigned int*       d_big_counters_s[2*HZ_NFB];    /* HZ_NCOUNTERS each, see mr_queue_t */
    unsigned int        bigrec_capacity, bigitem_capacity, midrec_capacity, clip_capacity;
    /* the last draw: a conversion that clears the framebuffer behind itself
     * (k_resolve<true>) consumes it; whoever wants to read it after that gets it
     * drawn again first (fb_refill) */
    hz_view_t           last_view;
    int                 have_view;
    int                 fb_consumed;
    int                 resolve_clears;         /* HZ_RESOLVE_CLEARS=0 switches the fused clear off */
    float*              d_tanel;
    float*              h_tanel;        /* the table d_tanel holds (or is about to, in stream order) */
    int                 tanel_resident;

    /* results into caller-owned host memory (hz_hip_resolve_to_host): a ring of
     * pinned staging chunks between the copy engine and the threads that move
     * the bytes on into the caller's (pageable) buffers */
    hipStream_t    cstream;
    unsigned char* h_stage[HZ_STAGE_SLOTS];
    hipEvent_t     ev_stage[HZ_STAGE_SLOTS];
    hipEvent_t     ev_resolved;

    /* internal output buffers for *_to_host */
    unsigned char* d_bgr;
    float*         d_ranges;
    int32_t*       d_index;
    uint32_t*      d_z24;

    /* texture path: the mosaic of map tiles, one uint32 B|G<<8|R<<16 per texel */
    uint32_t*      d_texels;
    hz_texparams_t tex;
    int            tex_on;

    hipEvent_t ev[10];
    int        have_times;
    hz_times_t times;
};

extern "C" int hz_hip_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* everything queued on any of the context's streams is done */
static hipError_t sync_all(hz_dev_t* d)
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
    (void)sync_all(d);          /* nothing of this context is still running when its memory goes */
    (void)hipFree(d->d_mosaic);
    for(int i=0; i<HZ_NFB; i++) { (void)hipFree(d->d_fbs[i]); (void)hipFree(d->d_touched[i]); }
    for(int i=0; i<2*HZ_NFB; i++)
    {
        (void)hipFree(d->d_bigrec_s[i]);
        (void)hipFree(d->d_bigitem_s[i]);
        (void)hipFree(d->d_midrec_s[i]);
        (void)hipFree(d->d_clip_s[i]);
        (void)hipFree(d->d_big_counters_s[i]);
    }
    for(int i=0; i<HZ_NFB; i++)
    {
        if(d->ev_qfree[i])  (void)hipEventDestroy(d->ev_qfree[i]);
        if(d->ev_nqfree[i]) (void)hipEventDestroy(d->ev_nqfree[i]);
    }
    if(d->ev_marched) (void)hipEventDestroy(d->ev_marched);
    if(d->ev_near)    (void)hipEventDestroy(d->ev_near);
    (void)hipFree(d->d_texels);
    (void)hipFree(d->d_tanel);
    free(d->h_tanel);
    if(d->cstream) (void)hipStreamSynchronize(d->cstream);
    for(int k=0; k<HZ_STAGE_SLOTS; k++)
    {
        if(d->h_stage[k])  (void)hipHostFree(d->h_stage[k]);
        if(d->ev_stage[k]) (void)hipEventDestroy(d->ev_stage[k]);
    }
    if(d->ev_resolved) (void)hipEventDestroy(d->ev_resolved);
    if(d->cstream) (void)hipStreamDestroy(d->cstream);
    (void)hipFree(d->d_bgr);
    (void)hipFree(d->d_ranges);
    (void)hipFree(d->d_index);
    (void)hipFree(d->d_z24);
    for(int k=0; k<10; k++) if(d->ev[k]) (void)hipEventDestroy(d->ev[k]);
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

static int create_impl(hz_dev_t* d)
{
    HZ_ON_DEVICE(d);
    /* HZ_SERIAL=1 (profiling): one stream, nothing overlaps - per-kernel times
     * of a trace are then those of each kernel alone on the chip */
    {
        const char* ser = getenv("HZ_SERIAL");
        d->serial = ser && atoi(ser) != 0;
    }
    HZ_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    if(d->serial) d->rstream = d->stream;
    else HZ_CHECK(hipStreamCreateWithFlags(&d->rstream, hipStreamNonBlocking));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_drawn,   hipEventDisableTiming));
    for(int i=0; i<HZ_NFB; i++) HZ_CHECK(hipEventCreateWithFlags(&d->ev_free[i], hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_readers, hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_tanel,   hipEventDisableTiming));
    HZ_CHECK(hipMalloc(&d->d_mosaic, (size_t)d->N*d->N*sizeof(int16_t)));
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
    /* queue of triangles too large for k_scatter's in-block pass.  cfg3
     * (16000x4000) produces ~0.3 M records and ~0.4 M items; sized for 32k-wide */
    d->bigrec_capacity  = 1u<<21;
    d->bigitem_capacity = 1u<<22;
    d->midrec_capacity  = 1u<<21;
    d->clip_capacity    = 1u<<21;
    {
        /* tests shrink the queues to exercise the overflow paths */
        const char* cap = getenv("HZ_QUEUE_CAPACITY");
        if(cap && atoi(cap) > 0)
            d->bigrec_capacity = d->bigitem_capacity = d->midrec_capacity = d->clip_capacity = (unsigned int)atoi(cap);
    }
    if(d->serial) d->qstream = d->nstream = d->stream;
    else
    {
        HZ_CHECK(hipStreamCreateWithFlags(&d->qstream, hipStreamNonBlocking));
        HZ_CHECK(hipStreamCreateWithFlags(&d->nstream, hipStreamNonBlocking));
    }
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_marched, hipEventDisableTiming));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_near,    hipEventDisableTiming));
    for(int i=0; i<2*HZ_NFB; i++)
    {
        HZ_CHECK(hipMalloc(&d->d_bigrec_s[i],  (size_t)d->bigrec_capacity*sizeof(hz_bigrec_t)));
        HZ_CHECK(hipMalloc(&d->d_bigitem_s[i], (size_t)d->bigitem_capacity*sizeof(hz_bigitem_t)));
        HZ_CHECK(hipMalloc(&d->d_midrec_s[i],  (size_t)d->midrec_capacity*sizeof(hz_rec_t)));
        HZ_CHECK(hipMalloc(&d->d_clip_s[i],    (size_t)d->clip_capacity*sizeof(uint32_t)));
        HZ_CHECK(hipMalloc(&d->d_big_counters_s[i], HZ_NCOUNTERS*sizeof(unsigned int)));
    }
    for(int i=0; i<HZ_NFB; i++)
    {
        HZ_CHECK(hipEventCreateWithFlags(&d->ev_qfree[i],  hipEventDisableTiming));
        HZ_CHECK(hipEventCreateWithFlags(&d->ev_nqfree[i], hipEventDisableTiming));
        HZ_CHECK(hipEventRecord(d->ev_qfree[i],  d->qstream));
        HZ_CHECK(hipEventRecord(d->ev_nqfree[i], d->nstream));
    }
    HZ_CHECK(hipEventRecord(d->ev_drawn, d->qstream));
    {
        const char* rc = getenv("HZ_RESOLVE_CLEARS");
        d->resolve_clears = !(rc && atoi(rc) == 0);
    }
    HZ_CHECK(hipMalloc(&d->d_tanel, (size_t)d->H*sizeof(float)));
    d->h_tanel = (float*)malloc((size_t)d->H*sizeof(float));
    d->tanel_resident = 0;
    for(int k=0; k<10; k++) HZ_CHECK(hipEventCreate(&d->ev[k]));
    return 0;
}

extern "C" hz_dev_t* hz_hip_create(int device, int N, int width, int height)
{
    if(N < 2 || width <= 0 || height <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_create: bad sizes N=%d W=%d H=%d", N, width, height);
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
    HZ_CHECK(sync_all(d));      /* draws in flight (first rounds run on a stream of their own) still read the old one */
    HZ_CHECK(hipMemcpyAsync(d->d_mosaic, mosaic, (size_t)d->N*d->N*sizeof(int16_t), hipMemcpyHostToDevice, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int hz_hip_download_mosaic(hz_dev_t* d, int16_t* mosaic)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(hipMemcpyAsync(mosaic, d->d_mosaic, (size_t)d->N*d->N*sizeof(int16_t), hipMemcpyDeviceToHost, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}

/* ingest: raw .hgt tiles -> mosaic, on the device (reference dem.c:264-309) */
__global__ __launch_bounds__(256)
void k_ingest(const unsigned char* const* __restrict__ tiles, int16_t* __restrict__ mosaic,
              int N, int ntx, int nty, int cpd, int oc_x, int oc_y)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if(i >= N) return;
    int cx = i + oc_x, tx = cx / cpd; cx -= tx*cpd; if(cx == 0 && tx > 0) { tx--; cx = cpd; }
    int cy = j + oc_y, ty = cy / cpd; cy -= ty*cpd; if(cy == 0 && ty > 0) { ty--; cy = cpd; }
    int16_t z = -1;
    if(tx < ntx && ty < nty)
    {
        const unsigned char* t = tiles[tx + ty*ntx];
        if(t == NULL) z = 0;
        else
        {
            const size_t q = (size_t)cx + (size_t)(cpd - cy)*(size_t)(cpd+1);
            const unsigned short be = *(const unsigned short*)(t + 2*q);
            z = (int16_t)(unsigned short)((be << 8) | (be >> 8));
            if(z < 0) z = 0;
        }
    }
    mosaic[(size_t)j*N + i] = z;
}

extern "C" int hz_hip_ingest_tiles(hz_dev_t* d, const unsigned char* const* tiles,
                                   int ntx, int nty, int cpd, int oc_x, int oc_y)
{
    HZ_ON_DEVICE(d);
    const int nt = ntx*nty;
    const size_t tile_bytes = (size_t)(cpd+1)*(cpd+1)*2;
    unsigned char** h_ptrs = (unsigned char**)calloc(nt, sizeof(*h_ptrs));
    unsigned char** d_ptrs = NULL;
    int rc = -1;
    if(!h_ptrs) return -1;
    if(sync_all(d) != hipSuccess) { free(h_ptrs); return -1; }      /* draws in flight still read the old mosaic */
    do {
        bool ok = true;
        for(int k=0; k<nt && ok; k++)
        {
            if(tiles[k] == NULL) continue;
            if(hipMalloc(&h_ptrs[k], tile_bytes) != hipSuccess) { ok = false; break; }
            if(hipMemcpyAsync(h_ptrs[k], tiles[k], tile_bytes, hipMemcpyHostToDevice, d->stream) != hipSuccess) ok = false;
        }
        if(!ok) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_ingest_tiles: tile upload failed"); break; }
        if(hipMalloc(&d_ptrs, nt*sizeof(*d_ptrs)) != hipSuccess) break;
        if(hipMemcpyAsync(d_ptrs, h_ptrs, nt*sizeof(*d_ptrs), hipMemcpyHostToDevice, d->stream) != hipSuccess) break;
        dim3 grid((d->N + 255)/256, d->N);
        hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, d->stream,
                           (const unsigned char* const*)d_ptrs, d->d_mosaic, d->N, ntx, nty, cpd, oc_x, oc_y);
        if(hipGetLastError() != hipSuccess) break;
        if(hipStreamSynchronize(d->stream) != hipSuccess) break;
        rc = 0;
    } while(0);
    (void)hipStreamSynchronize(d->stream);
    for(int k=0; k<nt; k++) if(h_ptrs[k]) (void)hipFree(h_ptrs[k]);
    if(d_ptrs) (void)hipFree(d_ptrs);
    free(h_ptrs);
    return rc;
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
static mr_zones_t mr_make_zones(const hz_params_t& p, bool near_first)
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
    int far_rows = 64*p.SW/p.W;
    if(far_rows < 16) far_rows = 16;
    if(far_rows > 64) far_rows = 64;
    const int rows[MR_NZONES] = { far_rows, 16, 4, 2, 4, 16, far_rows };
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

static hz_params_t make_params(const hz_dev_t* d, const hz_view_t* v)
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
    p.quad_max_dx = d->W >= 64 && d->W <= (1<<20) ? 256*(d->W/16 - 1) : 0;
    {
        const char* dbg = getenv("HZ_MARCH_DEBUG");
        p.debug = dbg ? atoi(dbg) : 0;
        const char* nf = getenv("HZ_NO_FAST_MATH");         /* diagnostics: the unabridged transform everywhere */
        p.fast_ok = hzf_draw_ok(&p.u) && !(nf && atoi(nf) != 0);
    }
    return p;
}

/* start of a round: empty its queues */
__global__ void k_reset_counters(unsigned int* counters)
{
    counters[0] = 0u; counters[1] = 0u; counters[2] = 0xFFFFFFFFu; counters[3] = 0u;
    counters[4] = 0u; counters[5] = 0xFFFFFFFFu;
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
 * HZ_TWO_PASS=0/1 forces one / two rounds; otherwise full-width contexts of at
 * least HZ_TWO_PASS_MIN_MPIX (default 24) megapixels whose far clip lies well
 * beyond the first round's strips draw in two rounds. */
static int draw_impl(hz_dev_t* d, const hz_view_t* view);

extern "C" int hz_hip_draw(hz_dev_t* d, const hz_view_t* view)
{
    HZ_ON_DEVICE(d);
    return draw_impl(d, view);
}

static int draw_impl(hz_dev_t* d, const hz_view_t* view)
{
    hz_params_t p = make_params(d, view);
    const bool prof = d->profiling != 0;
    d->last_view = *view; d->have_view = 1; d->fb_consumed = 0;

    /* The framebuffer of the previous draw goes back to "cleared" (glClear,
     * reference horizonator-lib.c:896: depth = 1.0 -> all-ones words): its
     * conversion did that already (k_resolve<true>), or a memset does it now on
     * rstream, behind the conversions of that draw and behind whatever `stream`
     * still had to read from it; this draw takes the next framebuffer. */
    const int prev = d->fbi, next = (prev + 1) % HZ_NFB;
    {
        HZ_CHECK(hipEventRecord(d->ev_readers, d->stream));
        HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_readers, 0));
        HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_drawn, 0));       /* the previous draw's last kernels (qstream) */
        if(prof) HZ_CHECK(hipEventRecord(d->ev[0], d->rstream));
        if(d->fb_used[prev])
        {
            HZ_CHECK(hipMemsetAsync(d->d_fbs[prev], 0xFF, d->fb_used[prev]*sizeof(unsigned long long), d->rstream));
            HZ_CHECK(hipMemsetAsync(d->d_touched[prev], 0, (size_t)d->seg_stride*d->H, d->rstream));
        }
        if(prof) HZ_CHECK(hipEventRecord(d->ev[1], d->rstream));
        d->fb_used[prev] = 0;
        HZ_CHECK(hipEventRecord(d->ev_free[prev], d->rstream));
        d->fbi = next; d->d_fb = d->d_fbs[next];
        d->fb_used[next] = (size_t)p.SW*p.H;
        p.touched = d->d_touched[next];
    }

    auto queue_set = [&](int k) -> mr_queue_t
    {
        mr_queue_t q = { d->d_bigrec_s[k], d->d_bigitem_s[k], d->d_midrec_s[k], d->d_clip_s[k], d->d_big_counters_s[k],
                         d->bigrec_capacity, d->bigitem_capacity, d->midrec_capacity, d->clip_capacity };
        return q;
    };
    auto queue_kernels = [&](const mr_queue_t& q, const hz_params_t& pp, hipStream_t st) -> int
    {
        hipLaunchKernelGGL(k_clip, dim3(1024), dim3(64), 0, st, (const int16_t*)d->d_mosaic, d->d_fb, q, pp);
        HZ_CHECK(hipGetLastError());
        if(d->raster != HZ_RASTER_SCATTER && pp.inline_max < pp.big_min)      /* else nothing is ever queued for it */
        {
            hipLaunchKernelGGL(k_mid, dim3(2048), dim3(64), 0, st,
                               d->d_fb, (const hz_rec_t*)q.midrec, (const unsigned int*)q.counters,
                               d->midrec_capacity, pp);
            HZ_CHECK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 0, st,
                           d->d_fb, (const hz_bigrec_t*)q.bigrec, (const hz_bigitem_t*)q.bigitem,
                           (const unsigned int*)q.counters, d->bigrec_capacity, d->bigitem_capacity, pp);
        HZ_CHECK(hipGetLastError());
        return 0;
    };

    const mr_queue_t q = queue_set(next);           /* one-round draw, or second round */

    if(d->raster == HZ_RASTER_SCATTER)
    {
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_free[next], 0));
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_qfree[next], 0));  /* the queue set is free again */
        if(prof) { HZ_CHECK(hipEventRecord(d->ev[7], d->stream)); HZ_CHECK(hipEventRecord(d->ev[6], d->stream)); }
        hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, d->stream, q.counters);
        if(prof) HZ_CHECK(hipEventRecord(d->ev[9], d->stream));
        dim3 grid((p.N-1 + SC_CX-1)/SC_CX, (p.N-1 + SC_CY-1)/SC_CY);
        hipLaunchKernelGGL(k_scatter, grid, dim3(SC_THREADS), 0, d->stream,
                           (const int16_t*)d->d_mosaic, d->d_fb, q, p);
        HZ_CHECK(hipGetLastError());
    }
    else
    {
        const int nsx = (p.N-1 + MR_COLS-1)/MR_COLS;
        /* the strips next to the viewer: within near_cells cells of the viewer's cell */
        const char* e2 = getenv("HZ_TWO_PASS");
        const char* en = getenv("HZ_NEAR_CELLS");
        const char* em = getenv("HZ_TWO_PASS_MIN_MPIX");
        const int near_cells = en ? atoi(en) : MR_NEAR_CELLS;
        p.near_x0 = (int)floorf((p.u.viewer_cell_i - (float)near_cells)/(float)MR_COLS);
        p.near_x1 = (int)floorf((p.u.viewer_cell_i + (float)near_cells)/(float)MR_COLS);
        if(p.near_x0 < 0) p.near_x0 = 0;
        if(p.near_x1 > nsx-1) p.near_x1 = nsx-1;
        p.near_j0 = (int)floorf(p.u.viewer_cell_j - (float)near_cells);
        p.near_j1 = (int)ceilf (p.u.viewer_cell_j + (float)near_cells);
        const double min_mpix = em ? atof(em) : 24.0;
        /* two rounds pay where there is a lot of terrain behind the first round's
         * strips: a large image (many pixel tests to save) and a far clip well
         * beyond them - with the API's default 40 km far clip most of a large
         * mosaic is never transformed at all and one round is faster (measured:
         * 16000x4000 over 7x7 tiles, 0.85 vs 0.91 ms) */
        const float cells_to_zfar = view->zfar / (p.u.deg_per_cell * 111194.9f);
        const bool want_two = e2 ? atoi(e2) != 0
                                 : (p.SW == p.W && (double)p.W*(double)p.H >= min_mpix*1e6 && cells_to_zfar >= 24.0f*(float)near_cells);
        const bool two_pass = want_two && near_cells > 0 && p.near_x1 >= p.near_x0;
        const mr_zones_t zn = mr_make_zones(p, two_pass);
        if(two_pass)
        {
            /* round 1, on its own stream */
            const mr_queue_t qn = queue_set(HZ_NFB + next);
            HZ_CHECK(hipStreamWaitEvent(d->nstream, d->ev_free[next], 0));
            HZ_CHECK(hipStreamWaitEvent(d->nstream, d->ev_nqfree[next], 0));
            if(prof) HZ_CHECK(hipEventRecord(d->ev[7], d->nstream));
            hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, d->nstream, qn.counters);
            hz_params_t p1 = p;
            p1.pass = 1; p1.early_z = 0;
            hipLaunchKernelGGL(k_march, dim3(p.near_x1 - p.near_x0 + 1, zn.total), dim3(64), 0, d->nstream,
                               (const int16_t*)d->d_mosaic, d->d_fb, qn, zn, p1);
            HZ_CHECK(hipGetLastError());
            if(queue_kernels(qn, p1, d->nstream) != 0) return -1;
            if(prof) HZ_CHECK(hipEventRecord(d->ev[6], d->nstream));
            HZ_CHECK(hipEventRecord(d->ev_nqfree[next], d->nstream));
            HZ_CHECK(hipEventRecord(d->ev_near, d->nstream));
            HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_near, 0));
            p.pass = 2; p.early_z = 1;
        }
        else if(prof) { HZ_CHECK(hipEventRecord(d->ev[7], d->stream)); HZ_CHECK(hipEventRecord(d->ev[6], d->stream)); }
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_free[next], 0));
        HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_qfree[next], 0));  /* the queue set is free again */
        hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(1), 0, d->stream, q.counters);
        if(prof) HZ_CHECK(hipEventRecord(d->ev[9], d->stream));

        dim3 grid(nsx, zn.total);
        /* diagnostics: HZ_WAVE_TIMING=<file> dumps the duration (shader clock
         * cycles) of every k_march wave of this launch as uint64[grid.y][grid.x][4] */
        const char* timing_path = getenv("HZ_WAVE_TIMING");
        hz_params_t pm = p;
        unsigned long long* d_cycles = NULL;
        if(timing_path)
        {
            HZ_CHECK(hipMalloc(&d_cycles, (size_t)grid.x*grid.y*4*sizeof(unsigned long long)));
            HZ_CHECK(hipMemsetAsync(d_cycles, 0, (size_t)grid.x*grid.y*4*sizeof(unsigned long long), d->stream));
            pm.wave_cycles = d_cycles;
        }
        hipLaunchKernelGGL(k_march, grid, dim3(64), 0, d->stream,
                           (const int16_t*)d->d_mosaic, d->d_fb, q, zn, pm);
        if(timing_path)
        {
            const size_t n = (size_t)grid.x*grid.y*4;
            unsigned long long* h = (unsigned long long*)malloc(n*sizeof(*h));
            HZ_CHECK(hipMemcpyAsync(h, d_cycles, n*sizeof(*h), hipMemcpyDeviceToHost, d->stream));
            HZ_CHECK(hipStreamSynchronize(d->stream));
            FILE* f = fopen(timing_path, "wb");
            if(f) { unsigned int hdr[2] = { grid.x, grid.y }; fwrite(hdr, 4, 2, f); fwrite(h, sizeof(*h), n, f); fclose(f); }
            free(h);
            (void)hipFree(d_cycles);
        }
        HZ_CHECK(hipGetLastError());
    }
    if(prof) HZ_CHECK(hipEventRecord(d->ev[2], d->stream));
    /* the kernels that finish the draw run on qstream, so that the next draw's
     * marching kernel can start beside them */
    HZ_CHECK(hipEventRecord(d->ev_marched, d->stream));
    HZ_CHECK(hipStreamWaitEvent(d->qstream, d->ev_marched, 0));
    if(prof) HZ_CHECK(hipEventRecord(d->ev[8], d->qstream));
    if(queue_kernels(q, p, d->qstream) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[3], d->qstream));
    HZ_CHECK(hipEventRecord(d->ev_qfree[next], d->qstream));
    HZ_CHECK(hipEventRecord(d->ev_drawn, d->qstream));
    d->have_times = prof ? 1 : 0;
    return 0;
}

/* a reader of the framebuffer finds it consumed (cleared by the conversion
 * that ran before): the draw is repeated - same view, same bytes */
static int fb_refill(hz_dev_t* d)
{
    if(!d->fb_consumed) return 0;
    if(!d->have_view) { snprintf(g_last_error, sizeof(g_last_error), "nothing has been drawn yet"); return -1; }
    const hz_view_t v = d->last_view;
    return draw_impl(d, &v);
}

/* the conversion just queued on rstream left the framebuffer of the last draw
 * all ones */
static int fb_mark_consumed(hz_dev_t* d)
{
    d->fb_consumed = 1;
    d->fb_used[d->fbi] = 0;
    HZ_CHECK(hipEventRecord(d->ev_free[d->fbi], d->rstream));
    return 0;
}

/* The per-row tan(elevation) table only changes with the azimuth extents: a
 * table equal to the resident one is not sent again (a host->device copy from
 * pageable memory would otherwise stall the host on the stream once per render) */
static int upload_tanel(hz_dev_t* d, const float* tanel)
{
    if(!tanel) { snprintf(g_last_error, sizeof(g_last_error), "a tanel table is required"); return -1; }
    const size_t bytes = (size_t)d->H*sizeof(float);
    if(d->tanel_resident && memcmp(d->h_tanel, tanel, bytes) == 0) return 0;
    /* a different table (the azimuth extents changed): nothing queued on either
     * stream may still read the old one, and both streams must see the new one */
    HZ_CHECK(sync_all(d));
    memcpy(d->h_tanel, tanel, bytes);
    HZ_CHECK(hipMemcpy(d->d_tanel, d->h_tanel, bytes, hipMemcpyHostToDevice));
    d->tanel_resident = 1;
    return 0;
}

/* conversions of the last draw run on rstream, behind that draw */
static int rstream_after_draw(hz_dev_t* d)
{
    HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_drawn, 0));
    return 0;
}

extern "C" int hz_hip_resolve(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                              unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(ranges)
    {
        if(!tanel)
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve: ranges requested without a tanel table");
            return -1;
        }
        if(upload_tanel(d, tanel) != 0) return -1;
    }
    if(fb_refill(d) != 0) return -1;
    if(rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    /* the textured resolve reads the framebuffer after this kernel: no fused clear then */
    const bool clears = d->resolve_clears && !(d->tex_on && bgr);
    const bool wide = (SW % 4) == 0 && (((uintptr_t)bgr | (uintptr_t)ranges | (uintptr_t)index | (uintptr_t)z24 | (uintptr_t)d->d_fb) & 15u) == 0;
    if(wide)
    {
        const dim3 grid((unsigned)((SW/4 + 255)/256), (unsigned)(d->H < 2048 ? d->H : 2048));
        if(clears)
            hipLaunchKernelGGL(k_resolve4<true>, grid, dim3(256), 0, d->rstream, d->d_fb, (const float*)d->d_tanel,
                               bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride);
        else
            hipLaunchKernelGGL(k_resolve4<false>, grid, dim3(256), 0, d->rstream, d->d_fb, (const float*)d->d_tanel,
                               bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride);
    }
    else if(clears)
        hipLaunchKernelGGL(k_resolve<true>, dim3((unsigned)nblocks), dim3(256), 0, d->rstream,
                           d->d_fb, (const float*)d->d_tanel,
                           bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar);
    else
        hipLaunchKernelGGL(k_resolve<false>, dim3((unsigned)nblocks), dim3(256), 0, d->rstream,
                           d->d_fb, (const float*)d->d_tanel,
                           bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar);
    HZ_CHECK(hipGetLastError());
    if(clears && fb_mark_consumed(d) != 0) return -1;
    if(d->tex_on && bgr)
    {
        /* reference fragment.glsl:17-22 instead of :15-16 for the terrain pixels */
        const hz_params_t p = make_params(d, view);
        size_t nchunks = (npix + TX_CHUNK-1)/TX_CHUNK;
        if(nchunks > 256*64) nchunks = 256*64;
        hipLaunchKernelGGL(k_shade_tex, dim3((unsigned)nchunks), dim3(64), 0, d->rstream,
                           (const unsigned long long*)d->d_fb, (const int16_t*)d->d_mosaic,
                           (const uint32_t*)d->d_texels, d->tex, bgr, p);
        HZ_CHECK(hipGetLastError());
    }
    if(prof)
    {
        HZ_CHECK(hipEventRecord(d->ev[5], d->rstream));
        d->have_times = 2;
    }
    return 0;
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
    if(fb_refill(d) != 0) return -1;
    if(rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    if(d->resolve_clears)
    {
        hipLaunchKernelGGL(k_pack<true>, dim3((unsigned)nblocks), dim3(256), 0, d->rstream, d->d_fb, d_packed, SW, d->H);
        HZ_CHECK(hipGetLastError());
        if(fb_mark_consumed(d) != 0) return -1;
    }
    else
        hipLaunchKernelGGL(k_pack<false>, dim3((unsigned)nblocks), dim3(256), 0, d->rstream, d->d_fb, d_packed, SW, d->H);
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
    if(d_ranges && upload_tanel(d, tanel) != 0) return -1;
    const size_t npix = (size_t)ncols*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    hipLaunchKernelGGL(k_resolve_packed, dim3((unsigned)nblocks), dim3(256), 0, d->rstream,
                       d_packed, stride, ncols, (const float*)d->d_tanel, d_bgr, d_ranges,
                       d->W, out_col0, d->H, view->znear, view->zfar);
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
    if(d->tex_on || mask_stride < (SW + 31)/32)
    {
        snprintf(g_last_error, sizeof(g_last_error), d->tex_on ? "hz_hip_pack_sparse: strips carry the shade only, not a textured colour"
                                                               : "hz_hip_pack_sparse: mask stride too small");
        return -1;
    }
    const bool prof = d->profiling != 0;
    if(fb_refill(d) != 0) return -1;
    if(rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    HZ_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint32_t), d->rstream));
    if(d->resolve_clears)
    {
        hipLaunchKernelGGL(k_pack_sparse<true>, dim3((unsigned)d->H), dim3(256), 0, d->rstream, d->d_fb, d_out, SW, d->H, mask_stride);
        HZ_CHECK(hipGetLastError());
        if(fb_mark_consumed(d) != 0) return -1;
    }
    else
        hipLaunchKernelGGL(k_pack_sparse<false>, dim3((unsigned)d->H), dim3(256), 0, d->rstream, d->d_fb, d_out, SW, d->H, mask_stride);
    HZ_CHECK(hipGetLastError());
    if(prof) { HZ_CHECK(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
    return 0;
}

extern "C" int hz_hip_resolve_sparse(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     const uint32_t* d_in, int mask_stride, int ncols, int out_col0,
                                     unsigned char* d_bgr, float* d_ranges)
{
    HZ_ON_DEVICE(d);
    if(ncols <= 0 || mask_stride < (ncols + 31)/32 || out_col0 < 0 || out_col0 + ncols > d->W)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse: columns [%d,%d) do not fit a %d-wide image",
                 out_col0, out_col0 + ncols, d->W);
        return -1;
    }
    if(d_ranges && upload_tanel(d, tanel) != 0) return -1;
    hipLaunchKernelGGL(k_resolve_sparse, dim3((unsigned)d->H), dim3(256), 0, d->rstream,
                       d_in, mask_stride, ncols, (const float*)d->d_tanel, d_bgr, d_ranges,
                       d->W, out_col0, d->H, view->znear, view->zfar);
    HZ_CHECK(hipGetLastError());
    return 0;
}

static int ensure_out_buffers(hz_dev_t* d, bool bgr, bool ranges, bool index, bool z24)
{
    const size_t npix = (size_t)d->W*d->H;
    if(bgr    && !d->d_bgr)    HZ_CHECK(hipMalloc(&d->d_bgr,    npix*3));
    if(ranges && !d->d_ranges) HZ_CHECK(hipMalloc(&d->d_ranges, npix*sizeof(float)));
    if(index  && !d->d_index)  HZ_CHECK(hipMalloc(&d->d_index,  npix*sizeof(int32_t)));
    if(z24    && !d->d_z24)    HZ_CHECK(hipMalloc(&d->d_z24,    npix*sizeof(uint32_t)));
    return 0;
}

/* ---- device -> caller memory --------------------------------------------------
 * The reference hands its results to the caller in host memory (reference
 * horizonator-lib.c:936-1048: glReadPixels into the caller's buffers), and so
 * does horizonator_render_offscreen().  A 16000x4000 panorama is 448 MB of
 * results.  hipMemcpy into pageable memory moves that at ~11 GB/s (40 ms,
 * twenty times the render); here the copy engine writes 32 MB chunks into a
 * ring of pinned staging buffers at the link's rate while a few host threads
 * move each finished chunk on into the caller's buffer (non-temporal memcpy,
 * the pages faulted in by several threads at once), chunk k+1.. being in
 * flight meanwhile. */
struct hz_copy_pool
{
    std::mutex m, busy;                 /* busy: one copy() at a time (contexts on several threads share the pool) */
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    unsigned char* dst = nullptr; const unsigned char* src = nullptr;
    size_t bytes = 0;
    int nparts = 0, next = 0, pending = 0;
    unsigned long long generation = 0;
    bool stop = false;

    explicit hz_copy_pool(int n)
    {
        for(int k=0; k<n; k++) threads.emplace_back([this] { run(); });
    }
    ~hz_copy_pool()
    {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_work.notify_all();
        for(auto& t : threads) t.join();
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for(;;)
        {
            cv_work.wait(lk, [this] { return stop || next < nparts; });
            if(stop) return;
            const int part = next++;
            const size_t lo = bytes*(size_t)part/(size_t)nparts, hi = bytes*(size_t)(part+1)/(size_t)nparts;
            unsigned char* d = dst; const unsigned char* s = src;
            lk.unlock();
            memcpy(d + lo, s + lo, hi - lo);
            lk.lock();
            if(--pending == 0) cv_done.notify_all();
        }
    }
    /* dst[0..bytes) = src[0..bytes), split over the pool; returns when done */
    void copy(unsigned char* d, const unsigned char* s, size_t n)
    {
        std::lock_guard<std::mutex> one(busy);
        std::unique_lock<std::mutex> lk(m);
        dst = d; src = s; bytes = n;
        nparts = (int)threads.size(); if((size_t)nparts > n/65536 + 1) nparts = (int)(n/65536 + 1);
        next = 0; pending = nparts;
        cv_work.notify_all();
        cv_done.wait(lk, [this] { return pending == 0; });
        nparts = 0;
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
        int n = 12;
        const char* e = getenv("HZ_COPY_THREADS");
        if(e && atoi(e) > 0) n = atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if(hw && (unsigned)n > hw) n = (int)hw;
        pool = new hz_copy_pool(n);
    }
    return pool;
}

static int ensure_staging(hz_dev_t* d)
{
    if(d->cstream) return 0;
    HZ_CHECK(hipStreamCreateWithFlags(&d->cstream, hipStreamNonBlocking));
    HZ_CHECK(hipEventCreateWithFlags(&d->ev_resolved, hipEventDisableTiming));
    for(int k=0; k<HZ_STAGE_SLOTS; k++)
    {
        HZ_CHECK(hipHostMalloc((void**)&d->h_stage[k], HZ_STAGE_BYTES, hipHostMallocDefault));
        HZ_CHECK(hipEventCreateWithFlags(&d->ev_stage[k], hipEventDisableTiming));
    }
    return 0;
}

/* the device buffers of the last conversion -> the caller's host buffers */
static int copy_out(hz_dev_t* d, int nbuf, unsigned char* const* dst, const unsigned char* const* src, const size_t* bytes)
{
    if(ensure_staging(d) != 0) return -1;
    hz_copy_pool* pool = copy_pool();
    HZ_CHECK(hipEventRecord(d->ev_resolved, d->rstream));
    HZ_CHECK(hipStreamWaitEvent(d->cstream, d->ev_resolved, 0));
    /* the chunks of all buffers, in order */
    struct chunk_t { unsigned char* dst; const unsigned char* src; size_t n; };
    std::vector<chunk_t> chunks;
    for(int b=0; b<nbuf; b++)
        for(size_t off=0; off<bytes[b]; off+=HZ_STAGE_BYTES)
            chunks.push_back({ dst[b] + off, src[b] + off, bytes[b] - off < HZ_STAGE_BYTES ? bytes[b] - off : HZ_STAGE_BYTES });
    const size_t nc = chunks.size();
    size_t issued = 0;
    for(size_t k=0; k<nc; k++)
    {
        /* keep the copy engine HZ_STAGE_SLOTS chunks ahead of the host threads; slot
         * k % SLOTS is free again once chunk k - SLOTS has been moved out (done below, in order) */
        for(; issued < nc && issued < k + HZ_STAGE_SLOTS; issued++)
        {
            const int slot = (int)(issued % HZ_STAGE_SLOTS);
            HZ_CHECK(hipMemcpyAsync(d->h_stage[slot], chunks[issued].src, chunks[issued].n, hipMemcpyDeviceToHost, d->cstream));
            HZ_CHECK(hipEventRecord(d->ev_stage[slot], d->cstream));
        }
        const int slot = (int)(k % HZ_STAGE_SLOTS);
        HZ_CHECK(hipEventSynchronize(d->ev_stage[slot]));
        pool->copy(chunks[k].dst, d->h_stage[slot], chunks[k].n);
    }
    return 0;
}

extern "C" int hz_hip_resolve_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                      unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    if(ensure_out_buffers(d, bgr != NULL, ranges != NULL, index != NULL, z24 != NULL) != 0) return -1;
    if(hz_hip_resolve(d, view, tanel,
                      bgr ? d->d_bgr : NULL, ranges ? d->d_ranges : NULL,
                      index ? d->d_index : NULL, z24 ? d->d_z24 : NULL) != 0) return -1;
    const size_t npix = (size_t)(d->col1 - d->col0)*d->H;
    unsigned char* dst[4]; const unsigned char* src[4]; size_t bytes[4];
    int nbuf = 0;
    if(bgr)    { dst[nbuf] = bgr;                    src[nbuf] = d->d_bgr;                            bytes[nbuf++] = npix*3; }
    if(ranges) { dst[nbuf] = (unsigned char*)ranges; src[nbuf] = (const unsigned char*)d->d_ranges;   bytes[nbuf++] = npix*sizeof(float); }
    if(index)  { dst[nbuf] = (unsigned char*)index;  src[nbuf] = (const unsigned char*)d->d_index;    bytes[nbuf++] = npix*sizeof(int32_t); }
    if(z24)    { dst[nbuf] = (unsigned char*)z24;    src[nbuf] = (const unsigned char*)d->d_z24;      bytes[nbuf++] = npix*sizeof(uint32_t); }
    const char* plain = getenv("HZ_PLAIN_COPY");        /* diagnostics: hipMemcpy into the caller's memory as it is */
    if(plain && atoi(plain) != 0)
    {
        for(int b=0; b<nbuf; b++) HZ_CHECK(hipMemcpyAsync(dst[b], src[b], bytes[b], hipMemcpyDeviceToHost, d->rstream));
        HZ_CHECK(hipStreamSynchronize(d->rstream));
        return 0;
    }
    return copy_out(d, nbuf, dst, src, bytes);
}

extern "C" int hz_hip_read_depth(hz_dev_t* d, int x, int y, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    if(fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
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

/* ------------------------------------------------------------------------ */
/* consumers of the range image (annotator passes)                           */

/* range of framebuffer word `key` in GL row `row`, as k_resolve computes it */
__device__ static inline float hz_range_of(unsigned long long key, float tanel_row, float znear, float zfar)
{
    const uint32_t zi = (uint32_t)(key >> 40);
    if(zi == HZ_Z24_MAX) return -1.0f;
    const float depth = (float)((double)zi * (1.0/16777215.0));
    const float len   = depth * (zfar-znear) + znear;
    const float zt    = tanel_row * len;
    return (float)sqrt((double)len*(double)len + (double)zt*(double)zt);
}

/* reference horizonator_unproject (horizonator-lib.c:1157-1213), range_enh given */
__device__ static inline void hz_unproject_enh(float* lat, float* lon, int x, int y, double range_enh,
                                               double lat_viewer, double cos_lat_viewer, double lon_viewer,
                                               double az_deg0, double az_deg1, int width, int height)
{
    const float Rearth = 6371000.0;
    float az_ndc = ((float)x + 0.5f) / (float)width * 2.f - 1.f;
    float az     = (az_ndc * (az_deg1-az_deg0) / 2.f + (az_deg1+az_deg0)/2.f) * M_PI/180.0f;
    double aspect = (double)width / (double)height;
    double el_ndc = ((double)y + 0.5) / (double)height * 2. - 1.;
    double el     = el_ndc * (az_deg1-az_deg0) / 2. / aspect * M_PI/180.0;
    double range_en = cos(el) * range_enh;
    float e = range_en * sinf(az);
    float n = range_en * cosf(az);
    *lon = lon_viewer + e / Rearth / M_PI * 180. / cos_lat_viewer;
    *lat = lat_viewer + n / Rearth / M_PI * 180.;
}

__global__ __launch_bounds__(256)
void k_link_cells(const unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
                  float* __restrict__ lat, float* __restrict__ lon,
                  int W, int H, int cell_w, int cell_h, int nx, int ny,
                  float znear, float zfar, double viewer_lat, double cos_viewer_lat, double viewer_lon,
                  double az_deg0, double az_deg1)
{
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    if(c >= nx*ny) return;
    const int cy = c / nx, cx = c - cy*nx;
    const int x = cx*cell_w, y = cy*cell_h;            /* image pixel, y = 0 top */
    const int row = H-1-y;
    const float range = hz_range_of(fb[(size_t)row*W + x], tanel[row], znear, zfar);
    float la = __builtin_nanf(""), lo = __builtin_nanf("");
    if(range > 0.0f)                                    /* reference annotator.c:236-238 */
        hz_unproject_enh(&la, &lo, x + cell_w/2, y + cell_h/2, (double)range,
                         viewer_lat, cos_viewer_lat, viewer_lon, az_deg0, az_deg1, W, H);
    lat[c] = la; lon[c] = lo;
}

/* reference horizonator-lib.c:1053-1095 */
__device__ static inline double hz_unwrap_d(double x, double near)
{
    const double d = (x - near) / (2.*M_PI);
    return (d - round(d)) * 2.*M_PI + near;
}

__global__ __launch_bounds__(256)
void k_poi(const unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
           const hz_poi_t* __restrict__ pois, int npois,
           unsigned char* __restrict__ visible, float* __restrict__ label_x, float* __restrict__ label_y,
           int W, int H, int height_out, float znear, float zfar,
           double lat_viewer, double cos_lat_viewer, double lon_viewer, double ele_viewer,
           double az_rad0, double az_rad1)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if(k >= npois) return;
    visible[k] = 0; label_x[k] = 0.f; label_y[k] = 0.f;

    /* reference horizonator_project (horizonator-lib.c:1097-1155) */
    const float Rearth = 6371000.0;
    const double dlat = ((double)pois[k].lat - lat_viewer)*M_PI/180;
    const double dlon = ((double)pois[k].lon - lon_viewer)*M_PI/180;
    const double east  = dlon * Rearth * cos_lat_viewer;
    const double north = dlat * Rearth;
    const double d2    = east*east + north*north;
    double a1 = hz_unwrap_d(az_rad1-az_rad0, M_PI) + az_rad0;
    const double center = (az_rad0 + a1)/2.;
    const double az = hz_unwrap_d(atan2(east, north), center);
    const double kk = 2.0 / (a1 - az_rad0);
    const double az_ndc = (az - center) * kk;
    if(!(-1. <= az_ndc && az_ndc <= 1.)) return;
    const double cx = (az_ndc + 1.)/2.*W - 0.5;
    const double h = (double)pois[k].ele_m - ele_viewer;
    const double d_ne = sqrt(d2);
    const double range_have = sqrt(d2 + h*h);
    const double aspect = (double)W / (double)H;
    const double el_ndc = atan2(h, d_ne) * aspect * kk;
    if(!(-1. <= el_ndc && el_ndc <= 1.)) return;
    const double cy = (-el_ndc + 1.)/2.*H - 0.5;

    /* reference annotator.c:297-347 */
    if(range_have < 500.0 || range_have > 100000.0) return;
    int    fuzz_nearest = 0;
    double err_nearest  = 1.7976931348623157e308;
    const int xi = (int)round(cx), yi = (int)round(cy);
    for(int fuzz = -6; fuzz < 6; fuzz++)
    {
        if(cy + (double)fuzz < 0) continue;
        if(cy + (double)fuzz >= height_out) break;
        const int y = yi + fuzz;
        if(y < 0 || y >= H || xi < 0 || xi >= W) continue;   /* the reference reads out of bounds here */
        const int row = H-1-y;
        const float range = hz_range_of(fb[(size_t)row*W + xi], tanel[row], znear, zfar);
        if(range <= 0.0f) continue;
        const double err = fabs(range_have - (double)range);
        if(err < err_nearest) { err_nearest = err; fuzz_nearest = fuzz; }
        else break;
    }
    if(err_nearest < 500.)
    {
        visible[k] = 1;
        label_x[k] = (float)cx;
        label_y[k] = (float)(cy + (float)fuzz_nearest);
    }
}

extern "C" int hz_hip_link_cells(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                 double viewer_lat, double viewer_lon,
                                 int cell_w, int cell_h, int cut_off_bottom_px,
                                 int nx, int ny, float* lat, float* lon)
{
    HZ_ON_DEVICE(d);
    if(fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    if(d->col0 != 0 || d->col1 != d->W || cell_w <= 0 || cell_h <= 0 || nx <= 0 || ny <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells: needs a full-width context and positive sizes");
        return -1;
    }
    (void)cut_off_bottom_px;
    if(upload_tanel(d, tanel) != 0) return -1;
    float *d_lat = NULL, *d_lon = NULL;
    const size_t n = (size_t)nx*ny;
    HZ_CHECK(hipMalloc(&d_lat, n*sizeof(float)));
    HZ_CHECK(hipMalloc(&d_lon, n*sizeof(float)));
    hipLaunchKernelGGL(k_link_cells, dim3((unsigned)((n + 255)/256)), dim3(256), 0, d->stream,
                       (const unsigned long long*)d->d_fb, (const float*)d->d_tanel, d_lat, d_lon,
                       d->W, d->H, cell_w, cell_h, nx, ny, view->znear, view->zfar,
                       viewer_lat, cos(viewer_lat * M_PI/180.), viewer_lon,
                       (double)view->az_deg0, (double)view->az_deg1);
    int rc = hipGetLastError() == hipSuccess ? 0 : -1;
    if(rc == 0 && hipMemcpyAsync(lat, d_lat, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(lon, d_lon, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(d_lat); (void)hipFree(d_lon);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells failed");
    return rc;
}

extern "C" int hz_hip_poi_visibility(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     double viewer_lat, double viewer_lon, double viewer_ele_m,
                                     int cut_off_bottom_px,
                                     const hz_poi_t* pois, int npois,
                                     unsigned char* visible, float* label_x, float* label_y)
{
    HZ_ON_DEVICE(d);
    if(fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    if(d->col0 != 0 || d->col1 != d->W || npois < 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility: needs a full-width context");
        return -1;
    }
    if(npois == 0) return 0;
    if(upload_tanel(d, tanel) != 0) return -1;
    hz_poi_t* d_pois = NULL; unsigned char* d_vis = NULL; float *d_x = NULL, *d_y = NULL;
    HZ_CHECK(hipMalloc(&d_pois, (size_t)npois*sizeof(hz_poi_t)));
    HZ_CHECK(hipMalloc(&d_vis, (size_t)npois));
    HZ_CHECK(hipMalloc(&d_x, (size_t)npois*sizeof(float)));
    HZ_CHECK(hipMalloc(&d_y, (size_t)npois*sizeof(float)));
    int rc = 0;
    if(hipMemcpyAsync(d_pois, pois, (size_t)npois*sizeof(hz_poi_t), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0)
    {
        hipLaunchKernelGGL(k_poi, dim3((unsigned)((npois + 255)/256)), dim3(256), 0, d->stream,
                           (const unsigned long long*)d->d_fb, (const float*)d->d_tanel,
                           (const hz_poi_t*)d_pois, npois, d_vis, d_x, d_y,
                           d->W, d->H, d->H - cut_off_bottom_px, view->znear, view->zfar,
                           viewer_lat, cos(viewer_lat * M_PI/180.), viewer_lon, viewer_ele_m,
                           (double)view->az_deg0 * M_PI/180., (double)view->az_deg1 * M_PI/180.);
        if(hipGetLastError() != hipSuccess) rc = -1;
    }
    if(rc == 0 && hipMemcpyAsync(visible, d_vis, (size_t)npois, hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_x, d_x, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_y, d_y, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(d_pois); (void)hipFree(d_vis); (void)hipFree(d_x); (void)hipFree(d_y);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility failed");
    return rc;
}

/* ------------------------------------------------------------------------ */
/* self-check of hz_fast.h: the abridged sequences against `/` and sqrtf       */

__device__ static inline unsigned long long hz_mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
/* a float with seeded mantissa and sign and an exponent in [elo, ehi] (biased) */
__device__ static inline float hz_seeded_float(unsigned long long r, int elo, int ehi)
{
    const uint32_t mant = (uint32_t)r & 0x7FFFFFu, sign = (uint32_t)(r >> 23) & 1u;
    const uint32_t ex = (uint32_t)elo + (uint32_t)((r >> 24) % (unsigned long long)(ehi - elo + 1));
    return __uint_as_float((sign << 31) | (ex << 23) | mant);
}

__global__ __launch_bounds__(256)
void k_check_fastmath(int what, unsigned long long seed, unsigned long long n, unsigned long long* mismatches, float* first_bad)
{
    unsigned long long bad = 0;
    for(unsigned long long k = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; k < n; k += (unsigned long long)gridDim.x*blockDim.x)
    {
        float a = 0.f, b = 0.f, want = 0.f, got = 0.f;
        bool in_range = true;
        if(what == 0)                       /* reciprocal: every bit pattern (k = the pattern), those in range checked */
        {
            b = __uint_as_float((uint32_t)k);
            in_range = hzf_in_range(b);
            if(in_range) { want = 1.0f / b; got = hzf_rcp(b); }
        }
        else if(what == 1)                  /* square root: every bit pattern from 2^-96 up to the largest finite float */
        {
            b = __uint_as_float((uint32_t)k);
            in_range = b >= 1.26217745e-29f && b <= 3.40282347e38f;
            if(in_range) { want = __builtin_sqrtf(b); got = hzf_sqrt(b); }
        }
        else if(what == 2)                  /* division: seeded pairs, numerator 0 or 2^-60..2^60, denominator 2^-31..2^31 */
        {
            const unsigned long long r1 = hz_mix64(seed + 2*k), r2 = hz_mix64(seed + 2*k + 1);
            a = hz_seeded_float(r1, 127-60, 127+60);
            b = hz_seeded_float(r2, 127-31, 127+31);
            /* the shapes the transform divides: any pair, min/max of a pair and 1, a shared mantissa, zero */
            if((r2 >> 61) == 1) { const float t = hz_abs(a); a = hz_min(t, 1.0f); b = hz_max(t, 1.0f); }
            if((r2 >> 61) == 2) a = __uint_as_float((__float_as_uint(a) & 0xFF800000u) | (__float_as_uint(b) & 0x7FFFFFu));
            if((r1 >> 60) == 0) a = 0.0f;            /* +0 */
            want = a / b; got = hzf_div(a, b);
        }
        else                                /* division by a per-draw constant through hzf_div_by: k = numerator pattern */
        {
            b = __uint_as_float((uint32_t)seed);
            a = __uint_as_float((uint32_t)k);
            const float aa = hz_abs(a);
            /* (+0 only: a negative zero would come out positive - see hz_fast.h on why none gets here) */
            in_range = __float_as_uint(a) == 0u || (aa >= 8.67361738e-19f && aa <= 1.15292150e18f);
            if(in_range) { want = a / b; got = hzf_div_by(a, b, hzf_refined_rcp(b)); }
        }
        if(in_range && __float_as_uint(want) != __float_as_uint(got))
        {
            if(bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = a; first_bad[1] = b; first_bad[2] = want; first_bad[3] = got; }
            bad++;
        }
    }
    if(bad) atomicAdd(mismatches, bad);
}

extern "C" int hz_hip_check_fastmath(int device, int what, unsigned long long seed, unsigned long long n,
                                     unsigned long long* mismatches, float* first_bad)
{
    hz_device_guard device_guard_(device);
    if(!device_guard_.ok) return -1;
    if(what < 0 || what > 3) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_check_fastmath: what = %d", what); return -1; }
    if(what == 0 || what == 1 || what == 3) n = 1ull << 32;
    unsigned long long* d_bad = NULL; float* d_first = NULL;
    HZ_CHECK(hipMalloc(&d_bad, 2*sizeof(unsigned long long)));
    HZ_CHECK(hipMalloc(&d_first, 4*sizeof(float)));
    HZ_CHECK(hipMemset(d_bad, 0, 2*sizeof(unsigned long long)));
    HZ_CHECK(hipMemset(d_first, 0, 4*sizeof(float)));
    hipLaunchKernelGGL(k_check_fastmath, dim3(256*32), dim3(256), 0, 0, what, seed, n, d_bad, d_first);
    HZ_CHECK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HZ_CHECK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
    if(first_bad) HZ_CHECK(hipMemcpy(first_bad, d_first, 4*sizeof(float), hipMemcpyDeviceToHost));
    *mismatches = h[0];
    (void)hipFree(d_bad); (void)hipFree(d_first);
    return 0;
}

/* diagnostics: the large-triangle queue of the last draw (set 0: its only or
 * second round, set 1: the first round of a two-round draw): counters[6] and,
 * for the first min(max_rec, counters[0]) records, px0 py0 bw bh + the three
 * edge vectors dx, dy in 1/256 pixel (10 int32 each) */
extern "C" int hz_hip_debug_bigqueue(hz_dev_t* d, int set, unsigned int* counters, int max_rec, int32_t* recs)
{
    HZ_ON_DEVICE(d);
    if(hz_hip_sync(d) != 0) return -1;
    const int k = (set ? HZ_NFB : 0) + d->fbi;
    HZ_CHECK(hipMemcpy(counters, d->d_big_counters_s[k], HZ_NCOUNTERS*sizeof(unsigned int), hipMemcpyDeviceToHost));
    unsigned int n = counters[0] < d->bigrec_capacity ? counters[0] : d->bigrec_capacity;
    if((int)n > max_rec) n = (unsigned int)max_rec;
    if(n == 0 || recs == NULL) return 0;
    hz_bigrec_t* h = (hz_bigrec_t*)malloc((size_t)n*sizeof(hz_bigrec_t));
    if(!h) return -1;
    HZ_CHECK(hipMemcpy(h, d->d_bigrec_s[k], (size_t)n*sizeof(hz_bigrec_t), hipMemcpyDeviceToHost));
    for(unsigned int r=0; r<n; r++)
    {
        int32_t* o = recs + (size_t)r*10;
        o[0] = h[r].r.px0; o[1] = h[r].r.py0; o[2] = h[r].r.bw; o[3] = h[r].bh;
        for(int m=0; m<3; m++) { o[4+m] = h[r].r.e.dx[m]; o[7+m] = -h[r].r.e.ndy[m]; }
    }
    free(h);
    return 0;
}

extern "C" int hz_hip_sync(hz_dev_t* d)
{
    HZ_ON_DEVICE(d);
    HZ_CHECK(sync_all(d));
    return 0;
}

extern "C" int hz_hip_last_times(hz_dev_t* d, hz_times_t* t)
{
    memset(t, 0, sizeof(*t));
    if(!d->have_times) return -1;
    HZ_ON_DEVICE(d);
    HZ_CHECK(sync_all(d));
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
