/* hz_kernels.hip - the DEM -> panorama render path as HIP kernels for gfx950,
 * and the C-ABI (include/hz_hip.h) through which the C host drives them.
 *
 * What runs here is what the reference hands to OpenGL:
 *   vertex.glsl:111-162     per-vertex transform            -> hz_transform()
 *   horizonator-lib.c:487-512 index buffer (2 tris per cell) -> implicit from (i,j,t)
 *   geometry.glsl:21-27     wide/seam triangle discard      -> hz_tri_setup()
 *   fixed function          cull, raster, depth test        -> hz_raster.h
 *   fragment.glsl:15-16     colour = (red,0,0)              -> packed red8
 *   horizonator-lib.c:936-1048 readback, flip, depth->range -> k_resolve
 *
 * HBM layout
 *   mosaic  int16 [N][N], row j = constant latitude (south first), i fastest
 *   fb      uint64 [H][SW]  GL row order (row 0 = bottom), SW = sector width
 *           word = z24<<40 | primitive<<8 | red8, cleared to all ones
 */
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hz_hip.h"
#include "hz_raster.h"

/* ------------------------------------------------------------------------ */
/* errors                                                                    */

static char g_last_error[512];
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
    int   dry;              /* timing experiments: 1 = compute fragments, store nothing; 2 = no read-before-atomic */
};

/* work item of the cooperative pass: one 32-row band of one large triangle */
struct hz_bigitem_t { uint32_t prim; int32_t band; };

#define HZ_BIG_BAND_ROWS   32
#define HZ_BIG_THRESHOLD   512      /* bbox pixels above which a triangle is deferred */

/* ------------------------------------------------------------------------ */
/* device helpers                                                            */

__device__ static inline void hz_emit(unsigned long long* fb, const hz_params_t& p,
                                      const hz_tri_t& t, uint32_t prim, int px, int py)
{
    if(!hz_tri_covers(&t, px, py)) return;
    uint32_t zi, r8;
    if(!hz_tri_fragment(&t, px, py, &zi, &r8)) return;
    const unsigned long long key = hz_pack(zi, prim, r8);
    unsigned long long* dst = &fb[(size_t)py*p.SW + (px - p.col0)];
    if(p.dry == 1) { if(key == 0x0123456789ull) *dst = key; return; }
    if(p.dry == 2) { atomicMin(dst, key); return; }
    /* plain read first: most fragments of far terrain lose against what is
     * already there, and a stale (larger) value only costs the atomic */
    if(key < __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMin(dst, key);
}

/* the three window vertices of triangle t (0|1) of cell (i,j), in the order
 * of the reference's index buffer (reference horizonator-lib.c:500-506):
 *   t=0: (j,i) (j+1,i+1) (j+1,i)      t=1: (j,i) (j,i+1) (j+1,i+1) */
__device__ static inline hz_wvert_t hz_vertex_at(const hz_params_t& p, const int16_t* mosaic, int i, int j)
{
    const float z = (float)mosaic[(size_t)j*p.N + i];
    return hz_to_window(hz_transform(&p.u, (float)i, (float)j, z), p.halfW, p.halfH);
}

/* ------------------------------------------------------------------------ */
/* scatter rasteriser: one thread per DEM cell                               */

#define SC_CX 64            /* cells per block along i (one wave = one row)  */
#define SC_CY 4             /* cells per block along j                       */
#define SC_VX (SC_CX+1)
#define SC_VY (SC_CY+1)

__global__ __launch_bounds__(SC_CX*SC_CY)
void k_scatter(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb,
               hz_bigitem_t* __restrict__ big, unsigned int* __restrict__ big_count,
               unsigned int big_capacity, hz_params_t p)
{
    /* 2-D LDS staging of the block's (SC_CX+1) x (SC_CY+1) vertices: each
     * vertex is transformed once and shared by the up to 6 triangles around it */
    __shared__ float s_xn [SC_VY][SC_VX];
    __shared__ float s_fx [SC_VY][SC_VX];
    __shared__ float s_fy [SC_VY][SC_VX];
    __shared__ float s_zw [SC_VY][SC_VX];
    __shared__ float s_red[SC_VY][SC_VX];

    const int tid = threadIdx.x;
    const int i0  = blockIdx.x*SC_CX;
    const int j0  = blockIdx.y*SC_CY;

    int some_not_near = 0, some_not_far = 0;
    for(int v = tid; v < SC_VX*SC_VY; v += SC_CX*SC_CY)
    {
        const int vy = v / SC_VX, vx = v - vy*SC_VX;
        const int i = i0 + vx, j = j0 + vy;
        if(i < p.N && j < p.N)
        {
            const hz_wvert_t w = hz_vertex_at(p, mosaic, i, j);
            s_xn [vy][vx] = w.xn;
            s_fx [vy][vx] = w.fx;
            s_fy [vy][vx] = w.fy;
            s_zw [vy][vx] = w.zw;
            s_red[vy][vx] = w.red;
            some_not_near |= !(w.zw < 0.f);
            some_not_far  |= !(w.zw > 1.f);
        }
    }
    /* block-wide early out: every vertex in front of the near sphere, or
     * every vertex beyond the far one.  hz_tri_setup() drops exactly those
     * triangles anyway; this only saves the per-triangle work (with the
     * default zfar = 40 km most of a large mosaic goes this way). */
    some_not_near = __syncthreads_or(some_not_near);
    some_not_far  = __syncthreads_or(some_not_far);
    if(!some_not_near || !some_not_far) return;

    const int cx = tid & (SC_CX-1);
    const int cy = tid / SC_CX;
    const int i = i0 + cx, j = j0 + cy;
    if(i >= p.N-1 || j >= p.N-1) return;

    hz_wvert_t v00 = { s_xn[cy  ][cx  ], s_fx[cy  ][cx  ], s_fy[cy  ][cx  ], s_zw[cy  ][cx  ], s_red[cy  ][cx  ] };
    hz_wvert_t v10 = { s_xn[cy  ][cx+1], s_fx[cy  ][cx+1], s_fy[cy  ][cx+1], s_zw[cy  ][cx+1], s_red[cy  ][cx+1] };
    hz_wvert_t v01 = { s_xn[cy+1][cx  ], s_fx[cy+1][cx  ], s_fy[cy+1][cx  ], s_zw[cy+1][cx  ], s_red[cy+1][cx  ] };
    hz_wvert_t v11 = { s_xn[cy+1][cx+1], s_fx[cy+1][cx+1], s_fy[cy+1][cx+1], s_zw[cy+1][cx+1], s_red[cy+1][cx+1] };

    const uint32_t prim0 = (uint32_t)(((size_t)j*(p.N-1) + i)*2);

    #pragma unroll
    for(int t=0; t<2; t++)
    {
        hz_tri_t tri;
        const int ok = (t == 0)
            ? hz_tri_setup(&tri, v00, v11, v01, p.col0, p.col1-1, 0, p.H-1)
            : hz_tri_setup(&tri, v00, v10, v11, p.col0, p.col1-1, 0, p.H-1);
        if(!ok) continue;
        const uint32_t prim = prim0 + t;
        const int bw = tri.px1 - tri.px0 + 1;
        const int bh = tri.py1 - tri.py0 + 1;
        if((long long)bw*bh > HZ_BIG_THRESHOLD)
        {
            /* defer: one work item per band of rows */
            const int nbands = (bh + HZ_BIG_BAND_ROWS-1) / HZ_BIG_BAND_ROWS;
            const unsigned int at = atomicAdd(big_count, (unsigned int)nbands);
            if(at + nbands <= big_capacity)
            {
                for(int b=0; b<nbands; b++) { big[at+b].prim = prim; big[at+b].band = b; }
                continue;
            }
            /* list full: fall through and rasterise inline (slow, correct) */
        }
        for(int py = tri.py0; py <= tri.py1; py++)
            for(int px = tri.px0; px <= tri.px1; px++)
                hz_emit(fb, p, tri, prim, px, py);
    }
}

/* cooperative pass: a whole block walks one band of one large triangle */
__global__ __launch_bounds__(256)
void k_big(const int16_t* __restrict__ mosaic, unsigned long long* __restrict__ fb,
           const hz_bigitem_t* __restrict__ big, const unsigned int* __restrict__ big_count,
           unsigned int big_capacity, hz_params_t p)
{
    unsigned int n = *big_count;
    if(n > big_capacity) n = big_capacity;      /* overflowed items were drawn inline */
    for(unsigned int it = blockIdx.x; it < n; it += gridDim.x)
    {
        const uint32_t prim = big[it].prim;
        const int      band = big[it].band;
        const uint32_t cell = prim >> 1;
        const int t = prim & 1;
        const int j = cell / (uint32_t)(p.N-1);
        const int i = cell - (uint32_t)j*(uint32_t)(p.N-1);

        const hz_wvert_t v00 = hz_vertex_at(p, mosaic, i,   j  );
        const hz_wvert_t v11 = hz_vertex_at(p, mosaic, i+1, j+1);
        hz_tri_t tri;
        int ok;
        if(t == 0) ok = hz_tri_setup(&tri, v00, v11, hz_vertex_at(p, mosaic, i,   j+1), p.col0, p.col1-1, 0, p.H-1);
        else       ok = hz_tri_setup(&tri, v00, hz_vertex_at(p, mosaic, i+1, j  ), v11, p.col0, p.col1-1, 0, p.H-1);
        if(!ok) continue;

        const int py_lo = tri.py0 + band*HZ_BIG_BAND_ROWS;
        int       py_hi = py_lo + HZ_BIG_BAND_ROWS-1;
        if(py_hi > tri.py1) py_hi = tri.py1;
        const int bw = tri.px1 - tri.px0 + 1;
        const int npix = bw*(py_hi - py_lo + 1);
        for(int k = threadIdx.x; k < npix; k += blockDim.x)
        {
            const int ry = k / bw;
            hz_emit(fb, p, tri, prim, tri.px0 + (k - ry*bw), py_lo + ry);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* resolve: framebuffer words -> BGR8, range, primitive id, z24; flips rows  */

__global__ __launch_bounds__(256)
void k_resolve(const unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
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

/* ------------------------------------------------------------------------ */
/* host side of the C-ABI                                                    */

struct hz_dev
{
    int device;
    int N, W, H;
    int col0, col1;
    int raster;
    int profiling;

    hipStream_t stream;
    int16_t*            d_mosaic;
    unsigned long long* d_fb;           /* W*H words (sector uses a prefix) */
    hz_bigitem_t*       d_big;
    unsigned int*       d_big_count;
    unsigned int        big_capacity;
    float*              d_tanel;

    /* internal output buffers for *_to_host */
    unsigned char* d_bgr;
    float*         d_ranges;
    int32_t*       d_index;
    uint32_t*      d_z24;

    hipEvent_t ev[6];
    int        have_times;
    hz_times_t times;
};

extern "C" int hz_hip_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void hz_hip_destroy(hz_dev_t* d)
{
    if(!d) return;
    (void)hipSetDevice(d->device);
    if(d->stream) (void)hipStreamSynchronize(d->stream);
    (void)hipFree(d->d_mosaic);
    (void)hipFree(d->d_fb);
    (void)hipFree(d->d_big);
    (void)hipFree(d->d_big_count);
    (void)hipFree(d->d_tanel);
    (void)hipFree(d->d_bgr);
    (void)hipFree(d->d_ranges);
    (void)hipFree(d->d_index);
    (void)hipFree(d->d_z24);
    for(int k=0; k<6; k++) if(d->ev[k]) (void)hipEventDestroy(d->ev[k]);
    if(d->stream) (void)hipStreamDestroy(d->stream);
    free(d);
}

static int create_impl(hz_dev_t* d)
{
    HZ_CHECK(hipSetDevice(d->device));
    HZ_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HZ_CHECK(hipMalloc(&d->d_mosaic, (size_t)d->N*d->N*sizeof(int16_t)));
    HZ_CHECK(hipMalloc(&d->d_fb, (size_t)d->W*d->H*sizeof(unsigned long long)));
    d->big_capacity = 1u<<20;
    HZ_CHECK(hipMalloc(&d->d_big, (size_t)d->big_capacity*sizeof(hz_bigitem_t)));
    HZ_CHECK(hipMalloc(&d->d_big_count, sizeof(unsigned int)));
    HZ_CHECK(hipMalloc(&d->d_tanel, (size_t)d->H*sizeof(float)));
    for(int k=0; k<6; k++) HZ_CHECK(hipEventCreate(&d->ev[k]));
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
    HZ_CHECK(hipSetDevice(d->device));
    HZ_CHECK(hipMemcpyAsync(d->d_mosaic, mosaic, (size_t)d->N*d->N*sizeof(int16_t), hipMemcpyHostToDevice, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int hz_hip_download_mosaic(hz_dev_t* d, int16_t* mosaic)
{
    HZ_CHECK(hipSetDevice(d->device));
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
    HZ_CHECK(hipSetDevice(d->device));
    const int nt = ntx*nty;
    const size_t tile_bytes = (size_t)(cpd+1)*(cpd+1)*2;
    unsigned char** h_ptrs = (unsigned char**)calloc(nt, sizeof(*h_ptrs));
    unsigned char** d_ptrs = NULL;
    int rc = -1;
    if(!h_ptrs) return -1;
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
    if(which < HZ_RASTER_AUTO || which > HZ_RASTER_COLUMNS+2) return -1;   /* +1,+2: timing experiments */
    d->raster = which;
    return 0;
}

extern "C" int hz_hip_set_profiling(hz_dev_t* d, int on) { d->profiling = on; return 0; }
extern "C" void* hz_hip_stream(hz_dev_t* d) { return (void*)d->stream; }

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
    p.dry = d->raster > HZ_RASTER_COLUMNS ? d->raster - HZ_RASTER_COLUMNS : 0;
    return p;
}

extern "C" int hz_hip_draw(hz_dev_t* d, const hz_view_t* view)
{
    HZ_CHECK(hipSetDevice(d->device));
    const hz_params_t p = make_params(d, view);
    const bool prof = d->profiling != 0;

    if(prof) HZ_CHECK(hipEventRecord(d->ev[0], d->stream));
    /* glClear (reference horizonator-lib.c:896): depth = 1.0 -> all-ones word */
    HZ_CHECK(hipMemsetAsync(d->d_fb, 0xFF, (size_t)p.SW*p.H*sizeof(unsigned long long), d->stream));
    HZ_CHECK(hipMemsetAsync(d->d_big_count, 0, sizeof(unsigned int), d->stream));
    if(prof) HZ_CHECK(hipEventRecord(d->ev[1], d->stream));

    {
        dim3 grid((p.N-1 + SC_CX-1)/SC_CX, (p.N-1 + SC_CY-1)/SC_CY);
        hipLaunchKernelGGL(k_scatter, grid, dim3(SC_CX*SC_CY), 0, d->stream,
                           (const int16_t*)d->d_mosaic, d->d_fb, d->d_big, d->d_big_count, d->big_capacity, p);
        HZ_CHECK(hipGetLastError());
        if(prof) HZ_CHECK(hipEventRecord(d->ev[2], d->stream));
        hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, d->stream,
                           (const int16_t*)d->d_mosaic, d->d_fb, (const hz_bigitem_t*)d->d_big,
                           (const unsigned int*)d->d_big_count, d->big_capacity, p);
        HZ_CHECK(hipGetLastError());
        if(prof) HZ_CHECK(hipEventRecord(d->ev[3], d->stream));
    }
    d->have_times = prof ? 1 : 0;
    return 0;
}

extern "C" int hz_hip_resolve(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                              unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_CHECK(hipSetDevice(d->device));
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(ranges)
    {
        if(!tanel)
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve: ranges requested without a tanel table");
            return -1;
        }
        HZ_CHECK(hipMemcpyAsync(d->d_tanel, tanel, (size_t)d->H*sizeof(float), hipMemcpyHostToDevice, d->stream));
    }
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->stream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)nblocks), dim3(256), 0, d->stream,
                       (const unsigned long long*)d->d_fb, (const float*)d->d_tanel,
                       bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar);
    HZ_CHECK(hipGetLastError());
    if(prof)
    {
        HZ_CHECK(hipEventRecord(d->ev[5], d->stream));
        d->have_times = 2;
    }
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

extern "C" int hz_hip_resolve_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                      unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_CHECK(hipSetDevice(d->device));
    if(ensure_out_buffers(d, bgr != NULL, ranges != NULL, index != NULL, z24 != NULL) != 0) return -1;
    if(hz_hip_resolve(d, view, tanel,
                      bgr ? d->d_bgr : NULL, ranges ? d->d_ranges : NULL,
                      index ? d->d_index : NULL, z24 ? d->d_z24 : NULL) != 0) return -1;
    const size_t npix = (size_t)(d->col1 - d->col0)*d->H;
    if(bgr)    HZ_CHECK(hipMemcpyAsync(bgr,    d->d_bgr,    npix*3,                hipMemcpyDeviceToHost, d->stream));
    if(ranges) HZ_CHECK(hipMemcpyAsync(ranges, d->d_ranges, npix*sizeof(float),    hipMemcpyDeviceToHost, d->stream));
    if(index)  HZ_CHECK(hipMemcpyAsync(index,  d->d_index,  npix*sizeof(int32_t),  hipMemcpyDeviceToHost, d->stream));
    if(z24)    HZ_CHECK(hipMemcpyAsync(z24,    d->d_z24,    npix*sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int hz_hip_read_depth(hz_dev_t* d, int x, int y, uint32_t* z24)
{
    HZ_CHECK(hipSetDevice(d->device));
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

extern "C" int hz_hip_sync(hz_dev_t* d)
{
    HZ_CHECK(hipSetDevice(d->device));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    return 0;
}

extern "C" int hz_hip_last_times(hz_dev_t* d, hz_times_t* t)
{
    memset(t, 0, sizeof(*t));
    if(!d->have_times) return -1;
    HZ_CHECK(hipSetDevice(d->device));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    HZ_CHECK(hipEventElapsedTime(&t->clear_ms,  d->ev[0], d->ev[1]));
    HZ_CHECK(hipEventElapsedTime(&t->raster_ms, d->ev[1], d->ev[2]));
    HZ_CHECK(hipEventElapsedTime(&t->big_ms,    d->ev[2], d->ev[3]));
    if(d->have_times == 2)
    {
        HZ_CHECK(hipEventElapsedTime(&t->resolve_ms, d->ev[4], d->ev[5]));
        HZ_CHECK(hipEventElapsedTime(&t->total_ms,   d->ev[0], d->ev[5]));
    }
    else
        HZ_CHECK(hipEventElapsedTime(&t->total_ms, d->ev[0], d->ev[3]));
    return 0;
}
