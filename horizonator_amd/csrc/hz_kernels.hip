/* hz_kernels.hip - the DEM -> panorama render path as HIP kernels for gfx950, and one launcher per kernel (hz_launch.h)
 * through which the host code (hz_context / hz_draw / hz_convert / hz_hostpath / hz_ingest .cpp, compiled by g++) reaches them.
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
 *
 * One translation unit: the device code lives in hz_k_common.h (parameters,
 * records, helpers, k_clip), hz_k_hiz.h (coarse depth for zoomed views),
 * hz_k_scatter.h (k_scatter, k_big), hz_k_tile.h (the tile-binned option),
 * hz_k_march.h (k_march, k_mid), hz_k_resolve.h (conversions, strips, blobs for host memory),
 * hz_k_tell.h (what the host is told about those blobs), hz_k_tex.h (textured resolve); this file
 * holds k_ingest, k_polar_fill, the annotator's kernels and the launchers.
 */
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hz_hip.h"
#include "hz_types.h"
#include "hz_fast.h"

#include "hz_k_common.h"
#include "hz_k_hiz.h"
#include "hz_k_scatter.h"
#include "hz_k_tile.h"
#include "hz_k_march.h"
#include "hz_k_resolve.h"
#include "hz_k_tell.h"
#include "hz_k_tex.h"

/* ------------------------------------------------------------------------ */
/* device-side DEM ingest ("next" row N3)                                    */

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

/* reference horizonator_unproject (horizonator-lib.c:1157-1213), range_enh given.  The transcendental functions in it
 * depend on the pixel's column (sinf / cosf of its azimuth) and on its row (cos of its elevation) only: the host
 * evaluates them with the C library the reference itself calls (hz_host.c: horizonator_amd_link_cells) and the kernel
 * is left with IEEE multiplications and divisions - the same bits as the reference's loop, not "within a tolerance". */
__device__ static inline void hz_unproject_enh(float* lat, float* lon, double range_enh,
                                               float sin_az, float cos_az, double cos_el,
                                               double lat_viewer, double cos_lat_viewer, double lon_viewer)
{
    const float Rearth = 6371000.0;
    double range_en = cos_el * range_enh;
    float e = range_en * sin_az;
    float n = range_en * cos_az;
    *lon = lon_viewer + e / Rearth / M_PI * 180. / cos_lat_viewer;
    *lat = lat_viewer + n / Rearth / M_PI * 180.;
}

__global__ __launch_bounds__(256)
void k_link_cells(const unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
                  const float* __restrict__ sin_az, const float* __restrict__ cos_az, const double* __restrict__ cos_el,
                  float* __restrict__ lat, float* __restrict__ lon,
                  int W, int H, int cell_w, int cell_h, int nx, int ny,
                  float znear, float zfar, double viewer_lat, double cos_viewer_lat, double viewer_lon)
{
    const int c = blockIdx.x*blockDim.x + threadIdx.x;
    if(c >= nx*ny) return;
    const int cy = c / nx, cx = c - cy*nx;
    const int x = cx*cell_w, y = cy*cell_h;            /* image pixel, y = 0 top */
    const int row = H-1-y;
    const float range = hz_range_of(fb[(size_t)row*W + x], tanel[row], znear, zfar);
    float la = __builtin_nanf(""), lo = __builtin_nanf("");
    if(range > 0.0f)                                    /* reference annotator.c:236-238 */
        hz_unproject_enh(&la, &lo, (double)range, sin_az[cx], cos_az[cx], cos_el[cy],
                         viewer_lat, cos_viewer_lat, viewer_lon);
    lat[c] = la; lon[c] = lo;
}

/* reference annotator.c:297-347: the search of the range image around a projected point of interest.  The projection
 * itself (reference horizonator_project, horizonator-lib.c:1097-1155: atan2, sqrt, in double) is the host's, made with
 * the C library the reference calls (hz_host.c: horizonator_amd_poi_visibility); proj[k].range < 0: outside the view. */
__global__ __launch_bounds__(256)
void k_poi(const unsigned long long* __restrict__ fb, const float* __restrict__ tanel,
           const hz_poi_proj_t* __restrict__ proj, int npois,
           unsigned char* __restrict__ visible, float* __restrict__ label_x, float* __restrict__ label_y,
           int W, int H, int height_out, float znear, float zfar)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if(k >= npois) return;
    visible[k] = 0; label_x[k] = 0.f; label_y[k] = 0.f;
    const double cx = proj[k].x, cy = proj[k].y, range_have = proj[k].range;
    if(!(range_have >= 0.0)) return;
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

/* ------------------------------------------------------------------------ */
/* the vertex cache: the view-independent half of every vertex's transform    */

/* one thread per vertex, a wave = 64 consecutive columns of a row: q[j][i] = hz_polar_en() of vertex (i, j) for the viewer of u
 * (the unabridged sequences: the abridged ones of hz_fast.h give the same bits where they apply, and this kernel runs once
 * per viewpoint) */
__global__ __launch_bounds__(256)
void k_polar_fill(const int16_t* __restrict__ mosaic, hz_polar_t* __restrict__ q, int N, hz_xform_t u)
{
    const int i = (int)(blockIdx.x*blockDim.x + threadIdx.x);
    if(i >= N) return;
    const float e = hz_east(&u, (float)i);
    for(int j = (int)blockIdx.y; j < N; j += (int)gridDim.y)
        q[(size_t)j*N + i] = hz_polar_en(&u, e, hz_north(&u, (float)j), (float)mosaic[(size_t)j*N + i]);
}

/* ------------------------------------------------------------------------ */
/* the launchers (hz_launch.h): what the host translation units call          */

#include "hz_launch.h"

void hzk_clip(bool wave_items, dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, hz_params_t p)
{
    if(wave_items) hipLaunchKernelGGL(k_clip<true>, grid, block, 0, stream, mosaic, fb, q, p);
    else hipLaunchKernelGGL(k_clip<false>, grid, block, 0, stream, mosaic, fb, q, p);
}

void hzk_hiz(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const unsigned char* touched, int seg_stride, int SW, int H, hz_hiz_t hz, unsigned int nunits)
{
    hipLaunchKernelGGL(k_hiz, grid, block, 0, stream, fb, touched, seg_stride, SW, H, hz, nunits);
}

void hzk_mid(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_rec_t* midrec, const unsigned int* counters, unsigned int midrec_capacity, hz_params_t p)
{
    hipLaunchKernelGGL(k_mid, grid, block, 0, stream, fb, midrec, counters, midrec_capacity, p);
}

void hzk_march(bool counters, bool hiz, bool vcache, dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, mr_zones_t zn, hz_params_t p)
{
    /* (the instance appends through as many queue counters as the draw's other kernels expect: hz_types.h, HZ_QSHARDS) */
    const bool shards = p.qshards_log2 != 0;
#ifdef HZ_SELFTEST
    if(counters)
    {
        if(shards) hipLaunchKernelGGL((k_march<true, true, false, true>), grid, block, 0, stream, mosaic, fb, q, zn, p);
        else hipLaunchKernelGGL((k_march<true, true, false, false>), grid, block, 0, stream, mosaic, fb, q, zn, p);
        return;
    }
#endif
    (void)counters;
    #define HZK_MARCH(H, V, S) hipLaunchKernelGGL((k_march<false, H, V, S>), grid, block, 0, stream, mosaic, fb, q, zn, p)
    if(shards)
    {
        if(vcache) { if(hiz) HZK_MARCH(true, true, true); else HZK_MARCH(false, true, true); }
        else       { if(hiz) HZK_MARCH(true, false, true); else HZK_MARCH(false, false, true); }
    }
    else
    {
        if(vcache) { if(hiz) HZK_MARCH(true, true, false); else HZK_MARCH(false, true, false); }
        else       { if(hiz) HZK_MARCH(true, false, false); else HZK_MARCH(false, false, false); }
    }
    #undef HZK_MARCH
}

void hzk_polar_fill(dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, hz_polar_t* q, int N, hz_xform_t u)
{
    hipLaunchKernelGGL(k_polar_fill, grid, block, 0, stream, mosaic, q, N, u);
}

void hzk_resolve(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const float* tanel, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int SW, int H, float znear, float zfar, unsigned int* qa, unsigned int* qb)
{
    if(clear) hipLaunchKernelGGL(k_resolve<true>, grid, block, 0, stream, fb, tanel, bgr, ranges, index, z24, SW, H, znear, zfar, qa, qb);
    else hipLaunchKernelGGL(k_resolve<false>, grid, block, 0, stream, fb, tanel, bgr, ranges, index, z24, SW, H, znear, zfar, qa, qb);
}

void hzk_resolve4(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const float* tanel, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int SW, int H, float znear, float zfar, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb, int yo0, int yo1, int nt)
{
    if(clear) hipLaunchKernelGGL(k_resolve4<true>, grid, block, 0, stream, fb, tanel, bgr, ranges, index, z24, SW, H, znear, zfar, touched, seg_stride, qa, qb, yo0, yo1, nt);
    else hipLaunchKernelGGL(k_resolve4<false>, grid, block, 0, stream, fb, tanel, bgr, ranges, index, z24, SW, H, znear, zfar, touched, seg_stride, qa, qb, yo0, yo1, nt);
}

void hzk_pack_host(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, hz_hostpack_t o, int SW, int H, int col_off, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb)
{
    if(clear) hipLaunchKernelGGL(k_pack_host<true>, grid, block, 0, stream, fb, o, SW, H, col_off, touched, seg_stride, qa, qb);
    else hipLaunchKernelGGL(k_pack_host<false>, grid, block, 0, stream, fb, o, SW, H, col_off, touched, seg_stride, qa, qb);
}

void hzk_tell(hipStream_t stream, hz_tell_t s)
{
    hipLaunchKernelGGL(k_tell, dim3(1), dim3(TELL_THREADS), 0, stream, s);
}

void hzk_pack(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, uint32_t* packed, int SW, int H, unsigned int* qa, unsigned int* qb)
{
    if(clear) hipLaunchKernelGGL(k_pack<true>, grid, block, 0, stream, fb, packed, SW, H, qa, qb);
    else hipLaunchKernelGGL(k_pack<false>, grid, block, 0, stream, fb, packed, SW, H, qa, qb);
}

void hzk_resolve_packed(dim3 grid, dim3 block, hipStream_t stream, const uint32_t* packed, int stride, int ncols, const float* tanel, unsigned char* bgr, float* ranges, int out_W, int out_col0, int H, float znear, float zfar)
{
    hipLaunchKernelGGL(k_resolve_packed, grid, block, 0, stream, packed, stride, ncols, tanel, bgr, ranges, out_W, out_col0, H, znear, zfar);
}

void hzk_pack_sparse(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, uint32_t* out, int SW, int H, int mask_stride, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb)
{
    if(clear) hipLaunchKernelGGL(k_pack_sparse<true>, grid, block, 0, stream, fb, out, SW, H, mask_stride, touched, seg_stride, qa, qb);
    else hipLaunchKernelGGL(k_pack_sparse<false>, grid, block, 0, stream, fb, out, SW, H, mask_stride, touched, seg_stride, qa, qb);
}

void hzk_resolve_sparse(dim3 grid, dim3 block, hipStream_t stream, hz_strips_t st, int mask_stride, const float* tanel, unsigned char* bgr, float* ranges, int out_W, int H, float znear, float zfar)
{
    hipLaunchKernelGGL(k_resolve_sparse, grid, block, 0, stream, st, mask_stride, tanel, bgr, ranges, out_W, H, znear, zfar);
}

void hzk_scatter(dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, hz_params_t p)
{
    hipLaunchKernelGGL(k_scatter, grid, block, 0, stream, mosaic, fb, q, p);
}

void hzk_big(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_bigrec_t* bigrec, const hz_bigitem_t* bigitem, const unsigned int* big_counters, unsigned int bigrec_capacity, unsigned int bigitem_capacity, hz_params_t p, const unsigned int* tile_state, unsigned int* report)
{
    if(p.qshards_log2 != 0) hipLaunchKernelGGL(k_big<true>, grid, block, 0, stream, fb, bigrec, bigitem, big_counters, bigrec_capacity, bigitem_capacity, p, tile_state, report);
    else hipLaunchKernelGGL(k_big<false>, grid, block, 0, stream, fb, bigrec, bigitem, big_counters, bigrec_capacity, bigitem_capacity, p, tile_state, report);
}

void hzk_shade_tex(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const int16_t* mosaic, const uint32_t* texels, hz_texparams_t tp, unsigned char* bgr, hz_params_t p)
{
    hipLaunchKernelGGL(k_shade_tex, grid, block, 0, stream, fb, mosaic, texels, tp, bgr, p);
}

void hzk_tile_bin(dim3 grid, dim3 block, hipStream_t stream, const hz_bigrec_t* bigrec, const hz_bigitem_t* bigitem, const unsigned int* big_counters, unsigned int bigrec_capacity, tl_bins_t tb, hz_params_t p)
{
    hipLaunchKernelGGL(k_tile_bin, grid, block, 0, stream, bigrec, bigitem, big_counters, bigrec_capacity, tb, p);
}

void hzk_tile_raster(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_bigrec_t* bigrec, tl_bins_t tb, hz_params_t p)
{
    hipLaunchKernelGGL(k_tile_raster, grid, block, 0, stream, fb, bigrec, tb, p);
}

void hzk_ingest(dim3 grid, dim3 block, hipStream_t stream, const unsigned char* const* tiles, int16_t* mosaic, int N, int ntx, int nty, int cpd, int oc_x, int oc_y)
{
    hipLaunchKernelGGL(k_ingest, grid, block, 0, stream, tiles, mosaic, N, ntx, nty, cpd, oc_x, oc_y);
}

void hzk_link_cells(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const float* tanel, const float* sin_az, const float* cos_az, const double* cos_el, float* lat, float* lon, int W, int H, int cell_w, int cell_h, int nx, int ny, float znear, float zfar, double viewer_lat, double cos_viewer_lat, double viewer_lon)
{
    hipLaunchKernelGGL(k_link_cells, grid, block, 0, stream, fb, tanel, sin_az, cos_az, cos_el, lat, lon, W, H, cell_w, cell_h, nx, ny, znear, zfar, viewer_lat, cos_viewer_lat, viewer_lon);
}

void hzk_poi(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const float* tanel, const hz_poi_proj_t* proj, int npois, unsigned char* visible, float* label_x, float* label_y, int W, int H, int height_out, float znear, float zfar)
{
    hipLaunchKernelGGL(k_poi, grid, block, 0, stream, fb, tanel, proj, npois, visible, label_x, label_y, W, H, height_out, znear, zfar);
}

#ifdef HZ_SELFTEST
#include "hz_dev.h"
#include "hz_selftest.h"         /* include/: the declarations */
#include "hz_selftest_impl.h"
#endif
