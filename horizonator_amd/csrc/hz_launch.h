/* hz_launch.h - the thin launchers between the host code (hz_context / hz_draw / hz_convert / hz_hostpath / hz_ingest .cpp:
 * plain C++ over the HIP runtime API, compiled by g++) and the kernels (hz_kernels.hip, compiled by hipcc): one function per kernel, the kernel's own
 * parameters behind grid, block and stream; template parameters of a kernel are leading bools.  Errors are the caller's
 * to collect (hipGetLastError). */
#pragma once

#include <hip/hip_runtime_api.h>

#include "hz_types.h"

void hzk_clip(bool wave_items, dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, hz_params_t p);
void hzk_hiz(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const unsigned char* touched, int seg_stride, int SW, int H, hz_hiz_t hz, unsigned int nunits);
void hzk_mid(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_rec_t* midrec, const unsigned int* counters, unsigned int midrec_capacity, hz_params_t p);
void hzk_march(bool counters, bool hiz, bool vcache, dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, mr_zones_t zn, hz_params_t p);
void hzk_resolve(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const float* tanel, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int SW, int H, float znear, float zfar, unsigned int* qa, unsigned int* qb);
void hzk_resolve4(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const float* tanel, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int SW, int H, float znear, float zfar, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb, int yo0, int yo1, int nt);
void hzk_pack_host(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, hz_hostpack_t o, int SW, int H, int col_off, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb);
void hzk_tell(hipStream_t stream, hz_tell_t s);
void hzk_pack(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, uint32_t* packed, int SW, int H, unsigned int* qa, unsigned int* qb);
void hzk_resolve_packed(dim3 grid, dim3 block, hipStream_t stream, const uint32_t* packed, int stride, int ncols, const float* tanel, unsigned char* bgr, float* ranges, int out_W, int out_col0, int H, float znear, float zfar);
void hzk_pack_sparse(bool clear, dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, uint32_t* out, int SW, int H, int mask_stride, unsigned char* touched, int seg_stride, unsigned int* qa, unsigned int* qb);
void hzk_resolve_sparse(dim3 grid, dim3 block, hipStream_t stream, hz_strips_t st, int mask_stride, const float* tanel, unsigned char* bgr, float* ranges, int out_W, int H, float znear, float zfar);
void hzk_scatter(dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, unsigned long long* fb, mr_queue_t q, hz_params_t p);
void hzk_big(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_bigrec_t* bigrec, const hz_bigitem_t* bigitem, const unsigned int* big_counters, unsigned int bigrec_capacity, unsigned int bigitem_capacity, hz_params_t p, const unsigned int* tile_state, unsigned int* report);
void hzk_shade_tex(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const int16_t* mosaic, const uint32_t* texels, hz_texparams_t tp, unsigned char* bgr, hz_params_t p);
void hzk_tile_bin(dim3 grid, dim3 block, hipStream_t stream, const hz_bigrec_t* bigrec, const hz_bigitem_t* bigitem, const unsigned int* big_counters, unsigned int bigrec_capacity, tl_bins_t tb, hz_params_t p);
void hzk_tile_raster(dim3 grid, dim3 block, hipStream_t stream, unsigned long long* fb, const hz_bigrec_t* bigrec, tl_bins_t tb, hz_params_t p);
void hzk_ingest(dim3 grid, dim3 block, hipStream_t stream, const unsigned char* const* tiles, int16_t* mosaic, int N, int ntx, int nty, int cpd, int oc_x, int oc_y);
void hzk_link_cells(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const float* tanel, const float* sin_az, const float* cos_az, const double* cos_el, float* lat, float* lon, int W, int H, int cell_w, int cell_h, int nx, int ny, float znear, float zfar, double viewer_lat, double cos_viewer_lat, double viewer_lon);
void hzk_poi(dim3 grid, dim3 block, hipStream_t stream, const unsigned long long* fb, const float* tanel, const hz_poi_proj_t* proj, int npois, unsigned char* visible, float* label_x, float* label_y, int W, int H, int height_out, float znear, float zfar);
void hzk_polar_fill(dim3 grid, dim3 block, hipStream_t stream, const int16_t* mosaic, hz_polar_t* q, int N, hz_xform_t u);
