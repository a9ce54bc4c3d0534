/* horizonator_amd.h - build-side additions to the reference API.
 *
 * Nothing here exists in the reference; these calls expose what the HIP
 * implementation has for free (the visible-primitive index map, raw depth),
 * what multi-GPU sharding needs (azimuth sectors, device-resident outputs) and
 * what the benchmark needs (per-kernel times).  A caller that only uses
 * horizonator.h never needs this header.
 */
#pragma once

#include "horizonator.h"
#include "hz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One process reads the DEM tiles, the others get the mosaic (multi-GPU: rank 0
 * loads and decodes, then broadcasts int16 N x N over RCCL).
 * horizonator_amd_get_window() describes the window of a context;
 * horizonator_amd_get_mosaic() copies its samples out; with both,
 * horizonator_amd_init_from_mosaic() makes an equivalent offscreen context
 * (untextured) without touching a tile.  viewer_z as in horizonator_init(). */
typedef struct
{
    int cells_per_deg;          /* 1200 | 3600 */
    int radius_cells;           /* the mosaic is (2*radius_cells)^2 samples */
    int origin_tile[2];         /* (lon,lat) of the tile that holds the window's SW corner */
    int origin_cell[2];         /* sample index of that corner inside the tile */
} horizonator_amd_window_t;
bool horizonator_amd_get_window(const horizonator_context_t* ctx, horizonator_amd_window_t* win);
bool horizonator_amd_init_from_mosaic(horizonator_context_t* ctx,
                                      float viewer_lat, float viewer_lon, float* viewer_z,
                                      int offscreen_width, int offscreen_height,
                                      const horizonator_amd_window_t* win, const int16_t* mosaic);

/* Like horizonator_render_offscreen() plus two more outputs, all HOST
 * pointers, each may be NULL, all [H][sector width], top row first:
 *   index  int32, id of the triangle that owns the pixel, -1 for sky.
 *          id = 2*(j*(N-1)+i)+t with (i,j) the DEM cell (i east, j north,
 *          N = 2*radius_cells) and t the triangle of the cell in the order of
 *          the reference's index buffer (reference horizonator-lib.c:500-506),
 *          i.e. the GL primitive id of the reference's draw call.
 *   z24    uint32, the 24-bit depth-buffer value, 0xFFFFFF for sky */
bool horizonator_amd_render(const horizonator_context_t* ctx,
                            char* image, float* ranges,
                            int32_t* index, uint32_t* z24);

/* horizonator_render_offscreen() in two halves, for a caller that renders a series: _begin queues the draw of the
 * current view and returns; _end returns when image and ranges (HOST pointers as horizonator_render_offscreen's, either
 * may be NULL) hold the panorama.  Two panoramas may be between their begin and their end - each into buffers of its
 * own, ended in the order begun: the device then draws one while the other crosses PCIe, and a panorama of a series
 * costs what the link takes (hz_hip.h: hz_hip_host_begin).  The view may be changed (move, pan_zoom, set_zextents)
 * between a begin and the next; the buffers belong to the library until the matching end returns.  Untextured. */
bool horizonator_amd_render_begin(const horizonator_context_t* ctx, char* image, float* ranges);
bool horizonator_amd_render_end(const horizonator_context_t* ctx);

/* Draw + resolve into DEVICE buffers owned by the caller (e.g. torch tensors
 * that an RCCL gather then moves); asynchronous (queued on the context's streams; consecutive renders overlap),
 * follow with horizonator_amd_sync(). */
bool horizonator_amd_render_device(const horizonator_context_t* ctx,
                                   void* d_image, float* d_ranges,
                                   int32_t* d_index, uint32_t* d_z24);
bool horizonator_amd_sync(const horizonator_context_t* ctx);

/* Ordering against a HIP stream of the caller's (hipStream_t as a void*; e.g. the stream a
 * collective is queued on) without involving the host:
 *   horizonator_amd_stream_waits_for_outputs: work queued on `stream` from now on runs after
 *     every conversion (the *_device, *_packed, *_sparse calls) queued on the context so far;
 *   horizonator_amd_waits_for_stream: conversions queued on the context from now on (e.g.
 *     horizonator_amd_resolve_sparse_strips of strips that `stream` is still receiving) run after
 *     everything queued on `stream` so far. */
/* (`stream` must be a stream of the context's device) */
bool horizonator_amd_stream_waits_for_outputs(const horizonator_context_t* ctx, void* stream);
bool horizonator_amd_waits_for_stream(const horizonator_context_t* ctx, void* stream);

/* Texture path with a caller-supplied map ("next" row N4: reference
 * render_texture=true without its tile downloads).  The reference drapes a
 * mosaic of zoom-12 slippy-map tiles, 256x256 each, over the terrain: tiles
 * x in [lowest_x, lowest_x+ntiles_x), y in [lowest_y, lowest_y+ntiles_y), the
 * range that covers the DEM window of horizonator_init() (reference
 * horizonator-lib.c:372-389).  horizonator_init(render_texture=true) reads them
 * from dir_tiles/tiles_name/12/X/Y.png as the reference does (no downloads);
 * alternatively the caller hands over the finished mosaic:
 *   texels_bgr  HOST bytes [ntiles_y*256][ntiles_x*256][3], B,G,R, row 0 = the
 *               SOUTHERN edge (tile row lowest_y+ntiles_y-1, bottom pixel row),
 *               i.e. what the reference passes to glTexSubImage2D(GL_BGR).
 * NULL switches texturing off again.  While on, image outputs are
 * 0.7*map + 0.3*shade (reference fragment.glsl:17-22). */
bool horizonator_amd_texture_layout(const horizonator_context_t* ctx,
                                    int* lowest_x, int* lowest_y, int* ntiles_x, int* ntiles_y);
bool horizonator_amd_set_texture(horizonator_context_t* ctx, const unsigned char* texels_bgr);

/* A batch of viewpoints over the context's DEM window (BASELINE.json configs[3]):
 * for v in [0,n): horizonator_move(viewer_lat[v], viewer_lon[v]) followed by a
 * render into DEVICE buffers d_images[v] ([H][sector width][3] BGR) and
 * d_ranges[v] ([H][sector width] float32); either base pointer may be NULL.
 * viewer_z: NULL = stand 1 m above the terrain at every viewpoint; otherwise n
 * in/out values with the meaning of horizonator_move()'s argument.  Nothing is
 * synchronised between viewpoints: the whole batch is queued on the context's
 * stream; follow with horizonator_amd_sync().  The context is left at the last
 * viewpoint. */
bool horizonator_amd_render_batch(horizonator_context_t* ctx, int n,
                                  const float* viewer_lat, const float* viewer_lon, float* viewer_z,
                                  void* d_images, float* d_ranges);

/* The same across GPUs with fewer bytes on the wire.  A rank draws its sector and
 * writes it as one word per pixel, z24<<8 | red8 (DEVICE uint32 [H][sector
 * width], top row first) - 4 bytes instead of the 7 of BGR8 + float32 range;
 * the gathering rank converts what it received (and its own strip) into the
 * final image and ranges, columns [out_col0, out_col0+ncols) of the FULL-width
 * DEVICE outputs.  The view (z extents, azimuth extents) of the resolving
 * context must be the one the strips were drawn with.  Untextured draws only.
 * Both calls are asynchronous: queued on the context's streams. */
bool horizonator_amd_render_packed(const horizonator_context_t* ctx, uint32_t* d_packed);
bool horizonator_amd_resolve_packed(const horizonator_context_t* ctx,
                                    const uint32_t* d_packed, int packed_stride, int ncols, int out_col0,
                                    void* d_image, float* d_ranges);

/* ... the same for all gathered strips in one call (strip k: d_packed[k], ncols[k] columns
 * that go to out_col0[k]; ncols[k] = 0 is skipped) */
bool horizonator_amd_resolve_packed_strips(const horizonator_context_t* ctx, int nstrips,
                                           const uint32_t* const* d_packed, int packed_stride,
                                           const int* ncols, const int* out_col0,
                                           void* d_image, float* d_ranges);

/* ... and without the sky, which is most of a panorama and carries no information: the
 * sparse strip format of hz_hip.h (hz_hip_pack_sparse).  d_out: DEVICE, room for
 * 1 + H + H*mask_stride + H*(sector width) words; word 0 becomes the number T of terrain
 * pixels and only the first 1 + H + H*mask_stride + T words need to travel.  All strips of
 * one gather share mask_stride (>= ceil(widest sector / 32)). */
bool horizonator_amd_render_sparse(const horizonator_context_t* ctx, uint32_t* d_out, int mask_stride);
bool horizonator_amd_resolve_sparse_strips(const horizonator_context_t* ctx, int nstrips,
                                           const uint32_t* const* d_in, int mask_stride,
                                           const int* ncols, const int* out_col0,
                                           void* d_image, float* d_ranges);

/* Restrict this context to image columns [col0,col1) of the panorama: the
 * azimuth-sector shard one GPU renders.  Outputs then have width col1-col0. */
bool horizonator_amd_set_sector(const horizonator_context_t* ctx, int col0, int col1);

/* HZ_RASTER_* of hz_hip.h */
bool horizonator_amd_set_raster(const horizonator_context_t* ctx, int which);

/* the context's tunables (hz_hip.h: hz_options_t; none of them changes a byte of any result) */
bool horizonator_amd_get_options(const horizonator_context_t* ctx, hz_options_t* options);
bool horizonator_amd_set_options(const horizonator_context_t* ctx, const hz_options_t* options);

bool horizonator_amd_set_profiling(const horizonator_context_t* ctx, bool on);
bool horizonator_amd_last_times(const horizonator_context_t* ctx, hz_times_t* times);

/* the view ("uniform") values currently in effect, and the device state */
bool      horizonator_amd_get_view(const horizonator_context_t* ctx, hz_view_t* view);
hz_dev_t* horizonator_amd_device  (const horizonator_context_t* ctx);

/* Annotator passes over the range image of the last draw, on the device
 * (see hz_hip.h: hz_hip_link_cells, hz_hip_poi_visibility).  Viewer position,
 * height and azimuth extents are those of the context. */
bool horizonator_amd_link_cells_size(const horizonator_context_t* ctx, int cell_width, int cell_height,
                                     int cut_off_bottom_px, int* nx, int* ny);
bool horizonator_amd_link_cells(const horizonator_context_t* ctx, int cell_width, int cell_height,
                                int cut_off_bottom_px, float* lat, float* lon);
bool horizonator_amd_poi_visibility(const horizonator_context_t* ctx, int cut_off_bottom_px,
                                    const hz_poi_t* pois, int npois,
                                    unsigned char* visible, float* label_x, float* label_y);

/* A digest of the sources this library was built from (16 hex digits).  A program linked against the library can
 * record it at link time and compare at run time (tests/caller_stubs: the reference's CLI does). */
const char* horizonator_amd_build_id(void);

/* copy of the N x N int16 mosaic as it sits in HBM (tests) */
bool horizonator_amd_get_mosaic(const horizonator_context_t* ctx, int16_t* mosaic);

#ifdef __cplusplus
}
#endif
