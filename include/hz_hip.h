/* hz_hip.h - C-ABI of the HIP side of the render path.
 *
 * The host library (hz_host.c, plain C) implements horizonator.h on top of
 * these entry points; they are also what bench.py / the Python mirror bind
 * when they need device-resident outputs (multi-GPU strips, timing).  Plain
 * pointers and sizes only; <hip/hip_runtime.h> is included by the .hip
 * translation unit alone.
 *
 * Each entry point names the reference code whose work it takes over
 * (citations into /root/reference).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hz_dev hz_dev_t;        /* opaque device-side state */

/* The "uniforms": everything a draw depends on besides the DEM.  float32, and
 * derived on the host exactly as the reference does it:
 *   viewer_cell_i/j, viewer_z, cos_viewer_lat  reference horizonator-lib.c:765-799
 *   deg_per_cell                               reference horizonator-lib.c:577
 *   az_deg0/1 (stored raw)                     reference horizonator-lib.c:833-834
 *   aspect = (float)W/(float)H                 reference horizonator-lib.c:658-659
 *   znear..zfar_color                          reference horizonator-lib.c:879-882 */
typedef struct
{
    float viewer_cell_i, viewer_cell_j;
    float viewer_z;
    float cos_viewer_lat;
    float deg_per_cell;
    float az_deg0, az_deg1;
    float aspect;
    float znear, zfar;
    float znear_color, zfar_color;
} hz_view_t;

/* which rasteriser hz_hip_draw() runs */
enum
{
    HZ_RASTER_AUTO   = 0,
    HZ_RASTER_SCATTER= 1,   /* block per 64x4 DEM cells, vertices staged in LDS            */
    HZ_RASTER_MARCH  = 2    /* wave per 63-cell-wide strip, marching north, no barriers   */
};

/* per-draw kernel times in ms, measured with HIP events on the context's own
 * stream (only filled while profiling is on) */
typedef struct
{
    float clear_ms;
    float raster_ms;        /* the dominant kernel: k_scatter, or k_march over everything but the strips
                             * next to the viewer (round 2 of the draw) */
    float big_ms;           /* the queue kernels after it: clipped, medium and large triangles */
    float resolve_ms;
    float total_ms;         /* first event to last event */
    float near_ms;          /* round 1 of the draw: k_march next to the viewer + its queue kernels */
} hz_times_t;

int  hz_hip_device_count(void);

/* Takes over the GL object creation of reference horizonator-lib.c:403-512
 * (VBO/IBO) and :617-666 (FBO): allocates the N x N int16 mosaic and the
 * W x H 64-bit depth/id/colour framebuffer on `device`.  NULL on failure.
 * Images of up to 2^29 pixels (framebuffer words are addressed with 32-bit byte
 * offsets; the reference's own limit on llvmpipe is 16384^2 = 2^28). */
hz_dev_t* hz_hip_create(int device, int N, int width, int height);
void      hz_hip_destroy(hz_dev_t* d);

/* mosaic[j*N+i], j north, i east (what reference horizonator-lib.c:435-480
 * pushes into the VBO, minus the redundant i,j).  Host pointer. */
int  hz_hip_upload_mosaic(hz_dev_t* d, const int16_t* mosaic);

/* Alternative ingest ("next" row N3): raw big-endian .hgt tiles are copied to
 * the device and a kernel does the byte swap, north-south flip, edge dedup and
 * void clamp of reference dem.c:264-309.  tiles[ti + tj*ntiles_lon] is a host
 * pointer to (cpd+1)^2 big-endian samples or NULL for sea. */
int  hz_hip_ingest_tiles(hz_dev_t* d,
                         const unsigned char* const* tiles,
                         int ntiles_lon, int ntiles_lat,
                         int cells_per_deg,
                         int origin_cell_lon, int origin_cell_lat);

/* read back the device mosaic (tests) */
int  hz_hip_download_mosaic(hz_dev_t* d, int16_t* mosaic);

/* Restrict drawing to image columns [col0, col1): the azimuth-sector shard of
 * one GPU.  The view (centre, scale, discard rule) stays that of the full
 * panorama, so a sector's pixels are bit-identical to the same pixels of a
 * full draw.  Default: [0, width). */
int  hz_hip_set_sector(hz_dev_t* d, int col0, int col1);

int  hz_hip_set_raster(hz_dev_t* d, int which);
int  hz_hip_set_profiling(hz_dev_t* d, int on);

/* The tunables of a context.  Every one of them changes HOW a picture is made, none WHAT is in it (the GPU suite runs
 * under each of them: tools/gpu_modes.sh).  -1 / 0 = "the draw decides" where noted.  A context starts with the
 * defaults below, overridden - for debugging only - by environment variables read once when it is created
 * (HZ_SERIAL, HZ_TWO_PASS=0|1, HZ_NEAR_CELLS, HZ_HIZ, HZ_TILES, HZ_TILE_LIST, HZ_ADAPT, HZ_ADAPT_HI, HZ_PRETEST_MARCH,
 * HZ_NO_WORKLIST, HZ_NO_FAST_MATH, HZ_RESOLVE_CLEARS, HZ_QUEUE_CAPACITY, HZ_HOST_DENSE, HZ_HOST_SECTORS,
 * HZ_HOST_TIMES, HZ_VERTEX_CACHE); hz_hip_set_options() replaces them (queued work is waited for first; `serial` and
 * `queue_capacity` only take effect at creation). */
typedef struct
{
    int serial;           /* 0    1: one stream instead of four (per-kernel times of a trace are then those of each kernel alone) */
    int rounds;           /* 0    1 / 2: every draw in one round / in two (first the strips next to the viewer); 0: by image size and far clip */
    int near_cells;       /* -1   the first round's reach in cells; -1: from the view (cells wider than ~20 px, at most 384, zoomed views 512) */
    int coarse_depth;     /* -1   0 / 1: second rounds never / always test larger boxes against coarse depth (k_hiz); -1: zoomed views and series of renders */
    int tiles;            /* -1   first rounds' large triangles by screen tile with depth in LDS (k_tile_*): 0 never, 1 always, -1 zoomed views */
    int tile_list;        /* 0    triangles a tile's list holds (tests: small values exercise the fall-back to k_big); 0: 2048 */
    int adapt;            /* 1    the first round's reach of a zoomed view: 0 always short, 2 always long, 1 by what the draws before had to queue */
    int adapt_hi;         /* -1   ... from this many work items in the second round on; -1: 500 000 per 64 Mpix */
    int pretest_march;    /* -1   0 / 1: second rounds' marching waves never / always read a word before the atomic; -1: framebuffers beyond the 256 MB last-level cache */
    int worklists;        /* 1    0: sectors and narrow views launch the whole grid of strips instead of a host-built list */
    int fast_math;        /* 1    0: the unabridged division / square-root sequences everywhere (same bits) */
    int resolve_clears;   /* 1    0: framebuffers are cleared by a memset instead of by the conversion that reads them last */
    int queue_capacity;   /* 0    records per queue between the kernels of a draw (tests: small values exercise the overflow paths); 0: by image size */
    int host_dense;       /* 0    1: results for host memory travel whole (every pixel) instead of without the sky */
    int host_sectors;     /* 0    azimuth sectors a call that delivers into host memory is drawn and shipped in (draw of sector s+1 beside the transfer of sector s); 0: by image size */
    int host_times;       /* 0    1: such a call says on stderr where its time went */
    int vertex_cache;     /* 1    0: every draw computes every vertex's transform in full; 1: from the second draw from a viewpoint on, the half of it that
                           *      depends on the viewer's position alone (two atan, two square roots) is kept in HBM, 16 bytes per vertex, and read back */
} hz_options_t;
int  hz_hip_get_options(hz_dev_t* d, hz_options_t* o);
int  hz_hip_set_options(hz_dev_t* d, const hz_options_t* o);

/* Takes over glClear + glDrawElements (reference horizonator-lib.c:896-897)
 * and with it vertex.glsl / geometry.glsl / fragment.glsl and the fixed
 * function raster + depth test.  Asynchronous: queued on the context's streams (marching kernel, queue kernels and
 * conversions have one each, so that consecutive draws overlap; hz_hip_sync() waits for all). */
int  hz_hip_draw(hz_dev_t* d, const hz_view_t* view);

/* Takes over the two glReadPixels + flip + depth->range conversion of
 * reference horizonator-lib.c:936-1048, on the device.  Outputs are DEVICE
 * pointers, each may be NULL; they describe the current sector only and are
 * laid out [height][col1-col0], top row first:
 *   bgr    3 bytes per pixel
 *   ranges float32, < 0 where no terrain
 *   index  int32 primitive id (2*(j*(N-1)+i)+t), -1 where no terrain
 *   z24    uint32 raw 24-bit depth, 0xFFFFFF where no terrain
 * tanel is a HOST array of `height` floats: tan(elevation) of each GL row
 * (row 0 = bottom) as the reference computes it (reference
 * horizonator-lib.c:1007-1012). */
int  hz_hip_resolve(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                    unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24);

/* The same into HOST pointers (each may be NULL); synchronous.  What crosses PCIe is the terrain pixels only, 4 bytes
 * each (hz_scatter.h: z24<<8 | red8 - the host makes BGR bytes, depth and range of them with the reference's own
 * arithmetic, reference horizonator-lib.c:1013-1025); host threads fill the caller's buffers with the sky's constants
 * (reference :185, :1016) while the device draws and put the terrain in its places as it arrives - the same bytes as
 * the dense copy (hz_options_t::host_dense).
 *   hz_hip_resolve_to_host   the conversion of the draw already queued (hz_hip_draw)
 *   hz_hip_render_to_host    draw + conversion: glClear, glDrawElements and the two glReadPixels of reference
 *                            horizonator-lib.c:896-897, 936-1048 in one call.  Whole images of 12 Mpix and more are
 *                            drawn and shipped in azimuth sectors (hz_options_t::host_sectors), sector s+1 drawn while
 *                            sector s crosses PCIe; same bytes.
 *   hz_hip_host_begin / hz_hip_host_end   that call in two halves: begin queues the draw and starts the sky, end moves
 *                            the results and returns when they are in place.  Up to two panoramas may be between their
 *                            begin and their end (each with buffers of its own; ended in the order begun): the device
 *                            draws the second while the first crosses PCIe, and a series pays the link alone.  The
 *                            buffers handed to begin are the library's until end returns. */
int  hz_hip_resolve_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                            unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24);
int  hz_hip_render_to_host(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                           unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24);
int  hz_hip_host_begin(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                       unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24);
int  hz_hip_host_end(hz_dev_t* d);
/* What the first of those calls would otherwise set up - the pool of host threads, the copy streams, the pinned
 * memory a panorama with these outputs lands in - ahead of it (horizonator_init does this: the reference's CLI makes
 * ONE render call per process, reference standalone.c:433-460).  warm_view != NULL: the device's share of such a call
 * (draws, conversion, the words that tell the host) is run once for that view and thrown away, so that the first real
 * call finds every kernel loaded and every table and list allocated. */
int  hz_hip_host_prepare(hz_dev_t* d, int want_bgr, int want_ranges, int want_index, int want_z24, const hz_view_t* warm_view);

/* 24-bit depth of image pixel (x, y), y = 0 top row, from the last draw
 * (reference horizonator-lib.c:1268-1270) */
int  hz_hip_read_depth(hz_dev_t* d, int x, int y, uint32_t* z24);

/* ---- consumers of the range image ("next" row N2) ---------------------------
 * The two passes the reference's annotator makes over the range image
 * (reference annotator.c:228-264 and :280-348), on the device, reading the
 * framebuffer of the last draw: the full-size range image never has to leave
 * the GPU for them. */

/* For every cell of cell_w x cell_h pixels (x = 0, cell_w, ... < W-cell_w;
 * y likewise below height_out = H - cut_off_bottom_px: nx x ny of them) whose
 * top-left pixel shows terrain: latitude/longitude under the cell centre
 * (reference horizonator_unproject with range_enh = range of that pixel).
 * The transcendental functions of that formula belong to the caller (the C
 * library the reference calls; the device's own sinf/cosf differ in the last
 * bit): HOST tables sin_az[nx], cos_az[nx] - sinf / cosf of the azimuth of
 * the centre column of cell column cx, reference horizonator-lib.c:1181-1183 -
 * and cos_el[ny] - cos of the elevation of the centre row of cell row cy,
 * :1190-1195.  lat/lon are HOST arrays [ny][nx]; NaN where the cell has no
 * terrain.  Full-width contexts only (sector = whole image). */
int  hz_hip_link_cells(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                       const float* sin_az, const float* cos_az, const double* cos_el,
                       double viewer_lat, double cos_viewer_lat, double viewer_lon,
                       int cell_w, int cell_h, int nx, int ny, float* lat, float* lon);

typedef struct { float lat, lon, ele_m; } hz_poi_t;
/* a point of interest projected into the image (reference horizonator_project,
 * horizonator-lib.c:1097-1155): pixel coordinates and range; range < 0: not in the view */
typedef struct { double x, y, range; } hz_poi_proj_t;

/* For every projected point of interest: is it visible in the last draw, and
 * where does its label crosshair go (reference annotator.c:297-347: distance
 * window 500 m .. 100 km, vertical search of +-6 pixels in the range image for
 * the range closest to the expected one, accepted within 500 m).  Outputs are
 * HOST arrays of npois: visible (0/1), label_x, label_y (pixels). */
int  hz_hip_poi_visibility(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                           int cut_off_bottom_px, const hz_poi_proj_t* proj, int npois,
                           unsigned char* visible, float* label_x, float* label_y);

/* Multi-GPU gather in 4 instead of 7 bytes per pixel: hz_hip_pack() writes the
 * last draw as z24<<8 | red8 per pixel (DEVICE uint32 [H][sector width], top row
 * first); hz_hip_resolve_packed() runs the readback conversion (reference
 * horizonator-lib.c:936-1048) on such words on any device that holds them:
 * columns [0,ncols) of d_packed[H][stride] go to columns [out_col0,
 * out_col0+ncols) of the full-width d_bgr[H][W][3] / d_ranges[H][W] (DEVICE,
 * either may be NULL).  Same bytes as hz_hip_resolve() of an untextured draw. */
int  hz_hip_pack(hz_dev_t* d, uint32_t* d_packed);
int  hz_hip_resolve_packed(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                           const uint32_t* d_packed, int stride, int ncols, int out_col0,
                           unsigned char* d_bgr, float* d_ranges);

/* ... and without the sky.  A sparse strip is a stream of uint32:
 *   [0] T = number of terrain pixels; [1..1+H) row_base[row]; then the terrain
 *   mask, mask_stride words per row (bit c%32 of word c/32, rows top first); then,
 *   from HDR = 1 + H + H*mask_stride on, the T words z24<<8 | red8, row `row`
 *   starting at HDR + row_base[row], left to right.
 * hz_hip_pack_sparse() writes the last draw in that form (d_out: DEVICE, room for
 * HDR + H*(sector width) words; only the first HDR + T carry information - read
 * T back, send that much).  hz_hip_resolve_sparse() converts such a strip like
 * hz_hip_resolve_packed() does a packed one. */
int  hz_hip_pack_sparse(hz_dev_t* d, uint32_t* d_out, int mask_stride);
int  hz_hip_resolve_sparse(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                           const uint32_t* d_in, int mask_stride, int ncols, int out_col0,
                           unsigned char* d_bgr, float* d_ranges);
/* ... all strips of a panorama in one launch: strip k = d_in[k] (DEVICE), ncols[k] columns
 * (0: skipped) that go to out_col0[k] */
int  hz_hip_resolve_sparse_strips(hz_dev_t* d, const hz_view_t* view, const float* tanel, int nstrips,
                                  const uint32_t* const* d_in, int mask_stride, const int* ncols, const int* out_col0,
                                  unsigned char* d_bgr, float* d_ranges);

/* uniforms of the texture half of the reference's vertex shader
 * (vertex.glsl:16-21; values as horizonator-lib.c:577-588,801-809 sets them)
 * and the size of the texture, NtilesX*256 x NtilesY*256 texels */
typedef struct
{
    float   viewer_lat_rad;
    float   origin_cell_lon_deg, origin_cell_lat_deg;
    float   lon0, lon1, dlat0, dlat1, dlat2;            /* texturemap_* */
    int32_t ntiles_x, ntiles_y, lowest_x, lowest_y;     /* NtilesX/Y, osmtile_lowestX/Y */
    int32_t tex_w, tex_h;
} hz_texparams_t;

/* Texture path (reference render_texture = true; vertex.glsl:41-61,116-126,
 * fragment.glsl:17-22, horizonator-lib.c:247-266,361-366).  params: the uniform
 * values of the texture half of the vertex shader and the size of the texture
 * (hz_tex.h).  texels_bgr: HOST bytes [tex_h][tex_w][3], B,G,R, row 0 = texture
 * coordinate t = 0 (the southern edge of the tile mosaic) - the layout the
 * reference hands to glTexSubImage2D(GL_BGR); NULL keeps the resident texels and
 * only replaces params.  params == NULL switches texturing off.  While on,
 * hz_hip_resolve*() writes 0.7*texture + 0.3*shade into the BGR output. */
int  hz_hip_set_texture(hz_dev_t* d, const hz_texparams_t* params, const unsigned char* texels_bgr);

int  hz_hip_sync(hz_dev_t* d);
int  hz_hip_last_times(hz_dev_t* d, hz_times_t* t);

/* A context works on several HIP streams of its own.  Everything a conversion
 * (hz_hip_resolve*, hz_hip_pack*) writes is written on ONE of them, in call
 * order: hz_hip_stream() returns that stream (as a void* = hipStream_t), so work
 * queued on it after a conversion sees the outputs.  hz_hip_wait_outputs() makes
 * any other stream of the caller's wait (on the device, not on the host) for
 * every conversion queued so far.  hz_hip_sync() waits on the host. */
void* hz_hip_stream(hz_dev_t* d);
int   hz_hip_wait_outputs(hz_dev_t* d, void* stream);
/* ... and the other way round: conversions queued from now on run after
 * everything queued on `stream` so far (strips a collective is still delivering).
 * `stream` must belong to the context's device in both calls (an event of one
 * device cannot be recorded on a stream of another: the call then fails). */
int   hz_hip_wait_for(hz_dev_t* d, void* stream);

/* What the last draw of the context was (bench.py records it with every timing; tests assert on it): out[0] rounds
 * (1 / 2), out[1] its second round kept coarse depth (zoomed views, draws of a series), out[2] the first round's
 * reach in cells, out[3] only the strips behind the drawn columns were launched, out[4] its vertices came from the
 * vertex cache (hz_options_t::vertex_cache).  out: room for 5 ints. */
int  hz_hip_last_plan(hz_dev_t* d, int* out);
/* What the first round's reach of zoomed views goes by (hz_draw.cpp, adapt): the latest second round whose queue
 * counters have reached the host - out[0] the reach of that draw's first round in cells, out[1] / out[2] the records and
 * work items its second round queued for k_big, out[3] 1 = the next zoomed draw takes the long reach */
int  hz_hip_last_queue_counts(hz_dev_t* d, unsigned int* out);

const char* hz_hip_last_error(void);

#ifdef __cplusplus
}
#endif
