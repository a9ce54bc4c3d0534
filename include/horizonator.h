/* horizonator.h - terrain-panorama render API, MI355X build.
 *
 * Drop-in for the reference's horizonator.h (reference horizonator.h:1-214):
 * identical context layout and identical prototypes, so callers such as the
 * reference's standalone.c and horizonator-pywrap.c compile against this file
 * unchanged.  The implementation behind it is hand-written HIP for gfx950
 * (horizonator_amd/csrc), not OpenGL.
 *
 * What is different from the GL implementation, by design:
 *  - only the offscreen mode exists (use_glut = true, offscreen_width > 0).
 *    The two on-screen modes need a GL context and return false with a message.
 *  - render_texture = true works as in the reference (0.7*map + 0.3*shade,
 *    reference fragment.glsl:17-22) with the map tiles read from
 *    dir_tiles/tiles_name/12/X/Y.png; nothing is ever downloaded (no
 *    system("wget"), reference horizonator-lib.c:291-320): a tile that is not on
 *    disk makes horizonator_init() fail with a message, whatever allow_downloads
 *    says, and tiles_url_fmt is not used.
 *  - ctx->program holds a handle to the device-side state; the uniform_* slots
 *    are unused and left 0.  Everything a caller is known to read
 *    (Ntriangles, offscreen.*, viewer_lat/lon, dems.*) stays truthful.
 *  - a context works on HIP streams of its own.  horizonator_render_offscreen()
 *    and horizonator_pick() return when their results are in the caller's memory
 *    (as the reference's do); horizonator_redraw() only queues the draw; the
 *    horizonator_amd_*_device calls of horizonator_amd.h queue work and need
 *    horizonator_amd_sync().  A context must be driven from one thread at a
 *    time; different contexts may live on different threads.
 *  - every call leaves the calling thread's current HIP device as it found it.
 *  - annotate() (reference annotator.h:10-25; the reference's Makefile:21 puts
 *    annotator.c into its libhorizonator) is NOT in this library: it is cairo
 *    drawing, out of scope here.  The reference's standalone.c calls it
 *    (standalone.c:499) and therefore links against this library PLUS the
 *    reference's own annotator.o; annotator.c itself only
 *    needs horizonator_project/_unproject, which are exported.
 */
#pragma once

#include <stdbool.h>
#include <stdint.h>

#include "dem.h"

#ifdef __cplusplus
extern "C" {
#endif

/* default clip ranges in metres (reference horizonator.h:9-10) */
#define HORIZONATOR_ZNEAR_DEFAULT 100.0f
#define HORIZONATOR_ZFAR_DEFAULT  40000.0f


/* Layout-identical to reference horizonator.h:13-53.  Callers allocate it. */
typedef struct
{
    int  Ntriangles;
    bool render_texture, use_glut;

    int  glut_window;               /* 1 while the context is live, else 0 */

    /* GL uniform locations in the reference; unused here, kept for layout */
    int32_t uniform_aspect, uniform_az_deg0, uniform_az_deg1;
    int32_t uniform_viewer_cell_i;
    int32_t uniform_viewer_cell_j;
    int32_t uniform_viewer_z;
    int32_t uniform_viewer_lat;
    int32_t uniform_cos_viewer_lat;
    int32_t uniform_texturemap_lon0;
    int32_t uniform_texturemap_lon1;
    int32_t uniform_texturemap_dlat0;
    int32_t uniform_texturemap_dlat1;
    int32_t uniform_texturemap_dlat2;
    int32_t uniform_znear, uniform_zfar;
    int32_t uniform_znear_color, uniform_zfar_color;

    uint32_t program;               /* here: handle of the device-side state */

    float viewer_lat, viewer_lon;

    horizonator_dem_context_t dems;

    struct
    {
        bool     inited;
        uint32_t frameBufID;        /* unused */
        uint32_t renderBufID;       /* unused */
        uint32_t depthBufID;        /* unused */
        int      width, height;
    } offscreen;
} horizonator_context_t;


__attribute__((unused))
static bool horizonator_context_isvalid(const horizonator_context_t* ctx)
{
    return ctx->Ntriangles > 0;
}

/* Replaces reference horizonator-lib.c:61-680.
 *
 * Loads the (2*radius)^2 DEM window centred on the viewer, uploads it to HBM
 * as one row-major int16 mosaic, allocates the W x H device framebuffer and
 * sets the initial view to az -45..45 deg.
 *
 * viewer_z: NULL or *viewer_z < 0 -> pick max(4 surrounding samples)+1 and, if
 * non-NULL, report it back; *viewer_z >= 0 -> use as given.
 * Exactly one of render_radius_cells / render_radius_m must be > 0.
 * dir_dems == NULL -> "~/.horizonator/DEMs_SRTM3" (or ..._SRTM1).
 * render_texture: dir_tiles (NULL -> "~/.horizonator/tiles") and tiles_name
 * (NULL -> "mapnik") say where the zoom-12 map tiles are, defaults as
 * reference horizonator-lib.c:102-120; tiles_url_fmt and allow_downloads are
 * accepted and not acted upon: a missing tile is an error. */
bool horizonator_init(horizonator_context_t* ctx,
                      float viewer_lat, float viewer_lon,
                      float* viewer_z,
                      int offscreen_width, int offscreen_height,
                      int render_radius_cells,
                      float render_radius_m,
                      bool use_glut,
                      bool render_texture,
                      bool SRTM1,
                      const char* dir_dems,
                      const char* dir_tiles,
                      const char* tiles_name,
                      const char* tiles_url_fmt,
                      bool allow_downloads);

/* Replaces reference horizonator-lib.c:682-689; also frees the device state
 * and the DEM mappings (the reference leaks both). */
void horizonator_deinit(horizonator_context_t* ctx);

/* Replaces reference horizonator-lib.c:838-856.  Offscreen contexts cannot be
 * resized (the reference asserts); here: false + message. */
bool horizonator_resized(const horizonator_context_t* ctx, int width, int height);

/* Replaces reference horizonator-lib.c:818-836.  az_deg0/az_deg1 are the
 * azimuths of the LEFT and RIGHT EDGES of the image (x = -0.5 and x = W-0.5),
 * az_deg1 > az_deg0; the vertical scale follows from the aspect ratio. */
bool horizonator_pan_zoom(const horizonator_context_t* ctx,
                          float az_deg0, float az_deg1);

/* Replaces reference horizonator-lib.c:691-816.  Moves the viewer inside the
 * already-loaded window; viewer_z as in horizonator_init(). */
bool horizonator_move(horizonator_context_t* ctx,
                      float* viewer_z,
                      float viewer_lat, float viewer_lon);

/* Replaces reference horizonator-lib.c:864-885.  Near/far clip ranges and the
 * two ranges that map to red = 0 and red = 1.  All four must be > 0, otherwise
 * nothing changes and false is returned (what the reference code does, as
 * opposed to what its comment says). */
bool horizonator_set_zextents(horizonator_context_t* ctx,
                              float znear,       float zfar,
                              float znear_color, float zfar_color);

/* Replaces reference horizonator-lib.c:887-899: clear + draw into the device
 * framebuffer, nothing is read back. */
bool horizonator_redraw(const horizonator_context_t* ctx);

/* Replaces reference horizonator-lib.c:1216-1296.  (x,y) in image pixels,
 * y = 0 is the top row.  Uses the depth of the most recent draw. */
bool horizonator_pick(const horizonator_context_t* ctx,
                      float* lat, float* lon,
                      int x, int y);

/* Replaces reference horizonator-lib.c:911-1051.  Draws and returns packed
 * BGR8 (W*H*3 bytes) and/or float32 ranges (W*H), top row first.  Either
 * pointer may be NULL.  Pixels that show no terrain have range < 0 and colour
 * (B,G,R) = (255,0,0). */
bool horizonator_render_offscreen(const horizonator_context_t* ctx,
                                  char* image, float* ranges);

/* Pure host math, replaces reference horizonator-lib.c:1062-1213 */
bool horizonator_x_from_az(double* x,
                           double* az_ndc_per_rad,
                           double az_rad,
                           double az_rad0,
                           double az_rad1,
                           int width);

bool horizonator_project(double* x,
                         double* y,
                         double* range,
                         double lat_viewer, double cos_lat_viewer,
                         double lon_viewer,
                         double ele_viewer,
                         double lat,
                         double lon,
                         double ele,
                         double az_rad0,
                         double az_rad1,
                         int width,
                         int height);

/* exactly one of range_enh (3-D) / range_en (horizontal) must be > 0 */
bool horizonator_unproject(float* lat, float* lon,
                           int x, int y,
                           double range_enh,
                           double range_en,
                           double lat_viewer, double cos_lat_viewer,
                           double lon_viewer,
                           double az_deg0,
                           double az_deg1,
                           int width,
                           int height);

#ifdef __cplusplus
}
#endif
