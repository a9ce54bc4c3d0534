/* hz_host.c - the reference's C API (include/horizonator.h) on top of the HIP
 * render path (include/hz_hip.h).  Plain C; no HIP or GL headers here.
 *
 * The reference keeps its per-context values inside the GL program object as
 * uniforms and reads them back with glGetUniformfv (reference
 * horizonator-lib.c:966-975).  Here they live in a side record found through
 * ctx->program, because horizonator_context_t is public, caller-allocated and
 * has no spare pointer field.
 */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include <stdio.h>
#include <time.h>

#include "horizonator.h"
#include "horizonator_amd.h"
#include "hz_dem.h"
#include "hz_hip.h"
#include "util.h"
#include "hz_png.h"

/* ------------------------------------------------------------------------ */
/* side records                                                              */

typedef struct
{
    bool         live;
    hz_dev_t*    dev;
    hz_view_t    view;
    hz_tileset_t tiles;         /* tile mappings, kept for horizonator_move()      */
    bool         tiles_owned;   /* false: aliases ctx->dems (<= 4x4 tiles)         */
    int          N;
    int          width, height;
    int          col0, col1;
    /* texture path (reference render_texture): layout fixed at init, coefficients per move */
    bool           textured;
    hz_texparams_t tex;
    /* a context made from a mosaic another process loaded (horizonator_amd_init_from_mosaic)
     * has no tiles: horizonator_move() samples this host copy instead */
    int16_t*       host_mosaic;
    float*       tanel;         /* [height] */
    bool         tanel_valid;   /* ... computed for these azimuth extents: */
    float        tanel_az0, tanel_az1;
} hz_state_t;

/* HZ_INIT_TIMES=1: what horizonator_init() is made of, on stderr (the device side's share: hz_context.cpp) */
static double init_lap(double* since, const char* what)
{
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    const double now = 1e3*(double)t.tv_sec + 1e-6*(double)t.tv_nsec;
    const char* e = getenv("HZ_INIT_TIMES");
    if(what != NULL && e != NULL && atoi(e) != 0) fprintf(stderr, "horizonator_init: %-29s %8.2f ms\n", what, now - *since);
    *since = now;
    return now;
}

#define HZ_MAX_CONTEXTS 64
static hz_state_t g_state[HZ_MAX_CONTEXTS];
/* slots are handed out and given back under this lock, so that contexts may be
 * created and destroyed from different threads; one context is used by one
 * thread at a time (the reference is not thread-safe at all: one process-global
 * GL context, reference horizonator-lib.c:127) */
static bool            g_reserved[HZ_MAX_CONTEXTS];
static pthread_mutex_t g_slots = PTHREAD_MUTEX_INITIALIZER;

static hz_state_t* state_of(const horizonator_context_t* ctx)
{
    if(ctx == NULL || ctx->program == 0 || ctx->program > HZ_MAX_CONTEXTS) return NULL;
    hz_state_t* s = &g_state[ctx->program-1];
    return s->live ? s : NULL;
}

/* every entry point of the reference starts with this check (e.g. reference
 * horizonator-lib.c:700-705): a destroyed context makes the call fail */
static hz_state_t* live_state(const horizonator_context_t* ctx)
{
    if(ctx->use_glut && ctx->glut_window == 0) return NULL;
    return state_of(ctx);
}

static void state_release(hz_state_t* s, horizonator_context_t* ctx)
{
    if(s->dev) hz_hip_destroy(s->dev);
    if(s->tiles_owned) hz_tileset_close(&s->tiles);
    else
    {
        free(s->tiles.tile); free(s->tiles.tile_bytes); free(s->tiles.tile_fd);
        if(ctx) horizonator_dem_deinit(&ctx->dems);
    }
    free(s->tanel);
    free(s->host_mosaic);
    pthread_mutex_lock(&g_slots);
    memset(s, 0, sizeof(*s));
    g_reserved[s - g_state] = false;
    pthread_mutex_unlock(&g_slots);
}

/* ------------------------------------------------------------------------ */
/* texture path (row N4): host side                                          */

#define OSM_RENDER_ZOOM  12         /* reference horizonator-lib.c:25-27 */
#define OSM_TILE_PX      256

/* slippy-map tile that holds (E,N) in degrees, reference horizonator-lib.c:225-246 */
static void osm_tile_id(int* x, int* y, float E, float N)
{
    const float n = (float)(1 << OSM_RENDER_ZOOM);
    E *= (float)M_PI/180.0f;
    N *= (float)M_PI/180.0f;
    const float lon0 = n / 2.0f;
    const float lon1 = n / ((float)M_PI * 2.0f);
    *x = (int)( fminf( n, fmaxf( 0.0f, E*lon1 + lon0 )));
    *y = (int)( n/2.0f * (1.0f - logf( (sinf(N) + 1.0f)/cosf(N) ) / (float)M_PI) );
}

/* the tile range that covers the DEM window around the init viewpoint (reference
 * :372-389), the texture size (:259-262) and the origin of the grid (:577-582) */
static void tex_layout(hz_state_t* s, float init_lat, float init_lon)
{
    const hz_window_t* w = &s->tiles.win;
    const float lowest_E  = init_lon - (float)w->radius_cells/w->cells_per_deg;
    const float lowest_N  = init_lat - (float)w->radius_cells/w->cells_per_deg;
    const float highest_E = init_lon + (float)w->radius_cells/w->cells_per_deg;
    const float highest_N = init_lat + (float)w->radius_cells/w->cells_per_deg;
    int hx, hy;
    osm_tile_id(&s->tex.lowest_x, &s->tex.lowest_y, lowest_E,  highest_N);  /* tile y grows southwards */
    osm_tile_id(&hx,              &hy,              highest_E, lowest_N);
    s->tex.ntiles_x = hx - s->tex.lowest_x + 1;
    s->tex.ntiles_y = hy - s->tex.lowest_y + 1;
    s->tex.tex_w = s->tex.ntiles_x*OSM_TILE_PX;
    s->tex.tex_h = s->tex.ntiles_y*OSM_TILE_PX;
    s->tex.origin_cell_lon_deg = (float)w->origin_tile[0] + (float)w->origin_cell[0] / (float)w->cells_per_deg;
    s->tex.origin_cell_lat_deg = (float)w->origin_tile[1] + (float)w->origin_cell[1] / (float)w->cells_per_deg;
}

/* reference horizonator-lib.c:707-759 texture_coeffs() and :801: they follow the viewer */
static void tex_coeffs(hz_state_t* s, float viewer_lat)
{
    const float n = (float)(1 << OSM_RENDER_ZOOM);
    s->tex.lon0 = n / 2.0f;
    s->tex.lon1 = n / ((float)M_PI * 2.0f);
    const float lat_center = viewer_lat * ((float)M_PI / 180.0f);
    const float k = -n / ((float)M_PI * 2.0f);
    const float t = tanf( lat_center );
    const float c = cosf( lat_center );
    s->tex.dlat0 = n/2.0f + k*logf( t + 1.0f/c );
    s->tex.dlat1 = k / c;
    s->tex.dlat2 = k * t / c / 2.0f;
    s->tex.viewer_lat_rad = (float)(viewer_lat * M_PI / 180.0f);
}

/* reads the map tiles of the layout from dir_tiles/tiles_name/12/X/Y.png
 * (reference :268-369) into texels[tex_h][tex_w][3], B,G,R, southern row first:
 * the bytes and the placement FreeImage's bottom-up BGR bitmap and
 * glTexSubImage2D(..., (highestY - Y)*256, ..., GL_BGR, ...) give the reference */
static bool tex_load_tiles(const hz_state_t* s, unsigned char* texels,
                           const char* dir_tiles, const char* tiles_name, bool allow_downloads)
{
    const hz_texparams_t* t = &s->tex;
    unsigned char* tile = malloc((size_t)OSM_TILE_PX*OSM_TILE_PX*3);
    if(tile == NULL) return false;
    bool ok = true;
    const int highest_y = t->lowest_y + t->ntiles_y - 1;
    for(int ty = t->lowest_y; ty <= highest_y && ok; ty++)
        for(int tx = t->lowest_x; tx < t->lowest_x + t->ntiles_x && ok; tx++)
        {
            char filename[1024], err[1200];
            if((int)sizeof(filename) <= snprintf(filename, sizeof(filename), "%s/%s/%d/%d/%d.png",
                                                 dir_tiles, tiles_name, OSM_RENDER_ZOOM, tx, ty))
            { MSG("tile path too long"); ok = false; break; }
            if(access(filename, R_OK) != 0)
            {
                if(!allow_downloads)
                    MSG("Tile '%s' doesn't exist on disk, and downloads aren't allowed. Giving up", filename);
                else
                    MSG("Tile '%s' doesn't exist on disk, and this build does not download (no network access from the library). Giving up", filename);
                ok = false; break;
            }
            if(0 != hz_png_load_rgb(filename, OSM_TILE_PX, OSM_TILE_PX, tile, err, sizeof(err)))
            { MSG("Couldn't load tile: %s", err); ok = false; break; }
            const int x0 = (tx - t->lowest_x)*OSM_TILE_PX, y0 = (highest_y - ty)*OSM_TILE_PX;
            for(int r=0; r<OSM_TILE_PX; r++)
            {
                unsigned char* dst = texels + ((size_t)(y0 + OSM_TILE_PX-1 - r)*t->tex_w + x0)*3;
                const unsigned char* src = tile + (size_t)r*OSM_TILE_PX*3;
                for(int c=0; c<OSM_TILE_PX; c++) { dst[3*c+0] = src[3*c+2]; dst[3*c+1] = src[3*c+1]; dst[3*c+2] = src[3*c+0]; }
            }
        }
    free(tile);
    return ok;
}

/* ------------------------------------------------------------------------ */

static bool load_tiles(hz_state_t* s, horizonator_context_t* ctx,
                       float viewer_lat, float viewer_lon,
                       int render_radius_cells, float render_radius_m,
                       const char* dir_dems, bool SRTM1)
{
    hz_window_t w;
    if(!hz_window_compute(&w, viewer_lat, viewer_lon, render_radius_cells, render_radius_m, SRTM1))
        return false;

    if(w.ntiles[0] <= max_Ndems_ij && w.ntiles[1] <= max_Ndems_ij)
    {
        /* fits the public struct: load through the public API exactly as the
         * reference does (reference horizonator-lib.c:187-197) so that callers
         * can keep using horizonator_dem_sample(&ctx->dems, ...) */
        if(!horizonator_dem_init(&ctx->dems, viewer_lat, viewer_lon,
                                 render_radius_cells, render_radius_m, dir_dems, SRTM1))
            return false;
        const int nt = w.ntiles[0]*w.ntiles[1];
        s->tiles.win        = w;
        s->tiles.tile       = calloc(nt, sizeof(*s->tiles.tile));
        s->tiles.tile_bytes = calloc(nt, sizeof(*s->tiles.tile_bytes));
        s->tiles.tile_fd    = calloc(nt, sizeof(*s->tiles.tile_fd));
        if(!s->tiles.tile || !s->tiles.tile_bytes || !s->tiles.tile_fd) return false;
        for(int tj=0; tj<w.ntiles[1]; tj++)
            for(int ti=0; ti<w.ntiles[0]; ti++)
                s->tiles.tile[ti + tj*w.ntiles[0]] = ctx->dems.dems[ti][tj];
        s->tiles_owned = false;
        return true;
    }

    /* larger than the reference can load (reference dem.h:8, dem.c:173-178):
     * own tile table; ctx->dems carries the window description only */
    if(!hz_tileset_open(&s->tiles, &w, dir_dems)) return false;
    s->tiles_owned = true;
    memset(&ctx->dems, 0, sizeof(ctx->dems));
    ctx->dems.cells_per_deg = w.cells_per_deg;
    ctx->dems.radius_cells  = w.radius_cells;
    for(int a=0; a<2; a++)
    {
        ctx->dems.origin_dem_lon_lat[a] = w.origin_tile[a];
        ctx->dems.origin_dem_cellij [a] = w.origin_cell[a];
        ctx->dems.Ndems_ij          [a] = w.ntiles[a];
    }
    return true;
}

/* everything after the DEM window is known: device state, DEM -> HBM, the
 * context's public fields, initial view.  `mosaic`: the N x N samples if the
 * caller has them already (then no tile is touched), else NULL: built from
 * s->tiles, on the host or - HORIZONATOR_INGEST=device - in a kernel */
static bool init_device_side(horizonator_context_t* ctx, hz_state_t* s, int slot,
                             float viewer_lat, float viewer_lon, float* viewer_z,
                             int offscreen_width, int offscreen_height,
                             const int16_t* given_mosaic,
                             bool render_texture, const char* dir_tiles, const char* tiles_name, bool allow_downloads)
{
    int16_t* mosaic = NULL;
    unsigned char* texels = NULL;
    bool     result = false;

    const int R = s->tiles.win.radius_cells;
    const int N = 2*R;
    s->N = N;
    s->width = offscreen_width; s->height = offscreen_height;
    s->col0 = 0; s->col1 = offscreen_width;

    int device = 0;
    const char* env = getenv("HORIZONATOR_HIP_DEVICE");
    if(env != NULL) device = atoi(env);

    if(hz_hip_device_count() <= 0)
    {
        MSG("No HIP device is visible; this library has no CPU or OpenGL fallback");
        goto done;
    }
    double lap; init_lap(&lap, NULL);
    s->dev = hz_hip_create(device, N, offscreen_width, offscreen_height);
    init_lap(&lap, "device state");
    if(s->dev == NULL)
    {
        MSG("Couldn't create the device state: %s", hz_hip_last_error());
        goto done;
    }

    /* (the pinned memory a panorama's terrain pixels will land in, made now: the tiles travel through it first) */
    (void)hz_hip_host_prepare(s->dev, 1, 1, 0, 0, NULL);
    init_lap(&lap, "host path: threads, pinned memory");

    /* DEM -> HBM.  Default (round 6): the tiles' raw bytes go to the device through pinned memory and a kernel decodes
     * them (hz_ingest.cpp).  HORIZONATOR_INGEST=host: decoded on the host into one row-major int16 mosaic, which is
     * uploaded (round 1's way; also what happens if the device's way fails). */
    const char* ingest = getenv("HORIZONATOR_INGEST");
    bool in_hbm = false;
    if(given_mosaic != NULL)
    {
        if(0 != hz_hip_upload_mosaic(s->dev, given_mosaic))
        {
            MSG("Mosaic upload failed: %s", hz_hip_last_error());
            goto done;
        }
        in_hbm = true;
    }
    else if(ingest == NULL || strcmp(ingest, "host") != 0)
    {
        in_hbm = 0 == hz_hip_ingest_tiles(s->dev, (const unsigned char* const*)s->tiles.tile,
                                          s->tiles.win.ntiles[0], s->tiles.win.ntiles[1],
                                          s->tiles.win.cells_per_deg,
                                          s->tiles.win.origin_cell[0], s->tiles.win.origin_cell[1]);
        if(!in_hbm) MSG("Device-side DEM ingest failed (%s): decoding on the host", hz_hip_last_error());
    }
    if(!in_hbm)
    {
        mosaic = malloc((size_t)N*N*sizeof(int16_t));
        if(mosaic == NULL) { MSG("out of memory for a %dx%d mosaic", N, N); goto done; }
        hz_tileset_build_mosaic(&s->tiles, mosaic);
        init_lap(&lap, "mosaic built on the host");
        if(0 != hz_hip_upload_mosaic(s->dev, mosaic))
        {
            MSG("Mosaic upload failed: %s", hz_hip_last_error());
            goto done;
        }
    }

    init_lap(&lap, "DEM in HBM");
    s->tanel = malloc((size_t)offscreen_height*sizeof(float));
    if(s->tanel == NULL) goto done;


    /* reference horizonator-lib.c:203; the count overflows int for 11x11
     * SRTM1 mosaics, which the reference cannot load anyway: saturate */
    const long long ntri = 2LL*(N-1)*(N-1);
    ctx->Ntriangles     = ntri > INT_MAX ? INT_MAX : (int)ntri;
    ctx->render_texture = render_texture;
    ctx->use_glut       = true;
    ctx->glut_window    = 1;
    ctx->program        = (uint32_t)(slot+1);
    s->live = true;

    s->view.deg_per_cell = 1.0f / (float)s->tiles.win.cells_per_deg;        /* reference :577 */
    s->view.aspect = (float)offscreen_width / (float)offscreen_height;     /* reference :658-659 */

    /* the texture's tile range belongs to the window around the init viewpoint
     * whether or not tiles are loaded now: horizonator_amd_set_texture() may
     * supply a mosaic later */
    tex_layout(s, viewer_lat, viewer_lon);
    if(render_texture)
    {
        /* reference horizonator-lib.c:259-264 (a texture of zeros), :393-400 (filled tile by tile) */
        texels = calloc((size_t)s->tex.tex_w*s->tex.tex_h, 3);
        if(texels == NULL) { MSG("out of memory for a %dx%d texture", s->tex.tex_w, s->tex.tex_h); goto done; }
        if(!tex_load_tiles(s, texels, dir_tiles, tiles_name, allow_downloads)) goto done;
        tex_coeffs(s, viewer_lat);
        if(0 != hz_hip_set_texture(s->dev, &s->tex, texels))
        {
            MSG("Texture upload failed: %s", hz_hip_last_error());
            goto done;
        }
        s->textured = true;
    }

    if(!horizonator_move(ctx, viewer_z, viewer_lat, viewer_lon)) goto done; /* reference :611 */
    if(!horizonator_set_zextents(ctx,
                                 HORIZONATOR_ZNEAR_DEFAULT, HORIZONATOR_ZFAR_DEFAULT,
                                 HORIZONATOR_ZNEAR_DEFAULT, HORIZONATOR_ZFAR_DEFAULT)) goto done;

    ctx->offscreen.inited = true;
    ctx->offscreen.width  = offscreen_width;
    ctx->offscreen.height = offscreen_height;

    if(!horizonator_pan_zoom(ctx, -45.f, 45.f)) goto done;                  /* reference :670 */
    init_lap(&lap, "texture, first view");
    /* what horizonator_render_offscreen() needs beyond the draw - host threads, the copy streams, pinned memory for a
     * panorama's terrain pixels - is made here, not inside the first call: the reference's CLI makes exactly one
     * (reference standalone.c:433-460); and the device's share of such a call is run once, for the whole circle from
     * here, and thrown away (HORIZONATOR_NO_WARMUP=1: not).  A failure is not fatal: the call then does it itself. */
    if(!s->textured)
    {
        hz_view_t warm = s->view;
        warm.az_deg0 = -180.f; warm.az_deg1 = 180.f;
        const char* no = getenv("HORIZONATOR_NO_WARMUP");
        (void)hz_hip_host_prepare(s->dev, 1, 1, 0, 0, (no != NULL && atoi(no) != 0) ? NULL : &warm);
    }
    init_lap(&lap, "host path prepared");
    result = true;

 done:
    free(mosaic);
    free(texels);
    return result;
}

/* reserves a slot; state_release() gives it back */
static int free_slot(void)
{
    int slot = -1;
    pthread_mutex_lock(&g_slots);
    for(int k=0; k<HZ_MAX_CONTEXTS && slot < 0; k++)
        if(!g_state[k].live && !g_reserved[k]) { g_reserved[k] = true; slot = k; }
    pthread_mutex_unlock(&g_slots);
    if(slot < 0) MSG("Too many live contexts (max %d)", HZ_MAX_CONTEXTS);
    return slot;
}

bool horizonator_init(horizonator_context_t* ctx,
                      float viewer_lat, float viewer_lon,
                      float* viewer_z,
                      int offscreen_width, int offscreen_height,
                      int render_radius_cells, float render_radius_m,
                      bool use_glut, bool render_texture, bool SRTM1,
                      const char* dir_dems,
                      const char* dir_tiles, const char* tiles_name,
                      const char* tiles_url_fmt, bool allow_downloads)
{
    (void)tiles_url_fmt;

    memset(ctx, 0, sizeof(*ctx));       /* reference horizonator-lib.c:84 */

    /* reference horizonator-lib.c:90-121 */
    char dir_tiles_expanded[512];
    if(tiles_name == NULL) tiles_name = "mapnik";
    if(dir_tiles == NULL)  dir_tiles  = "~/.horizonator/tiles";
    if(dir_tiles[0] == '~' && dir_tiles[1] == '/')
    {
        const char* home = getenv("HOME");
        if(home == NULL)
        {
            if(render_texture) { MSG("User asked for ~, but the 'HOME' env var isn't defined"); return false; }
            home = "";
        }
        if((int)sizeof(dir_tiles_expanded) <= snprintf(dir_tiles_expanded, sizeof(dir_tiles_expanded), "%s/%s", home, &dir_tiles[2]))
        { MSG("static buffer overflow: dir_tiles"); return false; }
        dir_tiles = dir_tiles_expanded;
    }

    if(dir_dems == NULL)                /* reference horizonator-lib.c:94-97 */
        dir_dems = SRTM1 ? "~/.horizonator/DEMs_SRTM1" : "~/.horizonator/DEMs_SRTM3";

    if(!use_glut || offscreen_width <= 0 || offscreen_height <= 0)
    {
        MSG("This build renders offscreen only: horizonator_init(use_glut=true, offscreen_width,height > 0). There is no OpenGL window mode");
        return false;
    }
    const int slot = free_slot();
    if(slot < 0) return false;
    hz_state_t* s = &g_state[slot];
    memset(s, 0, sizeof(*s));

    bool result = false;
    double lap; init_lap(&lap, NULL);
    const bool loaded = load_tiles(s, ctx, viewer_lat, viewer_lon, render_radius_cells, render_radius_m, dir_dems, SRTM1);
    init_lap(&lap, "tiles mapped");
    if(!loaded)
        MSG("Couldn't init DEMs. Giving up");
    else
        result = init_device_side(ctx, s, slot, viewer_lat, viewer_lon, viewer_z, offscreen_width, offscreen_height,
                                  NULL, render_texture, dir_tiles, tiles_name, allow_downloads);
    if(!result)
    {
        state_release(s, ctx);
        memset(ctx, 0, sizeof(*ctx));
    }
    return result;
}

bool horizonator_amd_get_window(const horizonator_context_t* ctx, horizonator_amd_window_t* win)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || win == NULL) return false;
    const hz_window_t* w = &s->tiles.win;
    win->cells_per_deg = w->cells_per_deg; win->radius_cells = w->radius_cells;
    for(int a=0; a<2; a++) { win->origin_tile[a] = w->origin_tile[a]; win->origin_cell[a] = w->origin_cell[a]; }
    return true;
}

bool horizonator_amd_init_from_mosaic(horizonator_context_t* ctx,
                                      float viewer_lat, float viewer_lon, float* viewer_z,
                                      int offscreen_width, int offscreen_height,
                                      const horizonator_amd_window_t* win, const int16_t* mosaic)
{
    memset(ctx, 0, sizeof(*ctx));
    if(win == NULL || mosaic == NULL || offscreen_width <= 0 || offscreen_height <= 0 ||
       win->radius_cells <= 0 || (win->cells_per_deg != 1200 && win->cells_per_deg != 3600))
    {
        MSG("horizonator_amd_init_from_mosaic: bad arguments");
        return false;
    }
    const int slot = free_slot();
    if(slot < 0) return false;
    hz_state_t* s = &g_state[slot];
    memset(s, 0, sizeof(*s));

    /* the window as the process that read the tiles computed it; no tile table here */
    hz_window_t* w = &s->tiles.win;
    w->cells_per_deg = win->cells_per_deg; w->radius_cells = win->radius_cells;
    for(int a=0; a<2; a++) { w->origin_tile[a] = win->origin_tile[a]; w->origin_cell[a] = win->origin_cell[a]; w->ntiles[a] = 0; }
    s->tiles_owned = true;              /* nothing of ctx->dems to release */
    ctx->dems.cells_per_deg = w->cells_per_deg;
    ctx->dems.radius_cells  = w->radius_cells;
    for(int a=0; a<2; a++)
    {
        ctx->dems.origin_dem_lon_lat[a] = w->origin_tile[a];
        ctx->dems.origin_dem_cellij [a] = w->origin_cell[a];
    }

    const size_t N = 2*(size_t)w->radius_cells;
    bool result = false;
    s->host_mosaic = malloc(N*N*sizeof(int16_t));
    if(s->host_mosaic == NULL) MSG("out of memory for a %zux%zu mosaic", N, N);
    else
    {
        memcpy(s->host_mosaic, mosaic, N*N*sizeof(int16_t));
        result = init_device_side(ctx, s, slot, viewer_lat, viewer_lon, viewer_z, offscreen_width, offscreen_height,
                                  s->host_mosaic, false, NULL, NULL, false);
    }
    if(!result)
    {
        state_release(s, ctx);
        memset(ctx, 0, sizeof(*ctx));
    }
    return result;
}

void horizonator_deinit(horizonator_context_t* ctx)
{
    hz_state_t* s = state_of(ctx);
    if(s != NULL) state_release(s, ctx);
    ctx->glut_window = 0;
    ctx->program     = 0;
    ctx->Ntriangles  = 0;
    ctx->offscreen.inited = false;
}

/* elevation of window sample (i,j) for the viewer's height: from the tiles, or
 * from the host copy of the mosaic of a context that has no tiles (-1 outside
 * the window, like horizonator_dem_sample) */
static float sample_for_move(const hz_state_t* s, int i, int j)
{
    if(s->host_mosaic == NULL) return hz_tileset_sample(&s->tiles, i, j);
    if(i < 0 || j < 0 || i >= s->N || j >= s->N) return -1.f;
    return s->host_mosaic[(size_t)j*s->N + i];
}

bool horizonator_move(horizonator_context_t* ctx, float* viewer_z,
                      float viewer_lat, float viewer_lon)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    const hz_window_t* w = &s->tiles.win;

    /* reference horizonator-lib.c:765-770, float32 */
    const float viewer_cell_i =
        (viewer_lon - (float)w->origin_tile[0]) * (float)w->cells_per_deg - (float)w->origin_cell[0];
    const float viewer_cell_j =
        (viewer_lat - (float)w->origin_tile[1]) * (float)w->cells_per_deg - (float)w->origin_cell[1];

    /* reference horizonator-lib.c:775-789: stand 1 m above the highest of
     * the four samples around the viewer unless told otherwise */
    float z;
    if(viewer_z == NULL || *viewer_z < 0)
    {
        const int i0 = (int)floorf(viewer_cell_i);
        const int j0 = (int)floorf(viewer_cell_j);
        const float z00 = sample_for_move(s, i0,   j0  );
        const float z10 = sample_for_move(s, i0+1, j0  );
        const float z01 = sample_for_move(s, i0,   j0+1);
        const float z11 = sample_for_move(s, i0+1, j0+1);
        z = fmaxf(fmaxf(z00, z10), fmaxf(z01, z11)) + 1.0f;
        if(viewer_z != NULL) *viewer_z = z;
    }
    else
        z = *viewer_z;

    s->view.viewer_cell_i  = viewer_cell_i;
    s->view.viewer_cell_j  = viewer_cell_j;
    s->view.viewer_z       = z;
    /* reference horizonator-lib.c:799: the product is formed in double
     * (M_PI is a double), rounded to float by cosf's prototype */
    s->view.cos_viewer_lat = cosf((float)((double)viewer_lat * M_PI / 180.0));

    if(s->textured)
    {
        /* reference horizonator-lib.c:761-763,801-809: the texture coefficients follow the viewer */
        tex_coeffs(s, viewer_lat);
        if(0 != hz_hip_set_texture(s->dev, &s->tex, NULL))
        {
            MSG("texture parameters: %s", hz_hip_last_error());
            return false;
        }
    }

    ctx->viewer_lat = viewer_lat;
    ctx->viewer_lon = viewer_lon;
    return true;
}

bool horizonator_pan_zoom(const horizonator_context_t* ctx, float az_deg0, float az_deg1)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    s->view.az_deg0 = az_deg0;      /* stored raw, reference horizonator-lib.c:833-834 */
    s->view.az_deg1 = az_deg1;
    return true;
}

bool horizonator_resized(const horizonator_context_t* ctx, int width, int height)
{
    (void)width; (void)height;
    if(live_state(ctx) == NULL) return false;
    /* reference horizonator-lib.c:847-851 asserts here; every context of this
     * build is offscreen */
    MSG("Resising an offscreen window is not yet supported");
    return false;
}

bool horizonator_set_zextents(horizonator_context_t* ctx,
                              float znear, float zfar, float znear_color, float zfar_color)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    /* reference horizonator-lib.c:875-877 */
    if(!(znear > 0.0f && znear_color > 0.0f && zfar > 0.0f && zfar_color > 0.0f))
        return false;
    s->view.znear = znear;             s->view.zfar = zfar;
    s->view.znear_color = znear_color; s->view.zfar_color = zfar_color;
    return true;
}

bool horizonator_redraw(const horizonator_context_t* ctx)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(0 != hz_hip_draw(s->dev, &s->view))
    {
        MSG("draw failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

/* tan(elevation) of every GL row (row 0 = bottom), reference
 * horizonator-lib.c:1006-1012 and the mirrored use at :1026-1047: the lower
 * half evaluates get_tanel(y), the upper half reuses the mirror row's value
 * (its sign does not matter, it is squared) */
static void fill_tanel(hz_state_t* s)
{
    const int   height = s->height;
    const float aspect = (float)s->width / (float)height;
    const float az_deg0 = s->view.az_deg0, az_deg1 = s->view.az_deg1;
    if(s->tanel_valid && s->tanel_az0 == az_deg0 && s->tanel_az1 == az_deg1) return;
    s->tanel_valid = true; s->tanel_az0 = az_deg0; s->tanel_az1 = az_deg1;
    for(int row=0; row<height; row++)
    {
        int y = row;
        if(row >= height - height/2) y = height-1 - row;
        const float el_ndc = ((float)y + 0.5f) / (float)height * 2.f - 1.f;
        const float el     = el_ndc * (az_deg1-az_deg0) / 2.f / aspect * M_PI/180.0f;
        s->tanel[row] = tanf(el);
    }
}

static bool render_common(const horizonator_context_t* ctx, bool to_host,
                          void* image, float* ranges, int32_t* index, uint32_t* z24)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(!ctx->offscreen.inited)
    {
        /* reference horizonator-lib.c:924-928 */
        MSG("Prior to calling horizonator_render_offscreen(), the context must have been inited for offscreen rendering with horizonator_init(use_glut=true, offscreen_width,height > 0)");
        return false;
    }
    if(ranges != NULL) fill_tanel(s);
    int rc;
    if(to_host)
        /* glClear + glDrawElements + the two glReadPixels and the conversion (reference horizonator-lib.c:896-897,
         * 936-1048) as one call: the HIP side draws and ships the panorama sector by sector (hz_hostpath.cpp) */
        rc = hz_hip_render_to_host(s->dev, &s->view, s->tanel, image, ranges, index, z24);
    else
    {
        if(!horizonator_redraw(ctx)) return false;
        rc = hz_hip_resolve(s->dev, &s->view, s->tanel, image, ranges, index, z24);
    }
    if(rc != 0)
    {
        MSG("render failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_amd_render_begin(const horizonator_context_t* ctx, char* image, float* ranges)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(!ctx->offscreen.inited) { MSG("the context must have been inited for offscreen rendering"); return false; }
    if(ranges != NULL) fill_tanel(s);
    if(0 != hz_hip_host_begin(s->dev, &s->view, s->tanel, (unsigned char*)image, ranges, NULL, NULL))
    {
        MSG("render failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}
bool horizonator_amd_render_end(const horizonator_context_t* ctx)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(0 != hz_hip_host_end(s->dev))
    {
        MSG("render failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_render_offscreen(const horizonator_context_t* ctx, char* image, float* ranges)
{
    return render_common(ctx, true, image, ranges, NULL, NULL);
}

bool horizonator_pick(const horizonator_context_t* ctx, float* lat, float* lon, int x, int y)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    uint32_t zi;
    if(0 != hz_hip_read_depth(s->dev, x, y, &zi)) return false;
    /* reference horizonator-lib.c:1267-1273: depth as GL hands it out */
    const float depth = (float)((double)zi * (1.0/16777215.0));
    if(depth >= 1.0f) return false;
    /* reference horizonator-lib.c:1285-1295 */
    const double range_en = depth * (s->view.zfar - s->view.znear) + s->view.znear;
    return horizonator_unproject(lat, lon, x, y, -1., range_en,
                                 ctx->viewer_lat, s->view.cos_viewer_lat, ctx->viewer_lon,
                                 s->view.az_deg0, s->view.az_deg1,
                                 s->width, s->height);
}

/* ------------------------------------------------------------------------ */
/* pure host math: reference horizonator-lib.c:1053-1213                     */

static double unwrap_near_rad_d(double x, double near)
{
    /* reference horizonator-lib.c:1056-1060.  C round() = half away from
     * zero, unlike the shader's round-half-even; kept as the reference has it */
    const double d = (x - near) / (2.*M_PI);
    return (d - round(d)) * 2.*M_PI + near;
}

bool horizonator_x_from_az(double* x, double* az_ndc_per_rad,
                           double az_rad, double az_rad0, double az_rad1, int width)
{
    az_rad1 = unwrap_near_rad_d(az_rad1-az_rad0, M_PI) + az_rad0;
    const double center = (az_rad0 + az_rad1)/2.;
    az_rad = unwrap_near_rad_d(az_rad, center);
    const double k = 2.0 / (az_rad1 - az_rad0);
    const double az_ndc = (az_rad - center) * k;
    if(!(-1. <= az_ndc && az_ndc <= 1.)) return false;
    if(az_ndc_per_rad != NULL) *az_ndc_per_rad = k;
    *x = (az_ndc + 1.)/2.*width - 0.5;
    return true;
}

bool horizonator_project(double* x, double* y, double* range,
                         double lat_viewer, double cos_lat_viewer, double lon_viewer,
                         double ele_viewer,
                         double lat, double lon, double ele,
                         double az_rad0, double az_rad1, int width, int height)
{
    const float Rearth = 6371000.0;
    const double dlat  = (lat - lat_viewer)*M_PI/180;
    const double dlon  = (lon - lon_viewer)*M_PI/180;
    const double east  = dlon * Rearth * cos_lat_viewer;
    const double north = dlat * Rearth;
    const double d2    = east*east + north*north;

    double k;
    if(!horizonator_x_from_az(x, &k, atan2(east, north), az_rad0, az_rad1, width))
        return false;

    const double h    = ele - ele_viewer;
    const double d_ne = sqrt(d2);
    *range = sqrt(d2 + h*h);

    const double aspect = (double)width / (double)height;
    const double el_ndc = atan2(h, d_ne) * aspect * k;
    if(!(-1. <= el_ndc && el_ndc <= 1.)) return false;
    *y = (-el_ndc + 1.)/2.*height - 0.5;
    return true;
}

bool horizonator_unproject(float* lat, float* lon, int x, int y,
                           double range_enh, double range_en,
                           double lat_viewer, double cos_lat_viewer, double lon_viewer,
                           double az_deg0, double az_deg1, int width, int height)
{
    if(1 != (range_enh > 0.) + (range_en > 0.)) return false;
    const float Rearth = 6371000.0;

    /* reference horizonator-lib.c:1183-1184: mixed float/double on purpose */
    float az_ndc = ((float)x + 0.5f) / (float)width * 2.f - 1.f;
    float az     = (az_ndc * (az_deg1-az_deg0) / 2.f + (az_deg1+az_deg0)/2.f) * M_PI/180.0f;

    if(range_en <= 0)
    {
        double aspect = (double)width / (double)height;
        double el_ndc = ((double)y + 0.5) / (double)height * 2. - 1.;
        double el     = el_ndc * (az_deg1-az_deg0) / 2. / aspect * M_PI/180.0;
        range_en = cos(el) * range_enh;
    }
    float e = range_en * sinf(az);
    float n = range_en * cosf(az);
    *lon = lon_viewer + e / Rearth / M_PI * 180. / cos_lat_viewer;
    *lat = lat_viewer + n / Rearth / M_PI * 180.;
    return true;
}

/* ------------------------------------------------------------------------ */
/* build-side additions (include/horizonator_amd.h)                          */

bool horizonator_amd_render(const horizonator_context_t* ctx,
                            char* image, float* ranges, int32_t* index, uint32_t* z24)
{
    return render_common(ctx, true, image, ranges, index, z24);
}

bool horizonator_amd_render_device(const horizonator_context_t* ctx,
                                   void* d_image, float* d_ranges, int32_t* d_index, uint32_t* d_z24)
{
    return render_common(ctx, false, d_image, d_ranges, d_index, d_z24);
}

bool horizonator_amd_render_batch(horizonator_context_t* ctx, int n,
                                  const float* viewer_lat, const float* viewer_lon, float* viewer_z,
                                  void* d_images, float* d_ranges)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || n < 0 || viewer_lat == NULL || viewer_lon == NULL) return false;
    const size_t npix = (size_t)(s->col1 - s->col0) * (size_t)s->height;
    for(int v=0; v<n; v++)
    {
        /* exactly what a caller of the reference does per viewpoint
         * (horizonator-pywrap.c:219-233: move, then render) */
        if(!horizonator_move(ctx, viewer_z ? &viewer_z[v] : NULL, viewer_lat[v], viewer_lon[v]))
            return false;
        if(!render_common(ctx, false,
                          d_images ? (char*)d_images + (size_t)v*npix*3 : NULL,
                          d_ranges ? d_ranges + (size_t)v*npix : NULL, NULL, NULL))
            return false;
    }
    return true;
}

bool horizonator_amd_texture_layout(const horizonator_context_t* ctx,
                                    int* lowest_x, int* lowest_y, int* ntiles_x, int* ntiles_y)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(lowest_x) *lowest_x = s->tex.lowest_x;
    if(lowest_y) *lowest_y = s->tex.lowest_y;
    if(ntiles_x) *ntiles_x = s->tex.ntiles_x;
    if(ntiles_y) *ntiles_y = s->tex.ntiles_y;
    return true;
}

bool horizonator_amd_set_texture(horizonator_context_t* ctx, const unsigned char* texels_bgr)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(texels_bgr == NULL)
    {
        (void)hz_hip_set_texture(s->dev, NULL, NULL);
        s->textured = false;
        ctx->render_texture = false;
        return true;
    }
    tex_coeffs(s, ctx->viewer_lat);
    if(0 != hz_hip_set_texture(s->dev, &s->tex, texels_bgr))
    {
        MSG("Texture upload failed: %s", hz_hip_last_error());
        return false;
    }
    s->textured = true;
    ctx->render_texture = true;
    return true;
}

bool horizonator_amd_render_packed(const horizonator_context_t* ctx, uint32_t* d_packed)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || d_packed == NULL) return false;
    if(!ctx->offscreen.inited) return false;
    if(!horizonator_redraw(ctx)) return false;
    if(0 != hz_hip_pack(s->dev, d_packed))
    {
        MSG("pack failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_amd_resolve_packed(const horizonator_context_t* ctx,
                                    const uint32_t* d_packed, int packed_stride, int ncols, int out_col0,
                                    void* d_image, float* d_ranges)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || d_packed == NULL) return false;
    if(d_ranges != NULL) fill_tanel(s);
    if(0 != hz_hip_resolve_packed(s->dev, &s->view, s->tanel, d_packed, packed_stride, ncols, out_col0,
                                  d_image, d_ranges))
    {
        MSG("resolve of packed strips failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_amd_render_sparse(const horizonator_context_t* ctx, uint32_t* d_out, int mask_stride)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || d_out == NULL || !ctx->offscreen.inited) return false;
    if(!horizonator_redraw(ctx)) return false;
    if(0 != hz_hip_pack_sparse(s->dev, d_out, mask_stride))
    {
        MSG("sparse pack failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_amd_resolve_sparse_strips(const horizonator_context_t* ctx, int nstrips,
                                           const uint32_t* const* d_in, int mask_stride,
                                           const int* ncols, const int* out_col0,
                                           void* d_image, float* d_ranges)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || nstrips < 0 || d_in == NULL || ncols == NULL || out_col0 == NULL) return false;
    if(d_ranges != NULL) fill_tanel(s);
    if(0 != hz_hip_resolve_sparse_strips(s->dev, &s->view, s->tanel, nstrips, d_in, mask_stride, ncols, out_col0,
                                         d_image, d_ranges))
    {
        MSG("resolve of sparse strips failed: %s", hz_hip_last_error());
        return false;
    }
    return true;
}

bool horizonator_amd_resolve_packed_strips(const horizonator_context_t* ctx, int nstrips,
                                           const uint32_t* const* d_packed, int packed_stride,
                                           const int* ncols, const int* out_col0,
                                           void* d_image, float* d_ranges)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || nstrips < 0 || d_packed == NULL || ncols == NULL || out_col0 == NULL) return false;
    if(d_ranges != NULL) fill_tanel(s);
    for(int k=0; k<nstrips; k++)
    {
        if(ncols[k] == 0) continue;             /* a rank that drew nothing */
        if(0 != hz_hip_resolve_packed(s->dev, &s->view, s->tanel, d_packed[k], packed_stride, ncols[k], out_col0[k],
                                      d_image, d_ranges))
        {
            MSG("resolve of packed strip %d failed: %s", k, hz_hip_last_error());
            return false;
        }
    }
    return true;
}

bool horizonator_amd_sync(const horizonator_context_t* ctx)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_sync(s->dev);
}

bool horizonator_amd_stream_waits_for_outputs(const horizonator_context_t* ctx, void* stream)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_wait_outputs(s->dev, stream);
}

bool horizonator_amd_waits_for_stream(const horizonator_context_t* ctx, void* stream)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_wait_for(s->dev, stream);
}

bool horizonator_amd_set_sector(const horizonator_context_t* ctx, int col0, int col1)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    if(0 != hz_hip_set_sector(s->dev, col0, col1))
    {
        MSG("%s", hz_hip_last_error());
        return false;
    }
    s->col0 = col0; s->col1 = col1;
    return true;
}

bool horizonator_amd_set_raster(const horizonator_context_t* ctx, int which)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_set_raster(s->dev, which);
}

bool horizonator_amd_get_options(const horizonator_context_t* ctx, hz_options_t* options)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && options != NULL && 0 == hz_hip_get_options(s->dev, options);
}
bool horizonator_amd_set_options(const horizonator_context_t* ctx, const hz_options_t* options)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && options != NULL && 0 == hz_hip_set_options(s->dev, options);
}

bool horizonator_amd_set_profiling(const horizonator_context_t* ctx, bool on)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_set_profiling(s->dev, on ? 1 : 0);
}

bool horizonator_amd_last_times(const horizonator_context_t* ctx, hz_times_t* times)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_last_times(s->dev, times);
}

bool horizonator_amd_get_view(const horizonator_context_t* ctx, hz_view_t* view)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL) return false;
    *view = s->view;
    return true;
}

hz_dev_t* horizonator_amd_device(const horizonator_context_t* ctx)
{
    hz_state_t* s = live_state(ctx);
    return s ? s->dev : NULL;
}

bool horizonator_amd_link_cells_size(const horizonator_context_t* ctx, int cell_width, int cell_height,
                                     int cut_off_bottom_px, int* nx, int* ny)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || cell_width <= 0 || cell_height <= 0) return false;
    /* the loops of reference annotator.c:230-232 */
    const int height_out = s->height - cut_off_bottom_px;
    int cx = 0, cy = 0;
    for(int x=0; x<s->width-cell_width; x += cell_width) cx++;
    for(int y=0; y<height_out-cell_height; y += cell_height) cy++;
    *nx = cx; *ny = cy;
    return true;
}

/* reference annotator.c:228-264.  What depends on the cell's column and row alone - sinf / cosf of the azimuth,
 * cos of the elevation, reference horizonator-lib.c:1181-1195 - is evaluated here, with the C library the
 * reference calls; the device does the rest (IEEE arithmetic only): the same bits as the reference's loop. */
bool horizonator_amd_link_cells(const horizonator_context_t* ctx, int cell_width, int cell_height,
                                int cut_off_bottom_px, float* lat, float* lon)
{
    hz_state_t* s = live_state(ctx);
    int nx, ny;
    if(s == NULL || !horizonator_amd_link_cells_size(ctx, cell_width, cell_height, cut_off_bottom_px, &nx, &ny))
        return false;
    if(nx == 0 || ny == 0) return true;
    fill_tanel(s);
    const int    width = s->width, height = s->height;
    const double az_deg0 = (double)s->view.az_deg0, az_deg1 = (double)s->view.az_deg1;
    float*  sin_az = (float*) malloc((size_t)nx*sizeof(float));
    float*  cos_az = (float*) malloc((size_t)nx*sizeof(float));
    double* cos_el = (double*)malloc((size_t)ny*sizeof(double));
    bool ok = sin_az != NULL && cos_az != NULL && cos_el != NULL;
    if(ok)
    {
        for(int cx=0; cx<nx; cx++)
        {
            const int x = cx*cell_width + cell_width/2;
            /* reference horizonator-lib.c:1183-1184: mixed float/double on purpose */
            float az_ndc = ((float)x + 0.5f) / (float)width * 2.f - 1.f;
            float az     = (az_ndc * (az_deg1-az_deg0) / 2.f + (az_deg1+az_deg0)/2.f) * M_PI/180.0f;
            sin_az[cx] = sinf(az);
            cos_az[cx] = cosf(az);
        }
        for(int cy=0; cy<ny; cy++)
        {
            const int y = cy*cell_height + cell_height/2;
            double aspect = (double)width / (double)height;
            double el_ndc = ((double)y + 0.5) / (double)height * 2. - 1.;
            double el     = el_ndc * (az_deg1-az_deg0) / 2. / aspect * M_PI/180.0;
            cos_el[cy] = cos(el);
        }
        const double viewer_lat = (double)ctx->viewer_lat;
        if(0 != hz_hip_link_cells(s->dev, &s->view, s->tanel, sin_az, cos_az, cos_el,
                                  viewer_lat, cos(viewer_lat * M_PI/180.), (double)ctx->viewer_lon,
                                  cell_width, cell_height, nx, ny, lat, lon))
        {
            MSG("%s", hz_hip_last_error());
            ok = false;
        }
    }
    free(sin_az); free(cos_az); free(cos_el);
    return ok;
}

/* reference annotator.c:280-348.  The projection of each point (reference horizonator_project: atan2 and sqrt in
 * double) is made here, with the C library the reference calls; the device searches the range image. */
bool horizonator_amd_poi_visibility(const horizonator_context_t* ctx, int cut_off_bottom_px,
                                    const hz_poi_t* pois, int npois,
                                    unsigned char* visible, float* label_x, float* label_y)
{
    hz_state_t* s = live_state(ctx);
    if(s == NULL || npois < 0) return false;
    if(npois == 0) return true;
    fill_tanel(s);
    hz_poi_proj_t* proj = (hz_poi_proj_t*)malloc((size_t)npois*sizeof(*proj));
    if(proj == NULL) return false;
    const double lat = (double)ctx->viewer_lat, lon = (double)ctx->viewer_lon, ele = (double)s->view.viewer_z;
    const double cos_lat = cos(lat * M_PI/180.);
    const double az_rad0 = (double)s->view.az_deg0 * M_PI/180., az_rad1 = (double)s->view.az_deg1 * M_PI/180.;
    for(int k=0; k<npois; k++)
    {
        if(!horizonator_project(&proj[k].x, &proj[k].y, &proj[k].range, lat, cos_lat, lon, ele,
                                (double)pois[k].lat, (double)pois[k].lon, (double)pois[k].ele_m,
                                az_rad0, az_rad1, s->width, s->height))
            proj[k].x = proj[k].y = 0.0, proj[k].range = -1.0;
    }
    const bool ok = 0 == hz_hip_poi_visibility(s->dev, &s->view, s->tanel, cut_off_bottom_px, proj, npois, visible, label_x, label_y);
    if(!ok) MSG("%s", hz_hip_last_error());
    free(proj);
    return ok;
}

#include "hz_build_id.h"         /* made by the Makefile: HZ_BUILD_ID */
const char* horizonator_amd_build_id(void) { return HZ_BUILD_ID; }

bool horizonator_amd_get_mosaic(const horizonator_context_t* ctx, int16_t* mosaic)
{
    hz_state_t* s = live_state(ctx);
    return s != NULL && 0 == hz_hip_download_mosaic(s->dev, mosaic);
}
