/* SRTM tile mosaic: window arithmetic, tile mapping, sampling, and the
 * row-major int16 mosaic that is uploaded to HBM.
 *
 * Behaviour follows reference dem.c (cited per function).  Structure is our
 * own: one window computation and one sampler shared by the public 4x4 API
 * (include/dem.h) and the uncapped internal loader (hz_dem.h).
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dem.h"
#include "hz_dem.h"
#include "util.h"

#define HGT_WIDTH_SRTM3 1201
#define HGT_WIDTH_SRTM1 3601

/* ------------------------------------------------------------------------ */
/* window                                                                    */

bool hz_window_compute(hz_window_t* win,
                       float viewer_lat, float viewer_lon,
                       int render_radius_cells, float render_radius_m,
                       bool SRTM1)
{
    /* reference dem.c:90-99 */
    if(render_radius_cells < 0 && render_radius_m < 0)
    {
        MSG("Exactly one of (render_radius_cells,render_radius_m) should be >0. Both were <0");
        return false;
    }
    if(render_radius_cells > 0 && render_radius_m > 0)
    {
        MSG("Exactly one of (render_radius_cells,render_radius_m) should be >0. Both were >0");
        return false;
    }

    memset(win, 0, sizeof(*win));
    win->cells_per_deg = (SRTM1 ? HGT_WIDTH_SRTM1 : HGT_WIDTH_SRTM3) - 1;
    const int cpd = win->cells_per_deg;

    if(render_radius_cells > 0)
        win->radius_cells = render_radius_cells;
    else
    {
        /* reference dem.c:124-126: the square must contain a circle of
         * render_radius_m; the east-west cell pitch (x cos lat) is the short one */
        const double Rearth  = 6371000.0;
        const double cos_lat = cos(M_PI / 180.0 * (double)viewer_lat);
        const double m_per_cell = Rearth * M_PI/180. * cos_lat / (double)cpd;
        win->radius_cells = (int)(0.5 + (double)render_radius_m / m_per_cell);
    }
    if(win->radius_cells <= 0)
    {
        MSG("render radius came out as %d cells; need >0", win->radius_cells);
        return false;
    }

    const float lonlat[2] = {viewer_lon, viewer_lat};
    for(int a=0; a<2; a++)
    {
        /* reference dem.c:143-152, float32 on purpose: same origin as the
         * reference for the same inputs */
        const int   cell0   = (int)floorf(lonlat[a] * (float)cpd) - (win->radius_cells-1);
        const float origin  = (float)cell0 / (float)cpd;
        win->origin_tile[a] = (int)floorf(origin);
        win->origin_cell[a] = (int)roundf((origin - (float)win->origin_tile[a]) * (float)cpd);

        /* reference dem.c:162-171: the last sample may sit on the shared
         * edge of the next tile, in which case that tile is not needed */
        const int last      = win->origin_cell[a] + 2*win->radius_cells - 1;
        const int tile_last = last / cpd;
        win->ntiles[a] = tile_last + 1;
        if(last == tile_last*cpd && tile_last > 0)
            win->ntiles[a]--;
    }
    return true;
}

bool hz_tile_path(char* path, int bufsize, int tile_lat, int tile_lon, const char* datadir)
{
    /* reference dem.c:22-76 */
    const char ns = tile_lat >= 0 ? 'N' : 'S';
    const char ew = tile_lon >= 0 ? 'E' : 'W';
    const int  alat = abs(tile_lat);
    const int  alon = abs(tile_lon);

    int n;
    if(datadir[0] == '~' && datadir[1] == '/')
    {
        const char* home = getenv("HOME");
        if(home == NULL)
        {
            MSG("User asked for ~, but the 'HOME' env var isn't defined");
            return false;
        }
        n = snprintf(path, bufsize, "%s/%s/%c%.2d%c%.3d.hgt", home, &datadir[2], ns, alat, ew, alon);
    }
    else
        n = snprintf(path, bufsize, "%s/%c%.2d%c%.3d.hgt", datadir, ns, alat, ew, alon);
    return n < bufsize;
}

/* ------------------------------------------------------------------------ */
/* tiles                                                                     */

/* Maps one tile.  *data == NULL on return means "reads as sea level".
 * Returns false only for hard errors (mmap failure, wrong size). */
static bool map_tile(unsigned char** data, size_t* bytes, int* fd,
                     const char* filename, size_t expected_bytes)
{
    *data = NULL; *bytes = 0; *fd = 0;

    /* reference dem.c:198-206: unreadable tile -> warning, elevation 0 */
    int f = open(filename, O_RDONLY);
    if(f <= 0)
    {
        MSG("Warning: couldn't open DEM file '%s'. Assuming elevation=0 (sea surface?)", filename);
        return true;
    }
    struct stat sb;
    if(fstat(f, &sb) != 0)
    {
        MSG("Couldn't stat the DEM file '%s'", filename);
        close(f);
        return false;
    }
    /* reference dem.c:210-222: empty file -> sea, silently */
    if(sb.st_size == 0)
    {
        close(f);
        return true;
    }
    /* reference dem.c:234-239 */
    if((size_t)sb.st_size != expected_bytes)
    {
        MSG("The DEM file '%s' has unexpected size. Is this a 3-arc-sec SRTM DEM?", filename);
        close(f);
        return false;
    }
    void* p = mmap(NULL, sb.st_size, PROT_READ, MAP_PRIVATE, f, 0);
    if(p == MAP_FAILED)
    {
        MSG("Couldn't mmap the DEM file '%s'", filename);
        close(f);
        return false;
    }
    *data = p; *bytes = sb.st_size; *fd = f;
    return true;
}

static void unmap_tile(unsigned char** data, size_t* bytes, int* fd)
{
    if(*data != NULL && *data != MAP_FAILED) munmap(*data, *bytes);
    if(*fd > 0) close(*fd);
    *data = NULL; *bytes = 0; *fd = 0;
}

static size_t hgt_bytes(int cpd) { return (size_t)(cpd+1)*(size_t)(cpd+1)*2; }

bool hz_tileset_open(hz_tileset_t* ts, const hz_window_t* win, const char* datadir)
{
    memset(ts, 0, sizeof(*ts));
    ts->win = *win;
    const int nt = win->ntiles[0]*win->ntiles[1];
    ts->tile       = calloc(nt, sizeof(*ts->tile));
    ts->tile_bytes = calloc(nt, sizeof(*ts->tile_bytes));
    ts->tile_fd    = calloc(nt, sizeof(*ts->tile_fd));
    if(!ts->tile || !ts->tile_bytes || !ts->tile_fd)
    {
        MSG("out of memory for %d tiles", nt);
        hz_tileset_close(ts);
        return false;
    }
    for(int tj=0; tj<win->ntiles[1]; tj++)
        for(int ti=0; ti<win->ntiles[0]; ti++)
        {
            char filename[1024];
            if(!hz_tile_path(filename, sizeof(filename),
                             tj + win->origin_tile[1], ti + win->origin_tile[0], datadir))
            {
                MSG("Couldn't construct DEM filename");
                hz_tileset_close(ts);
                return false;
            }
            const int k = ti + tj*win->ntiles[0];
            if(!map_tile(&ts->tile[k], &ts->tile_bytes[k], &ts->tile_fd[k],
                         filename, hgt_bytes(win->cells_per_deg)))
            {
                hz_tileset_close(ts);
                return false;
            }
        }
    return true;
}

void hz_tileset_close(hz_tileset_t* ts)
{
    if(ts->tile)
    {
        const int nt = ts->win.ntiles[0]*ts->win.ntiles[1];
        for(int k=0; k<nt; k++)
            unmap_tile(&ts->tile[k], &ts->tile_bytes[k], &ts->tile_fd[k]);
    }
    free(ts->tile);       ts->tile = NULL;
    free(ts->tile_bytes); ts->tile_bytes = NULL;
    free(ts->tile_fd);    ts->tile_fd = NULL;
}

/* Window sample -> (tile index, in-tile index) along one axis.
 * reference dem.c:280-293: in-tile index 0 is served from the previous
 * tile's last row/column (tiles overlap by one sample).  The reference does
 * that even when there is no previous tile and then reads out of bounds; here
 * the redirect is applied only when a previous tile exists, and index 0 of
 * the first tile is read from that tile itself (it is valid data). */
static inline bool split_axis(int* tile, int* cell, int v, int origin_cell, int cpd, int ntiles)
{
    int c = v + origin_cell;
    int t = c / cpd;
    c -= t*cpd;
    if(c == 0 && t > 0) { t--; c = cpd; }
    *tile = t; *cell = c;
    return t < ntiles;
}

static inline int16_t decode_be16_clamped(const unsigned char* p)
{
    /* reference dem.c:307-308: big-endian, voids/negatives read as 0 */
    int16_t z = (int16_t)((p[0] << 8) | p[1]);
    return z < 0 ? 0 : z;
}

static inline int16_t sample_tiles(unsigned char* const* tile, int tile_stride,
                                   const hz_window_t* w, int i, int j)
{
    if(i < 0 || j < 0) return -1;           /* reference dem.c:270 */
    int ti,ci, tj,cj;
    if(!split_axis(&ti,&ci, i, w->origin_cell[0], w->cells_per_deg, w->ntiles[0])) return -1;
    if(!split_axis(&tj,&cj, j, w->origin_cell[1], w->cells_per_deg, w->ntiles[1])) return -1;
    const unsigned char* t = tile[ti + tj*tile_stride];
    if(t == NULL) return 0;                 /* reference dem.c:296-298 */
    /* reference dem.c:300-304: file rows run north->south */
    const size_t p = (size_t)ci + (size_t)(w->cells_per_deg - cj)*(size_t)(w->cells_per_deg+1);
    return decode_be16_clamped(&t[2*p]);
}

int16_t hz_tileset_sample(const hz_tileset_t* ts, int i, int j)
{
    return sample_tiles(ts->tile, ts->win.ntiles[0], &ts->win, i, j);
}

void hz_tileset_build_mosaic(const hz_tileset_t* ts, int16_t* mosaic)
{
    const hz_window_t* w = &ts->win;
    const int N   = 2*w->radius_cells;
    const int cpd = w->cells_per_deg;

    #pragma omp parallel for schedule(static)
    for(int j=0; j<N; j++)
    {
        int16_t* row = &mosaic[(size_t)j*N];
        int tj,cj;
        if(!split_axis(&tj,&cj, j, w->origin_cell[1], cpd, w->ntiles[1]))
        {
            for(int i=0; i<N; i++) row[i] = -1;
            continue;
        }
        int i = 0;
        while(i < N)
        {
            int ti,ci;
            if(!split_axis(&ti,&ci, i, w->origin_cell[0], cpd, w->ntiles[0]))
            {
                for(; i<N; i++) row[i] = -1;
                break;
            }
            /* run of samples served by this tile: in-tile columns ci..cpd */
            int run = cpd - ci + 1;
            if(run > N-i) run = N-i;
            const unsigned char* t = ts->tile[ti + tj*w->ntiles[0]];
            if(t == NULL)
                memset(&row[i], 0, (size_t)run*sizeof(int16_t));
            else
            {
                const unsigned char* src = &t[2*((size_t)ci + (size_t)(cpd - cj)*(size_t)(cpd+1))];
                for(int k=0; k<run; k++)
                    row[i+k] = decode_be16_clamped(&src[2*k]);
            }
            i += run;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* public 4x4 API (include/dem.h)                                            */

bool horizonator_dem_init(horizonator_dem_context_t* ctx,
                          float viewer_lat, float viewer_lon,
                          int render_radius_cells, float render_radius_m,
                          const char* datadir, bool SRTM1)
{
    hz_window_t w;
    if(!hz_window_compute(&w, viewer_lat, viewer_lon, render_radius_cells, render_radius_m, SRTM1))
        return false;

    memset(ctx, 0, sizeof(*ctx));
    ctx->cells_per_deg = w.cells_per_deg;
    ctx->radius_cells  = w.radius_cells;
    for(int a=0; a<2; a++)
    {
        ctx->origin_dem_lon_lat[a] = w.origin_tile[a];
        ctx->origin_dem_cellij [a] = w.origin_cell[a];
        ctx->Ndems_ij          [a] = w.ntiles[a];
        /* reference dem.c:173-178 */
        if(w.ntiles[a] > max_Ndems_ij)
        {
            MSG("Requested radius too large. Increase the compile-time-constant max_Ndems_ij from the current value of %d", max_Ndems_ij);
            return false;
        }
    }

    for(int tj=0; tj<w.ntiles[1]; tj++)
        for(int ti=0; ti<w.ntiles[0]; ti++)
        {
            char filename[1024];
            if(!hz_tile_path(filename, sizeof(filename),
                             tj + w.origin_tile[1], ti + w.origin_tile[0], datadir))
            {
                MSG("Couldn't construct DEM filename");
                horizonator_dem_deinit(ctx);
                return false;
            }
            if(!map_tile(&ctx->dems[ti][tj], &ctx->mmap_sizes[ti][tj], &ctx->mmap_fd[ti][tj],
                         filename, hgt_bytes(w.cells_per_deg)))
            {
                horizonator_dem_deinit(ctx);
                return false;
            }
        }
    return true;
}

void horizonator_dem_deinit(horizonator_dem_context_t* ctx)
{
    for(int ti=0; ti<max_Ndems_ij; ti++)
        for(int tj=0; tj<max_Ndems_ij; tj++)
            unmap_tile(&ctx->dems[ti][tj], &ctx->mmap_sizes[ti][tj], &ctx->mmap_fd[ti][tj]);
}

int16_t horizonator_dem_sample(const horizonator_dem_context_t* ctx, int i, int j)
{
    if(i < 0 || j < 0) return -1;
    const int cpd = ctx->cells_per_deg;
    int ti,ci, tj,cj;
    if(!split_axis(&ti,&ci, i, ctx->origin_dem_cellij[0], cpd, ctx->Ndems_ij[0])) return -1;
    if(!split_axis(&tj,&cj, j, ctx->origin_dem_cellij[1], cpd, ctx->Ndems_ij[1])) return -1;
    if(ti >= max_Ndems_ij || tj >= max_Ndems_ij) return 0;
    const unsigned char* t = ctx->dems[ti][tj];
    if(t == NULL) return 0;
    const size_t p = (size_t)ci + (size_t)(cpd - cj)*(size_t)(cpd+1);
    return decode_be16_clamped(&t[2*p]);
}

void horizonator_dem_bounds_latlon_deg(const horizonator_dem_context_t* ctx,
                                       float* lat0, float* lon0,
                                       float* lat1, float* lon1)
{
    /* reference dem.c:313-330, float32 */
    const float cpd  = (float)ctx->cells_per_deg;
    const float span = (float)(2*ctx->radius_cells - 1);
    *lon0 = (float)ctx->origin_dem_lon_lat[0] +  (float)ctx->origin_dem_cellij[0]         / cpd;
    *lat0 = (float)ctx->origin_dem_lon_lat[1] +  (float)ctx->origin_dem_cellij[1]         / cpd;
    *lon1 = (float)ctx->origin_dem_lon_lat[0] + ((float)ctx->origin_dem_cellij[0] + span) / cpd;
    *lat1 = (float)ctx->origin_dem_lon_lat[1] + ((float)ctx->origin_dem_cellij[1] + span) / cpd;
}
