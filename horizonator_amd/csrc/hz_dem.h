/* Internal DEM loader: same window arithmetic and sample semantics as the
 * public dem.h API (reference dem.c:78-309) but with a heap-allocated tile
 * grid, so mosaics larger than the public struct's 4x4 (reference dem.h:8)
 * can be loaded: 7x7 SRTM3 and 11x11 SRTM1 need it. */
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    int cells_per_deg;          /* 1200 | 3600 */
    int radius_cells;
    int origin_tile[2];         /* (lon,lat) of the tile holding the SW corner */
    int origin_cell[2];         /* sample index of the SW corner in that tile   */
    int ntiles[2];              /* tiles along (lon,lat)                        */
} hz_window_t;

typedef struct
{
    hz_window_t     win;
    unsigned char** tile;       /* [ntiles[0]*ntiles[1]], index ti + tj*ntiles[0]; NULL = sea */
    size_t*         tile_bytes;
    int*            tile_fd;
} hz_tileset_t;

/* window arithmetic of reference dem.c:101-171, no tile-count limit */
bool hz_window_compute(hz_window_t* win,
                       float viewer_lat, float viewer_lon,
                       int render_radius_cells, float render_radius_m,
                       bool SRTM1);

/* "<dir>/[NS]dd[EW]ddd.hgt" with ~/ expansion (reference dem.c:22-76) */
bool hz_tile_path(char* path, int bufsize, int tile_lat, int tile_lon, const char* datadir);

bool hz_tileset_open (hz_tileset_t* ts, const hz_window_t* win, const char* datadir);
void hz_tileset_close(hz_tileset_t* ts);

/* elevation of window sample (i,j): reference dem.c:264-309 semantics */
int16_t hz_tileset_sample(const hz_tileset_t* ts, int i, int j);

/* Fills mosaic[j*N + i] = hz_tileset_sample(i,j), N = 2*radius_cells, row j =
 * constant latitude, i fastest.  Row-wise byte-swapping copy, OpenMP over rows. */
void hz_tileset_build_mosaic(const hz_tileset_t* ts, int16_t* mosaic);

#ifdef __cplusplus
}
#endif
