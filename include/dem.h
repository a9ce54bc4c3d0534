/* dem.h - SRTM tile mosaic access, MI355X build.
 *
 * ABI-compatible with the reference's dem.h (reference dem.h:8-66): same
 * struct layout, same four entry points, same semantics.  The struct is
 * public and caller-allocated, so its layout (including the fixed 4x4 tile
 * arrays) is part of the ABI and must not change.
 *
 * Grid convention (reference dem.h:32-42): the render window is a square of
 * (2*radius_cells)^2 samples addressed as (i,j), i growing east, j growing
 * north, (0,0) at the south-west corner.  The viewer sits between samples
 * radius_cells-1 and radius_cells on both axes.
 */
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Tile-grid capacity of the PUBLIC struct.  Part of the ABI (sizeof depends on
 * it), so it stays 4 exactly as in reference dem.h:8.  The render library
 * itself is not limited by it: horizonator_init() builds its device mosaic
 * through an internal loader with no tile-count cap (see hz_dem.c). */
#define max_Ndems_ij 4

typedef struct
{
    /* mmap'ed tiles, [lon index][lat index], NULL = sea level everywhere */
    unsigned char* dems      [max_Ndems_ij][max_Ndems_ij];
    size_t         mmap_sizes[max_Ndems_ij][max_Ndems_ij];
    int            mmap_fd   [max_Ndems_ij][max_Ndems_ij];

    /* integer (lon,lat) of the tile holding the SW corner of the window */
    int            origin_dem_lon_lat[2];

    /* sample index of the SW corner inside that tile */
    int            origin_dem_cellij [2];

    /* tiles spanned along (lon,lat) */
    int            Ndems_ij          [2];

    int radius_cells;

    /* 1200 for 3" SRTM, 3600 for 1" SRTM */
    int cells_per_deg;
} horizonator_dem_context_t;


/* Replaces reference dem.c:78-243.  Exactly one of render_radius_cells /
 * render_radius_m must be > 0.  Fails (false + message on stderr) if the
 * window needs more than max_Ndems_ij tiles along an axis, if a tile has the
 * wrong size, or if a path cannot be built.  A missing or zero-length tile is
 * not an error: it reads as elevation 0. */
bool horizonator_dem_init(horizonator_dem_context_t* ctx,
                          float viewer_lat,
                          float viewer_lon,
                          int   render_radius_cells,
                          float render_radius_m,
                          const char* datadir,
                          bool  SRTM1);

/* Replaces reference dem.c:245-261 */
void horizonator_dem_deinit(horizonator_dem_context_t* ctx);

/* Replaces reference dem.c:264-309.  Returns the elevation in metres, voids
 * and negatives clamped to 0; -1 for (i,j) outside the loaded tiles. */
int16_t horizonator_dem_sample(const horizonator_dem_context_t* ctx,
                               int i,
                               int j);

/* Replaces reference dem.c:313-330.  Inclusive lat/lon of the first and last
 * sample of the window. */
void horizonator_dem_bounds_latlon_deg(const horizonator_dem_context_t* ctx,
                                       float* lat0, float* lon0,
                                       float* lat1, float* lon1);

#ifdef __cplusplus
}
#endif
