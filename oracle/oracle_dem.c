/* oracle_dem.c - TEST INFRASTRUCTURE (see oracle.h).
 * CPU restatement of reference dem.c: which tiles a viewer-centred window
 * needs, and what elevation window sample (i,j) has.  Tiles are read whole
 * into memory (no mmap), sampling is the scalar formula of the reference.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static int tile_filename(char* out, size_t n, const char* dir, int lat, int lon)
{
    /* reference dem.c:30-75: N/S + 2 digits, E/W + 3 digits, ".hgt" */
    char ns = 'N', ew = 'E';
    if(lat < 0) { ns = 'S'; lat = -lat; }
    if(lon < 0) { ew = 'W'; lon = -lon; }
    if(dir[0] == '~' && dir[1] == '/')
    {
        const char* home = getenv("HOME");
        if(!home) return -1;
        return snprintf(out, n, "%s/%s/%c%.2d%c%.3d.hgt", home, dir+2, ns, lat, ew, lon) < (int)n ? 0 : -1;
    }
    return snprintf(out, n, "%s/%c%.2d%c%.3d.hgt", dir, ns, lat, ew, lon) < (int)n ? 0 : -1;
}

int orc_dem_open(orc_dem_t* d, float viewer_lat, float viewer_lon,
                 int radius_cells, float radius_m, const char* dir, int srtm1)
{
    memset(d, 0, sizeof(*d));
    /* reference dem.c:90-99 */
    if((radius_cells < 0 && radius_m < 0) || (radius_cells > 0 && radius_m > 0)) return -1;

    /* reference dem.c:101-104 */
    d->cells_per_deg = srtm1 ? 3600 : 1200;
    const int cpd = d->cells_per_deg;

    /* reference dem.c:106-127 */
    if(radius_cells > 0) d->radius_cells = radius_cells;
    else
    {
        const double Rearth = 6371000.0;
        const double c = cos(M_PI / 180.0 * viewer_lat);
        d->radius_cells = (int)(0.5 + (double)radius_m / (Rearth * M_PI/180. * c / (double)cpd));
    }

    const float v[2] = { viewer_lon, viewer_lat };
    for(int a=0; a<2; a++)
    {
        /* reference dem.c:143-152 (tgmath picks the float functions there) */
        int   icell_origin   = floorf(v[a] * cpd) - (d->radius_cells-1);
        float origin_lon_lat = (float)icell_origin / (float)cpd;
        d->origin_tile[a] = (int)floorf(origin_lon_lat);
        d->origin_cell[a] = (int)roundf( (origin_lon_lat - d->origin_tile[a]) * cpd );

        /* reference dem.c:162-171 */
        int cell_last = d->origin_cell[a] + d->radius_cells*2-1;
        int tile_last = cell_last / cpd;
        d->ntiles[a] = tile_last + 1;
        if(cell_last == tile_last*cpd) d->ntiles[a]--;
        if(d->ntiles[a] < 1) d->ntiles[a] = 1;
    }

    const size_t want = (size_t)(cpd+1)*(cpd+1)*2;     /* reference dem.c:129-132 */
    const int nt = d->ntiles[0]*d->ntiles[1];
    d->tiles = calloc(nt, sizeof(*d->tiles));
    if(!d->tiles) return -1;
    /* reference dem.c:183-240 */
    for(int tj=0; tj<d->ntiles[1]; tj++)
        for(int ti=0; ti<d->ntiles[0]; ti++)
        {
            char fn[1024];
            if(tile_filename(fn, sizeof(fn), dir, tj + d->origin_tile[1], ti + d->origin_tile[0]) != 0)
            { orc_dem_close(d); return -1; }
            FILE* f = fopen(fn, "rb");
            if(!f) continue;                       /* missing -> sea level */
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            if(sz == 0) { fclose(f); continue; }   /* empty -> sea level   */
            if((size_t)sz != want) { fclose(f); orc_dem_close(d); return -2; }
            unsigned char* buf = malloc(want);
            if(!buf || fread(buf, 1, want, f) != want) { free(buf); fclose(f); orc_dem_close(d); return -1; }
            fclose(f);
            d->tiles[ti + tj*d->ntiles[0]] = buf;
        }
    return 0;
}

void orc_dem_close(orc_dem_t* d)
{
    if(d->tiles)
    {
        const int nt = d->ntiles[0]*d->ntiles[1];
        for(int k=0; k<nt; k++) free(d->tiles[k]);
        free(d->tiles);
    }
    memset(d, 0, sizeof(*d));
}

int orc_dem_sample(const orc_dem_t* d, int i, int j)
{
    /* reference dem.c:264-309 */
    if(i < 0 || j < 0) return -1;
    int cell[2] = { i + d->origin_cell[0], j + d->origin_cell[1] };
    int tile[2];
    for(int a=0; a<2; a++)
    {
        tile[a]  = cell[a] / d->cells_per_deg;
        cell[a] -= tile[a] * d->cells_per_deg;
        /* neighbouring tiles share a row/column: serve in-tile index 0 from
         * the previous tile (reference dem.c:287-291).  When there is no
         * previous tile the reference indexes tile -1 (out of bounds); the
         * sample is read from the tile itself instead, where it is valid */
        if(cell[a] == 0 && tile[a] > 0)
        {
            tile[a]--;
            cell[a] = d->cells_per_deg;
        }
        if(tile[a] >= d->ntiles[a]) return -1;
    }
    const unsigned char* t = d->tiles[tile[0] + tile[1]*d->ntiles[0]];
    if(t == NULL) return 0;
    const uint32_t p = cell[0] + (d->cells_per_deg - cell[1])*(d->cells_per_deg+1);
    const int16_t z = (int16_t)((t[2*p] << 8) | t[2*p + 1]);
    return z < 0 ? 0 : z;
}

void orc_dem_mosaic(const orc_dem_t* d, int16_t* mosaic)
{
    const int N = 2*d->radius_cells;
    for(int j=0; j<N; j++)
        for(int i=0; i<N; i++)
            mosaic[(size_t)j*N + i] = (int16_t)orc_dem_sample(d, i, j);
}

void orc_view_move(orc_view_t* v, const orc_dem_t* d, float viewer_lat, float viewer_lon, float viewer_z)
{
    /* reference horizonator-lib.c:765-770 */
    v->viewer_cell_i = (viewer_lon - d->origin_tile[0]) * d->cells_per_deg - d->origin_cell[0];
    v->viewer_cell_j = (viewer_lat - d->origin_tile[1]) * d->cells_per_deg - d->origin_cell[1];
    /* reference horizonator-lib.c:775-789 */
    if(viewer_z < 0)
    {
        const int i0 = (int)floorf(v->viewer_cell_i);
        const int j0 = (int)floorf(v->viewer_cell_j);
        viewer_z =
            fmaxf( fmaxf(orc_dem_sample(d, i0,   j0  ), orc_dem_sample(d, i0+1, j0  )),
                   fmaxf(orc_dem_sample(d, i0,   j0+1), orc_dem_sample(d, i0+1, j0+1)) ) + 1.0;
    }
    v->viewer_z = viewer_z;
    /* reference horizonator-lib.c:799 */
    v->cos_viewer_lat = cosf( viewer_lat * M_PI / 180.0f );
    /* reference horizonator-lib.c:577 */
    v->deg_per_cell = 1.0f / (float)d->cells_per_deg;
}
