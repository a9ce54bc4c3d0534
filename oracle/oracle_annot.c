/* oracle_annot.c - TEST INFRASTRUCTURE (see oracle.h).
 * Restates, NEARLY VERBATIM - the mixed float/double evaluation order is what has
 * to be reproduced -, the two passes the reference's annotator makes over the range
 * image (reference annotator.c:228-264 and :280-348) and the two library functions
 * they call (reference horizonator-lib.c:1053-1213).  PARITY UNPINNED by a run of
 * the reference: annotator.c needs cairo and libswscale, which this image lacks
 * (tests/caller_stubs holds stand-ins for LINKING the reference's CLI; a run
 * against stand-ins would pin nothing).  The device kernels are compared with this
 * file within 1e-6 degrees / 1e-3 pixels (tests/test_annot.py), not bit for bit.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "oracle.h"

/* reference horizonator-lib.c:1055-1060 */
static double unwrap_near_rad(double x, double near)
{
    double d = (x - near) / (2.*M_PI);
    return (d - round(d)) * 2.*M_PI + near;
}

/* reference horizonator-lib.c:1062-1095 */
static bool x_from_az(double* x, double* az_ndc_per_rad, double az_rad, double az_rad0, double az_rad1, int width)
{
    az_rad1 = unwrap_near_rad(az_rad1-az_rad0, M_PI) + az_rad0;
    const double az_rad_center = (az_rad0 + az_rad1)/2.;
    az_rad = unwrap_near_rad(az_rad, az_rad_center);
    const double _az_ndc_per_rad = 2.0 / (az_rad1 - az_rad0);
    const double az_ndc = (az_rad - az_rad_center) * _az_ndc_per_rad;
    if(! (-1. <= az_ndc && az_ndc <= 1.) ) return false;
    if(az_ndc_per_rad != NULL) *az_ndc_per_rad = _az_ndc_per_rad;
    *x = ( az_ndc + 1.)/2.*width  - 0.5;
    return true;
}

/* reference horizonator-lib.c:1097-1155 */
int orc_project(double* x, double* y, double* range,
                double lat_viewer, double cos_lat_viewer, double lon_viewer, double ele_viewer,
                double lat, double lon, double ele,
                double az_rad0, double az_rad1, int width, int height)
{
    const float Rearth = 6371000.0;
    const double dlat = (lat - lat_viewer)*M_PI/180;
    const double dlon = (lon - lon_viewer)*M_PI/180;
    const double east  = dlon * Rearth * cos_lat_viewer;
    const double north = dlat * Rearth;
    const double distance_sq_ne = east*east + north*north;
    double az_ndc_per_rad;
    if(!x_from_az(x, &az_ndc_per_rad, atan2(east, north), az_rad0, az_rad1, width)) return 0;
    const double h           = ele - ele_viewer;
    const double distance_ne = sqrt(distance_sq_ne);
    *range                   = sqrt(distance_sq_ne + h*h);
    const double aspect = (double)width / (double)height;
    const double el_ndc = atan2(h, distance_ne) * aspect * az_ndc_per_rad;
    if(! (-1. <= el_ndc && el_ndc <= 1.) ) return 0;
    *y = (-el_ndc + 1.)/2.*height - 0.5;
    return 1;
}

/* reference horizonator-lib.c:1157-1213 */
int orc_unproject(float* lat, float* lon, int x, int y, double range_enh, double range_en,
                  double lat_viewer, double cos_lat_viewer, double lon_viewer,
                  double az_deg0, double az_deg1, int width, int height)
{
    if( 1 != (range_enh > 0.) + (range_en > 0.) ) return 0;
    const float Rearth = 6371000.0;
    float az_ndc = ((float)x + 0.5f) / (float)width * 2.f - 1.f;
    float az     = (az_ndc * (az_deg1-az_deg0) / 2.f + (az_deg1+az_deg0)/2.f) * M_PI/180.0f;
    float e,n;
    if(range_en <= 0)
    {
        double aspect = (double)width / (double)height;
        double el_ndc = ((double)y + 0.5) / (double)height * 2. - 1.;
        double el     = el_ndc * (az_deg1-az_deg0) / 2. / aspect * M_PI/180.0;
        range_en = cos(el) * range_enh;
    }
    e = range_en * sinf(az);
    n = range_en * cosf(az);
    *lon = lon_viewer + e / Rearth / M_PI * 180. / cos_lat_viewer;
    *lat = lat_viewer + n / Rearth / M_PI * 180.;
    return 1;
}

/* reference annotator.c:228-264: lat/lon under the centre of every link cell
 * whose top-left pixel shows terrain; NaN elsewhere.  lat/lon: [ny][nx] with
 * nx, ny the trip counts of the reference's two loops.  Returns nx*ny. */
int orc_link_cells(const float* range_image, int width, int height, int cut_off_bottom_px,
                   int cell_width, int cell_height,
                   double lat, double lon, double az_deg0, double az_deg1,
                   float* out_lat, float* out_lon)
{
    const int height_out = height - cut_off_bottom_px;
    const double cos_lat = cos(lat * M_PI/180.);
    int nx = 0;
    for(int x=0; x<width-cell_width; x += cell_width) nx++;
    int cy = 0;
    for(int y=0; y<height_out-cell_height; y += cell_height, cy++)
    {
        int cx = 0;
        for(int x=0; x<width-cell_width; x += cell_width, cx++)
        {
            float* la = &out_lat[(size_t)cy*nx + cx];
            float* lo = &out_lon[(size_t)cy*nx + cx];
            *la = NAN; *lo = NAN;
            const float range = range_image[width*y + x];
            if(range <= 0.0f) continue;
            float lat_cell, lon_cell;
            if(!orc_unproject(&lat_cell, &lon_cell, x+cell_width/2, y+cell_height/2, range, -1.,
                              lat, cos_lat, lon, az_deg0, az_deg1, width, height))
                continue;
            *la = lat_cell; *lo = lon_cell;
        }
    }
    return nx*cy;
}

#define MAX_MARKER_DIST 100000.0
#define MIN_MARKER_DIST 500.0
#define FUZZ_RANGE   500.
#define FUZZ_PIXEL_Y 6

/* reference annotator.c:280-348.  pois: lat, lon, ele_m triples. */
void orc_poi_visibility(const float* range_image, int width, int height, int cut_off_bottom_px,
                        const float* pois, int Npois,
                        double lat, double lon, double az_deg0, double az_deg1, double ele_m,
                        uint8_t* visible, float* label_x, float* label_y)
{
    const int height_out = height - cut_off_bottom_px;
    const double cos_lat = cos(lat * M_PI/180.);
    for(int i=0; i<Npois; i++)
    {
        visible[i] = 0; label_x[i] = 0.f; label_y[i] = 0.f;
        double crosshair_x, crosshair_y, range_have;
        if(!orc_project(&crosshair_x, &crosshair_y, &range_have, lat, cos_lat, lon, ele_m,
                        pois[3*i+0], pois[3*i+1], pois[3*i+2],
                        az_deg0 * M_PI/180., az_deg1 * M_PI/180., width, height))
            continue;
        if(range_have < MIN_MARKER_DIST || range_have > MAX_MARKER_DIST) continue;

        int    fuzz_nearest = 0;
        double err_nearest  = DBL_MAX;
        for(int fuzz = -FUZZ_PIXEL_Y; fuzz < FUZZ_PIXEL_Y; fuzz++)
        {
            if(crosshair_y + (double)fuzz < 0) continue;
            if(crosshair_y + (double)fuzz >= height_out) break;
            /* the reference indexes the image here without further checks; rows
             * and columns outside it are skipped instead */
            const int yy = (int)round(crosshair_y) + fuzz, xx = (int)round(crosshair_x);
            if(yy < 0 || yy >= height || xx < 0 || xx >= width) continue;
            const float range = range_image[width*yy + xx];
            if(range <= 0.0f) continue;
            double err = fabs(range_have - range);
            if(err < err_nearest) { err_nearest = err; fuzz_nearest = fuzz; }
            else break;
        }
        if(err_nearest < FUZZ_RANGE)
        {
            visible[i] = 1;
            label_x[i] = crosshair_x;
            label_y[i] = crosshair_y + (float)fuzz_nearest;
        }
    }
}
