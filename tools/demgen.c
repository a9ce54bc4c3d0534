/* demgen.c - deterministic synthetic SRTM tiles for tests and benchmarks.
 *
 * There is no SRTM data in this environment (and none ships with the
 * reference), so every test, golden vector and benchmark runs on this
 * closed-form terrain.  Elevation is a function of
 * absolute latitude/longitude, so the row/column shared by neighbouring tiles
 * agrees, as it does in real SRTM data.
 *
 *   z = clip(rint(1200 + 900 sin(23 lat) cos(17 lon) + 500 sin(97 lat + 61 lon)
 *                      + 150 sin(400 lat) sin(380 lon) [+ rough]), 0, 8000)
 *
 * lat/lon in degrees, used directly as the radian argument, float64.
 * "rough" adds a per-sample hash noise of +-30 m (seed 1234) that stresses
 * silhouettes.  File format: (cpd+1)^2 big-endian int16, row 0 = north edge
 * (what reference dem.c:300-308 reads).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* noise keyed on the absolute sample position so shared tile edges agree */
static double rough_term(long long gi, long long gj)
{
    const uint32_t h = hash32((uint32_t)(gi*2654435761LL) ^ hash32((uint32_t)gj + 1234u));
    return ((double)(h & 0xFFFF) / 65535.0 * 2.0 - 1.0) * 30.0;
}

static void tile_name(char* out, size_t n, const char* dir, int lat0, int lon0)
{
    snprintf(out, n, "%s/%c%02d%c%03d.hgt", dir,
             lat0 >= 0 ? 'N' : 'S', abs(lat0), lon0 >= 0 ? 'E' : 'W', abs(lon0));
}

/* elevations of tile (lat0,lon0) into z[(cpd+1)^2], row 0 = north, host order */
void hz_demgen_tile_values(int16_t* z, int lat0, int lon0, int cpd, int rough)
{
    const int w = cpd+1;
    double* s23  = malloc(sizeof(double)*w);
    double* s400 = malloc(sizeof(double)*w);
    double* c17  = malloc(sizeof(double)*w);
    double* s380 = malloc(sizeof(double)*w);
    for(int r=0; r<w; r++)
    {
        const double lat = (double)lat0 + 1.0 - (double)r/(double)cpd;
        s23[r] = sin(23.0*lat); s400[r] = sin(400.0*lat);
    }
    for(int c=0; c<w; c++)
    {
        const double lon = (double)lon0 + (double)c/(double)cpd;
        c17[c] = cos(17.0*lon); s380[c] = sin(380.0*lon);
    }
    #pragma omp parallel for schedule(static)
    for(int r=0; r<w; r++)
    {
        const double lat = (double)lat0 + 1.0 - (double)r/(double)cpd;
        for(int c=0; c<w; c++)
        {
            const double lon = (double)lon0 + (double)c/(double)cpd;
            double v = 1200.0 + 900.0*s23[r]*c17[c] + 500.0*sin(97.0*lat + 61.0*lon) + 150.0*s400[r]*s380[c];
            if(rough)
                v += rough_term((long long)lon0*cpd + c, ((long long)lat0+1)*cpd - r);
            v = rint(v);
            if(v < 0.0) v = 0.0;
            if(v > 8000.0) v = 8000.0;
            z[(size_t)r*w + c] = (int16_t)v;
        }
    }
    free(s23); free(s400); free(c17); free(s380);
}

/* returns 0 on success */
int hz_demgen_write_tile(const char* dir, int lat0, int lon0, int srtm1, int rough)
{
    const int cpd = srtm1 ? 3600 : 1200;
    const int w = cpd+1;
    int16_t* z = malloc(sizeof(int16_t)*(size_t)w*w);
    if(!z) return -1;
    hz_demgen_tile_values(z, lat0, lon0, cpd, rough);
    unsigned char* be = (unsigned char*)z;      /* swap in place */
    for(size_t k=0; k<(size_t)w*w; k++)
    {
        const uint16_t v = (uint16_t)z[k];
        be[2*k] = v >> 8; be[2*k+1] = v & 0xFF;
    }
    char path[1024];
    tile_name(path, sizeof(path), dir, lat0, lon0);
    FILE* f = fopen(path, "wb");
    if(!f) { free(z); return -2; }
    const size_t n = fwrite(be, 2, (size_t)w*w, f);
    fclose(f);
    free(z);
    return n == (size_t)w*w ? 0 : -3;
}

/* writes every tile with lat_lo <= lat0 <= lat_hi, lon_lo <= lon0 <= lon_hi
 * that does not exist yet with the right size; returns the number of tiles
 * written, or <0 on error */
int hz_demgen_write_region(const char* dir, int lat_lo, int lat_hi, int lon_lo, int lon_hi,
                           int srtm1, int rough)
{
    const int cpd = srtm1 ? 3600 : 1200;
    const long long want = (long long)(cpd+1)*(cpd+1)*2;
    int written = 0;
    mkdir(dir, 0777);
    for(int la=lat_lo; la<=lat_hi; la++)
        for(int lo=lon_lo; lo<=lon_hi; lo++)
        {
            char path[1024];
            tile_name(path, sizeof(path), dir, la, lo);
            struct stat sb;
            if(stat(path, &sb) == 0 && (long long)sb.st_size == want) continue;
            const int rc = hz_demgen_write_tile(dir, la, lo, srtm1, rough);
            if(rc != 0) return rc;
            written++;
        }
    return written;
}

#ifdef DEMGEN_MAIN
int main(int argc, char** argv)
{
    if(argc < 6)
    {
        fprintf(stderr, "usage: %s DIR LAT_LO LAT_HI LON_LO LON_HI [srtm1=0] [rough=0]\n", argv[0]);
        return 2;
    }
    const int rc = hz_demgen_write_region(argv[1], atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]),
                                          argc > 6 ? atoi(argv[6]) : 0, argc > 7 ? atoi(argv[7]) : 0);
    fprintf(stderr, "demgen: %d tile(s) written to %s\n", rc, argv[1]);
    return rc < 0;
}
#endif
