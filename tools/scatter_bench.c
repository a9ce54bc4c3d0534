/* how fast do this box's host cores put blobs of terrain pixels into a 16000 x 4000 panorama (hz_scatter.c: BGR + ranges from
 * z24<<8 | red8), by thread count, with streaming and with ordinary stores - and beside a sky fill running on other threads?
 * (diagnostics behind DESIGN.md's host path section)
 *   gcc -O2 -fopenmp -ffp-contract=off tools/scatter_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc -lm */
#define _GNU_SOURCE
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "hz_scatter.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec*1e3 + t.tv_nsec*1e-6; }
int main(void)
{
    const int W = 16000, H = 4000, y_top = 2400;         /* terrain: the lower 40 % of the image, every pixel */
    unsigned char* bgr = malloc((size_t)W*H*3); float* rng = malloc((size_t)W*H*4); float* tanel = malloc(H*4);
    for(int y=0; y<H; y++) tanel[y] = 0.001f*(y - H/2);
    memset(bgr, 1, (size_t)W*H*3); memset(rng, 1, (size_t)W*H*4);
    /* the blobs, as k_pack_host would write them: 4 rows x 2048 columns (the last tile of a row: 1664), all terrain */
    const int tiles_x = (W + HZ_BLOB_COLS-1)/HZ_BLOB_COLS, tiles_y = (H - y_top)/HZ_BLOB_ROWS, nblobs = tiles_x*tiles_y;
    uint32_t** blobs = malloc(nblobs*sizeof(*blobs));
    size_t words = 0;
    for(int ty=0; ty<tiles_y; ty++)
        for(int tx=0; tx<tiles_x; tx++)
        {
            const int x0 = tx*HZ_BLOB_COLS, n = W - x0 < HZ_BLOB_COLS ? W - x0 : HZ_BLOB_COLS, mw = (n + 31)/32;
            const size_t size = (HZ_BLOB_HDR + 4*mw + 4*(size_t)n + 3) & ~(size_t)3;
            uint32_t* b = malloc(size*4);
            b[0] = (uint32_t)(y_top + 4*ty) | (HZ_BLOB_PACKED << 16); b[1] = x0; b[2] = b[3] = b[4] = b[5] = n; b[6] = size; b[7] = n;
            for(int k=0; k<4*mw; k++) b[8 + k] = 0xFFFFFFFFu;
            if(n & 31) for(int r=0; r<4; r++) b[8 + r*mw + mw-1] = (1u << (n & 31)) - 1u;
            for(size_t k=0; k<4*(size_t)n; k++) b[8 + 4*mw + k] = (uint32_t)((k*2654435761u) >> 8) << 8 | (k & 255);
            blobs[ty*tiles_x + tx] = b; words += size;
        }
    hz_scatter_dst_t dst = { W, H, bgr, rng, NULL, NULL, tanel, 100.f, 600000.f };
    printf("%d blobs, %.1f MB of blobs, %.1f M terrain pixels -> %.1f MB written\n", nblobs, words*4e-6, 1e-6*(double)W*(H - y_top), 7e-6*(double)W*(H - y_top));
    const int ts[] = { 8, 16, 24, 32, 48, 64, 96 };
    for(int nt=1; nt>=0; nt--)
        for(int ti=0; ti<7; ti++)
        {
            hz_scatter_set_streaming(nt);
            omp_set_num_threads(ts[ti]);
            double best = 1e9;
            for(int rep=0; rep<4; rep++)
            {
                const double t0 = now();
                #pragma omp parallel for schedule(dynamic, 4)
                for(int k=0; k<nblobs; k++) if(hz_blob_scatter(blobs[k], &dst) != 0) abort();
                const double t1 = now();
                if(t1 - t0 < best) best = t1 - t0;
            }
            printf("%s stores, %2d threads: %.2f ms (%.0f GB/s written, %.2f ns per pixel and thread)\n", nt ? "streaming" : "ordinary ", ts[ti], best,
                   7e-6*(double)W*(H - y_top)/best, best*1e6*ts[ti]/((double)W*(H - y_top)));
        }
    /* the sky fill of the whole image beside the scatter (what a call in several sectors does): T threads each */
    for(int T=16; T<=48; T+=16)
    {
        hz_scatter_set_streaming(1);
        omp_set_num_threads(2*T);
        omp_set_nested(0);
        const double t0 = now();
        double t_fill = 0, t_scatter = 0;
        #pragma omp parallel
        {
            const int me = omp_get_thread_num();
            if(me < T)
            {
                for(int k=me; k<256; k+=T) { hz_sky_fill(bgr, (size_t)W*H*3/256*k, (size_t)W*H*3/256*(k+1), HZ_SKY_BGR); hz_sky_fill((unsigned char*)rng, (size_t)W*H*4/256*k, (size_t)W*H*4/256*(k+1), HZ_SKY_RANGES); }
                #pragma omp critical
                { const double t = now() - t0; if(t > t_fill) t_fill = t; }
            }
            else
            {
                for(int k=me-T; k<nblobs; k+=T) if(hz_blob_scatter(blobs[k], &dst) != 0) abort();
                #pragma omp critical
                { const double t = now() - t0; if(t > t_scatter) t_scatter = t; }
            }
        }
        printf("side by side, %d + %d threads: fill of 448 MB done after %.2f ms, scatter after %.2f ms\n", T, T, t_fill, t_scatter);
    }
    return 0;
}
