/* how fast can this box's host cores fill / populate 448 MB, by thread count and placement?  (diagnostics behind
 * DESIGN.md's host path section: gcc -O2 -fopenmp tools/hostfill_bench.c horizonator_amd/csrc/hz_scatter.c -Ihorizonator_amd/csrc) */
#define _GNU_SOURCE
#include <omp.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include "hz_scatter.h"
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec*1e3 + t.tv_nsec*1e-6; }
static int parse_cpulist(const char* path, cpu_set_t* set)
{
    FILE* f = fopen(path, "r"); if(!f) return -1;
    char buf[4096]; if(!fgets(buf, sizeof(buf), f)) { fclose(f); return -1; } fclose(f);
    CPU_ZERO(set); int n = 0;
    for(char* tok = strtok(buf, ",\n"); tok; tok = strtok(NULL, ",\n"))
    { int a, b; if(sscanf(tok, "%d-%d", &a, &b) == 2) { for(int c=a; c<=b; c++) { CPU_SET(c, set); n++; } } else if(sscanf(tok, "%d", &a) == 1) { CPU_SET(a, set); n++; } }
    return n;
}
int main(int argc, char** argv)
{
    const size_t n = (size_t)448 << 20;
    for(int node=-1; node<4; node++)
    {
        cpu_set_t set; char path[128];
        if(node >= 0) { snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node); if(parse_cpulist(path, &set) <= 0) continue; }
        const int ts[] = { 8, 12, 16, 24, 32, 64 };
        for(int ti=0; ti<6; ti++)
        {
            const int T = ts[ti];
            omp_set_num_threads(T);
            #pragma omp parallel
            { if(node >= 0) sched_setaffinity(0, sizeof(set), &set); }
            /* fresh pages, populated by the threads, then filled; then filled again (kept buffer) */
            unsigned char* p = mmap(NULL, n, PROT_READ|PROT_WRITE, MAP_PRIVATE|MAP_ANONYMOUS, -1, 0);
            double t0 = now();
            #pragma omp parallel for schedule(static)
            for(int k=0; k<64; k++) madvise(p + n/64*k, n/64, MADV_POPULATE_WRITE);
            double t1 = now();
            #pragma omp parallel for schedule(dynamic)
            for(int k=0; k<128; k++) hz_sky_fill(p, n/128*k, n/128*(k+1), k & 1);
            double t2 = now();
            #pragma omp parallel for schedule(dynamic)
            for(int k=0; k<128; k++) hz_sky_fill(p, n/128*k, n/128*(k+1), k & 1);
            double t3 = now();
            #pragma omp parallel for schedule(dynamic)
            for(int k=0; k<128; k++) memset(p + n/128*k, 1, n/128);
            double t4 = now();
            printf("node %2d threads %2d: populate %.2f ms, fill %.2f ms (%.0f GB/s), again %.2f ms (%.0f GB/s), memset %.2f ms\n", node, T, t1-t0, t2-t1, n/(t2-t1)*1e-6, t3-t2, n/(t3-t2)*1e-6, t4-t3);
            munmap(p, n);
            /* the same with transparent huge pages asked for (where the system offers them on request: madvise mode) */
            p = mmap(NULL, n + (2 << 20), PROT_READ|PROT_WRITE, MAP_PRIVATE|MAP_ANONYMOUS, -1, 0);
            unsigned char* q = (unsigned char*)(((unsigned long)p + (2 << 20) - 1) & ~(unsigned long)((2 << 20) - 1));
            madvise(q, n, 14 /* MADV_HUGEPAGE */);
            double h0 = now();
            #pragma omp parallel for schedule(static)
            for(int k=0; k<64; k++) madvise(q + n/64*k, n/64, MADV_POPULATE_WRITE);
            double h1 = now();
            #pragma omp parallel for schedule(dynamic)
            for(int k=0; k<128; k++) hz_sky_fill(q, n/128*k, n/128*(k+1), k & 1);
            double h2 = now();
            munmap(p, n + (2 << 20));
            /* ... and filled without populating first (the first touch of every page is a streaming store) */
            p = mmap(NULL, n + (2 << 20), PROT_READ|PROT_WRITE, MAP_PRIVATE|MAP_ANONYMOUS, -1, 0);
            q = (unsigned char*)(((unsigned long)p + (2 << 20) - 1) & ~(unsigned long)((2 << 20) - 1));
            madvise(q, n, 14);
            double g0 = now();
            #pragma omp parallel for schedule(dynamic)
            for(int k=0; k<128; k++) hz_sky_fill(q, n/128*k, n/128*(k+1), k & 1);
            double g1 = now();
            munmap(p, n + (2 << 20));
            printf("node %2d threads %2d: with MADV_HUGEPAGE: populate %.2f ms, then fill %.2f ms; fill of untouched pages %.2f ms\n", node, T, h1-h0, h2-h1, g1-g0);
        }
    }
    return 0;
}
