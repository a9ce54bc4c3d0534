/* hz_ingest.cpp - SRTM tiles -> the int16 mosaic in HBM, decoded on the device ("next" row N3; reference dem.c:264-309
 * for the sampling, horizonator-lib.c:403-485 for what the reference pushes into its VBO instead).
 *
 * The host path of round 1 (hz_dem.c: hz_tileset_build_mosaic) walks the memory-mapped tiles with the host's cores,
 * writes a second copy of the DEM into pageable memory and has the runtime upload that - every byte faulted in twice and
 * copied three times; 11 x 11 SRTM1 tiles (3.1 GB) made horizonator_init() take 8.4 s.  Here the tiles' raw bytes go
 * through pinned memory in pieces - the host threads of hz_pool.h copy piece k+1 out of the page cache while the copy
 * engine moves piece k at the link's rate - and k_ingest (hz_kernels.hip) does the byte swap, the north-south flip, the
 * shared-edge rule and the void clamp for the whole window in one launch.  The pinned memory is the landing area of the
 * host path where the context has one already (hz_hip_host_prepare: no second pinned allocation, 0.25 ms per MB);
 * the default of horizonator_init() since round 6 (HORIZONATOR_INGEST=host: the old way). */
#include "hz_dev.h"
#include "hz_pool.h"

extern "C" int hz_hip_ingest_tiles(hz_dev_t* d, const unsigned char* const* tiles,
                                   int ntx, int nty, int cpd, int oc_x, int oc_y)
{
    HZ_ON_DEVICE(d);
    const int nt = ntx*nty;
    const size_t tile_bytes = (size_t)(cpd+1)*(cpd+1)*2;
    if(nt <= 0 || !tiles) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_ingest_tiles: no tiles"); return -1; }
    HZ_CHECK(hz_sync_all(d));           /* draws in flight still read the old mosaic */
    int npresent = 0;
    for(int k=0; k<nt; k++) if(tiles[k]) npresent++;
    unsigned char* d_raw = NULL;        /* the present tiles' bytes, one after the other */
    unsigned char** d_ptrs = NULL;
    unsigned char* own_stage = NULL;
    std::vector<unsigned char*> h_ptrs((size_t)nt, (unsigned char*)NULL);
    int rc = -1;
    hipEvent_t ev[3] = { NULL, NULL, NULL };
    do {
        if(npresent && hipMalloc(&d_raw, (size_t)npresent*tile_bytes) != hipSuccess) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_ingest_tiles: no device memory for %d tiles", npresent); break; }
        if(hipMalloc(&d_ptrs, (size_t)nt*sizeof(*d_ptrs)) != hipSuccess) break;
        for(int k=0, at=0; k<nt; k++) if(tiles[k]) h_ptrs[(size_t)k] = d_raw + (size_t)(at++)*tile_bytes;
        /* pinned staging: three pieces of the host path's landing area, or of an allocation of our own */
        unsigned char* stage = NULL; size_t stage_bytes = 0;
        hz_hostpath_landing(d, &stage, &stage_bytes);
        if(stage_bytes < ((size_t)3 << 20))
        {
            stage_bytes = (size_t)24 << 20;
            if(hipHostMalloc((void**)&own_stage, stage_bytes, hipHostMallocDefault) != hipSuccess) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_ingest_tiles: no pinned memory"); break; }
            stage = own_stage;
        }
        size_t piece = stage_bytes/3 & ~(size_t)4095;
        if(piece > ((size_t)64 << 20)) piece = (size_t)64 << 20;
        bool ok = true;
        for(int k=0; k<3 && ok; k++) ok = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
        if(!ok) break;
        hz_copy_pool* pool = copy_pool(hz_gpu_numa_node(d));
        const size_t total = (size_t)npresent*tile_bytes;
        std::vector<const unsigned char*> src;      /* the present tiles, in the order of d_raw */
        for(int k=0; k<nt; k++) if(tiles[k]) src.push_back(tiles[k]);
        size_t turn = 0;
        for(size_t off = 0; off < total && ok; off += piece, turn++)
        {
            const size_t n = total - off < piece ? total - off : piece;
            unsigned char* buf = stage + (turn % 3)*piece;
            if(turn >= 3) ok = hipEventSynchronize(ev[turn % 3]) == hipSuccess;     /* the copy engine is done with this piece of staging */
            if(!ok) break;
            /* the bytes [off, off+n) of the concatenated tiles: the host threads copy them out of the mapped files */
            hz_copy_pool::batch_t copied = { 0 };
            for(size_t o = off; o < off + n; )
            {
                const size_t t = o/tile_bytes, in = o - t*tile_bytes;
                const size_t len = tile_bytes - in < off + n - o ? tile_bytes - in : off + n - o;
                pool->push(&copied, buf + (o - off), src[t] + in, len, (size_t)1 << 20);
                o += len;
            }
            pool->wait(&copied);
            ok = hipMemcpyAsync(d_raw + off, buf, n, hipMemcpyHostToDevice, d->stream) == hipSuccess
              && hipEventRecord(ev[turn % 3], d->stream) == hipSuccess;
        }
        if(!ok) { snprintf(g_last_error, sizeof(g_last_error), "hz_hip_ingest_tiles: tile upload failed"); break; }
        if(hipMemcpyAsync(d_ptrs, h_ptrs.data(), (size_t)nt*sizeof(*d_ptrs), hipMemcpyHostToDevice, d->stream) != hipSuccess) break;
        dim3 grid((d->N + 255)/256, d->N);
        hzk_ingest(grid, dim3(256), d->stream, (const unsigned char* const*)d_ptrs, d->d_mosaic, d->N, ntx, nty, cpd, oc_x, oc_y);
        if(hipGetLastError() != hipSuccess) break;
        if(hipStreamSynchronize(d->stream) != hipSuccess) break;
        d->adapt.have_view = 0;         /* (as hz_hip_upload_mosaic: what was observed, and cached, was the old terrain's) */
        d->vc.state = 0;
        rc = 0;
    } while(0);
    (void)hipStreamSynchronize(d->stream);
    for(int k=0; k<3; k++) if(ev[k]) (void)hipEventDestroy(ev[k]);
    (void)hipFree(d_raw); (void)hipFree(d_ptrs);
    if(own_stage) (void)hipHostFree(own_stage);
    (void)hipGetLastError();
    return rc;
}
