/* hz_convert.cpp - what is made of a finished draw, on the device: the readback conversion of reference
 * horizonator-lib.c:911-1051 into DEVICE buffers (k_resolve4), packed and sparse strips for the multi-GPU gather and their
 * conversion on the gathering rank, pick's depth read, the annotator's two passes - the C-ABI of include/hz_hip.h from
 * hz_hip_resolve() to hz_hip_poi_visibility().  Results into HOST memory: hz_hostpath.cpp. */
#include "hz_dev.h"

/* The per-row tan(elevation) table only changes with the azimuth extents: a
 * table equal to the resident one is not sent again (a host->device copy from
 * pageable memory would otherwise stall the host on the stream once per render) */
int hz_upload_tanel(hz_dev_t* d, const float* tanel)
{
    if(!tanel) { snprintf(g_last_error, sizeof(g_last_error), "a tanel table is required"); return -1; }
    const size_t bytes = (size_t)d->H*sizeof(float);
    if(d->tanel_resident && memcmp(d->h_tanel, tanel, bytes) == 0) return 0;
    /* a different table (the azimuth extents changed): nothing queued on either
     * stream may still read the old one, and both streams must see the new one */
    HZ_CHECK(hz_sync_all(d));
    memcpy(d->h_tanel, tanel, bytes);
    HZ_CHECK(hipMemcpy(d->d_tanel, d->h_tanel, bytes, hipMemcpyHostToDevice));
    d->tanel_resident = 1;
    return 0;
}

/* conversions of the last draw run on rstream, behind that draw */
int hz_rstream_after_draw(hz_dev_t* d)
{
    HZ_CHECK(hipStreamWaitEvent(d->rstream, d->ev_drawn, 0));
    return 0;
}

/* the conversion of the last draw into DEVICE buffers; nbands > 1 (wide path only): in that many bands of
 * rows, top first, ev_band[k] recorded on rstream behind band k - copy_out lets the first band's bytes leave
 * for the host while the others are still being converted */
int hz_resolve_impl(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                        unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24, int nbands, hipEvent_t* ev_band, int* band_rows)
{
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(ranges)
    {
        if(!tanel)
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve: ranges requested without a tanel table");
            return -1;
        }
        if(hz_upload_tanel(d, tanel) != 0) return -1;
    }
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    /* the textured resolve reads the framebuffer after this kernel: no fused clear then */
    const bool clears = d->env.resolve_clears && !(d->tex_on && bgr);
    unsigned int* const qa = d->d_big_counters_s[d->fbi], * const qb = d->d_big_counters_s[HZ_NFB + d->fbi];    /* emptied with the framebuffer */
    const bool wide = (SW % 4) == 0 && (((uintptr_t)bgr | (uintptr_t)ranges | (uintptr_t)index | (uintptr_t)z24 | (uintptr_t)d->d_fb) & 15u) == 0;
    if(!wide || (d->tex_on && bgr) || nbands < 1) nbands = 1;
    if(nbands > d->H) nbands = d->H;
    if(band_rows) *band_rows = (d->H + nbands-1)/nbands;
    if(wide)
    {
        const int rows = (d->H + nbands-1)/nbands;
        for(int k=0; k<nbands; k++)
        {
            const int yo0 = k*rows, yo1 = (k+1)*rows < d->H ? (k+1)*rows : d->H;
            const dim3 grid((unsigned)((SW/4 + 255)/256), (unsigned)(yo1 - yo0 < 2048 ? yo1 - yo0 : 2048));
            if(clears)
                hzk_resolve4(true, grid, dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride, k == 0 ? qa : (unsigned int*)NULL, qb, yo0, yo1, 1);
            else
                hzk_resolve4(false, grid, dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, d->d_touched[d->fbi], d->seg_stride, (unsigned int*)NULL, (unsigned int*)NULL, yo0, yo1, 1);
            HZ_CHECK(hipGetLastError());
            if(ev_band) HZ_CHECK(hipEventRecord(ev_band[k], d->rstream));
        }
    }
    else
    {
        if(clears)
            hzk_resolve(true, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, qa, qb);
        else
            hzk_resolve(false, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, (const float*)d->d_tanel, bgr, ranges, index, z24, SW, d->H, view->znear, view->zfar, qa, qb);
        HZ_CHECK(hipGetLastError());
    }
    if(clears && hz_fb_mark_consumed(d) != 0) return -1;
    if(d->tex_on && bgr)
    {
        /* reference fragment.glsl:17-22 instead of :15-16 for the terrain pixels */
        const hz_params_t p = hz_make_params(d, view);
        size_t nchunks = (npix + TX_CHUNK-1)/TX_CHUNK;
        if(nchunks > 256*64) nchunks = 256*64;
        hzk_shade_tex(dim3((unsigned)nchunks), dim3(64), d->rstream, (const unsigned long long*)d->d_fb, (const int16_t*)d->d_mosaic, (const uint32_t*)d->d_texels, d->tex, bgr, p);
        HZ_CHECK(hipGetLastError());
    }
    if(!wide && ev_band) HZ_CHECK(hipEventRecord(ev_band[0], d->rstream));
    else if(wide && nbands == 1 && ev_band && d->tex_on && bgr) HZ_CHECK(hipEventRecord(ev_band[0], d->rstream));   /* (behind the shading kernel) */
    if(prof)
    {
        HZ_CHECK(hipEventRecord(d->ev[5], d->rstream));
        d->have_times = 2;
    }
    return nbands;
}

extern "C" int hz_hip_resolve(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                              unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    return hz_resolve_impl(d, view, tanel, bgr, ranges, index, z24, 1, NULL, NULL) < 0 ? -1 : 0;
}

/* the draw's result as one word per pixel, z24<<8 | red8, top row first:
 * what a rank sends to the gathering rank (d_packed: DEVICE, [H][sector width]) */
extern "C" int hz_hip_pack(hz_dev_t* d, uint32_t* d_packed)
{
    HZ_ON_DEVICE(d);
    if(d->tex_on)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_pack: packed strips carry the shade only, not a textured colour");
        return -1;
    }
    const int SW = d->col1 - d->col0;
    const bool prof = d->profiling != 0;
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    const size_t npix = (size_t)SW*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    if(d->env.resolve_clears)
    {
        hzk_pack(true, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, d_packed, SW, d->H, d->d_big_counters_s[d->fbi], d->d_big_counters_s[HZ_NFB + d->fbi]);
        HZ_CHECK(hipGetLastError());
        if(hz_fb_mark_consumed(d) != 0) return -1;
    }
    else
        hzk_pack(false, dim3((unsigned)nblocks), dim3(256), d->rstream, d->d_fb, d_packed, SW, d->H, (unsigned int*)NULL, (unsigned int*)NULL);
    HZ_CHECK(hipGetLastError());
    if(prof) { HZ_CHECK(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
    return 0;
}

/* The readback conversion on packed words, wherever they were drawn: columns
 * [0,ncols) of d_packed[H][stride] become columns [out_col0, out_col0+ncols) of
 * the FULL-width outputs d_bgr[H][W][3] / d_ranges[H][W] (either may be NULL). */
extern "C" int hz_hip_resolve_packed(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     const uint32_t* d_packed, int stride, int ncols, int out_col0,
                                     unsigned char* d_bgr, float* d_ranges)
{
    HZ_ON_DEVICE(d);
    if(ncols <= 0 || stride < ncols || out_col0 < 0 || out_col0 + ncols > d->W)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_packed: columns [%d,%d) do not fit a %d-wide image",
                 out_col0, out_col0 + ncols, d->W);
        return -1;
    }
    if(d_ranges && hz_upload_tanel(d, tanel) != 0) return -1;
    const size_t npix = (size_t)ncols*d->H;
    size_t nblocks = (npix + 255)/256;
    if(nblocks > 256*32) nblocks = 256*32;
    hzk_resolve_packed(dim3((unsigned)nblocks), dim3(256), d->rstream, d_packed, stride, ncols, (const float*)d->d_tanel, d_bgr, d_ranges, d->W, out_col0, d->H, view->znear, view->zfar);
    HZ_CHECK(hipGetLastError());
    return 0;
}

/* the draw's result as a sparse strip (see k_pack_sparse): d_out must hold
 * 1 + H + H*mask_stride + H*(sector width) words; the first word ends up as the
 * number of terrain pixels T, and only the first 1 + H + H*mask_stride + T words
 * carry information.  mask_stride >= ceil(sector width / 32). */
extern "C" int hz_hip_pack_sparse(hz_dev_t* d, uint32_t* d_out, int mask_stride)
{
    HZ_ON_DEVICE(d);
    const int SW = d->col1 - d->col0;
    if(d->tex_on || mask_stride < (SW + 31)/32 || SW > SP_MAXIT*256)
    {
        snprintf(g_last_error, sizeof(g_last_error), d->tex_on ? "hz_hip_pack_sparse: strips carry the shade only, not a textured colour"
                                                    : SW > SP_MAXIT*256 ? "hz_hip_pack_sparse: sectors of up to 65536 columns"
                                                               : "hz_hip_pack_sparse: mask stride too small");
        return -1;
    }
    const bool prof = d->profiling != 0;
    if(hz_fb_refill(d) != 0) return -1;
    if(hz_rstream_after_draw(d) != 0) return -1;
    if(prof) HZ_CHECK(hipEventRecord(d->ev[4], d->rstream));
    HZ_CHECK(hipMemsetAsync(d_out, 0, sizeof(uint32_t), d->rstream));
    if(d->env.resolve_clears)
    {
        hzk_pack_sparse(true, dim3((unsigned)((d->H + SP_WAVES-1)/SP_WAVES)), dim3(64*SP_WAVES), d->rstream, d->d_fb, d_out, SW, d->H, mask_stride, d->d_touched[d->fbi], d->seg_stride, d->d_big_counters_s[d->fbi], d->d_big_counters_s[HZ_NFB + d->fbi]);
        HZ_CHECK(hipGetLastError());
        if(hz_fb_mark_consumed(d) != 0) return -1;
    }
    else
        hzk_pack_sparse(false, dim3((unsigned)((d->H + SP_WAVES-1)/SP_WAVES)), dim3(64*SP_WAVES), d->rstream, d->d_fb, d_out, SW, d->H, mask_stride, d->d_touched[d->fbi], d->seg_stride, (unsigned int*)NULL, (unsigned int*)NULL);
    HZ_CHECK(hipGetLastError());
    if(prof) { HZ_CHECK(hipEventRecord(d->ev[5], d->rstream)); d->have_times = 2; }
    return 0;
}

extern "C" int hz_hip_resolve_sparse_strips(hz_dev_t* d, const hz_view_t* view, const float* tanel, int nstrips,
                                            const uint32_t* const* d_in, int mask_stride, const int* ncols, const int* out_col0,
                                            unsigned char* d_bgr, float* d_ranges)
{
    HZ_ON_DEVICE(d);
    if(nstrips < 0 || (nstrips > 0 && (!d_in || !ncols || !out_col0)))
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse_strips: bad arguments");
        return -1;
    }
    for(int k=0; k<nstrips; k++)
        if(ncols[k] < 0 || (ncols[k] > 0 && (mask_stride < (ncols[k] + 31)/32 || out_col0[k] < 0 || out_col0[k] + ncols[k] > d->W || !d_in[k])))
        {
            snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse_strips: columns [%d,%d) of strip %d do not fit a %d-wide image",
                     out_col0[k], out_col0[k] + ncols[k], k, d->W);
            return -1;
        }
    if(d_ranges && hz_upload_tanel(d, tanel) != 0) return -1;
    for(int k0=0; k0<nstrips; k0+=HZ_MAX_STRIPS)
    {
        hz_strips_t st;
        memset(&st, 0, sizeof(st));
        const int n = nstrips - k0 < HZ_MAX_STRIPS ? nstrips - k0 : HZ_MAX_STRIPS;
        for(int k=0; k<n; k++) { st.in[k] = d_in[k0+k]; st.ncols[k] = ncols[k0+k]; st.col0[k] = out_col0[k0+k]; }
        hzk_resolve_sparse(dim3((unsigned)d->H, (unsigned)n), dim3(256), d->rstream, st, mask_stride, (const float*)d->d_tanel, d_bgr, d_ranges, d->W, d->H, view->znear, view->zfar);
        HZ_CHECK(hipGetLastError());
    }
    return 0;
}

extern "C" int hz_hip_resolve_sparse(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     const uint32_t* d_in, int mask_stride, int ncols, int out_col0,
                                     unsigned char* d_bgr, float* d_ranges)
{
    if(ncols <= 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_resolve_sparse: columns [%d,%d) do not fit a %d-wide image",
                 out_col0, out_col0 + ncols, d->W);
        return -1;
    }
    return hz_hip_resolve_sparse_strips(d, view, tanel, 1, &d_in, mask_stride, &ncols, &out_col0, d_bgr, d_ranges);
}


extern "C" int hz_hip_read_depth(hz_dev_t* d, int x, int y, uint32_t* z24)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(x < d->col0 || x >= d->col1 || y < 0 || y >= d->H)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_read_depth: (%d,%d) outside the drawn sector", x, y);
        return -1;
    }
    const int SW = d->col1 - d->col0;
    unsigned long long key = 0;
    HZ_CHECK(hipMemcpyAsync(&key, &d->d_fb[(size_t)(d->H-1-y)*SW + (x - d->col0)], sizeof(key),
                            hipMemcpyDeviceToHost, d->stream));
    HZ_CHECK(hipStreamSynchronize(d->stream));
    *z24 = (uint32_t)(key >> 40);
    return 0;
}


extern "C" int hz_hip_link_cells(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                 const float* sin_az, const float* cos_az, const double* cos_el,
                                 double viewer_lat, double cos_viewer_lat, double viewer_lon,
                                 int cell_w, int cell_h, int nx, int ny, float* lat, float* lon)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(d->col0 != 0 || d->col1 != d->W || cell_w <= 0 || cell_h <= 0 || nx <= 0 || ny <= 0 || !sin_az || !cos_az || !cos_el)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells: needs a full-width context, positive sizes and the three tables");
        return -1;
    }
    if(hz_upload_tanel(d, tanel) != 0) return -1;
    const size_t n = (size_t)nx*ny;
    /* one allocation: lat, lon, then the tables */
    const size_t bytes = 2*n*sizeof(float) + 2*(size_t)nx*sizeof(float) + (size_t)ny*sizeof(double) + 16;
    unsigned char* buf = NULL;
    HZ_CHECK(hipMalloc(&buf, bytes));
    double* d_cos_el = (double*)buf;
    float* d_lat = (float*)(d_cos_el + ny), * d_lon = d_lat + n, * d_sin = d_lon + n, * d_cos = d_sin + nx;
    int rc = 0;
    if(hipMemcpyAsync(d_cos_el, cos_el, (size_t)ny*sizeof(double), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(d_sin, sin_az, (size_t)nx*sizeof(float), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(d_cos, cos_az, (size_t)nx*sizeof(float), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0)
    {
        hzk_link_cells(dim3((unsigned)((n + 255)/256)), dim3(256), d->stream, (const unsigned long long*)d->d_fb, (const float*)d->d_tanel, (const float*)d_sin, (const float*)d_cos, (const double*)d_cos_el, d_lat, d_lon, d->W, d->H, cell_w, cell_h, nx, ny, view->znear, view->zfar, viewer_lat, cos_viewer_lat, viewer_lon);
        if(hipGetLastError() != hipSuccess) rc = -1;
    }
    if(rc == 0 && hipMemcpyAsync(lat, d_lat, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(lon, d_lon, n*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(buf);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_link_cells failed");
    return rc;
}

extern "C" int hz_hip_poi_visibility(hz_dev_t* d, const hz_view_t* view, const float* tanel,
                                     int cut_off_bottom_px, const hz_poi_proj_t* proj, int npois,
                                     unsigned char* visible, float* label_x, float* label_y)
{
    HZ_ON_DEVICE(d);
    if(hz_fb_refill(d) != 0) return -1;
    HZ_CHECK(hipStreamWaitEvent(d->stream, d->ev_drawn, 0));      /* the last draw finishes on qstream */
    d->stream_reads_fb = 1;
    if(d->col0 != 0 || d->col1 != d->W || npois < 0)
    {
        snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility: needs a full-width context");
        return -1;
    }
    if(npois == 0) return 0;
    if(hz_upload_tanel(d, tanel) != 0) return -1;
    hz_poi_proj_t* d_proj = NULL; unsigned char* d_vis = NULL; float *d_x = NULL, *d_y = NULL;
    HZ_CHECK(hipMalloc(&d_proj, (size_t)npois*sizeof(hz_poi_proj_t)));
    HZ_CHECK(hipMalloc(&d_vis, (size_t)npois));
    HZ_CHECK(hipMalloc(&d_x, (size_t)npois*sizeof(float)));
    HZ_CHECK(hipMalloc(&d_y, (size_t)npois*sizeof(float)));
    int rc = 0;
    if(hipMemcpyAsync(d_proj, proj, (size_t)npois*sizeof(hz_poi_proj_t), hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = -1;
    if(rc == 0)
    {
        hzk_poi(dim3((unsigned)((npois + 255)/256)), dim3(256), d->stream, (const unsigned long long*)d->d_fb, (const float*)d->d_tanel, (const hz_poi_proj_t*)d_proj, npois, d_vis, d_x, d_y, d->W, d->H, d->H - cut_off_bottom_px, view->znear, view->zfar);
        if(hipGetLastError() != hipSuccess) rc = -1;
    }
    if(rc == 0 && hipMemcpyAsync(visible, d_vis, (size_t)npois, hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_x, d_x, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(rc == 0 && hipMemcpyAsync(label_y, d_y, (size_t)npois*sizeof(float), hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = -1;
    if(hipStreamSynchronize(d->stream) != hipSuccess) rc = -1;
    (void)hipFree(d_proj); (void)hipFree(d_vis); (void)hipFree(d_x); (void)hipFree(d_y);
    if(rc != 0) snprintf(g_last_error, sizeof(g_last_error), "hz_hip_poi_visibility failed");
    return rc;
}
