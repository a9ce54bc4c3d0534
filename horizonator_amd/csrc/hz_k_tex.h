/* hz_k_tex.h - part of hz_kernels.hip (included there, in this order; one translation unit):
 * textured resolve: deferred shading (k_shade_tex). */
#pragma once

/* ------------------------------------------------------------------------ */
/* textured resolve ("next" row N4): deferred shading                         */
/*
 * The rasteriser kernels do not know about the texture: the framebuffer word
 * says which triangle won each pixel, and that is all the reference's fragment
 * stage needs beyond the triangle itself.  So for every terrain pixel this
 * kernel builds the winning triangle again (three vertices through the same
 * transform, plus their texture coordinates), sets up the planes of shade, s
 * and t with hz_tri_planes() arithmetic, evaluates them at the pixel, samples
 * the texture and blends (hz_tex.h).  A triangle the clipper cut is clipped
 * again, and the piece that covers the pixel with the stored depth supplies the
 * planes.  This path is not the benchmark's.
 */
__device__ __noinline__ static bool hz_shade_clipped(const hz_cvert_t& a, const hz_cvert_t& b, const hz_cvert_t& c,
                                                     const hz_params_t& p, int px, int py, uint32_t zi,
                                                     hz_texplanes_t* planes)
{
    hz_cvert_t bufa[HZ_MAX_CLIPPED+1], bufb[HZ_MAX_CLIPPED+1], *poly;
    const int n = hz_clip_triangle(bufa, bufb, &poly, &a, &b, &c, p.halfW, p.halfH);
    for(int k=2; k<n; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        if(px < box.px0 || px > box.px1 || py < box.py0 || py > box.py1) continue;
        hz_tri_t tri;
        hz_tri_planes(&tri, &va, &vb, &vc);
        if(!hz_tri_covers(&tri, px, py)) continue;
        uint32_t z2, r8;
        if(!hz_tri_fragment(&tri, px, py, &z2, &r8) || z2 != zi) continue;
        hz_tri_planes_tex(planes, &poly[k-1], &poly[k], &poly[0]);
        return true;
    }
    return false;
}

/* the three vertices of grid triangle `prim` with their texture coordinates */
__device__ static inline void hz_prim_cverts(const int16_t* __restrict__ mosaic, const hz_texparams_t& tp,
                                             const hz_params_t& p, uint32_t prim,
                                             hz_cvert_t* a, hz_cvert_t* b, hz_cvert_t* c)
{
    const uint32_t cell = prim >> 1;
    const int t = prim & 1;
    const int j = cell / (uint32_t)(p.N-1);
    const int i = cell - (uint32_t)j*(uint32_t)(p.N-1);
    /* reference horizonator-lib.c:500-506 */
    const int ib = i+1,              jb = t == 0 ? j+1 : j;
    const int ic = t == 0 ? i : i+1, jc = j+1;
    *a = hz_cvert(hz_transform(&p.u, (float)i,  (float)j,  (float)mosaic[(size_t)j *p.N + i ]), p.halfW, p.halfH);
    *b = hz_cvert(hz_transform(&p.u, (float)ib, (float)jb, (float)mosaic[(size_t)jb*p.N + ib]), p.halfW, p.halfH);
    *c = hz_cvert(hz_transform(&p.u, (float)ic, (float)jc, (float)mosaic[(size_t)jc*p.N + ic]), p.halfW, p.halfH);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)i,  (float)j,  &a->s, &a->t);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)ib, (float)jb, &b->s, &b->t);
    hz_vertex_tex(&tp, p.u.deg_per_cell, (float)ic, (float)jc, &c->s, &c->t);
}

/* One wave shades TX_CHUNK consecutive output pixels at a time.  Neighbouring
 * pixels mostly belong to the same triangle, and the expensive part - building
 * the triangle again and setting up its planes - depends on the triangle only:
 *   A  the chunk's pixels are cut into runs of equal primitive id (ballot + popcount)
 *   B  lane = run: vertices, planes of shade/s/t into LDS (a triangle the clipper
 *      cut is flagged: its planes depend on the piece that covers the pixel)
 *   C  lane = pixel: planes of its run from LDS, evaluate, sample, blend, store
 * Next to the viewer a chunk holds a handful of runs; at the skyline every pixel
 * is its own run and the scheme falls back to one set-up per pixel. */
#define TX_MAXCLIP 4                    /* clipped triangles per chunk whose pieces are kept in LDS */
#define TX_PIECES  (HZ_MAX_CLIPPED-2)
struct tx_lds_t
{
    uint32_t run_prim[TX_CHUNK];
    float    planes[9][TX_CHUNK];       /* r_org drdx drdy s_org dsdx dsdy t_org dtdx dtdy;
                                         * r_org = NaN: clipped, drdx then holds the slot below (-1: none) */
    /* The few triangles next to the viewer that cross the image border cover a
     * large share of the picture (9 triangles, 18% of the terrain pixels in the
     * benchmark scene): their clipped pieces are set up once per chunk.  Per
     * piece: snapped vertices (6), depth plane (3), the nine texture planes. */
    int32_t  npieces[TX_MAXCLIP];
    uint32_t piece[TX_MAXCLIP][TX_PIECES][18];
};

/* B, for a triangle the clipper cuts: all its pieces into LDS slot `slot` */
__device__ __noinline__ static void tx_store_pieces(tx_lds_t& L, int slot, const hz_cvert_t& a, const hz_cvert_t& b,
                                                    const hz_cvert_t& c, const hz_params_t& p)
{
    hz_cvert_t bufa[HZ_MAX_CLIPPED+1], bufb[HZ_MAX_CLIPPED+1], *poly;
    const int n = hz_clip_triangle(bufa, bufb, &poly, &a, &b, &c, p.halfW, p.halfH);
    int count = 0;
    for(int k=2; k<n && count < TX_PIECES; k++)
    {
        const hz_wvert_t va = hz_wvert_of(&poly[k-1]), vb = hz_wvert_of(&poly[k]), vc = hz_wvert_of(&poly[0]);
        hz_box_t box;
        if(!hz_tri_cull_window(&box, &va, &vb, &vc, p.col0, p.col1-1, 0, p.H-1)) continue;
        hz_tri_t tri;
        hz_tri_planes(&tri, &va, &vb, &vc);
        hz_texplanes_t pl;
        hz_tri_planes_tex(&pl, &poly[k-1], &poly[k], &poly[0]);
        uint32_t* o = L.piece[slot][count++];
        #pragma unroll
        for(int m=0; m<3; m++) { o[m] = (uint32_t)tri.xs[m]; o[3+m] = (uint32_t)tri.ys[m]; }
        o[6] = __float_as_uint(tri.z_org); o[7] = __float_as_uint(tri.dzdx); o[8] = __float_as_uint(tri.dzdy);
        o[9]  = __float_as_uint(pl.r_org); o[10] = __float_as_uint(pl.drdx); o[11] = __float_as_uint(pl.drdy);
        o[12] = __float_as_uint(pl.s_org); o[13] = __float_as_uint(pl.dsdx); o[14] = __float_as_uint(pl.dsdy);
        o[15] = __float_as_uint(pl.t_org); o[16] = __float_as_uint(pl.dtdx); o[17] = __float_as_uint(pl.dtdy);
    }
    L.npieces[slot] = count;
}
/* C, for a pixel of such a triangle: the piece that covers it with the stored depth */
__device__ static inline bool tx_find_piece(const tx_lds_t& L, int slot, int px, int py, uint32_t zi, hz_texplanes_t* pl)
{
    const int n = L.npieces[slot];
    for(int k=0; k<n; k++)
    {
        const uint32_t* o = L.piece[slot][k];
        hz_tri_t tri;
        #pragma unroll
        for(int m=0; m<3; m++) { tri.xs[m] = (int32_t)o[m]; tri.ys[m] = (int32_t)o[3+m]; }
        tri.z_org = __uint_as_float(o[6]); tri.dzdx = __uint_as_float(o[7]); tri.dzdy = __uint_as_float(o[8]);
        tri.r_org = tri.drdx = tri.drdy = 0.f;
        if(!hz_tri_covers(&tri, px, py)) continue;
        uint32_t z2, r8;
        if(!hz_tri_fragment(&tri, px, py, &z2, &r8) || z2 != zi) continue;
        pl->r_org = __uint_as_float(o[9]);  pl->drdx = __uint_as_float(o[10]); pl->drdy = __uint_as_float(o[11]);
        pl->s_org = __uint_as_float(o[12]); pl->dsdx = __uint_as_float(o[13]); pl->dsdy = __uint_as_float(o[14]);
        pl->t_org = __uint_as_float(o[15]); pl->dtdx = __uint_as_float(o[16]); pl->dtdy = __uint_as_float(o[17]);
        return true;
    }
    return false;
}

__global__ __launch_bounds__(64)
void k_shade_tex(const unsigned long long* __restrict__ fb, const int16_t* __restrict__ mosaic,
                 const uint32_t* __restrict__ texels, hz_texparams_t tp,
                 unsigned char* __restrict__ bgr, hz_params_t p)
{
    __shared__ tx_lds_t L;
    const int lane = threadIdx.x;
    const size_t npix = (size_t)p.SW*p.H;
    const size_t nchunks = (npix + TX_CHUNK-1)/TX_CHUNK;
    for(size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
    {
        /* A: runs */
        uint32_t zi_k[TX_SUB], run_k[TX_SUB];
        uint32_t nruns = 0, prev_last = 0xFFFFFFFFu;
        /* output position of this lane's pixel in sub-span 0 (one 64-bit division per
         * chunk), then stepped by 64 pixels */
        const size_t o0 = chunk*TX_CHUNK + lane;
        const int yo0 = (int)(o0 / p.SW), x0 = (int)(o0 - (size_t)yo0*p.SW);
        int yo = yo0, x = x0;
        #pragma unroll
        for(int k=0; k<TX_SUB; k++)
        {
            const size_t o = o0 + (size_t)k*64;
            uint32_t prim = 0xFFFFFFFFu, zi = HZ_Z24_MAX;
            if(o < npix)
            {
                const unsigned long long key = fb[(size_t)(p.H-1 - yo)*p.SW + x];
                zi = (uint32_t)(key >> 40);
                if(zi != HZ_Z24_MAX) prim = (uint32_t)((key >> 8) & 0xFFFFFFFFull);
            }
            uint32_t left = __shfl_up(prim, 1);
            if(lane == 0) left = prev_last;
            prev_last = __shfl(prim, 63);
            const bool is_start = prim != 0xFFFFFFFFu && prim != left;
            const unsigned long long starts = __ballot(is_start);
            const uint32_t upto = (uint32_t)__popcll(starts & ((2ull << lane) - 1ull));   /* starts at lanes <= lane */
            run_k[k] = nruns + upto - 1u;           /* a sky pixel gets a meaningless index it never uses */
            zi_k[k]  = zi;
            if(is_start) L.run_prim[run_k[k]] = prim;
            nruns += (uint32_t)__popcll(starts);
            x += 64;
            while(x >= p.SW) { x -= p.SW; yo++; }
        }
        if(nruns == 0) continue;                    /* sky only */
        __syncthreads();

        /* B: planes per run */
        int clip_slots = 0;
        for(uint32_t r0 = 0; r0 < nruns; r0 += 64)
        {
            const uint32_t r = r0 + lane;
            hz_cvert_t a = {}, b = {}, c = {};
            bool clipped = false;
            if(r < nruns)
            {
                hz_prim_cverts(mosaic, tp, p, L.run_prim[r], &a, &b, &c);
                clipped = (hz_clip_mask(a.xn, a.yn, a.zn) | hz_clip_mask(b.xn, b.yn, b.zn) | hz_clip_mask(c.xn, c.yn, c.zn)) != 0;
            }
            const unsigned long long cm = __ballot(clipped);
            hz_texplanes_t pl = {};
            if(clipped)
            {
                /* one LDS slot per clipped run, while they last (the same triangle may
                 * start several runs of a chunk: row after row) */
                const int slot = clip_slots + (int)__popcll(cm & ((1ull << lane) - 1ull));
                pl.r_org = __uint_as_float(0x7FC00000u);
                pl.drdx  = (float)(slot < TX_MAXCLIP ? slot : -1);
                if(slot < TX_MAXCLIP) tx_store_pieces(L, slot, a, b, c, p);
            }
            else if(r < nruns)
                hz_tri_planes_tex(&pl, &a, &b, &c);
            clip_slots += (int)__popcll(cm);
            if(r >= nruns) continue;
            L.planes[0][r] = pl.r_org; L.planes[1][r] = pl.drdx; L.planes[2][r] = pl.drdy;
            L.planes[3][r] = pl.s_org; L.planes[4][r] = pl.dsdx; L.planes[5][r] = pl.dsdy;
            L.planes[6][r] = pl.t_org; L.planes[7][r] = pl.dtdx; L.planes[8][r] = pl.dtdy;
        }
        __syncthreads();

        /* C: pixels */
        yo = yo0; x = x0;
        #pragma unroll
        for(int k=0; k<TX_SUB; k++)
        {
            const size_t o = o0 + (size_t)k*64;
            const int py = p.H-1 - yo, px = x + p.col0;
            x += 64;
            while(x >= p.SW) { x -= p.SW; yo++; }
            if(o >= npix || zi_k[k] == HZ_Z24_MAX) continue;       /* sky: k_resolve wrote the clear colour */
            const uint32_t r = run_k[k];
            hz_texplanes_t pl;
            pl.r_org = L.planes[0][r]; pl.drdx = L.planes[1][r]; pl.drdy = L.planes[2][r];
            pl.s_org = L.planes[3][r]; pl.dsdx = L.planes[4][r]; pl.dsdy = L.planes[5][r];
            pl.t_org = L.planes[6][r]; pl.dtdx = L.planes[7][r]; pl.dtdy = L.planes[8][r];
            if(!(pl.r_org == pl.r_org))
            {
                /* the clipper cut this triangle: the piece that covers this pixel with the stored depth */
                const int slot = (int)pl.drdx;
                if(slot < 0 || !tx_find_piece(L, slot, px, py, zi_k[k], &pl))
                {
                    hz_cvert_t a, b, c;
                    hz_prim_cverts(mosaic, tp, p, L.run_prim[r], &a, &b, &c);
                    if(!hz_shade_clipped(a, b, c, p, px, py, zi_k[k], &pl)) continue;   /* cannot happen; keeps the untextured colour */
                }
            }
            const float shade = hz_plane_at(pl.r_org, pl.drdx, pl.drdy, px, py);
            const float s     = hz_plane_at(pl.s_org, pl.dsdx, pl.dsdy, px, py);
            const float tt    = hz_plane_at(pl.t_org, pl.dtdx, pl.dtdy, px, py);
            const uint32_t col = hz_fragment_textured(hz_tex_sample(texels, tp.tex_w, tp.tex_h, s, tt), shade);
            bgr[o*3+0] = (unsigned char)(col & 255u);
            bgr[o*3+1] = (unsigned char)((col >> 8) & 255u);
            bgr[o*3+2] = (unsigned char)((col >> 16) & 255u);
        }
        __syncthreads();                            /* the next chunk reuses the LDS tables */
    }
}
