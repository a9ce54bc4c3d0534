/* hz_raster.h - triangle setup, coverage and interpolation rules of the
 * software rasteriser (device code; also compiled on the host for unit use).
 *
 * These restate what the GL pipeline does between the geometry shader and the
 * depth buffer for the state the reference sets up (reference
 * horizonator-lib.c:183-185 depth test GL_LESS + back-face cull, :631/:646
 * RGB8 + 24-bit depth, :657 viewport), with Mesa/llvmpipe's conventions where
 * GL leaves a choice (8 sub-pixel bits, fill rule, Z24 rounding):
 *
 *   window position   xw = x*W/2 + W/2,  yw = y*H/2 + H/2,  zw = z/2 + 1/2
 *   pixel centres     at half-integers of (xw,yw); we work in f = w - 0.5 so
 *                     that centres are integers
 *   coverage          from positions snapped to 1/256 px, integer edge
 *                     functions, ties owned by left and bottom edges (y up)
 *   facing            sign of the snapped area; <= 0 is culled
 *   depth / colour    planes through the UNSNAPPED positions, set up and
 *                     evaluated with llvmpipe's arithmetic (hz_tri_planes)
 *   depth buffer      zi = rint(z * (2^24-1)), fragment dropped unless
 *                     0 <= z <= 1, passes iff zi < stored; 0xFFFFFF = cleared
 *
 * The framebuffer word packs  zi<<40 | primitive<<8 | red8  so that one
 * unsigned 64-bit min implements "GL_LESS, first drawn wins ties" regardless
 * of the order in which triangles are processed.
 */
#pragma once

#include "hz_num.h"

#define HZ_SUBPIXEL_BITS   8
#define HZ_SUBPIXEL_ONE    256
#define HZ_Z24_MAX         0xFFFFFFu
#define HZ_FB_CLEAR        0xFFFFFFFFFFFFFFFFull
/* positions beyond this many pixels from the origin are outside the guard
 * band: the triangle is dropped (keeps the 64-bit edge functions exact) */
#define HZ_GUARD_PX        2097152.0f
#define HZ_OUTSIDE_GUARD   INT32_MIN

/* window-space vertex as the kernels keep it: NDC x (for the discard rule),
 * unsnapped window position (pixel centres at half-integers), depth, colour,
 * and the position relative to the pixel-centre grid snapped to 1/256 px
 * (xs = HZ_OUTSIDE_GUARD marks a vertex outside the guard band or with a
 * non-finite position) */
typedef struct { float xn, wx, wy, zw, red; int32_t xs, ys; uint32_t cmask; } hz_wvert_t;

/* GL clips against the view volume -1 <= x,y,z <= 1 (w = 1 here): which of the
 * six planes a vertex lies beyond, bit order as in Mesa (x>1, x<-1, y>1, y<-1,
 * z<-1, z>1) */
HZ_HD uint32_t hz_clip_mask(float xn, float yn, float zn)
{
    uint32_t m = 0;
    if(xn > 1.0f)        m |= 1;
    if(xn + 1.0f < 0.f)  m |= 2;
    if(yn > 1.0f)        m |= 4;
    if(yn + 1.0f < 0.f)  m |= 8;
    if(zn + 1.0f < 0.f)  m |= 16;
    if(zn > 1.0f)        m |= 32;
    return m;
}

HZ_HD hz_wvert_t hz_to_window(hz_vertex_t v, float halfW, float halfH)
{
    hz_wvert_t w;
    w.xn  = v.x;
    w.wx  = v.x*halfW + halfW;
    w.wy  = v.y*halfH + halfH;
    w.zw  = v.z*0.5f + 0.5f;
    w.red = v.red;
    w.cmask = hz_clip_mask(v.x, v.y, v.z);
    const float fx = w.wx - 0.5f, fy = w.wy - 0.5f;     /* pixel centres at integers */
    if(hz_abs(fx) <= HZ_GUARD_PX && hz_abs(fy) <= HZ_GUARD_PX)
    {
        w.xs = (int32_t)hz_roundeven(fx*256.f);
        w.ys = (int32_t)hz_roundeven(fy*256.f);
    }
    else
    {
        w.xs = HZ_OUTSIDE_GUARD;
        w.ys = 0;
    }
    return w;
}

/* pixel box of a triangle, inclusive, clipped to the scissor */
typedef struct { int32_t px0, px1, py0, py1; } hz_box_t;

/* pixel centres inside the snapped bounding box, clipped to the scissor;
 * returns 0 if there is none */
HZ_HD int hz_tri_box(hz_box_t* box,
                     const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c,
                     int sx0, int sx1, int sy0, int sy1)
{
    int32_t xlo = a->xs < b->xs ? a->xs : b->xs; xlo = xlo < c->xs ? xlo : c->xs;
    int32_t xhi = a->xs > b->xs ? a->xs : b->xs; xhi = xhi > c->xs ? xhi : c->xs;
    int32_t ylo = a->ys < b->ys ? a->ys : b->ys; ylo = ylo < c->ys ? ylo : c->ys;
    int32_t yhi = a->ys > b->ys ? a->ys : b->ys; yhi = yhi > c->ys ? yhi : c->ys;

    int32_t px0 = (xlo + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS;
    int32_t px1 =  xhi                        >> HZ_SUBPIXEL_BITS;
    int32_t py0 = (ylo + (HZ_SUBPIXEL_ONE-1)) >> HZ_SUBPIXEL_BITS;
    int32_t py1 =  yhi                        >> HZ_SUBPIXEL_BITS;
    if(px0 < sx0) px0 = sx0;
    if(px1 > sx1) px1 = sx1;
    if(py0 < sy0) py0 = sy0;
    if(py1 > sy1) py1 = sy1;
    box->px0 = px0; box->px1 = px1; box->py0 = py0; box->py1 = py1;
    return !(px0 > px1 || py0 > py1);
}

/* the window-space rejections: guard band, back face, empty pixel box */
HZ_HD int hz_tri_cull_window(hz_box_t* box,
                             const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c,
                             int sx0, int sx1, int sy0, int sy1)
{
    if(a->xs == HZ_OUTSIDE_GUARD || b->xs == HZ_OUTSIDE_GUARD || c->xs == HZ_OUTSIDE_GUARD) return 0;

    /* back-face cull on the snapped area (counter-clockwise = front, y up) */
    const int64_t area =
        (int64_t)(b->xs - a->xs)*(int64_t)(c->ys - a->ys) -
        (int64_t)(c->xs - a->xs)*(int64_t)(b->ys - a->ys);
    if(area <= 0) return 0;

    return hz_tri_box(box, a, b, c, sx0, sx1, sy0, sy1);
}

/* Everything that can be decided about a triangle without looking at a pixel.
 * Returns HZ_TRI_DROP, HZ_TRI_DRAW (with the pixel box, clipped to the scissor
 * [sx0,sx1] x [sy0,sy1], inclusive), or HZ_TRI_CLIP: the triangle crosses a
 * plane of the view volume and has to go through the clipper first. */
#define HZ_TRI_DROP 0
#define HZ_TRI_DRAW 1
#define HZ_TRI_CLIP 2
HZ_HD int hz_tri_cull(hz_box_t* box,
                      const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c,
                      int sx0, int sx1, int sy0, int sy1)
{
    /* reference geometry.glsl:21-27: wider than a quarter of the viewport
     * (which includes everything straddling the +-180 deg seam) -> dropped */
    const float xmax = hz_max(hz_max(a->xn, b->xn), c->xn);
    const float xmin = hz_min(hz_min(a->xn, b->xn), c->xn);
    if(xmax - xmin > 0.5f) return HZ_TRI_DROP;

    /* wholly beyond one plane of the view volume (image border, near sphere,
     * far sphere) */
    if(a->cmask & b->cmask & c->cmask) return HZ_TRI_DROP;
    if(a->cmask | b->cmask | c->cmask) return HZ_TRI_CLIP;

    return hz_tri_cull_window(box, a, b, c, sx0, sx1, sy0, sy1) ? HZ_TRI_DRAW : HZ_TRI_DROP;
}

/* a triangle ready for rasterisation */
typedef struct
{
    int32_t xs[3], ys[3];           /* snapped positions, 1/256 px                       */
    float z_org, dzdx, dzdy;        /* depth  = fma(dzdy, py, fma(dzdx, px, z_org))      */
    float r_org, drdx, drdy;        /* colour likewise                                   */
} hz_tri_t;

/* Attribute planes with the arithmetic of llvmpipe's triangle setup and
 * fragment interpolation.  Pinned on the reference's draws (tests/golden):
 * with exactly these operations the depth of every triangle llvmpipe does not
 * clip comes out bit-identical.
 *   - llvmpipe sees the front faces of an FBO draw as clockwise and swaps the
 *     first two vertices: its v0 is the SECOND vertex (b) of the draw call
 *   - gradients via ooa = 1/area and four pre-multiplied edge deltas
 *   - value at the window origin, then two fused multiply-adds per pixel */
HZ_HD void hz_tri_planes(hz_tri_t* t, const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c)
{
    t->xs[0] = a->xs; t->xs[1] = b->xs; t->xs[2] = c->xs;
    t->ys[0] = a->ys; t->ys[1] = b->ys; t->ys[2] = c->ys;
    const hz_wvert_t *v0 = b, *v1 = a, *v2 = c;
    const float dx01 = v0->wx - v1->wx, dy01 = v0->wy - v1->wy;
    const float dx20 = v2->wx - v0->wx, dy20 = v2->wy - v0->wy;
    const float ooa  = 1.0f / (dx01*dy20 - dx20*dy01);
    const float dy20_ooa = dy20*ooa, dy01_ooa = dy01*ooa, dx20_ooa = dx20*ooa, dx01_ooa = dx01*ooa;
    const float x0c = v0->wx - 0.5f, y0c = v0->wy - 0.5f;
    const float dz01 = v0->zw  - v1->zw,  dz20 = v2->zw  - v0->zw;
    const float dr01 = v0->red - v1->red, dr20 = v2->red - v0->red;
    t->dzdx  = dz01*dy20_ooa - dz20*dy01_ooa;
    t->dzdy  = dz20*dx01_ooa - dz01*dx20_ooa;
    t->z_org = v0->zw  - (t->dzdx*x0c + t->dzdy*y0c);
    t->drdx  = dr01*dy20_ooa - dr20*dy01_ooa;
    t->drdy  = dr20*dx01_ooa - dr01*dx20_ooa;
    t->r_org = v0->red - (t->drdx*x0c + t->drdy*y0c);
}

/* Early depth rejection: a depth value that no fragment of triangle (a,b,c)
 * can fall below, as a 24-bit integer.  If every pixel centre of the triangle's
 * box already holds a SMALLER depth, none of its fragments can pass GL_LESS and
 * the triangle may be skipped without changing a single byte of the result
 * (stored depths only ever decrease during a draw).
 *
 * Why min(vertex depth) alone is not such a value: coverage is decided on
 * positions snapped to 1/256 px, the depth plane goes through the unsnapped
 * ones.  A covered pixel centre p lies in the snapped triangle, i.e. within
 * 1/512 px (per axis) of a point q of the unsnapped one; depth(q) is a convex
 * combination of the vertex depths, depth(p) = depth(q) + grad . (p-q).  For
 * slivers grad is huge (1/area) and depth(p) undershoots the vertex depths -
 * the oracle counts ~1800 such fragments in the 16000x4000 benchmark scene.
 * So: floor = min vertex depth - G*(1/512 + rounding) with G >= |dz/dx|+|dz/dy|
 * bounded from the very expressions hz_tri_planes() evaluates (same `area`,
 * bit for bit), and the float rounding of the plane set-up and evaluation
 * (z_org and two fmas on coordinates up to max(W,H)) bounded by
 * 2^-22*max(W,H)*G + 3*2^-24.  `guard` = 1/500 + max(W,H)*2^-22 is passed in.
 * Returns 0 when no useful floor exists (degenerate set-up, floor <= 0). */
HZ_HD int hz_tri_depth_floor(const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c,
                             float guard, uint32_t* zfloor)
{
    const hz_wvert_t *v0 = b, *v1 = a, *v2 = c;             /* as hz_tri_planes */
    const float dx01 = v0->wx - v1->wx, dy01 = v0->wy - v1->wy;
    const float dx20 = v2->wx - v0->wx, dy20 = v2->wy - v0->wy;
    const float area = dx01*dy20 - dx20*dy01;
    const float dz01 = hz_abs(v0->zw - v1->zw), dz20 = hz_abs(v2->zw - v0->zw);
    /* |dzdx| <= (|dz01||dy20| + |dz20||dy01|)/|area|, |dzdy| likewise */
    const float num  = dz01*(hz_abs(dy20) + hz_abs(dx20)) + dz20*(hz_abs(dy01) + hz_abs(dx01));
    const float G    = num / hz_abs(area) * 1.01f;          /* 1%: rounding of this bound and of the set-up */
    const float zmin = hz_min(v0->zw, hz_min(v1->zw, v2->zw));
    const float zf   = zmin*16777215.f - (G*guard*16777215.f + 8.f);
    if(!(zf >= 1.0f)) return 0;                             /* also catches NaN/inf from area == 0 */
    *zfloor = (uint32_t)zf;                                 /* truncation: rounds down */
    return 1;
}

/* The same test without the division, for a box whose pixel centres all hold a
 * depth of at most `zs` (24-bit): true only if no fragment of (a,b,c) can pass
 * GL_LESS there.  hz_tri_depth_floor() asks  zs < floor(zmin*M - (G*guard*M + 8)),
 * G = 1.01*num/|area|, M = 2^24-1; multiplied through by |area| that follows from
 *     (zmin*M - (zs + 12)) * |area|  >=  num * (1.03*guard*M)      and  |area| > 0
 * with room for this evaluation's own rounding: zmin*M and the subtraction are
 * off by at most 1.5 units (12 instead of 9), the two products and num by a few
 * 2^-24 relative (1.03 instead of 1.01).  `area` is the expression hz_tri_planes()
 * divides by, bit for bit, as there.  k = 1.03*guard*M.  A NaN anywhere: false. */
HZ_HD int hz_tri_hidden(const hz_wvert_t* a, const hz_wvert_t* b, const hz_wvert_t* c, float k, uint32_t zs)
{
    const hz_wvert_t *v0 = b, *v1 = a, *v2 = c;             /* as hz_tri_planes */
    const float dx01 = v0->wx - v1->wx, dy01 = v0->wy - v1->wy;
    const float dx20 = v2->wx - v0->wx, dy20 = v2->wy - v0->wy;
    const float area = hz_abs(dx01*dy20 - dx20*dy01);
    const float dz01 = hz_abs(v0->zw - v1->zw), dz20 = hz_abs(v2->zw - v0->zw);
    const float num  = dz01*(hz_abs(dy20) + hz_abs(dx20)) + dz20*(hz_abs(dy01) + hz_abs(dx01));
    const float zmin = hz_min(v0->zw, hz_min(v1->zw, v2->zw));
    const float room = zmin*16777215.f - ((float)zs + 12.f);
    return room > 0.f && area > 0.f && room*area >= num*k;
}

/* edge function of edge m (vertex m -> m+1) at pixel centre (px,py), and
 * whether a zero belongs to the triangle */
HZ_HD int64_t hz_edge(const hz_tri_t* t, int m, int px, int py)
{
    const int a = m, b = (m == 2) ? 0 : m+1;
    const int64_t dx = t->xs[b] - t->xs[a];
    const int64_t dy = t->ys[b] - t->ys[a];
    return dx*(((int64_t)py << HZ_SUBPIXEL_BITS) - t->ys[a]) - dy*(((int64_t)px << HZ_SUBPIXEL_BITS) - t->xs[a]);
}
HZ_HD int hz_edge_owns_zero(const hz_tri_t* t, int m)
{
    const int a = m, b = (m == 2) ? 0 : m+1;
    const int32_t dx = t->xs[b] - t->xs[a];
    const int32_t dy = t->ys[b] - t->ys[a];
    return dy < 0 || (dy == 0 && dx > 0);
}

/* coverage of pixel centre (px,py) */
HZ_HD int hz_tri_covers(const hz_tri_t* t, int px, int py)
{
    #pragma unroll
    for(int m=0; m<3; m++)
    {
        const int64_t e = hz_edge(t, m, px, py);
        if(e < 0) return 0;
        if(e == 0 && !hz_edge_owns_zero(t, m)) return 0;
    }
    return 1;
}

/* The coverage test once more, with everything that depends on the triangle
 * alone taken out of it.  hz_edge() is
 *     E = dx*(256*py - ya) - dy*(256*px - xa) = 256*(dx*py - dy*px) + (dy*xa - dx*ya)
 * and a pixel centre is covered by edge m iff E >= c, c = 0 where the edge owns
 * its zeros, else 1 (hz_tri_covers).  With F = dy*xa - dx*ya - c that is
 * 256*T + F >= 0 for the integer T = dx*py - dy*px, i.e. T >= ceil(-F/256) =
 * -floor(F/256):   covered  <=>  g + dx*py - dy*px >= 0,   g = F >> 8.
 * Three edges: six 64-bit multiply-adds and a sign test per pixel. */
typedef struct { int32_t dx[3], ndy[3]; uint32_t glo[3], ghi[3]; } hz_edges_t;

HZ_HD void hz_edges_of(hz_edges_t* e, const hz_tri_t* t)
{
    #pragma unroll
    for(int m=0; m<3; m++)
    {
        const int a = m, b = (m == 2) ? 0 : m+1;
        const int32_t dx = t->xs[b] - t->xs[a], dy = t->ys[b] - t->ys[a];
        const int64_t F = (int64_t)dy*(int64_t)t->xs[a] - (int64_t)dx*(int64_t)t->ys[a] - (hz_edge_owns_zero(t, m) ? 0 : 1);
        const int64_t g = F >> HZ_SUBPIXEL_BITS;            /* arithmetic: floor */
        e->dx[m] = dx; e->ndy[m] = -dy;
        e->glo[m] = (uint32_t)(uint64_t)g; e->ghi[m] = (uint32_t)((uint64_t)g >> 32);
    }
}
HZ_HD int64_t hz_edges_g(const hz_edges_t* e, int m) { return (int64_t)(((uint64_t)e->ghi[m] << 32) | (uint64_t)e->glo[m]); }
HZ_HD int hz_edges_cover(const hz_edges_t* e, int px, int py)
{
    int64_t any = 0;
    #pragma unroll
    for(int m=0; m<3; m++)
        any |= hz_edges_g(e, m) + (int64_t)e->dx[m]*(int64_t)py + (int64_t)e->ndy[m]*(int64_t)px;
    return any >= 0;
}

/* depth + colour of a covered pixel; returns 0 if the fragment cannot beat the
 * cleared depth */
HZ_HD int hz_tri_fragment(const hz_tri_t* t, int px, int py, uint32_t* zi, uint32_t* r8)
{
    const float fpx = (float)px, fpy = (float)py;
    /* all vertices are inside the depth range by now (clipper); what rounding
     * leaves outside is clamped, as llvmpipe does */
    float z = __builtin_fmaf(t->dzdy, fpy, __builtin_fmaf(t->dzdx, fpx, t->z_org));
    if(!(z == z)) return 0;
    z = hz_min(hz_max(z, 0.f), 1.f);
    const uint32_t q = (uint32_t)hz_roundeven(z * 16777215.f);
    if(q >= HZ_Z24_MAX) return 0;
    float r = __builtin_fmaf(t->drdy, fpy, __builtin_fmaf(t->drdx, fpx, t->r_org));
    r = hz_max(hz_min(r, 1.0f), 0.0f);
    *zi = q;
    *r8 = (uint32_t)hz_roundeven(r * 255.f);
    return 1;
}

/* ---- clipper -----------------------------------------------------------------
 * Mesa's polygon clipper (draw_pipe_clip.c) restated operation for operation:
 * Sutherland-Hodgman against the planes in hz_clip_mask() order, new vertices
 * interpolated in clip space from the inside vertex towards the outside one and
 * sent through the viewport transform again, the polygon rasterised as a fan
 * that keeps vertex 0 last.  The visible pixels do not change, but depth and
 * colour are interpolated over the pieces: with this the depth of triangles
 * that cross the image border or the near/far sphere matches the reference's
 * draw on llvmpipe bit for bit as well. */
#define HZ_MAX_CLIPPED 12

typedef struct { float xn, yn, zn, wx, wy, zw, red, s, t; } hz_cvert_t;     /* s,t: texture coordinates (hz_tex.h) */

HZ_HD hz_cvert_t hz_cvert(hz_vertex_t v, float halfW, float halfH)
{
    hz_cvert_t c;
    c.xn = v.x; c.yn = v.y; c.zn = v.z; c.red = v.red; c.s = 0.f; c.t = 0.f;
    c.wx = v.x*halfW + halfW; c.wy = v.y*halfH + halfH; c.zw = v.z*0.5f + 0.5f;
    return c;
}

HZ_HD float hz_clip_dist(const hz_cvert_t* v, int p)
{
    /* dot4(position, plane) evaluated left to right, w = 1 */
    const float px = (p == 0) ? -1.f : (p == 1) ? 1.f : 0.f;
    const float py = (p == 2) ? -1.f : (p == 3) ? 1.f : 0.f;
    const float pz = (p == 4) ?  1.f : (p == 5) ? -1.f : 0.f;
    return px*v->xn + py*v->yn + pz*v->zn + 1.0f*1.0f;
}

HZ_HD hz_cvert_t hz_clip_interp(float t, const hz_cvert_t* out, const hz_cvert_t* in, float halfW, float halfH)
{
    hz_cvert_t d;
    d.xn  = out->xn  + t*(in->xn  - out->xn);
    d.yn  = out->yn  + t*(in->yn  - out->yn);
    d.zn  = out->zn  + t*(in->zn  - out->zn);
    const float w   = 1.0f + t*(1.0f - 1.0f);
    const float oow = 1.0f / w;
    d.wx  = d.xn*oow*halfW + halfW;
    d.wy  = d.yn*oow*halfH + halfH;
    d.zw  = d.zn*oow*0.5f  + 0.5f;
    d.red = out->red + t*(in->red - out->red);
    d.s   = out->s   + t*(in->s   - out->s);
    d.t   = out->t   + t*(in->t   - out->t);
    return d;
}

/* clips triangle (a,b,c); the result lands in bufa or bufb, *poly points at it;
 * returns the vertex count (< 3: nothing left) */
HZ_HD int hz_clip_triangle(hz_cvert_t* bufa, hz_cvert_t* bufb, hz_cvert_t** poly,
                           const hz_cvert_t* a, const hz_cvert_t* b, const hz_cvert_t* c,
                           float halfW, float halfH)
{
    uint32_t todo = hz_clip_mask(a->xn, a->yn, a->zn) | hz_clip_mask(b->xn, b->yn, b->zn) | hz_clip_mask(c->xn, c->yn, c->zn);
    hz_cvert_t *in = bufa, *out = bufb;
    in[0] = *a; in[1] = *b; in[2] = *c;
    int n = 3;
    while(todo && n >= 3)
    {
        const int p = __builtin_ctz(todo);
        todo &= ~(1u << p);
        int outcount = 0;
        in[n] = in[0];
        const hz_cvert_t* vert_prev = &in[0];
        float dp_prev = hz_clip_dist(vert_prev, p);
        if(!(hz_abs(dp_prev) <= 3.0e38f)) return 0;                 /* NaN or infinite */
        for(int i=1; i<=n; i++)
        {
            const hz_cvert_t* vert = &in[i];
            const float dp = hz_clip_dist(vert, p);
            if(!(hz_abs(dp) <= 3.0e38f)) return 0;
            int different_sign;
            if(dp_prev >= 0.0f)
            {
                if(outcount >= HZ_MAX_CLIPPED) return 0;
                out[outcount++] = *vert_prev;
                different_sign = dp < 0.0f;
            }
            else
                different_sign = !(dp < 0.0f);
            if(different_sign)
            {
                if(outcount >= HZ_MAX_CLIPPED) return 0;
                const float denom = dp - dp_prev;
                if(dp < 0.0f)
                {
                    if(-dp < dp_prev) out[outcount++] = hz_clip_interp(dp / denom,       vert,      vert_prev, halfW, halfH);
                    else              out[outcount++] = hz_clip_interp(-dp_prev / denom, vert_prev, vert,      halfW, halfH);
                }
                else
                {
                    if(-dp_prev < dp) out[outcount++] = hz_clip_interp(-dp_prev / denom, vert_prev, vert,      halfW, halfH);
                    else              out[outcount++] = hz_clip_interp(dp / denom,       vert,      vert_prev, halfW, halfH);
                }
            }
            vert_prev = vert;
            dp_prev   = dp;
        }
        hz_cvert_t* tmp = in; in = out; out = tmp;
        n = outcount;
    }
    *poly = in;
    return n;
}

/* planes of the shade and of the texture coordinates over triangle (a,b,c):
 * hz_tri_planes() arithmetic on three more attributes (textured resolve) */
typedef struct { float r_org, drdx, drdy, s_org, dsdx, dsdy, t_org, dtdx, dtdy; } hz_texplanes_t;
HZ_HD void hz_tri_planes_tex(hz_texplanes_t* o, const hz_cvert_t* a, const hz_cvert_t* b, const hz_cvert_t* c)
{
    const hz_cvert_t *v0 = b, *v1 = a, *v2 = c;
    const float dx01 = v0->wx - v1->wx, dy01 = v0->wy - v1->wy;
    const float dx20 = v2->wx - v0->wx, dy20 = v2->wy - v0->wy;
    const float ooa  = 1.0f / (dx01*dy20 - dx20*dy01);
    const float dy20_ooa = dy20*ooa, dy01_ooa = dy01*ooa, dx20_ooa = dx20*ooa, dx01_ooa = dx01*ooa;
    const float x0c = v0->wx - 0.5f, y0c = v0->wy - 0.5f;
    const float dr01 = v0->red - v1->red, dr20 = v2->red - v0->red;
    const float ds01 = v0->s   - v1->s,   ds20 = v2->s   - v0->s;
    const float dt01 = v0->t   - v1->t,   dt20 = v2->t   - v0->t;
    o->drdx  = dr01*dy20_ooa - dr20*dy01_ooa;
    o->drdy  = dr20*dx01_ooa - dr01*dx20_ooa;
    o->r_org = v0->red - (o->drdx*x0c + o->drdy*y0c);
    o->dsdx  = ds01*dy20_ooa - ds20*dy01_ooa;
    o->dsdy  = ds20*dx01_ooa - ds01*dx20_ooa;
    o->s_org = v0->s - (o->dsdx*x0c + o->dsdy*y0c);
    o->dtdx  = dt01*dy20_ooa - dt20*dy01_ooa;
    o->dtdy  = dt20*dx01_ooa - dt01*dx20_ooa;
    o->t_org = v0->t - (o->dtdx*x0c + o->dtdy*y0c);
}
HZ_HD float hz_plane_at(float org, float ddx, float ddy, int px, int py)
{
    return __builtin_fmaf(ddy, (float)py, __builtin_fmaf(ddx, (float)px, org));
}

/* a clipper vertex as the rasteriser wants it */
HZ_HD hz_wvert_t hz_wvert_of(const hz_cvert_t* c)
{
    hz_wvert_t w;
    w.xn = c->xn; w.wx = c->wx; w.wy = c->wy; w.zw = c->zw; w.red = c->red; w.cmask = 0;
    const float fx = w.wx - 0.5f, fy = w.wy - 0.5f;
    if(hz_abs(fx) <= HZ_GUARD_PX && hz_abs(fy) <= HZ_GUARD_PX)
    {
        w.xs = (int32_t)hz_roundeven(fx*256.f);
        w.ys = (int32_t)hz_roundeven(fy*256.f);
    }
    else { w.xs = HZ_OUTSIDE_GUARD; w.ys = 0; }
    return w;
}

HZ_HD uint64_t hz_pack(uint32_t zi, uint32_t prim, uint32_t r8)
{
    return ((uint64_t)zi << 40) | ((uint64_t)prim << 8) | (uint64_t)r8;
}
