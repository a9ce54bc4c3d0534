/* oracle_render.c - TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of one draw of the reference: every vertex through
 * vertex.glsl, every triangle of the index buffer in draw order through
 * geometry.glsl's discard, back-face cull, rasterisation and a GL_LESS depth
 * test against a 24-bit depth buffer, then the reference's readback
 * conversions.  float32 throughout, as on the GPU the reference runs on.
 *
 * Build with -ffp-contract=off: every operation below is meant to round once.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle.h"

/* ---- shader arithmetic -------------------------------------------------- */
/*
 * The vertex stage below is not a free reading of vertex.glsl: it is the
 * operation sequence Mesa 23.2's GLSL compiler produces for that file (dump
 * with ST_DEBUG=nir while oracle/_ref/glsl_golden runs), which is what
 * llvmpipe then executes in IEEE float32 (fdiv, frcp = 1/x and fsqrt are
 * exact there).  Following it operation for operation makes this function
 * BIT-IDENTICAL to the reference's vertex shader on llvmpipe
 * (tests/test_oracle_golden.py checks that on captured vertices).  What the
 * compiler did to the source, and what therefore shows up below:
 *   - Rearth*pi folded to one constant, multiplied before DEG_PER_CELL
 *   - atan(y,x) lowered to a degree-11 polynomial with s*(1/t), not s/t
 *   - length() as sqrt of sums in the order n*n + e*e (+ h*h first for vec3)
 *   - unwrap_near_rad(az, c) - c and az1' - az0 simplified algebraically
 */

static const float C_REARTH_PI = 20015088.0f;       /* 0x4b98b3f8 = float(Rearth*pi) */
static const float C_PI        = 3.14159274f;       /* 0x40490fdb */
static const float C_TWO_PI    = 6.28318548f;       /* 0x40c90fdb */
static const float C_HALF_PI   = 1.57079637f;       /* 0x3fc90fdb */
static const float C_DEG2RAD   = 0.0174532924f;     /* 0x3c8efa35, radians() */

/* GLSL round() on llvmpipe: to nearest, ties to even */
static float glsl_round(float x) { return rintf(x); }

/* GLSL atan(y,x) as lowered by Mesa (nir_atan2 + nir_atan) */
static float glsl_atan2(float y, float x)
{
    int   flip = 0.f >= x;
    float ax = fabsf(x);
    float s = flip ? ax : y;
    float t = flip ? y  : ax;
    float scale = fabsf(t) >= 1e18f ? 0.25f : 1.0f;
    float rcp = 1.0f / (t*scale);
    float s_over_t = (s*scale) * rcp;
    float tn = (ax == fabsf(y)) ? 1.0f : fabsf(s_over_t);

    /* atan(tn), tn >= 0: argument reduced to [0,1] */
    float hi = tn > 1.0f ? tn : 1.0f;
    float lo = tn < 1.0f ? tn : 1.0f;
    float u   = lo / hi;
    float u2  = u*u;
    float u3  = u2*u;
    float u5  = u3*u2;
    float u7  = u5*u2;
    float u9  = u7*u2;
    float p = u*0.9999793128310355f + u3*-0.3326756418091246f;
    p = p + u5*0.1938924977115610f;
    p = p + u7*-0.1173503194786851f;
    p = p + u9*0.0536813784310406f;
    p = p + (u9*-0.0121323213173444f)*u2;
    float big = (1.0f < tn) ? 1.0f : 0.0f;
    float a = big*(p*-2.0f + C_HALF_PI) + p;
    float sign = tn > 0.f ? 1.0f : (tn < 0.f ? -1.0f : 0.0f);
    a = a*sign;

    float arc = (flip ? 1.0f : 0.0f)*C_HALF_PI + a;
    float m = y < rcp ? y : rcp;
    return m < 0.f ? -arc : arc;
}

typedef struct { float x, y, z, red; } ndc_t;

/* reference vertex.glsl:139-150: the two per-draw constants that survive the
 * compiler's simplification: span = az_rad1' - az_rad0 and the view centre */
static void az_constants(const orc_view_t* v, float* az_rad_center, float* az_ndc_per_rad)
{
    float az_rad0 = v->az_deg0 * C_DEG2RAD;
    float az_rad1 = v->az_deg1 * C_DEG2RAD;
    /* unwrap_near_rad(az_rad1-az_rad0, pi), reference vertex.glsl:143 */
    float d    = ((az_rad1 + -C_PI) + -az_rad0) / C_TWO_PI;
    float span = C_TWO_PI*(d - glsl_round(d)) + C_PI;
    az_rad1 = span + az_rad0;
    *az_rad_center  = (az_rad0 + az_rad1) / 2.0f;           /* vertex.glsl:146 */
    *az_ndc_per_rad = 2.0f / span;                          /* vertex.glsl:150 */
}

static ndc_t vertex_shader(const orc_view_t* v, float az_rad_center, float az_ndc_per_rad,
                           float i, float j, float height)
{
    /* reference vertex.glsl:128-131 */
    float e = (i - v->viewer_cell_i) * C_REARTH_PI * v->deg_per_cell / 180.0f * v->cos_viewer_lat;
    float n = (j - v->viewer_cell_j) * C_REARTH_PI * v->deg_per_cell / 180.0f;
    float h = height - v->viewer_z;

    /* reference vertex.glsl:133-134 */
    float nn = n*n, ee = e*e;
    float distance_ne = sqrtf(nn + ee);
    float az_rad = glsl_atan2(e, n);

    /* reference vertex.glsl:148,152: (unwrap_near_rad(az, c) - c) * k */
    float d = (az_rad + -az_rad_center) / C_TWO_PI;
    ndc_t o;
    o.x = (C_TWO_PI*(d - glsl_round(d))) * az_ndc_per_rad;
    /* reference vertex.glsl:153 */
    o.y = glsl_atan2(h, distance_ne) * v->aspect * az_ndc_per_rad;
    /* reference vertex.glsl:155 */
    o.z = (sqrtf(h*h + nn + ee) - v->znear) / (v->zfar - v->znear) * 2.0f + -1.0f;

    /* reference vertex.glsl:159-160 */
    float r = (distance_ne - v->znear_color) / (v->zfar_color - v->znear_color);
    r = r > 0.0f ? r : 0.0f;
    r = r < 1.0f ? r : 1.0f;
    o.red = r;
    return o;
}

void orc_vertex(const orc_view_t* v, int i, int j, int z, float out_xyzr[4])
{
    float c, k;
    az_constants(v, &c, &k);
    ndc_t o = vertex_shader(v, c, k, (float)i, (float)j, (float)z);
    out_xyzr[0] = o.x; out_xyzr[1] = o.y; out_xyzr[2] = o.z; out_xyzr[3] = o.red;
}

/* ---- texture path: host side and vertex stage ------------------------------- */

void orc_osm_tile_id(int* x, int* y, float E, float N)
{
    /* reference horizonator-lib.c:225-246 */
    float n = (float)(1 << ORC_OSM_ZOOM);
    E *= (float)M_PI/180.0f;
    N *= (float)M_PI/180.0f;
    float lon0 = n / 2.0f;
    float lon1 = n / ((float)M_PI * 2.0f);
    *x = (int)( fminf( n, fmaxf( 0.0f, E*lon1 + lon0 )));
    *y = (int)( n/2.0f * (1.0f - logf( (sinf(N) + 1.0f)/cosf(N) ) / (float)M_PI) );
}

void orc_tex_setup(orc_tex_t* t, const orc_dem_t* d, float init_lat, float init_lon, float viewer_lat)
{
    /* reference horizonator-lib.c:372-389 */
    const float lowest_E  = init_lon - (float)d->radius_cells/d->cells_per_deg;
    const float lowest_N  = init_lat - (float)d->radius_cells/d->cells_per_deg;
    const float highest_E = init_lon + (float)d->radius_cells/d->cells_per_deg;
    const float highest_N = init_lat + (float)d->radius_cells/d->cells_per_deg;
    int hx, hy;
    orc_osm_tile_id(&t->lowest_x, &t->lowest_y, lowest_E,  highest_N);      /* ytile decreases with lat */
    orc_osm_tile_id(&hx,          &hy,          highest_E, lowest_N);
    t->ntiles_x = hx - t->lowest_x + 1;
    t->ntiles_y = hy - t->lowest_y + 1;
    t->tex_w = t->ntiles_x*ORC_OSM_TILE_PX;
    t->tex_h = t->ntiles_y*ORC_OSM_TILE_PX;
    /* reference :577-582 */
    t->origin_cell_lon_deg = (float)d->origin_tile[0] + (float)d->origin_cell[0] / (float)d->cells_per_deg;
    t->origin_cell_lat_deg = (float)d->origin_tile[1] + (float)d->origin_cell[1] / (float)d->cells_per_deg;
    /* reference :707-759 texture_coeffs(viewer_lat) */
    {
        float n = (float)(1 << ORC_OSM_ZOOM);
        t->lon0 = n / 2.0f;
        t->lon1 = n / ((float)M_PI * 2.0f);
        float lat_center = viewer_lat * ((float)M_PI / 180.0f);
        float k = -n / ((float)M_PI * 2.0f);
        float tn = tanf( lat_center );
        float c  = cosf( lat_center );
        t->dlat0 = n/2.0f + k*logf( tn + 1.0f/c );
        t->dlat1 = k / c;
        t->dlat2 = k * tn / c / 2.0f;
    }
    /* reference :801: the product is formed in double, glUniform1f rounds it */
    t->viewer_lat_rad = (float)(viewer_lat * M_PI / 180.0f);
}

/* reference vertex.glsl:116-126 with get_xtexture/get_ytexture (:53-61) in the
 * operation order of Mesa's compiler (ST_DEBUG=nir) */
static void vertex_tex(const orc_tex_t* t, float deg_per_cell, float i, float j, float* s_out, float* t_out)
{
    const float lat  = (t->origin_cell_lat_deg + j*deg_per_cell) * C_DEG2RAD;
    const float dlat = lat + -t->viewer_lat_rad;
    const float lon  = (t->origin_cell_lon_deg + i*deg_per_cell) * C_DEG2RAD;
    const float xt   = t->lon1*lon + t->lon0;
    *s_out = (xt + -(float)t->lowest_x) / (float)t->ntiles_x;
    const float yt   = dlat*(dlat*t->dlat2 + t->dlat1) + t->dlat0;
    *t_out = 1.0f + -((yt + -(float)t->lowest_y) / (float)t->ntiles_y);
}

void orc_vertex_tex(const orc_tex_t* t, float deg_per_cell, int i, int j, float out_st[2])
{
    vertex_tex(t, deg_per_cell, (float)i, (float)j, &out_st[0], &out_st[1]);
}

/* ---- texture path: the sampler (llvmpipe, GL_LINEAR, GL_REPEAT, RGB8) ------------ */

static void tex_wrap_linear(float s, int size, int* i0, int* i1, int* w)
{
    const int pot = (size & (size-1)) == 0;
    if(!pot)
    {
        s = s - floorf(s);
        if(!(s <= 0.99999994f)) s = 0.99999994f;
    }
    /* 24.8 fixed point: round to nearest even of s*size*256, then -0.5 texel */
    const int fixed = (int)lrintf((s*(float)size)*256.0f) - 128;
    const int ip = fixed >> 8;
    *w = fixed & 255;
    if(pot) { *i0 = ip & (size-1); *i1 = (ip+1) & (size-1); }
    else    { *i0 = ip < 0 ? size-1 : ip; *i1 = ip+1 > size-1 ? 0 : ip+1; }
}

void orc_tex_sample(const orc_tex_t* t, float s, float tt, uint8_t out_bgr[3])
{
    int i0, i1, wx, j0, j1, wy;
    tex_wrap_linear(s,  t->tex_w, &i0, &i1, &wx);
    tex_wrap_linear(tt, t->tex_h, &j0, &j1, &wy);
    const uint8_t* t00 = &t->texels[((size_t)j0*t->tex_w + i0)*3];
    const uint8_t* t10 = &t->texels[((size_t)j0*t->tex_w + i1)*3];
    const uint8_t* t01 = &t->texels[((size_t)j1*t->tex_w + i0)*3];
    const uint8_t* t11 = &t->texels[((size_t)j1*t->tex_w + i1)*3];
    for(int c=0; c<3; c++)
    {
        const int a = t00[c] + ((wx*((int)t10[c] - (int)t00[c]) + 128) >> 8);
        const int b = t01[c] + ((wx*((int)t11[c] - (int)t01[c]) + 128) >> 8);
        out_bgr[c] = (uint8_t)(a + ((wy*(b - a) + 128) >> 8));
    }
}

/* fragment.glsl:17-22 as compiled: R = 0.7*tex.r + 0.3*shade, G,B = 0.7*tex.g,b,
 * separate multiplies and add; then the RGB8 render target */
static uint8_t unorm8(float x)
{
    x = x < 1.0f ? x : 1.0f;
    x = x > 0.0f ? x : 0.0f;
    return (uint8_t)rintf(x * 255.f);
}
static void fragment_textured(const orc_tex_t* t, float s, float tt, float shade, uint8_t out_bgr[3])
{
    uint8_t texel[3];
    orc_tex_sample(t, s, tt, texel);
    const float b = 0.7f*((float)texel[0]*(1.0f/255.0f));
    const float g = 0.7f*((float)texel[1]*(1.0f/255.0f));
    const float r = 0.7f*((float)texel[2]*(1.0f/255.0f)) + 0.3f*shade;
    out_bgr[0] = unorm8(b); out_bgr[1] = unorm8(g); out_bgr[2] = unorm8(r);
}

/* ---- rasteriser --------------------------------------------------------- */

/* a vertex as it leaves the vertex stage: clip-space position (w = 1), its
 * viewport transform (window coordinates with pixel centres at half-integers,
 * depth in [0,1]) and colour */
typedef struct { float xn, yn, zn, wx, wy, zw, red, s, t; } wvert_t;

#define GUARD_PX 2097152.0f

typedef struct
{
    uint32_t* depth;        /* [H][SW] GL row order, 24-bit values */
    uint32_t* prim;         /* [H][SW]; ids reach beyond 2^31 on BASELINE's largest mosaic */
    uint8_t*  red;          /* [H][SW] */
    uint8_t*  color;        /* [H][SW][3] B,G,R - textured draws only */
    const orc_tex_t* tex;   /* NULL: fragment.glsl:15-16, else :17-22 */
    int SW, H, col0, col1;
} target_t;

/* rasterise one window-space triangle (a whole one, or a piece the clipper made) */
static void raster_triangle(target_t* fb, int x_lo, int x_hi,
                            const wvert_t* A, const wvert_t* B, const wvert_t* C, uint32_t prim)
{
    /* positions relative to the pixel-centre grid (llvmpipe's "pixel offset"):
     * centres at integers */
    const wvert_t* V[3] = {A,B,C};
    float fx[3], fy[3];
    for(int m=0; m<3; m++) { fx[m] = V[m]->wx - 0.5f; fy[m] = V[m]->wy - 0.5f; }

    /* guard band (also catches non-finite positions) */
    for(int m=0; m<3; m++)
        if(!(fabsf(fx[m]) <= GUARD_PX && fabsf(fy[m]) <= GUARD_PX)) return;

    /* positions snapped to 1/256 pixel decide coverage and facing */
    int64_t X[3], Y[3];
    for(int m=0; m<3; m++)
    {
        X[m] = (int64_t)rintf(fx[m] * 256.f);
        Y[m] = (int64_t)rintf(fy[m] * 256.f);
    }
    /* glEnable(GL_CULL_FACE), default GL_BACK / GL_CCW (reference
     * horizonator-lib.c:184): counter-clockwise in window space is front */
    int64_t area = (X[1]-X[0])*(Y[2]-Y[0]) - (X[2]-X[0])*(Y[1]-Y[0]);
    if(area <= 0) return;

    int64_t bx0 = X[0], bx1 = X[0], by0 = Y[0], by1 = Y[0];
    for(int m=1; m<3; m++)
    {
        if(X[m] < bx0) bx0 = X[m];
        if(X[m] > bx1) bx1 = X[m];
        if(Y[m] < by0) by0 = Y[m];
        if(Y[m] > by1) by1 = Y[m];
    }
    /* integer pixel centres inside the box; floor division by 256 */
    int64_t px0 = (bx0 + 255) >> 8, px1 = bx1 >> 8;
    int64_t py0 = (by0 + 255) >> 8, py1 = by1 >> 8;
    if(px0 < x_lo)    px0 = x_lo;
    if(px1 > x_hi)    px1 = x_hi;
    if(py0 < 0)       py0 = 0;
    if(py1 > fb->H-1) py1 = fb->H-1;
    if(px0 > px1 || py0 > py1) return;

    /* Depth and colour are planes through the unsnapped positions, set up and
     * evaluated the way llvmpipe does it (pinned on the golden draws: with this
     * arithmetic the depth of every unclipped triangle comes out bit-identical):
     *   - llvmpipe sees the front faces of an FBO draw as clockwise and swaps
     *     the first two vertices: its v0 is the SECOND vertex of the draw call
     *   - gradients through ooa = 1/area and four pre-multiplied edge deltas
     *   - a0 = value at the window origin, value(px,py) = fma(dady,py, fma(dadx,px,a0)) */
    const wvert_t *v0 = B, *v1 = A, *v2 = C;
    float dx01 = v0->wx - v1->wx, dy01 = v0->wy - v1->wy;
    float dx20 = v2->wx - v0->wx, dy20 = v2->wy - v0->wy;
    float ooa  = 1.0f / (dx01*dy20 - dx20*dy01);
    float dy20_ooa = dy20*ooa, dy01_ooa = dy01*ooa, dx20_ooa = dx20*ooa, dx01_ooa = dx01*ooa;
    float x0_center = v0->wx - 0.5f, y0_center = v0->wy - 0.5f;
    float dz01 = v0->zw  - v1->zw,  dz20 = v2->zw  - v0->zw;
    float dr01 = v0->red - v1->red, dr20 = v2->red - v0->red;
    float dzdx = dz01*dy20_ooa - dz20*dy01_ooa;
    float dzdy = dz20*dx01_ooa - dz01*dx20_ooa;
    float z_org = v0->zw  - (dzdx*x0_center + dzdy*y0_center);
    float drdx = dr01*dy20_ooa - dr20*dy01_ooa;
    float drdy = dr20*dx01_ooa - dr01*dx20_ooa;
    float r_org = v0->red - (drdx*x0_center + drdy*y0_center);
    /* texture coordinates: two more attributes, same arithmetic */
    float ds01 = v0->s - v1->s, ds20 = v2->s - v0->s;
    float dt01 = v0->t - v1->t, dt20 = v2->t - v0->t;
    float dsdx = ds01*dy20_ooa - ds20*dy01_ooa;
    float dsdy = ds20*dx01_ooa - ds01*dx20_ooa;
    float s_org = v0->s - (dsdx*x0_center + dsdy*y0_center);
    float dtdx = dt01*dy20_ooa - dt20*dy01_ooa;
    float dtdy = dt20*dx01_ooa - dt01*dx20_ooa;
    float t_org = v0->t - (dtdx*x0_center + dtdy*y0_center);

    for(int64_t py=py0; py<=py1; py++)
        for(int64_t px=px0; px<=px1; px++)
        {
            /* edge functions at the pixel centre; a centre exactly on an edge
             * belongs to the triangle if that edge is a left edge or a
             * horizontal bottom edge (y pointing up): llvmpipe's rule for a
             * framebuffer object, pinned by tests/golden/raster_probe.npz */
            int inside = 1;
            for(int m=0; m<3 && inside; m++)
            {
                int a = m, b = (m+1)%3;
                int64_t dx = X[b]-X[a], dy = Y[b]-Y[a];
                int64_t E = dx*(py*256 - Y[a]) - dy*(px*256 - X[a]);
                if(E < 0) inside = 0;
                else if(E == 0 && !(dy < 0 || (dy == 0 && dx > 0))) inside = 0;
            }
            if(!inside) continue;

            /* every vertex lies inside the depth range by now (clipper); what
             * interpolation rounding leaves outside is clamped, as llvmpipe does */
            float z = fmaf(dzdy, (float)py, fmaf(dzdx, (float)px, z_org));
            if(!(z == z)) continue;
            z = z < 0.f ? 0.f : (z > 1.f ? 1.f : z);
            uint32_t zi = (uint32_t)rintf(z * 16777215.f); /* 24-bit unorm */

            /* GL_LESS against what was drawn before; among equal depths the
             * triangle drawn first (lowest primitive id) stays */
            const size_t at = (size_t)py*fb->SW + (size_t)(px - fb->col0);
            if(!(zi < fb->depth[at] || (zi == fb->depth[at] && zi != 0xFFFFFFu && prim < fb->prim[at]))) continue;

            float r = fmaf(drdy, (float)py, fmaf(drdx, (float)px, r_org));
            fb->depth[at] = zi;
            fb->prim [at] = prim;
            if(fb->tex)
            {
                const float s  = fmaf(dsdy, (float)py, fmaf(dsdx, (float)px, s_org));
                const float tt = fmaf(dtdy, (float)py, fmaf(dtdx, (float)px, t_org));
                fragment_textured(fb->tex, s, tt, r, &fb->color[3*at]);
            }
            r = r < 1.0f ? r : 1.0f;
            r = r > 0.0f ? r : 0.0f;
            fb->red  [at] = (uint8_t)rintf(r * 255.f);      /* RGB8 unorm */
        }
}

/* ---- clipping --------------------------------------------------------------
 * Between the geometry shader and the rasteriser GL clips every primitive
 * against the view volume -1 <= x,y,z <= 1 (w = 1 here).  For the reference
 * that concerns the triangles that cross the image border, the near sphere
 * (range = znear) or the far sphere (range = zfar).  The visible pixels are the
 * same with or without it, but the clipper cuts such a triangle into a fan of
 * smaller ones with NEW vertices, and depth/colour are then interpolated over
 * those.  This is Mesa's clipper (draw_pipe_clip.c) restated operation for
 * operation; with it the depth of clipped triangles matches llvmpipe bit for
 * bit too (probe: 6 769 pixels of random triangles cut by the y, z and y+z
 * planes: all equal).
 */
#define MAX_CLIPPED 12

static unsigned clip_mask(const wvert_t* v)
{
    unsigned m = 0;
    if(v->xn > 1.0f)        m |= 1;
    if(v->xn + 1.0f < 0.f)  m |= 2;
    if(v->yn > 1.0f)        m |= 4;
    if(v->yn + 1.0f < 0.f)  m |= 8;
    if(v->zn + 1.0f < 0.f)  m |= 16;
    if(v->zn > 1.0f)        m |= 32;
    return m;
}

/* signed distance to clip plane p (w = 1): dot4 with the plane, left to right */
static float clip_dist(const wvert_t* v, int p)
{
    static const float plane[6][4] = { {-1,0,0,1}, {1,0,0,1}, {0,-1,0,1}, {0,1,0,1}, {0,0,1,1}, {0,0,-1,1} };
    return plane[p][0]*v->xn + plane[p][1]*v->yn + plane[p][2]*v->zn + plane[p][3]*1.0f;
}

/* new vertex at parameter t on the way from `out` to `in` */
static wvert_t clip_interp(float t, const wvert_t* out, const wvert_t* in, float halfW, float halfH)
{
    wvert_t d;
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

static void clip_and_raster(target_t* fb, int x_lo, int x_hi, float halfW, float halfH,
                            const wvert_t* A, const wvert_t* B, const wvert_t* C, uint32_t prim);

static void draw_triangle(target_t* fb, int x_lo, int x_hi, float halfW, float halfH,
                          const wvert_t* A, const wvert_t* B, const wvert_t* C, uint32_t prim)
{
    /* reference geometry.glsl:21-27 */
    float xmax = A->xn > B->xn ? A->xn : B->xn; xmax = xmax > C->xn ? xmax : C->xn;
    float xmin = A->xn < B->xn ? A->xn : B->xn; xmin = xmin < C->xn ? xmin : C->xn;
    if(xmax - xmin > 0.5f) return;
    clip_and_raster(fb, x_lo, x_hi, halfW, halfH, A, B, C, prim);
}

/* what GL does with a primitive after the geometry stage */
static void clip_and_raster(target_t* fb, int x_lo, int x_hi, float halfW, float halfH,
                            const wvert_t* A, const wvert_t* B, const wvert_t* C, uint32_t prim)
{
    const unsigned ma = clip_mask(A), mb = clip_mask(B), mc = clip_mask(C);
    if(ma & mb & mc) return;                        /* wholly outside one plane */
    unsigned todo = ma | mb | mc;
    if(!todo) { raster_triangle(fb, x_lo, x_hi, A, B, C, prim); return; }

    /* Sutherland-Hodgman, one plane after the other in Mesa's order */
    wvert_t bufa[MAX_CLIPPED+1], bufb[MAX_CLIPPED+1];
    wvert_t *in = bufa, *out = bufb;
    in[0] = *A; in[1] = *B; in[2] = *C;
    int n = 3;
    while(todo && n >= 3)
    {
        const int p = __builtin_ctz(todo);
        todo &= ~(1u << p);
        int outcount = 0;
        in[n] = in[0];
        const wvert_t* vert_prev = &in[0];
        float dp_prev = clip_dist(vert_prev, p);
        if(!(dp_prev == dp_prev) || isinf(dp_prev)) return;
        for(int i=1; i<=n; i++)
        {
            const wvert_t* vert = &in[i];
            const float dp = clip_dist(vert, p);
            if(!(dp == dp) || isinf(dp)) return;
            int different_sign;
            if(dp_prev >= 0.0f)
            {
                if(outcount >= MAX_CLIPPED) return;
                out[outcount++] = *vert_prev;
                different_sign = dp < 0.0f;
            }
            else
                different_sign = !(dp < 0.0f);
            if(different_sign)
            {
                if(outcount >= MAX_CLIPPED) return;
                const float denom = dp - dp_prev;
                /* always interpolate from the inside vertex towards the outside one */
                if(dp < 0.0f)
                {
                    if(-dp < dp_prev) out[outcount++] = clip_interp(dp / denom,       vert,      vert_prev, halfW, halfH);
                    else              out[outcount++] = clip_interp(-dp_prev / denom, vert_prev, vert,      halfW, halfH);
                }
                else
                {
                    if(-dp_prev < dp) out[outcount++] = clip_interp(-dp_prev / denom, vert_prev, vert,      halfW, halfH);
                    else              out[outcount++] = clip_interp(dp / denom,       vert,      vert_prev, halfW, halfH);
                }
            }
            vert_prev = vert;
            dp_prev   = dp;
        }
        wvert_t* tmp = in; in = out; out = tmp;
        n = outcount;
    }
    /* the polygon goes on as a fan that keeps vertex 0 last (GL provoking vertex) */
    for(int i=2; i<n; i++)
        raster_triangle(fb, x_lo, x_hi, &in[i-1], &in[i], &in[0], prim);
}

/* Raw clip-space triangles (x,y,z, shade, s,t per vertex; w = 1) through the
 * clipper, the rasteriser and - with a texture - the textured fragment stage,
 * in draw order.  The counterpart of glsl_golden's probe modes 2 and 5: lets
 * the tests hold this file's rasteriser and sampler directly against what
 * llvmpipe drew for the same triangles.  Outputs in GL row order (bottom
 * first): bgr [H][W][3] (clear colour where nothing was drawn), z24 [H][W]. */
int orc_draw_triangles(const float* tris, int ntri, int W, int H, const orc_tex_t* tex,
                       uint8_t* bgr, uint32_t* z24)
{
    const size_t npix = (size_t)W*H;
    target_t fb;
    fb.SW = W; fb.H = H; fb.col0 = 0; fb.col1 = W;
    fb.depth = malloc(npix*sizeof(uint32_t));
    fb.prim  = malloc(npix*sizeof(uint32_t));
    fb.red   = malloc(npix);
    fb.tex   = tex;
    fb.color = tex ? malloc(npix*3) : NULL;
    if(!fb.depth || !fb.prim || !fb.red || (tex && !fb.color))
    {
        free(fb.depth); free(fb.prim); free(fb.red); free(fb.color);
        return -1;
    }
    for(size_t k=0; k<npix; k++) { fb.depth[k] = 0xFFFFFFu; fb.prim[k] = 0xFFFFFFFFu; fb.red[k] = 0; }
    const float halfW = (float)W*0.5f, halfH = (float)H*0.5f;
    for(int n=0; n<ntri; n++)
    {
        wvert_t v[3];
        for(int m=0; m<3; m++)
        {
            const float* p = &tris[((size_t)n*3 + m)*6];
            v[m].xn = p[0]; v[m].yn = p[1]; v[m].zn = p[2];
            v[m].wx = p[0]*halfW + halfW; v[m].wy = p[1]*halfH + halfH; v[m].zw = p[2]*0.5f + 0.5f;
            v[m].red = p[3]; v[m].s = p[4]; v[m].t = p[5];
        }
        clip_and_raster(&fb, 0, W-1, halfW, halfH, &v[0], &v[1], &v[2], (uint32_t)n);
    }
    for(size_t k=0; k<npix; k++)
    {
        const int sky = fb.depth[k] == 0xFFFFFFu;
        if(z24) z24[k] = fb.depth[k];
        if(bgr)
        {
            bgr[3*k+0] = sky ? 255 : 0; bgr[3*k+1] = 0; bgr[3*k+2] = sky ? 0 : fb.red[k];
            if(tex && !sky) { bgr[3*k+0] = fb.color[3*k+0]; bgr[3*k+1] = fb.color[3*k+1]; bgr[3*k+2] = fb.color[3*k+2]; }
        }
    }
    free(fb.depth); free(fb.prim); free(fb.red); free(fb.color);
    return 0;
}

void orc_tanel(float* tanel, int W, int H, float az_deg0, float az_deg1)
{
    /* reference horizonator-lib.c:1006-1012 */
    float aspect = (float)W / (float)H;
    for(int row=0; row<H; row++)
    {
        /* rows of the upper half reuse the mirrored row's value, negated
         * (reference horizonator-lib.c:1033-1034); the sign is irrelevant to
         * the range, so the magnitude is stored */
        int y = row < H - H/2 ? row : H-1-row;
        float el_ndc = ((float)y + 0.5f) / (float)H * 2.f - 1.f;
        float el     = el_ndc * (az_deg1-az_deg0) / 2.f / aspect * M_PI/180.0f;
        tanel[row] = tanf(el);
    }
}

/* one pixel of the readback conversion (reference horizonator-lib.c:1013-1025) */
static float range_of_z24(uint32_t zi, float tanel_row, float znear, float zfar)
{
    /* glReadPixels(GL_DEPTH_COMPONENT, GL_FLOAT) of a Z24 buffer */
    float depth = (float)((double)zi * (1.0/16777215.0));
    if(depth == 1.0f) return -1.0f;                         /* reference :1016 */
    float length_en = depth * (zfar - znear) + znear;       /* :1018 */
    float zt = tanel_row * length_en;                       /* :1023 */
    return hypotf(length_en, zt);                           /* :1024 */
}

int orc_render(const int16_t* mosaic, int N, const orc_view_t* v,
               int W, int H, int col0, int col1,
               uint8_t* bgr, float* ranges, int32_t* index, uint32_t* z24,
               int nthreads)
{
    return orc_render_tex(mosaic, N, v, NULL, W, H, col0, col1, bgr, ranges, index, z24, nthreads);
}

int orc_render_tex(const int16_t* mosaic, int N, const orc_view_t* v, const orc_tex_t* tex,
                   int W, int H, int col0, int col1,
                   uint8_t* bgr, float* ranges, int32_t* index, uint32_t* z24,
                   int nthreads)
{
    if(N < 2 || W <= 0 || H <= 0 || col0 < 0 || col1 > W || col0 >= col1) return -1;
    const int SW = col1 - col0;
    const size_t npix = (size_t)SW*H;

    target_t fb;
    fb.SW = SW; fb.H = H; fb.col0 = col0; fb.col1 = col1;
    fb.depth = malloc(npix*sizeof(uint32_t));
    fb.prim  = malloc(npix*sizeof(uint32_t));
    fb.red   = malloc(npix);
    fb.tex   = tex;
    fb.color = tex ? malloc(npix*3) : NULL;
    /* vertices: position/depth/shade always, texture coordinates only when
     * texturing (the untextured draw is the CPU baseline of bench.py: keep it
     * lean).  The grid is walked in bands of BLK cell rows, and only the BLK+1
     * vertex rows of the current band are kept: 3.1 G triangles over 39600^2
     * samples (BASELINE configs[4]) would otherwise need 44 GB of vertices. */
    enum { BLK = 32 };
    typedef struct { float xn, yn, zn, wx, wy, zw, red; } pvert_t;
    /* cell rows per band: as many as ~2 GB of vertices allow (the whole grid up to
     * BASELINE's 7x7-tile mosaic: one band, no barrier in between), a multiple of BLK */
    int band = (int)(2000000000.0 / ((double)N*sizeof(pvert_t))) / BLK * BLK;
    if(band < BLK) band = BLK;
    if(band > N-1) band = ((N-1 + BLK-1)/BLK)*BLK;
    pvert_t* vert = malloc((size_t)(band+1)*N*sizeof(pvert_t));
    float (*vst)[2] = tex ? malloc((size_t)(band+1)*N*sizeof(*vst)) : NULL;
    float* tanel = malloc((size_t)H*sizeof(float));
    const int nbx = (N-1 + BLK-1)/BLK, nby = band/BLK, nb = nbx*nby;
    int* blk_lo = malloc((size_t)nb*sizeof(int));
    int* blk_hi = malloc((size_t)nb*sizeof(int));
    if(!fb.depth || !fb.prim || !fb.red || !vert || !tanel || !blk_lo || !blk_hi || (tex && (!fb.color || !vst)))
    {
        free(fb.depth); free(fb.prim); free(fb.red); free(fb.color); free(vert); free(vst); free(tanel); free(blk_lo); free(blk_hi);
        return -1;
    }

    /* glClear: depth 1.0 -> 0xFFFFFF, colour (0,0,1) (reference horizonator-lib.c:185,896) */
    for(size_t k=0; k<npix; k++) { fb.depth[k] = 0xFFFFFFu; fb.prim[k] = 0xFFFFFFFFu; fb.red[k] = 0; }

#ifdef _OPENMP
    if(nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

    float c, k;
    az_constants(v, &c, &k);
    const float halfW = (float)W*0.5f, halfH = (float)H*0.5f;

    for(int jb=0; jb<N-1; jb+=band)
    {
        const int jrows = (jb + band < N-1 ? band : N-1 - jb) + 1;      /* vertex rows of this band */
        /* vertex stage + viewport transform (glViewport(0,0,W,H), reference
         * horizonator-lib.c:657; depth range 0..1) */
        #pragma omp parallel for schedule(static) num_threads(nthreads) collapse(2)
        for(int jj=0; jj<jrows; jj++)
            for(int i=0; i<N; i++)
            {
                const int j = jb + jj;
                ndc_t o = vertex_shader(v, c, k, (float)i, (float)j, (float)mosaic[(size_t)j*N + i]);
                pvert_t* w = &vert[(size_t)jj*N + i];
                w->xn  = o.x; w->yn = o.y; w->zn = o.z;
                w->wx  = o.x*halfW + halfW;
                w->wy  = o.y*halfH + halfH;
                w->zw  = o.z*0.5f + 0.5f;
                w->red = o.red;
                if(tex) vertex_tex(tex, v->deg_per_cell, (float)i, (float)j, &vst[(size_t)jj*N + i][0], &vst[(size_t)jj*N + i][1]);
            }

        /* Triangles of the index buffer (reference horizonator-lib.c:496-508).
         * GL draws them in order and keeps the first of equal depths (GL_LESS);
         * draw_triangle() implements that as "lower depth wins, then lower
         * primitive id", which does not depend on the order of processing.  So the
         * band can be walked in blocks of 32x32 cells, threads owning disjoint
         * column strips of the image and skipping the blocks whose window x-range
         * (from their 33x33 vertices) misses their strip. */
        #pragma omp parallel for schedule(dynamic,8) num_threads(nthreads)
        for(int b=0; b<nb; b++)
        {
            const int ib = (b%nbx)*BLK, jj0 = (b/nbx)*BLK;
            if(jj0 >= jrows-1) { blk_lo[b] = INT32_MAX; blk_hi[b] = INT32_MIN; continue; }     /* beyond the last (short) band */
            float lo = INFINITY, hi = -INFINITY, nlo = INFINITY, nhi = -INFINITY;
            for(int jj=jj0; jj<=jj0+BLK && jj<jrows; jj++)
                for(int i=ib; i<=ib+BLK && i<N; i++)
                {
                    const pvert_t* w = &vert[(size_t)jj*N + i];
                    if(!(w->wx >= lo)) lo = w->wx;      /* NaN-proof: a NaN widens the range */
                    if(!(w->wx <= hi)) hi = w->wx;
                    if(w->xn < nlo) nlo = w->xn;
                    if(w->xn > nhi) nhi = w->xn;
                }
            /* a block that reaches across the +-180 degree seam (or holds the
             * viewer) spans the image: keep it for every strip */
            if(!(lo == lo) || !(hi == hi) || nhi - nlo > 0.5f) { blk_lo[b] = INT32_MIN; blk_hi[b] = INT32_MAX; }
            else { blk_lo[b] = (int)floorf(fmaxf(lo, -1e9f)) - 2; blk_hi[b] = (int)ceilf(fminf(hi, 1e9f)) + 2; }
        }
        #pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
            const int tid = 0, nth = 1;
#endif
            const int x_lo = col0 + (int)((long long)SW*tid/nth);
            const int x_hi = col0 + (int)((long long)SW*(tid+1)/nth) - 1;
            if(x_lo <= x_hi)
                for(int b=0; b<nb; b++)
                {
                    const int ib = (b%nbx)*BLK, jj0 = (b/nbx)*BLK;
                    if(jj0 >= jrows-1) break;
                    if(blk_hi[b] < x_lo || blk_lo[b] > x_hi) continue;
                    for(int jj=jj0; jj<jj0+BLK && jj<jrows-1; jj++)
                        for(int i=ib; i<ib+BLK && i<N-1; i++)
                        {
                            wvert_t q[4];               /* v00 v10 v01 v11 */
                            for(int m=0; m<4; m++)
                            {
                                const size_t at = (size_t)(jj + (m >> 1))*N + i + (m & 1);
                                const pvert_t* pv = &vert[at];
                                q[m].xn = pv->xn; q[m].yn = pv->yn; q[m].zn = pv->zn;
                                q[m].wx = pv->wx; q[m].wy = pv->wy; q[m].zw = pv->zw; q[m].red = pv->red;
                                q[m].s = tex ? vst[at][0] : 0.f; q[m].t = tex ? vst[at][1] : 0.f;
                            }
                            const uint32_t prim = (uint32_t)(((uint64_t)(jb + jj)*(uint64_t)(N-1) + (uint64_t)i)*2u);
                            draw_triangle(&fb, x_lo, x_hi, halfW, halfH, &q[0], &q[3], &q[2], prim  );
                            draw_triangle(&fb, x_lo, x_hi, halfW, halfH, &q[0], &q[1], &q[3], prim+1);
                        }
                }
        }
    }
    free(blk_lo); free(blk_hi);

    /* readback (reference horizonator-lib.c:936-1048) */
    orc_tanel(tanel, W, H, v->az_deg0, v->az_deg1);
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for(int yo=0; yo<H; yo++)
    {
        const int row = H-1-yo;         /* top row first */
        for(int x=0; x<SW; x++)
        {
            const size_t at = (size_t)row*SW + x, o = (size_t)yo*SW + x;
            const uint32_t zi = fb.depth[at];
            const int sky = zi == 0xFFFFFFu;
            if(bgr)
            {
                bgr[3*o+0] = sky ? 255 : 0;
                bgr[3*o+1] = 0;
                bgr[3*o+2] = sky ? 0 : fb.red[at];
                if(tex && !sky) { bgr[3*o+0] = fb.color[3*at+0]; bgr[3*o+1] = fb.color[3*at+1]; bgr[3*o+2] = fb.color[3*at+2]; }
            }
            if(index) index[o] = (int32_t)fb.prim[at];      /* sky: -1 */
            if(z24)   z24[o]   = zi;
            if(ranges)
            {
                ranges[o] = range_of_z24(zi, tanel[row], v->znear, v->zfar);
            }
        }
    }

    free(fb.depth); free(fb.prim); free(fb.red); free(fb.color); free(vert); free(vst); free(tanel);
    return 0;
}

/* The readback conversion alone (reference horizonator-lib.c:1006-1047) on a depth image given as raw 24-bit values in
 * GL row order (row 0 = bottom); ranges come out top row first.  tests/test_naive_host_math.py compares it with a
 * third, independent restatement. */
void orc_ranges_from_z24(float* ranges, const uint32_t* z24_gl, int W, int H,
                         float az_deg0, float az_deg1, float znear, float zfar)
{
    float* tanel = malloc((size_t)H*sizeof(float));
    orc_tanel(tanel, W, H, az_deg0, az_deg1);
    for(int yo=0; yo<H; yo++)
    {
        const int row = H-1-yo;
        for(int x=0; x<W; x++)
            ranges[(size_t)yo*W + x] = range_of_z24(z24_gl[(size_t)row*W + x], tanel[row], znear, zfar);
    }
    free(tanel);
}

/* ---- workload statistics (design aid, not part of any comparison) -------- */

/* counts[0] triangles, [1] after geometry-shader discard, [2] after guard band,
 * [3] front-facing, [4] non-empty pixel box, [5] inside depth range,
 * [6] pixel centres tested (box area), [7] covered, [8] depth-clipped away,
 * hist[k]: triangles whose box holds 2^(k-1) < n <= 2^k pixel centres (hist[0]: n = 1) */
void orc_stats(const int16_t* mosaic, int N, const orc_view_t* v, int W, int H,
               int64_t counts[9], int64_t hist[32])
{
    memset(counts, 0, 9*sizeof(int64_t));
    memset(hist, 0, 32*sizeof(int64_t));
    float c, k;
    az_constants(v, &c, &k);
    const float halfW = (float)W*0.5f, halfH = (float)H*0.5f;
    wvert_t* row0 = malloc((size_t)N*sizeof(wvert_t));
    wvert_t* row1 = malloc((size_t)N*sizeof(wvert_t));
    for(int j=0; j<N; j++)
    {
        for(int i=0; i<N; i++)
        {
            ndc_t o = vertex_shader(v, c, k, (float)i, (float)j, (float)mosaic[(size_t)j*N + i]);
            wvert_t* w = &row1[i];
            w->xn = o.x; w->yn = o.y; w->zn = o.z; w->wx = o.x*halfW + halfW; w->wy = o.y*halfH + halfH;
            w->zw = o.z*0.5f + 0.5f; w->red = o.red;
        }
        if(j > 0)
            for(int i=0; i<N-1; i++)
                for(int t=0; t<2; t++)
                {
                    const wvert_t* A = &row0[i];
                    const wvert_t* B = t == 0 ? &row1[i+1] : &row0[i+1];
                    const wvert_t* C = t == 0 ? &row1[i]   : &row1[i+1];
                    counts[0]++;
                    float xmax = fmaxf(fmaxf(A->xn,B->xn),C->xn), xmin = fminf(fminf(A->xn,B->xn),C->xn);
                    if(xmax - xmin > 0.5f) continue;
                    counts[1]++;
                    const wvert_t* V[3] = {A,B,C};
                    int guard_ok = 1;
                    for(int m=0; m<3; m++)
                        if(!(fabsf(V[m]->wx - 0.5f) <= GUARD_PX && fabsf(V[m]->wy - 0.5f) <= GUARD_PX)) guard_ok = 0;
                    if(!guard_ok) continue;
                    counts[2]++;
                    int64_t X[3], Y[3];
                    for(int m=0; m<3; m++) { X[m] = (int64_t)rintf((V[m]->wx - 0.5f)*256.f); Y[m] = (int64_t)rintf((V[m]->wy - 0.5f)*256.f); }
                    int64_t area = (X[1]-X[0])*(Y[2]-Y[0]) - (X[2]-X[0])*(Y[1]-Y[0]);
                    if(area <= 0) continue;
                    counts[3]++;
                    int64_t bx0=X[0],bx1=X[0],by0=Y[0],by1=Y[0];
                    for(int m=1;m<3;m++){ if(X[m]<bx0)bx0=X[m]; if(X[m]>bx1)bx1=X[m]; if(Y[m]<by0)by0=Y[m]; if(Y[m]>by1)by1=Y[m]; }
                    int64_t px0=(bx0+255)>>8, px1=bx1>>8, py0=(by0+255)>>8, py1=by1>>8;
                    if(px0<0) px0=0;
                    if(px1>W-1) px1=W-1;
                    if(py0<0) py0=0;
                    if(py1>H-1) py1=H-1;
                    if(px0>px1||py0>py1) continue;
                    counts[4]++;
                    if((A->zw<0.f&&B->zw<0.f&&C->zw<0.f)||(A->zw>1.f&&B->zw>1.f&&C->zw>1.f)) continue;
                    counts[5]++;
                    int64_t n = (px1-px0+1)*(py1-py0+1);
                    counts[6] += n;
                    int b = 0; while(((int64_t)1<<b) < n) b++;
                    hist[b]++;
                    for(int64_t py=py0;py<=py1;py++) for(int64_t px=px0;px<=px1;px++)
                    {
                        int inside = 1;
                        for(int m=0;m<3&&inside;m++)
                        {
                            int a=m,bb=(m+1)%3; int64_t dx=X[bb]-X[a], dy=Y[bb]-Y[a];
                            int64_t E = dx*(py*256-Y[a]) - dy*(px*256-X[a]);
                            if(E<0) inside=0; else if(E==0 && !(dy<0||(dy==0&&dx>0))) inside=0;
                        }
                        counts[7] += inside;
                    }
                }
        wvert_t* tmp = row0; row0 = row1; row1 = tmp;
    }
    free(row0); free(row1);
}
