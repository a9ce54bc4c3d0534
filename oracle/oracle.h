/* oracle.h - CPU restatement of the reference's DEM -> panorama path.
 *
 * TEST INFRASTRUCTURE.  This is the checker the HIP path is compared against;
 * it is not part of the product and nothing under horizonator_amd/ or include/
 * may include, link or call it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * What it restates (citations into /root/reference):
 *   dem.c:78-309                    window arithmetic, tile lookup, sample decode
 *   horizonator-lib.c:765-799       viewer cell / viewer height / cos(lat)
 *   vertex.glsl:30-38,111-162       per-vertex transform, float32
 *   horizonator-lib.c:487-512       two triangles per cell, their order
 *   geometry.glsl:21-27             wide / seam-crossing triangle discard
 *   horizonator-lib.c:183-185,896   depth test LESS, back-face cull, clear
 *   fragment.glsl:15-16             colour = (red, 0, 0)
 *   horizonator-lib.c:936-1048      BGR readback, row flip, depth -> range
 *   horizonator-lib.c:1053-1213     x_from_az / project / unproject (host math)
 *   annotator.c:228-264, 280-348    link cells and label visibility from the range image
 * The rasterisation between geometry.glsl and the depth buffer is not code of
 * the reference: it is OpenGL's, executed for the reference by Mesa/llvmpipe.
 * It is restated with GL's rules and llvmpipe's conventions (8 sub-pixel bits,
 * left/bottom fill rule, 24-bit depth, llvmpipe's plane set-up arithmetic).
 *
 * Parity pinning: oracle_dem is checked bit-for-bit against the reference's
 * own dem.c compiled in place (oracle/_ref/libdem_ref.so).  The render is
 * checked against golden vectors produced by the reference's three GLSL
 * shaders, unmodified, executed by Mesa llvmpipe (oracle/glsl_golden.c,
 * tests/golden/): the vertex stage bit-exact, and every byte of the BGR image
 * and of the 24-bit depth of whole draws identical, clipped triangles included.
 * horizonator-lib.c and annotator.c themselves cannot be built here (they need
 * epoxy, freeglut, FreeImage, cairo and swscale headers the image lacks), so
 * the host-side uniform derivation, the readback conversion and the annotator
 * passes are pinned by restatement only.
 *
 * The texture path ("next" row N4: vertex.glsl:41-61,116-126, fragment.glsl:17-22,
 * horizonator-lib.c:225-246,372-389,577-588,707-759,801-809) is restated the same
 * way: texture coordinates follow the compiler's operation order (bit-exact on
 * captured vertices), and the GL_LINEAR / GL_REPEAT sampling of the RGB8 texture
 * is llvmpipe's, pinned on 7 M probe samples: coordinates in 24.8 fixed point
 * (round to nearest even of s*size*256, minus 128), two 8-bit lerps in x then one
 * in y, each (w*(b-a)+128)>>8, result byte*(1/255); then 0.7*tex + 0.3*shade in
 * float32 and round(x*255) into the RGB8 target.
 */
#pragma once

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DEM ---------------------------------------------------------------- */

typedef struct
{
    int cells_per_deg;
    int radius_cells;
    int origin_tile[2];         /* lon, lat */
    int origin_cell[2];
    int ntiles[2];
    unsigned char** tiles;      /* [ntiles[0]*ntiles[1]], NULL = sea; whole files read into memory */
} orc_dem_t;

/* 0 on success.  No limit on the number of tiles (the reference stops at 4x4). */
int  orc_dem_open (orc_dem_t* d, float viewer_lat, float viewer_lon,
                   int radius_cells, float radius_m, const char* dir, int srtm1);
void orc_dem_close(orc_dem_t* d);
int  orc_dem_sample(const orc_dem_t* d, int i, int j);
/* mosaic[j*N+i] = sample(i,j), N = 2*radius_cells */
void orc_dem_mosaic(const orc_dem_t* d, int16_t* mosaic);

/* ---- view ("uniforms") -------------------------------------------------- */

typedef struct
{
    float viewer_cell_i, viewer_cell_j;
    float viewer_z;
    float cos_viewer_lat;
    float deg_per_cell;
    float az_deg0, az_deg1;
    float aspect;
    float znear, zfar;
    float znear_color, zfar_color;
} orc_view_t;

/* reference horizonator-lib.c:765-799.  viewer_z < 0 -> max(4 samples)+1 */
void orc_view_move(orc_view_t* v, const orc_dem_t* d, float viewer_lat, float viewer_lon, float viewer_z);

/* ---- vertex stage ------------------------------------------------------- */

/* gl_Position.xyz and rgb.r of one vertex (reference vertex.glsl:111-160) */
void orc_vertex(const orc_view_t* v, int i, int j, int z, float out_xyzr[4]);

/* ---- full render -------------------------------------------------------- */

/* Renders image columns [col0,col1) of a W x H panorama.  Outputs are
 * [H][col1-col0], top row first, each may be NULL:
 *   bgr 3 bytes/pixel; ranges float32 (<0 = sky); index int32 primitive id
 *   (-1 = sky); z24 uint32 (0xFFFFFF = sky).
 * nthreads <= 0: all cores.  Returns 0 on success. */
int orc_render(const int16_t* mosaic, int N, const orc_view_t* v,
               int W, int H, int col0, int col1,
               uint8_t* bgr, float* ranges, int32_t* index, uint32_t* z24,
               int nthreads);

/* ---- texture path ("next" row N4) ----------------------------------------------- */

typedef struct
{
    /* uniforms of the texture half of vertex.glsl */
    float viewer_lat_rad;
    float origin_cell_lon_deg, origin_cell_lat_deg;
    float lon0, lon1, dlat0, dlat1, dlat2;
    int   ntiles_x, ntiles_y, lowest_x, lowest_y;
    /* the texture as glTexSubImage2D(GL_BGR) receives it: [tex_h][tex_w][3],
     * B,G,R, row 0 = texture coordinate t = 0 = southern edge of the mosaic */
    int   tex_w, tex_h;
    const uint8_t* texels;
} orc_tex_t;

#define ORC_OSM_ZOOM      12
#define ORC_OSM_TILE_PX   256

/* reference horizonator-lib.c:225-246: slippy-map tile that holds (E,N), degrees */
void orc_osm_tile_id(int* x, int* y, float E, float N);
/* everything but the texels: tile range from the DEM window around the INIT
 * viewpoint (reference :372-389, fixed for the life of a context), origin of
 * the grid (:577-582), and the coefficients of the current viewpoint (:707-759,
 * :801-809; they change with every horizonator_move) */
void orc_tex_setup(orc_tex_t* t, const orc_dem_t* d,
                   float init_lat, float init_lon, float viewer_lat);
/* texture coordinate of one vertex (reference vertex.glsl:116-126) */
void orc_vertex_tex(const orc_tex_t* t, float deg_per_cell, int i, int j, float out_st[2]);
/* the sampler alone: B,G,R bytes of texture(tex, (s,t)) */
void orc_tex_sample(const orc_tex_t* t, float s, float tt, uint8_t out_bgr[3]);
/* orc_render() with render_texture = true; tex == NULL is orc_render() */
int orc_render_tex(const int16_t* mosaic, int N, const orc_view_t* v, const orc_tex_t* tex,
                   int W, int H, int col0, int col1,
                   uint8_t* bgr, float* ranges, int32_t* index, uint32_t* z24,
                   int nthreads);

/* raw clip-space triangles, float[ntri][3][6] = x,y,z, shade, s,t (w = 1), through clipper,
 * rasteriser and (tex != NULL) the textured fragment stage; outputs in GL row order */
int orc_draw_triangles(const float* tris, int ntri, int W, int H, const orc_tex_t* tex,
                       uint8_t* bgr, uint32_t* z24);

/* ---- annotator passes over the range image ("next" row N2) ----------------- */

/* reference horizonator-lib.c:1006-1047 alone: z24_gl[H][W] raw 24-bit depth in GL row order -> ranges[H][W], top row first */
void orc_ranges_from_z24(float* ranges, const uint32_t* z24_gl, int W, int H,
                         float az_deg0, float az_deg1, float znear, float zfar);

/* reference horizonator-lib.c:1097-1155 / :1157-1213; return 1 on success */
int orc_project(double* x, double* y, double* range,
                double lat_viewer, double cos_lat_viewer, double lon_viewer, double ele_viewer,
                double lat, double lon, double ele,
                double az_rad0, double az_rad1, int width, int height);
int orc_unproject(float* lat, float* lon, int x, int y, double range_enh, double range_en,
                  double lat_viewer, double cos_lat_viewer, double lon_viewer,
                  double az_deg0, double az_deg1, int width, int height);

/* reference annotator.c:228-264; out arrays [ny][nx], NaN where no terrain; returns nx*ny */
int orc_link_cells(const float* range_image, int width, int height, int cut_off_bottom_px,
                   int cell_width, int cell_height,
                   double lat, double lon, double az_deg0, double az_deg1,
                   float* out_lat, float* out_lon);

/* reference annotator.c:280-348; pois = (lat, lon, ele_m) float triples */
void orc_poi_visibility(const float* range_image, int width, int height, int cut_off_bottom_px,
                        const float* pois, int Npois,
                        double lat, double lon, double az_deg0, double az_deg1, double ele_m,
                        uint8_t* visible, float* label_x, float* label_y);

/* tan(elevation) per GL row as reference horizonator-lib.c:1006-1012,1026-1047 uses it */
void orc_tanel(float* tanel, int W, int H, float az_deg0, float az_deg1);

#ifdef __cplusplus
}
#endif
