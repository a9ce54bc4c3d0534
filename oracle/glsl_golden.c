/* glsl_golden.c - TEST INFRASTRUCTURE: golden-vector generator.
 *
 * A GL host program of our own that runs the REFERENCE'S OWN SHADERS
 * (vertex.glsl, geometry.glsl, fragment.glsl, read unmodified from the
 * reference checkout at run time, never copied) on Mesa llvmpipe, headless,
 * and dumps what comes out.  It replays the GL calls the reference makes
 * around its draw (citations into /root/reference/horizonator-lib.c):
 *   :183-185  glEnable(DEPTH_TEST), glEnable(CULL_FACE), glClearColor(0,0,1,0)
 *   :150      GL_PACK_ALIGNMENT 1
 *   :423-425  VBO of GLshort (i,j,z), attribute 0, not normalised
 *   :492-508  IBO of GLuint, two triangles per cell
 *   :555-559  vertex, fragment, geometry shaders in one program
 *   :577-588  static uniforms; :791-800,833-834,879-882,658 dynamic ones
 *   :631,646  GL_RGB and GL_DEPTH_COMPONENT renderbuffers on an FBO; :657 viewport
 *   :896-897  glClear + glDrawElements(GL_TRIANGLES, GL_UNSIGNED_INT)
 *   :938,962  glReadPixels(GL_BGR, UNSIGNED_BYTE) and (GL_DEPTH_COMPONENT, GL_FLOAT)
 * The reference's host code (horizonator-lib.c) itself cannot be compiled
 * here: it includes epoxy, freeglut and FreeImage headers the image does not
 * have.  Uniform VALUES are therefore inputs of this program (computed by the
 * caller), and only the shader + GL half of the reference is executed.
 *
 * The GL context comes straight from Mesa's software DRI driver
 * (swrast_dri.so) through the loader interface in GL/internal/dri_interface.h;
 * there is no X server, EGL or OSMesa in the image.
 *
 * Job file (little endian):
 *   int32 mode            0 = render, 1 = vertex capture (transform feedback),
 *                         2 = raw triangles (our pass-through shaders; probes
 *                             llvmpipe's raster rules, no reference code)
 *   int32 N, W, H
 *   float32 u[12]         viewer_cell_i, viewer_cell_j, viewer_z, DEG_PER_CELL,
 *                         cos_viewer_lat, az_deg0, az_deg1, aspect,
 *                         znear, zfar, znear_color, zfar_color
 *   mode 0,1: int16 z[N*N]          elevation, j-major (row j = constant latitude)
 *   mode 2:   int32 ntri; float32 v[ntri*3*4]   clip-space x,y,z,red per vertex
 *   modes 3,4,5 (the texture path, reference vertex.glsl:41-61,116-126,
 *   fragment.glsl:17-22, horizonator-lib.c:247-266,361-366,577-588,801-809):
 *             float32 t[8]  viewer_lat (radians), origin_cell_lon_deg, origin_cell_lat_deg,
 *                           texturemap_lon0, lon1, dlat0, dlat1, dlat2
 *             int32 ti[6]   NtilesX, NtilesY, osmtile_lowestX, osmtile_lowestY, texW, texH
 *             uint8 texels[texH*texW*3]   as handed to glTexSubImage2D(GL_BGR): row 0 first
 *     3 = textured render, 4 = vertex capture including `tex`: then int16 z[N*N]
 *     5 = raw triangles through the reference's FRAGMENT shader (our pass-through
 *         vertex stage): int32 ntri; float32 v[ntri*3*6]  clip x,y,z, red, s, t
 *     6 = as 5 with a fragment shader of ours that outputs texture() itself (probes
 *         Mesa's sampler; no reference code)
 * Result file:
 *   mode 0,2,3,5: uint8 bgr[H*W*3] (GL row order, bottom first), float32 depth[H*W],
 *             uint32 z24[H*W]
 *   mode 1:   float32 out[N*N*5]    gl_Position.xyzw, rgb.r
 *   mode 4:   float32 out[N*N*7]    gl_Position.xyzw, rgb.r, tex.xy
 *
 * usage: glsl_golden SHADER_DIR JOB RESULT
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

#define DIE(...) do { fprintf(stderr, "glsl_golden: " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while(0)

/* ---- GL entry points, resolved through libglapi -------------------------- */
#define GLFUNCS(X) \
    X(PFNGLGETERRORPROC, glGetError) X(PFNGLENABLEPROC, glEnable) X(PFNGLDISABLEPROC, glDisable) \
    X(PFNGLCLEARCOLORPROC, glClearColor) \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays) X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray) \
    X(PFNGLGENBUFFERSPROC, glGenBuffers) X(PFNGLBINDBUFFERPROC, glBindBuffer) \
    X(PFNGLBUFFERDATAPROC, glBufferData) X(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray) \
    X(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer) \
    X(PFNGLCREATEPROGRAMPROC, glCreateProgram) X(PFNGLCREATESHADERPROC, glCreateShader) \
    X(PFNGLSHADERSOURCEPROC, glShaderSource) X(PFNGLCOMPILESHADERPROC, glCompileShader) \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv) X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) \
    X(PFNGLATTACHSHADERPROC, glAttachShader) X(PFNGLLINKPROGRAMPROC, glLinkProgram) \
    X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) \
    X(PFNGLUSEPROGRAMPROC, glUseProgram) X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) \
    X(PFNGLUNIFORM1FPROC, glUniform1f) X(PFNGLUNIFORM1IPROC, glUniform1i) \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) \
    X(PFNGLGENRENDERBUFFERSPROC, glGenRenderbuffers) X(PFNGLBINDRENDERBUFFERPROC, glBindRenderbuffer) \
    X(PFNGLRENDERBUFFERSTORAGEPROC, glRenderbufferStorage) \
    X(PFNGLFRAMEBUFFERRENDERBUFFERPROC, glFramebufferRenderbuffer) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) \
    X(PFNGLGETRENDERBUFFERPARAMETERIVPROC, glGetRenderbufferParameteriv) \
    X(PFNGLVIEWPORTPROC, glViewport) X(PFNGLCLEARPROC, glClear) \
    X(PFNGLDRAWELEMENTSPROC, glDrawElements) X(PFNGLDRAWARRAYSPROC, glDrawArrays) \
    X(PFNGLDRAWBUFFERPROC, glDrawBuffer) \
    X(PFNGLREADPIXELSPROC, glReadPixels) X(PFNGLPIXELSTOREIPROC, glPixelStorei) \
    X(PFNGLGETSTRINGPROC, glGetString) X(PFNGLGETINTEGERVPROC, glGetIntegerv) X(PFNGLFINISHPROC, glFinish) \
    X(PFNGLTRANSFORMFEEDBACKVARYINGSPROC, glTransformFeedbackVaryings) \
    X(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) \
    X(PFNGLBEGINTRANSFORMFEEDBACKPROC, glBeginTransformFeedback) \
    X(PFNGLENDTRANSFORMFEEDBACKPROC, glEndTransformFeedback) \
    X(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData) \
    X(PFNGLGENTEXTURESPROC, glGenTextures) X(PFNGLACTIVETEXTUREPROC, glActiveTexture) \
    X(PFNGLBINDTEXTUREPROC, glBindTexture) X(PFNGLTEXPARAMETERIPROC, glTexParameteri) \
    X(PFNGLTEXIMAGE2DPROC, glTexImage2D) X(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D)

#define X(type, name) static type p_##name;
GLFUNCS(X)
#undef X

#define GLCHECK(what) do { GLenum e_ = p_glGetError(); if(e_ != GL_NO_ERROR) DIE("GL error %#x after %s", e_, what); } while(0)

/* ---- headless llvmpipe context ------------------------------------------- */

static void ld_getDrawableInfo(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* priv)
{ (void)d; (void)priv; *x = 0; *y = 0; *w = 64; *h = 64; }
static void ld_putImage(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* priv)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv; }
static void ld_getImage(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* priv)
{ (void)d; (void)x; (void)y; (void)priv; memset(data, 0, (size_t)w*h*4); }
static void ld_putImage2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* priv)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)priv; }
static void ld_getImage2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* priv)
{ (void)d; (void)x; (void)y; (void)w; (void)priv; memset(data, 0, (size_t)stride*h); }

static const __DRIswrastLoaderExtension swrast_loader =
{
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_getDrawableInfo,
    .putImage  = ld_putImage,
    .getImage  = ld_getImage,
    .putImage2 = ld_putImage2,
    .getImage2 = ld_getImage2,
};
static const __DRIextension* loader_extensions[] = { &swrast_loader.base, NULL };

static void make_context(void)
{
    const char* candidates[] = {
        "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so",
        "/usr/lib64/dri/swrast_dri.so",
        "swrast_dri.so", NULL };
    void* drv = NULL;
    for(int k=0; candidates[k] && !drv; k++) drv = dlopen(candidates[k], RTLD_NOW | RTLD_GLOBAL);
    if(!drv) DIE("cannot load Mesa's swrast_dri.so: %s", dlerror());

    const __DRIextension** (*get_ext)(void) = dlsym(drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if(!get_ext) DIE("driver has no %s_swrast", __DRI_DRIVER_GET_EXTENSIONS);
    const __DRIextension** drv_ext = get_ext();

    const __DRIcoreExtension*   core   = NULL;
    const __DRIswrastExtension* swrast = NULL;
    for(int k=0; drv_ext[k]; k++)
    {
        if(!strcmp(drv_ext[k]->name, __DRI_CORE))   core   = (const __DRIcoreExtension*)  drv_ext[k];
        if(!strcmp(drv_ext[k]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)drv_ext[k];
    }
    if(!core || !swrast || swrast->base.version < 4) DIE("driver lacks DRI_Core / DRI_SWRast v4");

    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = swrast->createNewScreen2(0, loader_extensions, drv_ext, &configs, NULL);
    if(!screen) DIE("createNewScreen2 failed");

    const __DRIconfig* config = NULL;
    for(int k=0; configs[k]; k++)
    {
        unsigned depth=0, red=0, dbl=1;
        core->getConfigAttrib(configs[k], __DRI_ATTRIB_DEPTH_SIZE,    &depth);
        core->getConfigAttrib(configs[k], __DRI_ATTRIB_RED_SIZE,      &red);
        core->getConfigAttrib(configs[k], __DRI_ATTRIB_DOUBLE_BUFFER, &dbl);
        if(depth >= 24 && red == 8 && !dbl) { config = configs[k]; break; }
    }
    if(!config) config = configs[0];

    /* the reference asks GLUT for a 4.2 core, forward-compatible context (:130-132) */
    const uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 2 };
    unsigned err = 0;
    __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, config, NULL,
                                                     2, attribs, &err, NULL);
    if(!ctx) DIE("createContextAttribs failed (%u)", err);
    __DRIdrawable* draw = swrast->createNewDrawable(screen, config, NULL);
    if(!draw) DIE("createNewDrawable failed");
    if(!core->bindContext(ctx, draw, draw)) DIE("bindContext failed");

    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if(!glapi) DIE("cannot load libglapi.so.0: %s", dlerror());
    void* (*gpa)(const char*) = dlsym(glapi, "_glapi_get_proc_address");
    if(!gpa) DIE("no _glapi_get_proc_address");
#define X(type, name) p_##name = (type)gpa(#name); if(!p_##name) DIE("GL function %s not found", #name);
    GLFUNCS(X)
#undef X
}

/* ---- helpers ------------------------------------------------------------ */

static char* read_text(const char* dir, const char* name)
{
    char path[2048];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if(!f) DIE("cannot read %s", path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char* s = malloc(n+1);
    if(fread(s, 1, n, f) != (size_t)n) DIE("short read on %s", path);
    s[n] = 0; fclose(f);
    return s;
}

static GLuint compile(GLenum type, const char* src, const char* what)
{
    GLuint sh = p_glCreateShader(type);
    p_glShaderSource(sh, 1, &src, NULL);
    p_glCompileShader(sh);
    GLint ok = 0; p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if(!ok) { char log[4096]; p_glGetShaderInfoLog(sh, sizeof(log), NULL, log); DIE("%s shader: %s", what, log); }
    return sh;
}
static void link_program(GLuint prog)
{
    p_glLinkProgram(prog);
    GLint ok = 0; p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if(!ok) { char log[4096]; p_glGetProgramInfoLog(prog, sizeof(log), NULL, log); DIE("link: %s", log); }
}
static void set1f(GLuint prog, const char* name, float v)
{
    GLint loc = p_glGetUniformLocation(prog, name);
    if(loc >= 0) p_glUniform1f(loc, v);
}
static void set1i(GLuint prog, const char* name, int v)
{
    GLint loc = p_glGetUniformLocation(prog, name);
    if(loc >= 0) p_glUniform1i(loc, v);
}

static void set_reference_uniforms(GLuint prog, const float* u)
{
    set1f(prog, "viewer_cell_i", u[0]);  set1f(prog, "viewer_cell_j", u[1]);
    set1f(prog, "viewer_z", u[2]);       set1f(prog, "DEG_PER_CELL", u[3]);
    set1f(prog, "cos_viewer_lat", u[4]); set1f(prog, "az_deg0", u[5]);
    set1f(prog, "az_deg1", u[6]);        set1f(prog, "aspect", u[7]);
    set1f(prog, "znear", u[8]);          set1f(prog, "zfar", u[9]);
    set1f(prog, "znear_color", u[10]);   set1f(prog, "zfar_color", u[11]);
    /* untextured: reference horizonator-lib.c:585-588 with texture_ctx = {} */
    set1i(prog, "NtilesX", 0); set1i(prog, "NtilesY", 0);
    set1i(prog, "osmtile_lowestX", 0); set1i(prog, "osmtile_lowestY", 0);
}

/* the texture half of the uniforms and the texture object itself, made with the
 * calls the reference makes (horizonator-lib.c:247-266 initOSMtexture, :361-366
 * one glTexSubImage2D per tile - here the whole mosaic at once) */
static void set_texture(GLuint prog, const float* t, const int32_t* ti, const unsigned char* texels)
{
    set1f(prog, "viewer_lat", t[0]);
    set1f(prog, "origin_cell_lon_deg", t[1]); set1f(prog, "origin_cell_lat_deg", t[2]);
    set1f(prog, "texturemap_lon0", t[3]);  set1f(prog, "texturemap_lon1", t[4]);
    set1f(prog, "texturemap_dlat0", t[5]); set1f(prog, "texturemap_dlat1", t[6]); set1f(prog, "texturemap_dlat2", t[7]);
    set1i(prog, "NtilesX", ti[0]); set1i(prog, "NtilesY", ti[1]);
    set1i(prog, "osmtile_lowestX", ti[2]); set1i(prog, "osmtile_lowestY", ti[3]);
    GLuint tex;
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0);
    p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB, ti[4], ti[5], 0, GL_BGR, GL_UNSIGNED_BYTE, NULL);
    p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, ti[4], ti[5], GL_BGR, GL_UNSIGNED_BYTE, texels);
    set1i(prog, "tex", 0);
    GLCHECK("texture");
}

/* vertex stage of the texture probe: position, shade and texture coordinate
 * straight from the attributes, under the names the reference's fragment
 * shader reads */
static const char* texprobe_vs =
    "#version 420\n"
    "layout (location = 0) in vec4 v;\n"
    "layout (location = 1) in vec2 t;\n"
    "out vec3 rgb_fragment;\n"
    "out vec2 tex_fragment;\n"
    "void main(void) { gl_Position = vec4(v.xyz, 1.0); rgb_fragment = vec3(v.w, 0., 0.); tex_fragment = t; }\n";

/* fragment stage that shows the sampler's result unscaled (mode 6) */
static const char* texprobe_fs =
    "#version 420\n"
    "layout(location = 0) out vec4 frag_color;\n"
    "in vec3 rgb_fragment;\n"
    "in vec2 tex_fragment;\n"
    "uniform sampler2D tex;\n"
    "void main(void) { frag_color = texture(tex, tex_fragment.xy) + 0.*vec4(rgb_fragment, 0.); }\n";

static const char* passthrough_vs =
    "#version 420\n"
    "layout (location = 0) in vec4 v;\n"
    "out vec3 rgb_fragment;\n"
    "void main(void) { gl_Position = vec4(v.xyz, 1.0); rgb_fragment = vec3(v.w, 0., 0.); }\n";
static const char* passthrough_fs =
    "#version 420\n"
    "layout(location = 0) out vec4 frag_color;\n"
    "in vec3 rgb_fragment;\n"
    "void main(void) { frag_color = vec4(rgb_fragment, 1.0); }\n";
/* fragment stage for the capture program (rasteriser is discarded anyway) */
static const char* dummy_fs =
    "#version 420\n"
    "layout(location = 0) out vec4 frag_color;\n"
    "void main(void) { frag_color = vec4(0.); }\n";

static void setup_fbo(int W, int H)
{
    GLuint fbo, color, depth;
    p_glGenFramebuffers(1, &fbo);  p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glGenRenderbuffers(1, &color); p_glBindRenderbuffer(GL_RENDERBUFFER, color);
    p_glRenderbufferStorage(GL_RENDERBUFFER, GL_RGB, W, H);
    p_glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, color);
    p_glGenRenderbuffers(1, &depth); p_glBindRenderbuffer(GL_RENDERBUFFER, depth);
    p_glRenderbufferStorage(GL_RENDERBUFFER, GL_DEPTH_COMPONENT, W, H);
    p_glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, depth);
    if(p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) DIE("FBO incomplete");
    GLint bits = 0;
    p_glGetRenderbufferParameteriv(GL_RENDERBUFFER, GL_RENDERBUFFER_DEPTH_SIZE, &bits);
    fprintf(stderr, "glsl_golden: depth renderbuffer has %d bits\n", bits);
    p_glViewport(0, 0, W, H);
    GLCHECK("fbo");
}

static void readback(FILE* out, int W, int H)
{
    const size_t npix = (size_t)W*H;
    unsigned char* bgr = malloc(npix*3);
    float*    df = malloc(npix*sizeof(float));
    uint32_t* du = malloc(npix*sizeof(uint32_t));
    p_glFinish();
    p_glReadPixels(0,0, W,H, GL_BGR, GL_UNSIGNED_BYTE, bgr);
    p_glReadPixels(0,0, W,H, GL_DEPTH_COMPONENT, GL_FLOAT, df);
    p_glReadPixels(0,0, W,H, GL_DEPTH_COMPONENT, GL_UNSIGNED_INT, du);
    GLCHECK("readback");
    for(size_t k=0; k<npix; k++) du[k] >>= 8;      /* 32-bit unorm -> the 24 stored bits */
    fwrite(bgr, 3, npix, out); fwrite(df, 4, npix, out); fwrite(du, 4, npix, out);
    free(bgr); free(df); free(du);
}

int main(int argc, char** argv)
{
    if(argc != 4) DIE("usage: %s SHADER_DIR JOB RESULT", argv[0]);
    FILE* job = fopen(argv[2], "rb");
    if(!job) DIE("cannot open %s", argv[2]);
    int32_t hdr[4]; float u[12];
    if(fread(hdr, 4, 4, job) != 4 || fread(u, 4, 12, job) != 12) DIE("short job header");
    const int mode = hdr[0], N = hdr[1], W = hdr[2], H = hdr[3];
    float t[8] = {0}; int32_t ti[6] = {0}; unsigned char* texels = NULL;
    if(mode >= 3)
    {
        if(fread(t, 4, 8, job) != 8 || fread(ti, 4, 6, job) != 6) DIE("short texture header");
        const size_t nb = (size_t)ti[4]*ti[5]*3;
        texels = malloc(nb);
        if(fread(texels, 1, nb, job) != nb) DIE("short texture body");
    }

    make_context();
    fprintf(stderr, "glsl_golden: %s / %s\n", (const char*)p_glGetString(GL_VERSION), (const char*)p_glGetString(GL_RENDERER));
    GLint subpix = 0; p_glGetIntegerv(GL_SUBPIXEL_BITS, &subpix);
    fprintf(stderr, "glsl_golden: GL_SUBPIXEL_BITS %d\n", subpix);

    FILE* out = fopen(argv[3], "wb");
    if(!out) DIE("cannot open %s", argv[3]);

    /* reference horizonator-lib.c:150,183-185 */
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glEnable(GL_DEPTH_TEST);
    p_glEnable(GL_CULL_FACE);
    p_glClearColor(0, 0, 1, 0);

    GLuint vao; p_glGenVertexArrays(1, &vao); p_glBindVertexArray(vao);
    GLuint vbo; p_glGenBuffers(1, &vbo);      p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
    p_glEnableVertexAttribArray(0);

    if(mode == 0 || mode == 1 || mode == 3 || mode == 4)
    {
        int16_t* z = malloc((size_t)N*N*sizeof(int16_t));
        if(fread(z, 2, (size_t)N*N, job) != (size_t)N*N) DIE("short job body");
        /* reference horizonator-lib.c:423-425,435-480 */
        GLshort* v = malloc((size_t)N*N*3*sizeof(GLshort));
        size_t at = 0;
        for(int j=0; j<N; j++) for(int i=0; i<N; i++) { v[at++] = i; v[at++] = j; v[at++] = z[(size_t)j*N+i]; }
        p_glBufferData(GL_ARRAY_BUFFER, (size_t)N*N*3*sizeof(GLshort), v, GL_STATIC_DRAW);
        p_glVertexAttribPointer(0, 3, GL_SHORT, GL_FALSE, 0, NULL);
        free(v); free(z);

        char* vs_src = read_text(argv[1], "vertex.glsl");
        GLuint prog = p_glCreateProgram();
        p_glAttachShader(prog, compile(GL_VERTEX_SHADER, vs_src, "vertex"));
        if(mode == 0 || mode == 3)
        {
            p_glAttachShader(prog, compile(GL_FRAGMENT_SHADER, read_text(argv[1], "fragment.glsl"), "fragment"));
            p_glAttachShader(prog, compile(GL_GEOMETRY_SHADER, read_text(argv[1], "geometry.glsl"), "geometry"));
            link_program(prog);
            p_glUseProgram(prog);
            set_reference_uniforms(prog, u);
            if(mode == 3) set_texture(prog, t, ti, texels);
            GLCHECK("uniforms");

            /* reference horizonator-lib.c:492-508 */
            const size_t ntri = (size_t)(N-1)*(N-1)*2;
            GLuint* idx = malloc(ntri*3*sizeof(GLuint));
            at = 0;
            for(int j=0; j<N-1; j++) for(int i=0; i<N-1; i++)
            {
                idx[at++] = (j+0)*N + (i+0); idx[at++] = (j+1)*N + (i+1); idx[at++] = (j+1)*N + (i+0);
                idx[at++] = (j+0)*N + (i+0); idx[at++] = (j+0)*N + (i+1); idx[at++] = (j+1)*N + (i+1);
            }
            GLuint ibo; p_glGenBuffers(1, &ibo); p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo);
            p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, ntri*3*sizeof(GLuint), idx, GL_STATIC_DRAW);
            free(idx);

            setup_fbo(W, H);
            /* HZ_GL_TIMING=n: the reference's per-frame GL calls (glClear + glDrawElements, reference
             * horizonator-lib.c:896-897) and its two readbacks (:938,:962) timed n times on stderr -
             * tools/llvmpipe_timing.py turns the lines into the baseline figures of BASELINE.md */
            const char* tim = getenv("HZ_GL_TIMING");
            const int reps = tim && atoi(tim) > 0 ? atoi(tim) : 0;
            for(int r=0; r<reps; r++)
            {
                struct timespec t0, t1, t2;
                p_glFinish();
                clock_gettime(CLOCK_MONOTONIC, &t0);
                p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
                p_glDrawElements(GL_TRIANGLES, (GLsizei)(ntri*3), GL_UNSIGNED_INT, NULL);
                p_glFinish();
                clock_gettime(CLOCK_MONOTONIC, &t1);
                {
                    const size_t npix = (size_t)W*H;
                    unsigned char* bgr = malloc(npix*3); float* df = malloc(npix*sizeof(float));
                    p_glReadPixels(0,0, W,H, GL_BGR, GL_UNSIGNED_BYTE, bgr);
                    p_glReadPixels(0,0, W,H, GL_DEPTH_COMPONENT, GL_FLOAT, df);
                    free(bgr); free(df);
                }
                clock_gettime(CLOCK_MONOTONIC, &t2);
                fprintf(stderr, "timing rep %d draw_s %.6f readback_s %.6f\n", r,
                        (double)(t1.tv_sec - t0.tv_sec) + 1e-9*(double)(t1.tv_nsec - t0.tv_nsec),
                        (double)(t2.tv_sec - t1.tv_sec) + 1e-9*(double)(t2.tv_nsec - t1.tv_nsec));
            }
            p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
            p_glDrawElements(GL_TRIANGLES, (GLsizei)(ntri*3), GL_UNSIGNED_INT, NULL);
            GLCHECK("draw");
            p_glDrawBuffer(GL_COLOR_ATTACHMENT0);
            readback(out, W, H);
        }
        else
        {
            /* vertex stage only: capture gl_Position and rgb of every vertex */
            p_glAttachShader(prog, compile(GL_FRAGMENT_SHADER, dummy_fs, "dummy fragment"));
            const char* varyings[] = { "gl_Position", "rgb", "tex" };
            const int nvar = mode == 4 ? 3 : 2, stride = mode == 4 ? 9 : 7;
            p_glTransformFeedbackVaryings(prog, nvar, varyings, GL_INTERLEAVED_ATTRIBS);
            link_program(prog);
            p_glUseProgram(prog);
            set_reference_uniforms(prog, u);
            if(mode == 4) set_texture(prog, t, ti, texels);
            const size_t nv = (size_t)N*N;
            GLuint tfb; p_glGenBuffers(1, &tfb);
            p_glBindBuffer(GL_TRANSFORM_FEEDBACK_BUFFER, tfb);
            p_glBufferData(GL_TRANSFORM_FEEDBACK_BUFFER, nv*stride*sizeof(float), NULL, GL_STATIC_READ);
            p_glBindBufferBase(GL_TRANSFORM_FEEDBACK_BUFFER, 0, tfb);
            p_glEnable(GL_RASTERIZER_DISCARD);
            p_glBeginTransformFeedback(GL_POINTS);
            p_glDrawArrays(GL_POINTS, 0, (GLsizei)nv);
            p_glEndTransformFeedback();
            p_glFinish();
            GLCHECK("transform feedback");
            float* cap = malloc(nv*stride*sizeof(float));
            p_glGetBufferSubData(GL_TRANSFORM_FEEDBACK_BUFFER, 0, nv*stride*sizeof(float), cap);
            for(size_t k=0; k<nv; k++)
            {
                fwrite(&cap[k*stride], sizeof(float), 5, out);                       /* xyzw + r */
                if(mode == 4) fwrite(&cap[k*stride + 7], sizeof(float), 2, out);     /* tex.xy   */
            }
            free(cap);
        }
    }
    else if(mode == 2)
    {
        int32_t ntri;
        if(fread(&ntri, 4, 1, job) != 1) DIE("short job");
        float* v = malloc((size_t)ntri*12*sizeof(float));
        if(fread(v, 4, (size_t)ntri*12, job) != (size_t)ntri*12) DIE("short job body");
        p_glBufferData(GL_ARRAY_BUFFER, (size_t)ntri*12*sizeof(float), v, GL_STATIC_DRAW);
        p_glVertexAttribPointer(0, 4, GL_FLOAT, GL_FALSE, 0, NULL);
        free(v);
        GLuint prog = p_glCreateProgram();
        p_glAttachShader(prog, compile(GL_VERTEX_SHADER, passthrough_vs, "passthrough vertex"));
        p_glAttachShader(prog, compile(GL_FRAGMENT_SHADER, passthrough_fs, "passthrough fragment"));
        link_program(prog);
        p_glUseProgram(prog);
        setup_fbo(W, H);
        p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
        p_glDrawArrays(GL_TRIANGLES, 0, ntri*3);
        GLCHECK("draw");
        readback(out, W, H);
    }
    else if(mode == 5 || mode == 6)
    {
        int32_t ntri;
        if(fread(&ntri, 4, 1, job) != 1) DIE("short job");
        float* v = malloc((size_t)ntri*18*sizeof(float));
        if(fread(v, 4, (size_t)ntri*18, job) != (size_t)ntri*18) DIE("short job body");
        p_glBufferData(GL_ARRAY_BUFFER, (size_t)ntri*18*sizeof(float), v, GL_STATIC_DRAW);
        p_glVertexAttribPointer(0, 4, GL_FLOAT, GL_FALSE, 6*sizeof(float), NULL);
        p_glEnableVertexAttribArray(1);
        p_glVertexAttribPointer(1, 2, GL_FLOAT, GL_FALSE, 6*sizeof(float), (const void*)(4*sizeof(float)));
        free(v);
        GLuint prog = p_glCreateProgram();
        p_glAttachShader(prog, compile(GL_VERTEX_SHADER, texprobe_vs, "texture probe vertex"));
        p_glAttachShader(prog, mode == 5 ? compile(GL_FRAGMENT_SHADER, read_text(argv[1], "fragment.glsl"), "fragment")
                                         : compile(GL_FRAGMENT_SHADER, texprobe_fs, "texture probe fragment"));
        link_program(prog);
        p_glUseProgram(prog);
        set_texture(prog, t, ti, texels);
        setup_fbo(W, H);
        p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
        p_glDrawArrays(GL_TRIANGLES, 0, ntri*3);
        GLCHECK("draw");
        readback(out, W, H);
    }
    else DIE("unknown mode %d", mode);

    fclose(out); fclose(job);
    fflush(NULL);
    _exit(0);       /* skip driver teardown */
}
