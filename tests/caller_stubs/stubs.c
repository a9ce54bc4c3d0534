/* no-op definitions of the callers' own dependencies (see README.md): drawing, image files, windows.
 * A surface / context "exists" (non-NULL) so that the callers' error paths are not what gets exercised. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "FreeImage.h"
#include "epoxy/gl.h"
#include "GL/freeglut.h"
#include "cairo.h"
#include "libswscale/swscale.h"

static int something;

/* which build of the library this program was linked against (HZ_BUILD_ID: horizonator_amd/csrc/build/hz_build_id.h at
 * link time, tests/caller_stubs/Makefile), said on request: tests/test_gpu_standalone.py compares it with what the
 * library it finds at run time says of itself - a binary left over from an older tree does not pass for a current one */
#include <stdlib.h>
#ifndef HZ_BUILD_ID
#define HZ_BUILD_ID "unknown"
#endif
__attribute__((constructor)) static void say_build_id(void)
{
    if(getenv("HZ_SHOW_BUILD_ID")) fprintf(stderr, "standalone_ref: linked against libhorizonator build %s\n", HZ_BUILD_ID);
}
/* FreeImage: the one stand-in that does something - FreeImage_Save() writes what the caller handed to
 * FreeImage_ConvertFromRawBitsEx() to the named file as it is ("HZRAW width height bytes_per_pixel\n", then
 * the rows top first, pitch removed), so that a test can look at the image the reference's CLI produced */
static struct { BYTE* bits; int w, h, pitch, bytes_pp, topdown; } kept;
void      FreeImage_Initialise(BOOL b) { (void)b; }
void      FreeImage_DeInitialise(void) {}
FIBITMAP* FreeImage_ConvertFromRawBitsEx(BOOL c, BYTE* bits, FREE_IMAGE_TYPE t, int w, int h, int pitch, unsigned bpp,
                                         unsigned r, unsigned g, unsigned b, BOOL topdown)
{
    (void)c; (void)t; (void)r; (void)g; (void)b;
    kept.bits = bits; kept.w = w; kept.h = h; kept.pitch = pitch; kept.bytes_pp = (int)bpp/8; kept.topdown = topdown;
    return (FIBITMAP*)&something;
}
BOOL      FreeImage_Save(FREE_IMAGE_FORMAT f, FIBITMAP* d, const char* name, int flags)
{
    (void)f; (void)d; (void)flags;
    FILE* fp = fopen(name, "wb");
    if(!fp) return 0;
    fprintf(fp, "HZRAW %d %d %d\n", kept.w, kept.h, kept.bytes_pp);
    for(int y=0; y<kept.h; y++)
        fwrite(kept.bits + (size_t)(kept.topdown ? y : kept.h-1-y)*kept.pitch, 1, (size_t)kept.w*kept.bytes_pp, fp);
    return fclose(fp) == 0;
}
void      FreeImage_Unload(FIBITMAP* d) { (void)d; }

void glPolygonMode(GLenum face, GLenum mode) { (void)face; (void)mode; }
void glFrontFace(GLenum mode) { (void)mode; }
void glutSwapBuffers(void) {}
void glutExit(void) {}
void glutPostRedisplay(void) {}
void glutDisplayFunc(void (*cb)(void)) { (void)cb; }
void glutKeyboardFunc(void (*cb)(unsigned char, int, int)) { (void)cb; }
void glutReshapeFunc(void (*cb)(int, int)) { (void)cb; }
void glutMainLoop(void) {}

cairo_surface_t* cairo_pdf_surface_create(const char* f, double w, double h) { (void)f; (void)w; (void)h; return (cairo_surface_t*)&something; }
cairo_surface_t* cairo_svg_surface_create(const char* f, double w, double h) { (void)f; (void)w; (void)h; return (cairo_surface_t*)&something; }
cairo_surface_t* cairo_image_surface_create_for_data(unsigned char* d, cairo_format_t f, int w, int h, int s)
{ (void)d; (void)f; (void)w; (void)h; (void)s; return (cairo_surface_t*)&something; }
void     cairo_surface_destroy(cairo_surface_t* s) { (void)s; }
void     cairo_surface_show_page(cairo_surface_t* s) { (void)s; }
cairo_t* cairo_create(cairo_surface_t* t) { (void)t; return (cairo_t*)&something; }
void     cairo_destroy(cairo_t* cr) { (void)cr; }
void     cairo_scale(cairo_t* cr, double sx, double sy) { (void)cr; (void)sx; (void)sy; }
void     cairo_set_source_rgb(cairo_t* cr, double r, double g, double b) { (void)cr; (void)r; (void)g; (void)b; }
void     cairo_set_source_surface(cairo_t* cr, cairo_surface_t* s, double x, double y) { (void)cr; (void)s; (void)x; (void)y; }
void     cairo_set_font_size(cairo_t* cr, double size) { (void)cr; (void)size; }
void     cairo_paint(cairo_t* cr) { (void)cr; }
void     cairo_fill(cairo_t* cr) { (void)cr; }
void     cairo_stroke(cairo_t* cr) { (void)cr; }
void     cairo_rectangle(cairo_t* cr, double x, double y, double w, double h) { (void)cr; (void)x; (void)y; (void)w; (void)h; }
void     cairo_move_to(cairo_t* cr, double x, double y) { (void)cr; (void)x; (void)y; }
void     cairo_line_to(cairo_t* cr, double x, double y) { (void)cr; (void)x; (void)y; }
void     cairo_rel_line_to(cairo_t* cr, double dx, double dy) { (void)cr; (void)dx; (void)dy; }
void     cairo_show_text(cairo_t* cr, const char* s) { (void)cr; (void)s; }
void     cairo_text_extents(cairo_t* cr, const char* s, cairo_text_extents_t* e) { (void)cr; memset(e, 0, sizeof(*e)); e->width = 6.0*(double)strlen(s); }
void     cairo_tag_begin(cairo_t* cr, const char* t, const char* a) { (void)cr; (void)t; (void)a; }
void     cairo_tag_end(cairo_t* cr, const char* t) { (void)cr; (void)t; }

struct SwsContext* sws_getContext(int sw, int sh, enum AVPixelFormat sf, int dw, int dh, enum AVPixelFormat df, int flags, void* a, void* b, const double* p)
{ (void)sw; (void)sh; (void)sf; (void)dw; (void)dh; (void)df; (void)flags; (void)a; (void)b; (void)p; return (struct SwsContext*)&something; }
int  sws_scale(struct SwsContext* c, const uint8_t* const src[], const int ss[], int y, int h, uint8_t* const dst[], const int ds[])
{ (void)c; (void)src; (void)ss; (void)y; (void)dst; (void)ds; return h; }
void sws_freeContext(struct SwsContext* c) { (void)c; }
