/* stand-in for cairo's headers: the names reference annotator.c uses (PDF / SVG drawing) */
#pragma once
typedef struct _cairo cairo_t;
typedef struct _cairo_surface cairo_surface_t;
typedef struct { double x_bearing, y_bearing, width, height, x_advance, y_advance; } cairo_text_extents_t;
typedef enum { CAIRO_FORMAT_ARGB32 = 0, CAIRO_FORMAT_RGB24 = 1 } cairo_format_t;
#define CAIRO_TAG_LINK "Link"
cairo_surface_t* cairo_pdf_surface_create(const char* filename, double width_in_points, double height_in_points);
cairo_surface_t* cairo_svg_surface_create(const char* filename, double width_in_points, double height_in_points);
cairo_surface_t* cairo_image_surface_create_for_data(unsigned char* data, cairo_format_t format, int width, int height, int stride);
void     cairo_surface_destroy(cairo_surface_t* surface);
void     cairo_surface_show_page(cairo_surface_t* surface);
cairo_t* cairo_create(cairo_surface_t* target);
void     cairo_destroy(cairo_t* cr);
void     cairo_scale(cairo_t* cr, double sx, double sy);
void     cairo_set_source_rgb(cairo_t* cr, double red, double green, double blue);
void     cairo_set_source_surface(cairo_t* cr, cairo_surface_t* surface, double x, double y);
void     cairo_set_font_size(cairo_t* cr, double size);
void     cairo_paint(cairo_t* cr);
void     cairo_fill(cairo_t* cr);
void     cairo_stroke(cairo_t* cr);
void     cairo_rectangle(cairo_t* cr, double x, double y, double width, double height);
void     cairo_move_to(cairo_t* cr, double x, double y);
void     cairo_line_to(cairo_t* cr, double x, double y);
void     cairo_rel_line_to(cairo_t* cr, double dx, double dy);
void     cairo_show_text(cairo_t* cr, const char* utf8);
void     cairo_text_extents(cairo_t* cr, const char* utf8, cairo_text_extents_t* extents);
void     cairo_tag_begin(cairo_t* cr, const char* tag_name, const char* attributes);
void     cairo_tag_end(cairo_t* cr, const char* tag_name);
