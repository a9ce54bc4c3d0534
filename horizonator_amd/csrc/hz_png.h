/* hz_png.h - just enough PNG to read map tiles (texture path, row N4).
 *
 * The reference reads its 256x256 OpenStreetMap tiles through FreeImage
 * (reference horizonator-lib.c:323-369), which this image does not have; zlib it
 * has.  Supported: non-interlaced, 8 bits per sample grey / RGB / RGBA (alpha
 * dropped) and palette images of 1, 2, 4 or 8 bits - what tile servers emit. */
#pragma once

#include <stddef.h>

/* Decodes `path` into rgb[height][width][3] (R,G,B, top row first), which the
 * caller has allocated; the image must be exactly width x height.  Returns 0,
 * or -1 with a message in err. */
int hz_png_load_rgb(const char* path, int width, int height, unsigned char* rgb, char* err, size_t errlen);
