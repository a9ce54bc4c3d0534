/* stand-in for <FreeImage.h>: the names reference standalone.c:471-488 uses (PNG output of the render) */
#pragma once
typedef int BOOL;
typedef unsigned char BYTE;
typedef struct FIBITMAP FIBITMAP;
typedef enum { FIT_UNKNOWN = 0, FIT_BITMAP = 1 } FREE_IMAGE_TYPE;
typedef enum { FIF_UNKNOWN = -1, FIF_PNG = 13 } FREE_IMAGE_FORMAT;
void      FreeImage_Initialise(BOOL load_local_plugins_only);
void      FreeImage_DeInitialise(void);
FIBITMAP* FreeImage_ConvertFromRawBitsEx(BOOL copy_source, BYTE* bits, FREE_IMAGE_TYPE type, int width, int height, int pitch,
                                         unsigned bpp, unsigned red_mask, unsigned green_mask, unsigned blue_mask, BOOL topdown);
BOOL      FreeImage_Save(FREE_IMAGE_FORMAT fif, FIBITMAP* dib, const char* filename, int flags);
void      FreeImage_Unload(FIBITMAP* dib);
