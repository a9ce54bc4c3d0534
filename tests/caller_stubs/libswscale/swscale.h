/* stand-in for <libswscale/swscale.h>: what reference annotator.c:125-138 uses (BGR24 -> RGB32 for cairo) */
#pragma once
#include <stdint.h>
#include "libavutil/pixfmt.h"
#define SWS_POINT 0x10
struct SwsContext;
struct SwsContext* sws_getContext(int srcW, int srcH, enum AVPixelFormat srcFormat, int dstW, int dstH, enum AVPixelFormat dstFormat,
                                  int flags, void* srcFilter, void* dstFilter, const double* param);
int  sws_scale(struct SwsContext* c, const uint8_t* const srcSlice[], const int srcStride[], int srcSliceY, int srcSliceH,
               uint8_t* const dst[], const int dstStride[]);
void sws_freeContext(struct SwsContext* c);
