/* stand-in for <libavutil/pixfmt.h>: the two formats reference annotator.c:127-129 names */
#pragma once
enum AVPixelFormat { AV_PIX_FMT_NONE = -1, AV_PIX_FMT_BGR24 = 3, AV_PIX_FMT_RGB32 = 28 };
