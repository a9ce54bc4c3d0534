/* hz_scatter.h - results into the caller's host memory without the sky: what the device code that writes the
 * blobs (k_pack_host, hz_k_resolve.h) and the host code that reads them (hz_scatter.c) share.  Format: hz_scatter.c. */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HZ_BLOB_ROWS   4
#define HZ_BLOB_COLS   2048
#define HZ_BLOB_HDR    8                /* words in front of the masks */
#define HZ_BLOB_RANGES 1u
#define HZ_BLOB_INDEX  2u
#define HZ_BLOB_Z24    4u
#define HZ_BLOB_RED    8u
#define HZ_BLOB_VOID   0xFFFFFFFEu      /* word [0] of a stretch of the stream that holds nothing; word [1]: its length in words */

#define HZ_SKY_BGR     0
#define HZ_SKY_RANGES  1
#define HZ_SKY_INDEX   2
#define HZ_SKY_Z24     3

void   hz_sky_fill(unsigned char* buf, size_t lo, size_t hi, int kind);
size_t hz_blob_walk(const uint32_t* chunk, size_t nwords, size_t first, size_t* offsets, size_t max, size_t* beyond);
int    hz_blob_scatter(const uint32_t* blob, int SW, int H, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24);

#ifdef __cplusplus
}
#endif
