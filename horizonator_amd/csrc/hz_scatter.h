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
#define HZ_BLOB_PACKED 1u               /* one word per terrain pixel: z24<<8 | red8 - range, depth and shade follow from it */
#define HZ_BLOB_INDEX  2u               /* one word per terrain pixel: the id of the triangle that owns it */
#define HZ_BLOB_RED    8u               /* one BYTE per terrain pixel: the shade (a caller that wants the image only) */
#define HZ_BLOB_VOID   0xFFFFFFFEu      /* word [0] of a stretch of the stream that holds nothing; word [1]: its length in words */

#define HZ_SKY_BGR     0
#define HZ_SKY_RANGES  1
#define HZ_SKY_INDEX   2
#define HZ_SKY_Z24     3

/* the caller's buffers ([H][W] pixels each, top row first; any may be NULL) and what the readback conversion
 * (reference horizonator-lib.c:1006-1047) needs to make ranges of depths */
typedef struct
{
    int            W, H;
    unsigned char* bgr;
    float*         ranges;
    int32_t*       index;
    uint32_t*      z24;
    const float*   tanel;               /* tan(elevation) of every GL row (row 0 = bottom, i.e. output row H-1); needed for ranges */
    float          znear, zfar;
} hz_scatter_dst_t;

void   hz_sky_fill(unsigned char* buf, size_t lo, size_t hi, int kind);
void   hz_scatter_set_streaming(int on);     /* diagnostics (tools/scatter_bench.c): 0 = ordinary instead of streaming stores */
size_t hz_blob_walk(const uint32_t* chunk, size_t nwords, size_t first, size_t* offsets, size_t max, size_t* beyond);
int    hz_blob_scatter(const uint32_t* blob, const hz_scatter_dst_t* dst);
int    hz_blob_scatter_mode(const uint32_t* blob, const hz_scatter_dst_t* dst, int full);
/* out[k] = range of packed[k] = z24<<8 | red8 in a row whose tan(elevation) is tan_row: reference
 * horizonator-lib.c:1013-1025, bit for bit what the device's conversions compute (hz_k_resolve.h: hz_range_from_z24) */
void   hz_ranges_from_packed(float* out, const uint32_t* packed, size_t n, float tan_row, float znear, float zfar);

#ifdef __cplusplus
}
#endif
