/* hz_scatter.c - the host half of "results into the caller's memory without the sky".
 *
 * The reference hands its results to the caller in host memory (reference horizonator-lib.c:936-1048: two
 * glReadPixels into the caller's buffers, the clear colour and the cleared depth where nothing was drawn,
 * :185, :1016).  In a panorama most pixels are exactly that - sky: BGR (255,0,0), range -1 - and a sky pixel
 * carries no information.  The device therefore sends only the terrain pixels, as "blobs" (k_pack_host,
 * hz_k_resolve.h; layout below), and the host threads of hz_kernels.hip's pool
 *   - fill the caller's buffers with the sky's constants while the draw is still running (hz_sky_fill), and
 *   - put each blob's terrain pixels in their places as its bytes arrive (hz_blob_scatter):
 * the same bytes in the caller's buffers as the dense copy, a third to a fifth of the bytes over PCIe.
 *
 * A blob = the terrain pixels of up to 4 image rows x up to 2048 columns, uint32 words:
 *   [0]     first row (top row = 0) | flags << 16   (HZ_BLOB_RANGES | _INDEX | _Z24 | _RED: the arrays it carries)
 *   [1]     first column     [2..5] terrain pixels T0..T3 of its four rows     [6] size of the blob in words
 *   [7]     columns n (<= 2048)
 *   then    4 x ceil(n/32) mask words (row after row; bit c%32 of word c/32: column c shows terrain)
 *   then    per carried array T0+T1+T2+T3 words, row after row, left to right: float32 ranges, int32 index,
 *           uint32 z24 - and last the shades, one BYTE per terrain pixel (padded to a word)
 * Blobs of tiles without terrain are not sent.  The stream travels in chunks of HZ_STAGE_BYTES and no blob
 * straddles a chunk boundary: where a blob would have (the writers take their places with an atomic add), the
 * stream holds a void instead - word [0] = HZ_BLOB_VOID, word [1] = its length in words - which may reach into
 * the next chunk.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <emmintrin.h>

#include "hz_scatter.h"

/* bytes [lo,hi) of a buffer of `kind` whose byte 0 is the start of a pixel.  Streaming stores: 448 MB per
 * 16000x4000 panorama that nobody reads before the blobs land on a third of them. */
void hz_sky_fill(unsigned char* buf, size_t lo, size_t hi, int kind)
{
    if(hi <= lo) return;
    unsigned char pat[48];
    if(kind == HZ_SKY_BGR)
        for(int k=0; k<48; k++) pat[k] = (k % 3) == 0 ? 255 : 0;       /* reference horizonator-lib.c:185: clear colour (0,0,1) read back as B,G,R */
    else
    {
        uint32_t w = kind == HZ_SKY_RANGES ? 0xBF800000u /* -1.0f, reference horizonator-lib.c:1016 */
                   : kind == HZ_SKY_INDEX  ? 0xFFFFFFFFu /* -1 */ : 0x00FFFFFFu /* the cleared 24-bit depth */;
        for(int k=0; k<12; k++) memcpy(pat + 4*k, &w, 4);
    }
    unsigned char* p = buf + lo;
    unsigned char* const end = buf + hi;
    size_t phase = lo % 48;
    /* head: up to the next 16-byte boundary of the address */
    while(p < end && ((uintptr_t)p & 15u)) { *p++ = pat[phase]; phase = phase + 1 == 48 ? 0 : phase + 1; }
    /* the pattern as it repeats from here: three vectors */
    unsigned char rot[48];
    for(int k=0; k<48; k++) rot[k] = pat[(phase + k) % 48];
    const __m128i v0 = _mm_loadu_si128((const __m128i*)rot), v1 = _mm_loadu_si128((const __m128i*)(rot + 16)), v2 = _mm_loadu_si128((const __m128i*)(rot + 32));
    while(p + 48 <= end)
    {
        _mm_stream_si128((__m128i*)p, v0); _mm_stream_si128((__m128i*)(p + 16), v1); _mm_stream_si128((__m128i*)(p + 32), v2);
        p += 48;
    }
    for(int k=0; p < end; k++) *p++ = rot[k];
    _mm_sfence();
}

/* the blobs of one chunk of the stream, which starts `first` words into the chunk (what a void at the end of the
 * chunk before reached over): their offsets (in words) into `offsets`, at most `max`; *beyond = how far the last
 * void reaches past this chunk's nwords (the next chunk's `first`).  Returns how many blobs there are, or
 * (size_t)-1 if the chunk is not a sequence of blobs and voids. */
size_t hz_blob_walk(const uint32_t* chunk, size_t nwords, size_t first, size_t* offsets, size_t max, size_t* beyond)
{
    size_t n = 0, at = first;
    *beyond = 0;
    while(at < nwords)
    {
        if(at + 2 > nwords) return (size_t)-1;
        if(chunk[at] == HZ_BLOB_VOID)
        {
            const size_t size = chunk[at + 1];
            if(size < 4 || (size & 3)) return (size_t)-1;
            at += size;
            continue;
        }
        if(at + HZ_BLOB_HDR > nwords) return (size_t)-1;
        const size_t size = chunk[at + 6];
        if(size < HZ_BLOB_HDR || (size & 3) || at + size > nwords) return (size_t)-1;
        if(n < max) offsets[n] = at;
        n++;
        at += size;
    }
    *beyond = at - nwords;
    return n;
}

/* 16 shades -> 48 bytes B,G,R = 0,0,shade (reference fragment.glsl:15-16: colour = (red,0,0)) */
__attribute__((target("ssse3")))
static void expand16_ssse3(unsigned char* dst, const unsigned char* red)
{
    const __m128i r = _mm_loadu_si128((const __m128i*)red);
    const __m128i m0 = _mm_setr_epi8(-128,-128,0, -128,-128,1, -128,-128,2, -128,-128,3, -128,-128,4, -128);
    const __m128i m1 = _mm_setr_epi8(-128,5, -128,-128,6, -128,-128,7, -128,-128,8, -128,-128,9, -128,-128);
    const __m128i m2 = _mm_setr_epi8(10, -128,-128,11, -128,-128,12, -128,-128,13, -128,-128,14, -128,-128,15);
    _mm_storeu_si128((__m128i*)dst,        (__m128i)__builtin_ia32_pshufb128((__v16qi)r, (__v16qi)m0));
    _mm_storeu_si128((__m128i*)(dst + 16), (__m128i)__builtin_ia32_pshufb128((__v16qi)r, (__v16qi)m1));
    _mm_storeu_si128((__m128i*)(dst + 32), (__m128i)__builtin_ia32_pshufb128((__v16qi)r, (__v16qi)m2));
}
static void expand_scalar(unsigned char* dst, const unsigned char* red, int n)
{
    for(int k=0; k<n; k++) { dst[3*k] = 0; dst[3*k+1] = 0; dst[3*k+2] = red[k]; }
}

static int have_ssse3 = -1;

/* one blob into the caller's buffers (any of them may be NULL; [H][SW] pixels, top row first).  Returns 0, or -1
 * if the blob does not describe pixels of a SW x H image. */
int hz_blob_scatter(const uint32_t* blob, int SW, int H, unsigned char* bgr, float* ranges, int32_t* index, uint32_t* z24)
{
    if(have_ssse3 < 0) have_ssse3 = __builtin_cpu_supports("ssse3") ? 1 : 0;
    const uint32_t flags = blob[0] >> 16;
    const int yo0 = (int)(blob[0] & 0xFFFFu), x0 = (int)blob[1], n = (int)blob[7];
    if(n < 1 || n > HZ_BLOB_COLS || x0 < 0 || x0 + n > SW || yo0 >= H) return -1;
    const int mw = (n + 31) >> 5;
    size_t total = 0;
    for(int r=0; r<HZ_BLOB_ROWS; r++) total += blob[2 + r];
    const uint32_t* mask = blob + 8;
    const uint32_t* p = mask + (size_t)HZ_BLOB_ROWS*mw;
    const float*    src_rng = NULL; const int32_t* src_idx = NULL; const uint32_t* src_z = NULL; const unsigned char* src_red = NULL;
    if(flags & HZ_BLOB_RANGES) { src_rng = (const float*)p;   p += total; }
    if(flags & HZ_BLOB_INDEX)  { src_idx = (const int32_t*)p; p += total; }
    if(flags & HZ_BLOB_Z24)    { src_z   = p;                 p += total; }
    if(flags & HZ_BLOB_RED)    { src_red = (const unsigned char*)p; p += (total + 3) >> 2; }
    if((size_t)(p - blob) > blob[6]) return -1;
    if(!src_rng) ranges = NULL;
    if(!src_idx) index = NULL;
    if(!src_z)   z24 = NULL;
    if(!src_red) bgr = NULL;
    size_t k = 0;                                   /* terrain pixels of the blob so far */
    for(int r=0; r<HZ_BLOB_ROWS; r++)
    {
        const int yo = yo0 + r;
        const uint32_t* m = mask + (size_t)r*mw;
        if(yo >= H) { if(blob[2 + r]) return -1; continue; }
        const size_t row = (size_t)yo*SW + x0;
        size_t seen = 0;
        for(int w=0; w<mw; w++)
        {
            uint32_t bits = m[w];
            if(!bits) continue;
            const size_t o = row + 32u*(size_t)w;
            if(bits == 0xFFFFFFFFu)
            {
                /* 32 terrain pixels in a row: below the horizon that is nearly every word */
                if(ranges) memcpy(ranges + o, src_rng + k, 128);
                if(index)  memcpy(index + o,  src_idx + k, 128);
                if(z24)    memcpy(z24 + o,    src_z + k,   128);
                if(bgr)
                {
                    if(have_ssse3) { expand16_ssse3(bgr + 3*o, src_red + k); expand16_ssse3(bgr + 3*o + 48, src_red + k + 16); }
                    else expand_scalar(bgr + 3*o, src_red + k, 32);
                }
                k += 32; seen += 32;
                continue;
            }
            while(bits)
            {
                const int c = __builtin_ctz(bits);
                bits &= bits - 1;
                if(ranges) ranges[o + c] = src_rng[k];
                if(index)  index[o + c]  = src_idx[k];
                if(z24)    z24[o + c]    = src_z[k];
                if(bgr)    { unsigned char* q = bgr + 3*(o + c); q[0] = 0; q[1] = 0; q[2] = src_red[k]; }
                k++; seen++;
            }
        }
        if(seen != blob[2 + r]) return -1;
    }
    return 0;
}
