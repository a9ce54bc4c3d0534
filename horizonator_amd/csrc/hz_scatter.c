/* hz_scatter.c - the host half of "results into the caller's memory without the sky".
 *
 * The reference hands its results to the caller in host memory (reference horizonator-lib.c:936-1048: two
 * glReadPixels into the caller's buffers, the clear colour and the cleared depth where nothing was drawn,
 * :185, :1016).  In a panorama most pixels are exactly that - sky: BGR (255,0,0), range -1 - and a sky pixel
 * carries no information.  The device therefore sends only the terrain pixels, as "blobs" (k_pack_host,
 * hz_k_resolve.h; layout below), and the host threads of hz_pool.h
 *   - fill the caller's buffers with the sky's constants while the draw is still running (hz_sky_fill), and
 *   - put each blob's terrain pixels in their places as its bytes arrive (hz_blob_scatter):
 * the same bytes in the caller's buffers as the dense copy, a third to a fifth of the bytes over PCIe.
 *
 * A blob = the terrain pixels of up to 4 image rows x up to 2048 columns, uint32 words:
 *   [0]     first row (top row = 0) | flags << 16   (HZ_BLOB_PACKED | _INDEX | _RED: the arrays it carries)
 *   [1]     first column (of the IMAGE: a sector's blobs carry their sector's offset)
 *   [2..5]  terrain pixels T0..T3 of its four rows     [6] size of the blob in words     [7] columns n (<= 2048)
 *   then    4 x ceil(n/32) mask words (row after row; bit c%32 of word c/32: column c shows terrain)
 *   then    per carried array T0+T1+T2+T3 entries, row after row, left to right:
 *             PACKED  a word z24<<8 | red8 - the 24-bit depth and the shade, from which this file makes the BGR
 *                     bytes, the raw depth and the float32 range (reference horizonator-lib.c:1013-1025) exactly
 *                     as the device's conversions make them: 4 bytes over PCIe where range + shade took 5
 *             INDEX   a word: the id of the triangle that owns the pixel
 *             RED     a BYTE: the shade alone (callers that want the image only), padded to a word
 * Blobs of tiles without terrain are not sent.  The stream travels in chunks of HZ_STAGE_BYTES and no blob
 * straddles a chunk boundary: where a blob would have (the writers take their places with an atomic add), the
 * stream holds a void instead - word [0] = HZ_BLOB_VOID, word [1] = its length in words - which may reach into
 * the next chunk.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <emmintrin.h>
#include <immintrin.h>

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

/* what this machine's vector unit can do, found out once when the library is loaded (not lazily by whichever pool
 * thread comes first) */
static int cpu_ssse3, cpu_avx2, cpu_avx512;
/* streaming (non-temporal) stores for runs of 32 terrain pixels; 0: ordinary stores (tools/scatter_bench.c measures both) */
static int scatter_streaming = 1;
void hz_scatter_set_streaming(int on) { scatter_streaming = on; }
__attribute__((constructor)) static void hz_scatter_probe_cpu(void)
{
    __builtin_cpu_init();
    cpu_ssse3 = __builtin_cpu_supports("ssse3") ? 1 : 0;
    cpu_avx2  = __builtin_cpu_supports("avx2") ? 1 : 0;
    cpu_avx512 = (cpu_avx2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) ? 1 : 0;
}

/* 8 ranges at a time where the machine has 512-bit vectors (the double-precision square root is what the conversion
 * costs: 8 per instruction instead of 4) */
__attribute__((target("avx512f,avx512vl,avx2")))
static inline __m256 ranges8_avx512(const uint32_t* w, __m256 vspan, __m256 vnear, __m256 vtan)
{
    const __m256i zi    = _mm256_srli_epi32(_mm256_loadu_si256((const __m256i*)w), 8);
    const __m256  depth = _mm512_cvtpd_ps(_mm512_mul_pd(_mm512_cvtepi32_pd(zi), _mm512_set1_pd(1.0/16777215.0)));
    const __m256  len   = _mm256_add_ps(_mm256_mul_ps(depth, vspan), vnear);
    const __m256  zt    = _mm256_mul_ps(vtan, len);
    const __m512d l = _mm512_cvtps_pd(len), z = _mm512_cvtps_pd(zt);
    return _mm512_cvtpd_ps(_mm512_sqrt_pd(_mm512_add_pd(_mm512_mul_pd(l, l), _mm512_mul_pd(z, z))));
}
__attribute__((target("avx512f,avx512vl,avx2")))
static void ranges_avx512(float* out, const uint32_t* packed, size_t n, float tan_row, float znear, float span)
{
    const __m256 vspan = _mm256_set1_ps(span), vnear = _mm256_set1_ps(znear), vtan = _mm256_set1_ps(tan_row);
    size_t k = 0;
    for(; k + 8 <= n; k += 8) _mm256_storeu_ps(out + k, ranges8_avx512(packed + k, vspan, vnear, vtan));
    for(; k < n; k++)
    {
        const float depth = (float)((double)(packed[k] >> 8) * (1.0/16777215.0));
        const float len   = depth * span + znear;
        const float zt    = tan_row * len;
        out[k] = (float)__builtin_sqrt((double)len*(double)len + (double)zt*(double)zt);
    }
}
__attribute__((target("avx512f,avx512vl,avx2")))
static void word32_ranges_avx512(const uint32_t* w, float* rng, float tan_row, float znear, float span)
{
    const __m256 vspan = _mm256_set1_ps(span), vnear = _mm256_set1_ps(znear), vtan = _mm256_set1_ps(tan_row);
    const int nt = scatter_streaming && ((uintptr_t)rng & 31u) == 0;
    for(int k=0; k<32; k+=8)
    {
        const __m256 r = ranges8_avx512(w + k, vspan, vnear, vtan);
        if(nt) _mm256_stream_ps(rng + k, r); else _mm256_storeu_ps(rng + k, r);
    }
}

/* ---- depth -> range on the host ------------------------------------------------------------------------------
 * reference horizonator-lib.c:1013-1025:  depth = float(z24 / (2^24-1))   (what glReadPixels hands out)
 *                                         length_en = depth*(zfar-znear) + znear;  z = tanel*length_en;
 *                                         range = hypotf(length_en, z)
 * hypotf of two floats is the correctly rounded double square root of the exact sum of squares, rounded to float (the
 * squares of floats are exact in double): the form the device kernels use (hz_k_resolve.h) and the one used here, with the
 * IEEE operations in the same order - no fused multiply-add (this file is built with -ffp-contract=off and the vector
 * version spells out every operation). */
static inline float range_of_packed(uint32_t w, float tan_row, float znear, float span)
{
    const float depth = (float)((double)(w >> 8) * (1.0/16777215.0));
    const float len   = depth * span + znear;
    const float zt    = tan_row * len;
    return (float)__builtin_sqrt((double)len*(double)len + (double)zt*(double)zt);
}

__attribute__((target("avx2")))
static void ranges_avx2(float* out, const uint32_t* packed, size_t n, float tan_row, float znear, float span)
{
    const __m256d inv = _mm256_set1_pd(1.0/16777215.0);
    const __m128  vspan = _mm_set1_ps(span), vnear = _mm_set1_ps(znear), vtan = _mm_set1_ps(tan_row);
    size_t k = 0;
    for(; k + 4 <= n; k += 4)
    {
        const __m128i zi    = _mm_srli_epi32(_mm_loadu_si128((const __m128i*)(packed + k)), 8);
        const __m128  depth = _mm256_cvtpd_ps(_mm256_mul_pd(_mm256_cvtepi32_pd(zi), inv));
        const __m128  len   = _mm_add_ps(_mm_mul_ps(depth, vspan), vnear);
        const __m128  zt    = _mm_mul_ps(vtan, len);
        const __m256d l = _mm256_cvtps_pd(len), z = _mm256_cvtps_pd(zt);
        const __m256d r = _mm256_sqrt_pd(_mm256_add_pd(_mm256_mul_pd(l, l), _mm256_mul_pd(z, z)));
        _mm_storeu_ps(out + k, _mm256_cvtpd_ps(r));
    }
    for(; k < n; k++) out[k] = range_of_packed(packed[k], tan_row, znear, span);
}

void hz_ranges_from_packed(float* out, const uint32_t* packed, size_t n, float tan_row, float znear, float zfar)
{
    const float span = zfar - znear;
    if(cpu_avx512) { ranges_avx512(out, packed, n, tan_row, znear, span); return; }
    if(cpu_avx2) { ranges_avx2(out, packed, n, tan_row, znear, span); return; }
    for(size_t k=0; k<n; k++) out[k] = range_of_packed(packed[k], tan_row, znear, span);
}

/* 32 terrain pixels in a row - below the horizon that is nearly every mask word - straight from the blob's words into the
 * caller's buffers: ranges 4 at a time through the double-precision square root, depths by a shift, the shades packed to
 * bytes and spread to B,G,R.  Streaming stores where the destination is aligned (a 16000-wide image: always): the sky was
 * written the same way, nothing of these lines is in any cache, and an ordinary store would first read the line it is
 * about to overwrite. */
__attribute__((target("avx2")))
static void word32_avx2(const uint32_t* w, float* rng, uint32_t* z24, unsigned char* bgr, float tan_row, float znear, float span)
{
    if(rng)
    {
        const __m256d inv = _mm256_set1_pd(1.0/16777215.0);
        const __m128  vspan = _mm_set1_ps(span), vnear = _mm_set1_ps(znear), vtan = _mm_set1_ps(tan_row);
        const int nt = scatter_streaming && ((uintptr_t)rng & 15u) == 0;
        for(int k=0; k<32; k+=4)
        {
            const __m128i zi    = _mm_srli_epi32(_mm_loadu_si128((const __m128i*)(w + k)), 8);
            const __m128  depth = _mm256_cvtpd_ps(_mm256_mul_pd(_mm256_cvtepi32_pd(zi), inv));
            const __m128  len   = _mm_add_ps(_mm_mul_ps(depth, vspan), vnear);
            const __m128  zt    = _mm_mul_ps(vtan, len);
            const __m256d l = _mm256_cvtps_pd(len), z = _mm256_cvtps_pd(zt);
            const __m128  r = _mm256_cvtpd_ps(_mm256_sqrt_pd(_mm256_add_pd(_mm256_mul_pd(l, l), _mm256_mul_pd(z, z))));
            if(nt) _mm_stream_ps(rng + k, r); else _mm_storeu_ps(rng + k, r);
        }
    }
    if(z24)
    {
        const int nt = scatter_streaming && ((uintptr_t)z24 & 15u) == 0;
        for(int k=0; k<32; k+=4)
        {
            const __m128i zi = _mm_srli_epi32(_mm_loadu_si128((const __m128i*)(w + k)), 8);
            if(nt) _mm_stream_si128((__m128i*)(z24 + k), zi); else _mm_storeu_si128((__m128i*)(z24 + k), zi);
        }
    }
    if(bgr)
    {
        const __m128i lo8 = _mm_set1_epi32(0xFF);
        const __m128i m0 = _mm_setr_epi8(-128,-128,0, -128,-128,1, -128,-128,2, -128,-128,3, -128,-128,4, -128);
        const __m128i m1 = _mm_setr_epi8(-128,5, -128,-128,6, -128,-128,7, -128,-128,8, -128,-128,9, -128,-128);
        const __m128i m2 = _mm_setr_epi8(10, -128,-128,11, -128,-128,12, -128,-128,13, -128,-128,14, -128,-128,15);
        const int nt = scatter_streaming && ((uintptr_t)bgr & 15u) == 0;
        for(int k=0; k<32; k+=16)
        {
            /* 16 shades as bytes (reference fragment.glsl:15-16: colour = (red,0,0)), then 48 bytes B,G,R = 0,0,shade */
            const __m128i a = _mm_and_si128(_mm_loadu_si128((const __m128i*)(w + k)),      lo8), b = _mm_and_si128(_mm_loadu_si128((const __m128i*)(w + k + 4)),  lo8);
            const __m128i c = _mm_and_si128(_mm_loadu_si128((const __m128i*)(w + k + 8)),  lo8), d = _mm_and_si128(_mm_loadu_si128((const __m128i*)(w + k + 12)), lo8);
            const __m128i r = _mm_packus_epi16(_mm_packus_epi32(a, b), _mm_packus_epi32(c, d));
            const __m128i o0 = _mm_shuffle_epi8(r, m0), o1 = _mm_shuffle_epi8(r, m1), o2 = _mm_shuffle_epi8(r, m2);
            unsigned char* q = bgr + 3*k;
            if(nt) { _mm_stream_si128((__m128i*)q, o0); _mm_stream_si128((__m128i*)(q + 16), o1); _mm_stream_si128((__m128i*)(q + 32), o2); }
            else   { _mm_storeu_si128((__m128i*)q, o0); _mm_storeu_si128((__m128i*)(q + 16), o1); _mm_storeu_si128((__m128i*)(q + 32), o2); }
        }
    }
}

/* one blob into the caller's buffers.  Returns 0, or -1 if the blob does not describe pixels of dst's image - decided
 * BEFORE anything is written: the columns and rows lie inside the image, every row's mask has exactly the bits its count
 * says and none beyond the blob's columns, and the arrays those counts imply fit the size the blob declares. */
/* npx (<= 32) sky pixels from pixel o on: the constants of hz_sky_fill() */
static void sky_run(const hz_scatter_dst_t* d, int want_bgr, int want_rng, int want_idx, int want_z, size_t o, int npx, int stream_ok)
{
    if(npx == 32 && stream_ok && scatter_streaming)
    {
        if(want_rng && ((uintptr_t)(d->ranges + o) & 15u) == 0)
        {
            const __m128 v = _mm_set1_ps(-1.0f);
            for(int k=0; k<32; k+=4) _mm_stream_ps(d->ranges + o + k, v);
            want_rng = 0;
        }
        if(want_z && ((uintptr_t)(d->z24 + o) & 15u) == 0)
        {
            const __m128i v = _mm_set1_epi32(0x00FFFFFF);
            for(int k=0; k<32; k+=4) _mm_stream_si128((__m128i*)(d->z24 + o + k), v);
            want_z = 0;
        }
        if(want_idx && ((uintptr_t)(d->index + o) & 15u) == 0)
        {
            const __m128i v = _mm_set1_epi32(-1);
            for(int k=0; k<32; k+=4) _mm_stream_si128((__m128i*)(d->index + o + k), v);
            want_idx = 0;
        }
        if(want_bgr && ((uintptr_t)(d->bgr + 3*o) & 15u) == 0)
        {
            /* B,G,R = 255,0,0 (reference horizonator-lib.c:185): 96 bytes = two periods of the 48-byte pattern */
            const __m128i v0 = _mm_setr_epi8(-1,0,0, -1,0,0, -1,0,0, -1,0,0, -1,0,0, -1), v1 = _mm_setr_epi8(0,0,-1, 0,0,-1, 0,0,-1, 0,0,-1, 0,0,-1, 0),
                          v2 = _mm_setr_epi8(0,-1,0, 0,-1,0, 0,-1,0, 0,-1,0, 0,-1,0, 0);
            __m128i* q = (__m128i*)(d->bgr + 3*o);
            _mm_stream_si128(q, v0); _mm_stream_si128(q+1, v1); _mm_stream_si128(q+2, v2);
            _mm_stream_si128(q+3, v0); _mm_stream_si128(q+4, v1); _mm_stream_si128(q+5, v2);
            want_bgr = 0;
        }
    }
    for(int c=0; c<npx; c++)
    {
        if(want_rng) d->ranges[o + c] = -1.0f;                                              /* reference horizonator-lib.c:1016 */
        if(want_z)   d->z24[o + c] = 0x00FFFFFFu;
        if(want_idx) d->index[o + c] = -1;
        if(want_bgr) { unsigned char* b3 = d->bgr + 3*(o + c); b3[0] = 255; b3[1] = 0; b3[2] = 0; }
    }
}

int hz_blob_scatter(const uint32_t* blob, const hz_scatter_dst_t* dst) { return hz_blob_scatter_mode(blob, dst, 0); }

/* full != 0: every pixel of the blob's tile (its rows inside the image x its columns) is written, the sky's constants where
 * the mask says sky - for tiles the sky has NOT been filled into beforehand: each byte of the caller's buffers is then
 * written once instead of twice */
int hz_blob_scatter_mode(const uint32_t* blob, const hz_scatter_dst_t* dst, int full)
{
    const int W = dst->W, H = dst->H;
    const uint32_t flags = blob[0] >> 16;
    const int yo0 = (int)(blob[0] & 0xFFFFu), n = (int)blob[7];
    if(flags & ~(HZ_BLOB_PACKED | HZ_BLOB_INDEX | HZ_BLOB_RED)) return -1;
    if(n < 1 || n > HZ_BLOB_COLS || blob[1] > (uint32_t)W || (int)blob[1] + n > W || yo0 >= H) return -1;
    const int x0 = (int)blob[1];
    const int mw = (n + 31) >> 5;
    const uint32_t* mask = blob + HZ_BLOB_HDR;
    size_t total = 0;
    for(int r=0; r<HZ_BLOB_ROWS; r++)
    {
        const uint32_t* m = mask + (size_t)r*mw;
        uint32_t bits = 0;
        for(int w=0; w<mw; w++) bits += (uint32_t)__builtin_popcount(m[w]);
        if(bits != blob[2 + r]) return -1;
        if((n & 31) && (m[mw-1] >> (n & 31))) return -1;
        if(yo0 + r >= H && bits) return -1;
        total += bits;
    }
    const uint32_t* p = mask + (size_t)HZ_BLOB_ROWS*mw;
    const uint32_t* src_pk = NULL; const int32_t* src_idx = NULL; const unsigned char* src_red = NULL;
    if(flags & HZ_BLOB_PACKED) { src_pk  = p;                 p += total; }
    if(flags & HZ_BLOB_INDEX)  { src_idx = (const int32_t*)p; p += total; }
    if(flags & HZ_BLOB_RED)    { src_red = (const unsigned char*)p; p += (total + 3) >> 2; }
    if((size_t)(p - blob) > blob[6]) return -1;
    unsigned char* bgr    = (src_pk || src_red) ? dst->bgr : NULL;
    float*         ranges = src_pk ? dst->ranges : NULL;
    uint32_t*      z24    = src_pk ? dst->z24 : NULL;
    int32_t*       index  = src_idx ? dst->index : NULL;
    if(ranges && !dst->tanel) return -1;

    const float span = dst->zfar - dst->znear;
    int streamed = 0;
    size_t k = 0;                                   /* terrain pixels of the blob so far */
    for(int r=0; r<HZ_BLOB_ROWS; r++)
    {
        const int yo = yo0 + r;
        const size_t T = blob[2 + r];
        if(yo >= H || (T == 0 && !full)) continue;
        const uint32_t* m = mask + (size_t)r*mw;
        const uint32_t*      pk  = src_pk  ? src_pk  + k : NULL;
        const int32_t*       idx = src_idx ? src_idx + k : NULL;
        const unsigned char* red = src_red ? src_red + k : NULL;
        const float tan_row = ranges ? dst->tanel[H-1 - yo] : 0.f;
        const size_t row = (size_t)yo*W + x0;
        size_t q = 0;                               /* terrain pixels of the row so far */
        for(int w=0; w<mw; w++)
        {
            uint32_t bits = m[w];
            const size_t o = row + 32u*(size_t)w;
            if(full && bits != 0xFFFFFFFFu)
            {
                /* (the arrays the blob does not carry are not touched: what it carries is what the caller asked for) */
                /* (a word with sky only: streaming stores; with both: ordinary ones - the terrain pixels that follow go into the same lines) */
                sky_run(dst, bgr != NULL, ranges != NULL, index != NULL, z24 != NULL, o, w == mw-1 && (n & 31) ? (n & 31) : 32, bits == 0);
                streamed = 1;
            }
            if(!bits) continue;
            if(bits == 0xFFFFFFFFu)
            {
                /* 32 terrain pixels in a row: below the horizon that is nearly every word */
                if(pk && cpu_avx2)
                {
                    if(ranges && cpu_avx512) word32_ranges_avx512(pk + q, ranges + o, tan_row, dst->znear, span);
                    word32_avx2(pk + q, ranges && !cpu_avx512 ? ranges + o : NULL, z24 ? z24 + o : NULL, bgr ? bgr + 3*o : NULL, tan_row, dst->znear, span);
                    streamed = 1;
                }
                else if(pk)
                {
                    for(int c=0; c<32; c++)
                    {
                        const uint32_t v = pk[q + c];
                        if(ranges) ranges[o + c] = range_of_packed(v, tan_row, dst->znear, span);
                        if(z24)    z24[o + c]    = v >> 8;
                        if(bgr)    { unsigned char* b3 = bgr + 3*(o + c); b3[0] = 0; b3[1] = 0; b3[2] = (unsigned char)v; }
                    }
                }
                else if(bgr)
                {
                    if(cpu_ssse3) { expand16_ssse3(bgr + 3*o, red + q); expand16_ssse3(bgr + 3*o + 48, red + q + 16); }
                    else expand_scalar(bgr + 3*o, red + q, 32);
                }
                if(index) memcpy(index + o, idx + q, 128);
                q += 32;
                continue;
            }
            while(bits)
            {
                const int c = __builtin_ctz(bits);
                bits &= bits - 1;
                if(pk)
                {
                    const uint32_t v = pk[q];
                    if(ranges) ranges[o + c] = range_of_packed(v, tan_row, dst->znear, span);
                    if(z24)    z24[o + c]    = v >> 8;
                    if(bgr)    { unsigned char* b3 = bgr + 3*(o + c); b3[0] = 0; b3[1] = 0; b3[2] = (unsigned char)v; }
                }
                else if(bgr) { unsigned char* b3 = bgr + 3*(o + c); b3[0] = 0; b3[1] = 0; b3[2] = red[q]; }
                if(index) index[o + c] = idx[q];
                q++;
            }
        }
        k += T;
    }
    if(streamed) _mm_sfence();
    return 0;
}
