/* hz_selftest.h - entry points of libhorizonator_selftest.so only: the library's own sources compiled with
 * -DHZ_SELFTEST, i.e. everything include/hz_hip.h declares plus the device-side self-checks of the arithmetic
 * shortcuts and the diagnostics that tools/ use.  The library that ships (libhorizonator.so) exports none of these. */
#pragma once

#include "hz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Self-check of the marching kernel's abridged division / square-root
 * sequences (horizonator_amd/csrc/hz_fast.h) against the device's own `/` and
 * sqrtf, bit for bit.  what: 0 reciprocal, every float32 pattern in the
 * sequences' operand range; 1 square root, every pattern from 2^-96 up; 2
 * division, n seeded pairs; 3 division by the constant whose bit pattern is
 * `seed`, every numerator pattern; 4 k_big's double reciprocal (hz_rcp_f64),
 * every divisor 1 <= d < 2^31: within 2^-50 of 1/d; 5 k_big's exact floor
 * division (hz_floor_div), n seeded (numerator, divisor) pairs against 64-bit
 * integer division.  *mismatches = how many differed (0 is the
 * only acceptable answer); first_bad (4 floats: a, b, want, got) may be NULL. */
int  hz_hip_check_fastmath(int device, int what, unsigned long long seed, unsigned long long n,
                           unsigned long long* mismatches, float* first_bad);

/* Self-checks of the marching kernel's two shortcuts that rest on an argument instead of
 * on the reference's arithmetic, on seeded inputs around every border of the argument:
 *   what 0  hz_tri_hidden() (the early depth test without its division): n triangles -
 *           far-field ones, slivers with |area| down to 2^-12 px^2, depth gradients up to
 *           10^5 LSB per pixel, coordinates up to W x H; for each the largest stored depth
 *           that still reads "hidden" is found and every covered pixel centre drawn with
 *           hz_tri_planes()/hz_tri_fragment(): no fragment may pass GL_LESS (reference
 *           horizonator-lib.c:183).  out: triangles tested, hidden for some depth, fragments
 *           drawn, VIOLATIONS, smallest (fragment depth - stored depth) + 2^32
 *   what 1  the cull of whole cells (mr_simple_cull) against hz_tri_cull() (reference
 *           geometry.glsl:21-27 + GL's cull and scissor): n cases of two rows of 64
 *           vertices - cells about a sixteenth of the image wide, rows touching the view
 *           volume's faces, positions on pixel centres, the +-180 degree seam, back faces;
 *           drawn columns [col0,col1).  out: cases, cases culled the short way, cells
 *           compared, DISAGREEMENTS, triangles kept
 *   what 2  the smallest depth any pixel centre of a rectangle gets from a set-up triangle's
 *           depth plane (hz_k_hiz.h: the smallest of the four corners', which lets k_big drop
 *           chunks of rows in zoomed views) against the minimum over all of the rectangle's
 *           pixel centres as hz_tri_fragment() computes them: n rectangles of up to 48 x 48,
 *           planes flat to 10^6 per pixel, clamped, overflowing, infinite, NaN.  out: cases,
 *           cases with numbers at all corners, pixel centres evaluated, DISAGREEMENTS, cases
 *           with a depth that is not a number
 * 0 violations / disagreements is the only acceptable answer. */
int  hz_hip_check_exactness(int device, int what, unsigned long long seed, unsigned long long n,
                            int W, int H, int col0, int col1, unsigned long long* out);

/* diagnostics (tools/history/bigqueue_stats.py): the queue of large triangles the last
 * draw left behind - set 0: its only or second round, 1: the first round of a
 * two-round draw.  counters: 6 words (mr_queue_t); recs: 10 int32 per record
 * (px0 py0 bw bh, the three edge vectors' dx, then dy, in 1/256 pixel), at most max_rec */
int  hz_hip_debug_bigqueue(hz_dev_t* d, int set, unsigned int* counters, int max_rec, int32_t* recs);

/* diagnostics / tests (no device needed): the list of marching waves a draw of columns
 * [col0,col1) of `view` launches on a context of N samples per axis and a W x H image -
 * azimuth sectors and views of less than 360 degrees launch only the strips of the DEM
 * that can reach their columns.  round: 0 = a one-round draw, 1 / 2 = the rounds of a
 * two-round draw; + 256: every strip of the grid, no azimuth test.  out: 3 int32 per wave (strip column, first cell row, cell row behind
 * the last).  Returns the number of waves (only capacity_items of them written if that
 * is less), -1 for a draw that launches the whole grid (the full circle). */
long hz_hip_debug_worklist(int N, int W, int H, const hz_view_t* view, int col0, int col1, int round,
                           int32_t* out, size_t capacity_items);

/* diagnostics (tools/wave_timing.py): `view` drawn once more by the instance of the
 * marching kernel that counts; per wave of its second (or only) round 4 words: duration
 * in shader clock cycles, flushes<<32 | triangles set up, to k_big<<32 | to k_mid,
 * hidden by the early depth test<<32 | pixel centres tested by the wave itself.
 * grid[2] = the launch grid; out holds grid[0]*grid[1]*4 words (capacity_words: its size) */
int  hz_hip_debug_wave_timing(hz_dev_t* d, const hz_view_t* view, unsigned long long* out, size_t capacity_words, unsigned int* grid);

#ifdef __cplusplus
}
#endif
