/* horizonator_rccl.h - the exchange steps of the multi-GPU render for a C caller
 * (libhorizonator_rccl.so; links RCCL, which libhorizonator.so itself does not).
 *
 * The reference has no multi-device code; this is the build's own (SURVEY.md
 * section 8e, DESIGN.md "Multi-GPU"): one process per GPU, every rank holds the
 * whole DEM mosaic and draws one azimuth sector (horizonator_amd_set_sector,
 * horizonator_amd.h), the strips travel to one rank, which converts them.  The
 * Python counterpart is horizonator_amd/sharding.py over torch.distributed.
 *
 *   comm    an ncclComm_t of the caller's (as a void*): ncclCommInitRank etc. are
 *           the caller's business
 *   stream  the hipStream_t (as a void*) the collective is queued on; the calls
 *           order it against the context's own streams on the device and return
 *           without waiting - hipStreamSynchronize(stream), or
 *           horizonator_amd_waits_for_stream() before converting the strips
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "horizonator.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rank `root` passes the int16 mosaic it read ([N][N], DEVICE memory); the others
 * receive it into d_mosaic - after which they build their context with
 * horizonator_amd_init_from_mosaic() on a host copy, or keep it on the device.
 * ncclBroadcast of N*N*2 bytes.  Returns 0, or -1 (message on stderr). */
int horizonator_rccl_broadcast_mosaic(void* comm, int root, int16_t* d_mosaic, int N, void* stream);

/* The gather of a panorama's strips: every rank sends `words` uint32 from
 * d_send (its sparse or packed strip as horizonator_amd_render_sparse /
 * _render_packed wrote it, cut or padded to a length all ranks agree on); rank
 * `root` receives rank r's words into d_recv[r] (world pointers, DEVICE memory,
 * `words` each; ignored elsewhere, may be NULL).  One ncclGroup of
 * ncclSend/ncclRecv: world-1 messages converge on the root over its xGMI links
 * at once.  ctx: the rank's context - the exchange is queued behind its
 * conversions (the strip is complete when it leaves); NULL skips that. */
int horizonator_rccl_gather_strips(const horizonator_context_t* ctx, void* comm, int rank, int world, int root,
                                   const uint32_t* d_send, size_t words, uint32_t* const* d_recv, void* stream);

/* A series of panoramas of the context's current view without the host in their way: what bench.py's Python loop
 * does per panorama (sharding.StripExchange: draw the sector, queue the gather behind the strip's conversion, convert
 * the gathered strips on the gathering rank) as one C call - five library calls and one ncclGroup per panorama, no
 * interpreter, no tensor bookkeeping (profiles/r4_c_loop_host_time.txt).  Nothing in it waits on the host for the
 * device; the caller synchronises `stream` and the context (horizonator_amd_sync) when it wants the results.
 *
 *   rank, world     as in the communicator
 *   rotate          0: every panorama is gathered and converted on rank 0; 1: panorama i on rank i % world
 *   nslots          strip buffers (and sets of bins) used in turn, 1..4: panorama i uses slot i % nslots, and is
 *                   ordered on the device behind the exchange that last read that slot
 *   d_strips        [nslots]        this rank's strip buffers (DEVICE; horizonator_amd_render_sparse's d_out)
 *   d_bins          [nslots*world]  where a gathering rank receives rank r's strip of a panorama in slot s:
 *                                   d_bins[s*world + r], `words` uint32 each; NULL on a rank that never gathers
 *   words           what every rank sends per strip - all ranks agree on it (the longest strip + a margin)
 *   header_words    1 + H + H*mask_stride (the strips' header; for the check below)
 *   col0, ncols     [world] the layout: rank r draws image columns [col0[r], col0[r]+ncols[r]) - this context is
 *                   set to its own (horizonator_amd_set_sector); ncols[rank] == 0: it draws nothing and sends filler
 *   d_image, d_ranges  the gathering ranks' full-width outputs (DEVICE; either may be NULL)
 *   first, count    panoramas first .. first+count-1 (the index decides slot and gathering rank: a caller that
 *                   splits a series over several calls passes where it is)
 *   check_fit       != 0: after the last panorama is queued, wait for `stream` and the context (no exchange of this
 *                   series is in flight when the call returns) and look at the length of this rank's last strip
 * Returns 0; 1 if check_fit and this rank's strip was longer than `words` (the panoramas gathered are then
 * incomplete: agree on more words and repeat); -1 on an error (message on stderr). */
typedef struct
{
    int rank, world, rotate, nslots;
    uint32_t* const* d_strips;
    uint32_t* const* d_bins;
    size_t words, header_words;
    int mask_stride;
    const int* col0;
    const int* ncols;
    void* d_image;
    float* d_ranges;
    void* stream;
    /* NULL: the strips travel over RCCL (`comm`).  Otherwise the exchange of one panorama is the caller's: called once per
     * panorama in place of horizonator_rccl_gather_strips() with the arguments that would have got (`comm` may then be NULL) -
     * every rank sends `words` from d_send, rank `root` receives rank r's into d_recv[r] (NULL on the other ranks) - and must
     * leave the transfer ordered on `stream` (queued on it, or complete when it returns).  0 on success.  What it is for: a
     * transport other than RCCL - tests/test_gpu_rccl.py runs the series with two ranks that share one GPU (which RCCL
     * refuses) over torch.distributed's gloo backend, so that the slot and rotation logic meets a second rank. */
    int (*exchange)(void* user, int root, const uint32_t* d_send, size_t words, uint32_t* const* d_recv, void* stream);
    void* exchange_user;
} horizonator_rccl_series_t;
int horizonator_rccl_render_series(const horizonator_context_t* ctx, void* comm, const horizonator_rccl_series_t* s,
                                   long first, int count, int check_fit);

#ifdef __cplusplus
}
#endif
