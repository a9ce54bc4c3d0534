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

#ifdef __cplusplus
}
#endif
