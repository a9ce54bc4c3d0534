/* hz_rccl.c - see include/horizonator_rccl.h */
#include "horizonator_rccl.h"
#include "horizonator_amd.h"
#include "hz_hip.h"
#include "util.h"

#include <stdbool.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#define NCCL_TRY(call) do { ncclResult_t r_ = (call); if(r_ != ncclSuccess) { MSG("%s -> %s", #call, ncclGetErrorString(r_)); return -1; } } while(0)

int horizonator_rccl_broadcast_mosaic(void* comm, int root, int16_t* d_mosaic, int N, void* stream)
{
    if(comm == NULL || d_mosaic == NULL || N < 2) { MSG("bad arguments"); return -1; }
    NCCL_TRY(ncclBroadcast(d_mosaic, d_mosaic, (size_t)N*(size_t)N*sizeof(int16_t), ncclInt8, root, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}

int horizonator_rccl_gather_strips(const horizonator_context_t* ctx, void* comm, int rank, int world, int root,
                                   const uint32_t* d_send, size_t words, uint32_t* const* d_recv, void* stream)
{
    if(comm == NULL || d_send == NULL || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world ||
       (rank == root && d_recv == NULL))
    {
        MSG("bad arguments");
        return -1;
    }
    /* the strip is written by the context's conversion stream */
    if(ctx != NULL && !horizonator_amd_stream_waits_for_outputs(ctx, stream)) return -1;
    NCCL_TRY(ncclGroupStart());
    /* a failure inside the group must still close it: an open group swallows every later
     * RCCL call of this thread (queued, never launched - a hang, not an error) */
    ncclResult_t first = ncclSuccess, r;
    if(rank == root)
        for(int k=0; k<world && first == ncclSuccess; k++)
        {
            r = ncclRecv(d_recv[k], words, ncclUint32, k, (ncclComm_t)comm, (hipStream_t)stream);
            if(r != ncclSuccess) { MSG("ncclRecv from rank %d -> %s", k, ncclGetErrorString(r)); first = r; }
        }
    if(first == ncclSuccess)
    {
        r = ncclSend(d_send, words, ncclUint32, root, (ncclComm_t)comm, (hipStream_t)stream);
        if(r != ncclSuccess) { MSG("ncclSend to rank %d -> %s", root, ncclGetErrorString(r)); first = r; }
    }
    r = ncclGroupEnd();
    if(first != ncclSuccess) return -1;
    if(r != ncclSuccess) { MSG("ncclGroupEnd -> %s", ncclGetErrorString(r)); return -1; }
    return 0;
}

/* one panorama of a series: which slot it uses and which rank gathers it */
static int series_root(const horizonator_rccl_series_t* s, long i) { return s->rotate ? (int)(i % s->world) : 0; }

/* the gathering rank's half of panorama i, queued behind the arrival of its strips (a wait on the device) */
static int series_convert(const horizonator_context_t* ctx, const horizonator_rccl_series_t* s, long i, hipStream_t conversions, hipEvent_t arrived)
{
    if(s->rank != series_root(s, i)) return 0;
    const int slot = (int)(i % s->nslots);
    hipError_t e = hipStreamWaitEvent(conversions, arrived, 0);
    if(e != hipSuccess) { MSG("hipStreamWaitEvent -> %s", hipGetErrorString(e)); return -1; }
    if(!horizonator_amd_resolve_sparse_strips(ctx, s->world, (const uint32_t* const*)(s->d_bins + (size_t)slot*s->world), s->mask_stride,
                                              s->ncols, s->col0, s->d_image, s->d_ranges)) return -1;
    return 0;
}

int horizonator_rccl_render_series(const horizonator_context_t* ctx, void* comm, const horizonator_rccl_series_t* s,
                                   long first, int count, int check_fit)
{
    if(ctx == NULL || s == NULL || (comm == NULL && s->exchange == NULL) || s->world < 1 || s->rank < 0 || s->rank >= s->world || s->nslots < 1 || s->nslots > 4 ||
       s->d_strips == NULL || s->col0 == NULL || s->ncols == NULL || first < 0 || count < 0 || s->words < s->header_words + 1)
    {
        MSG("bad arguments");
        return -1;
    }
    const bool gathers = s->rotate || s->rank == 0;
    if(gathers && s->d_bins == NULL) { MSG("this rank gathers panoramas and has no bins"); return -1; }
    const bool draws = s->ncols[s->rank] > 0;
    hipStream_t conversions = (hipStream_t)hz_hip_stream(horizonator_amd_device(ctx));     /* where the context converts: strips out, gathered strips in */
    /* exchanged[slot]: the exchange that last used the slot is done - its strip buffer may be refilled, its bins hold a panorama.
     * Everything below is ordered by such waits on the device; the host only queues.  The conversion of a gathered panorama
     * is queued nslots panoramas late (as bench.py's loop does it): queued at once it would sit on the context's
     * conversion stream between this panorama's strip and the next one's, waiting for the exchange - and every strip
     * would wait for the exchange before it (measured with one rank: a 1/8 sector every 0.37 ms instead of every 0.2). */
    hipEvent_t exchanged[4] = { NULL, NULL, NULL, NULL };
    int rc = 0;
    for(int k=0; k<s->nslots && rc == 0; k++)
        if(hipEventCreateWithFlags(&exchanged[k], hipEventDisableTiming) != hipSuccess) { MSG("hipEventCreate failed"); rc = -1; }
    /* (the slots' earlier users - a call before this one - have drained: every call converts what it gathered before it returns,
     * and the first exchanges of this call queue behind those conversions through horizonator_rccl_gather_strips's own wait) */
    int last_slot = -1;
    for(long i=first; i<first+count && rc == 0; i++)
    {
        const int slot = (int)(i % s->nslots);
        if(i - s->nslots >= first)
        {
            if(series_convert(ctx, s, i - s->nslots, conversions, exchanged[slot]) != 0) { rc = -1; break; }
            /* ... and the strip buffer is free (a rank that did not gather that panorama has not waited yet) */
            if(hipStreamWaitEvent(conversions, exchanged[slot], 0) != hipSuccess) { MSG("hipStreamWaitEvent failed"); rc = -1; break; }
        }
        else if(!horizonator_amd_waits_for_stream(ctx, s->stream)) { rc = -1; break; }     /* whatever the caller's stream still does with the slot */
        if(draws && !horizonator_amd_render_sparse(ctx, s->d_strips[slot], s->mask_stride)) { rc = -1; break; }
        /* (queued behind the strip's conversion, and behind the conversion that last read this slot's bins) */
        uint32_t* const* bins = s->rank == series_root(s, i) ? s->d_bins + (size_t)slot*s->world : NULL;
        if(s->exchange != NULL)
        {
            /* the caller's transport: behind the strip's conversion, like the RCCL gather */
            if(!horizonator_amd_stream_waits_for_outputs(ctx, s->stream)) { rc = -1; break; }
            if(s->exchange(s->exchange_user, series_root(s, i), s->d_strips[slot], s->words, bins, s->stream) != 0) { MSG("the caller's exchange failed"); rc = -1; break; }
        }
        else if(horizonator_rccl_gather_strips(ctx, comm, s->rank, s->world, series_root(s, i), s->d_strips[slot], s->words, bins, s->stream) != 0) { rc = -1; break; }
        if(hipEventRecord(exchanged[slot], (hipStream_t)s->stream) != hipSuccess) { MSG("hipEventRecord failed"); rc = -1; break; }
        last_slot = slot;
    }
    /* the panoramas still in their bins, oldest first */
    for(long i = first + count - s->nslots; i < first + count && rc == 0; i++)
        if(i >= first && series_convert(ctx, s, i, conversions, exchanged[i % s->nslots]) != 0) rc = -1;
    for(int k=0; k<s->nslots; k++) if(exchanged[k]) (void)hipEventDestroy(exchanged[k]);   /* (released when the work that waits on them is done) */
    if(rc != 0) return -1;
    if(check_fit && draws && last_slot >= 0)
    {
        uint32_t terrain = 0;
        /* (the exchange stream too: the last panoramas' send/receive groups may still be in flight on THIS communicator,
         * and a caller that now agrees on the fit over ANOTHER communicator - bench.py's all_reduce over torch's - would
         * have collectives of two communicators in flight on one device, which RCCL documents as deadlock-prone) */
        hipError_t es = hipStreamSynchronize((hipStream_t)s->stream);
        if(es != hipSuccess) { MSG("hipStreamSynchronize -> %s", hipGetErrorString(es)); return -1; }
        if(!horizonator_amd_sync(ctx)) return -1;
        hipError_t e = hipMemcpy(&terrain, s->d_strips[last_slot], sizeof(terrain), hipMemcpyDeviceToHost);
        if(e != hipSuccess) { MSG("hipMemcpy -> %s", hipGetErrorString(e)); return -1; }
        if((size_t)terrain + s->header_words > s->words) return 1;
    }
    return 0;
}
