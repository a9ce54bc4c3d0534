/* hz_rccl.c - see include/horizonator_rccl.h */
#include "horizonator_rccl.h"
#include "horizonator_amd.h"
#include "util.h"

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
