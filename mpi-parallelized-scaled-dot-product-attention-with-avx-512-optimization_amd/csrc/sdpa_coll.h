// sdpa_coll.h -- the collective layer under the host-level pipeline (internal).
//
// The reference merges its K/V shards with MPI_Iallreduce(MAX), MPI_Iallreduce(SUM) and
// MPI_Ireduce(SUM) (attention-mpi.c:340-380).  Here ONE host thread drives P logical ranks,
// each with its own compute stream, and a collective is one call that enqueues the operation
// on all P streams.  Two implementations behind the same interface:
//
//   rccl      P physical GPUs; ncclAllReduce / ncclAllGather / ncclReduce inside one
//             ncclGroupStart/End over the P communicators of ncclCommInitAll (xGMI).
//             RCCL is bound with dlopen only when this implementation is created.
//   loopback  P logical ranks that all live on ONE device (SDPA_VIRTUAL_GPUS=P): every rank
//             still has its own streams and buffers; a collective is an event fan-in to a hub
//             stream, one small kernel that reads the P send buffers and writes the receive
//             buffers (rank order, deterministic), and an event fan-out.  It exists so that the
//             P > 1 choreography of the host pipeline runs -- and is parity-tested -- on a
//             one-GPU box; its results are exact, it is not a stub.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sdpa {

constexpr int kMaxRanks = 16;

enum class RedOp { Sum, Max };

class Collectives {
public:
    virtual ~Collectives() {}
    virtual const char *name() const = 0;
    int ranks() const { return P_; }
    // recv[r][i] = op_p send[p][i], i < count, on every rank r.  send[r] may equal recv[r].
    virtual int all_reduce(float *const *send, float *const *recv, size_t count, RedOp op,
                           hipStream_t const *streams) = 0;
    // recv[r][p * count + i] = send[p][i] on every rank r.
    virtual int all_gather(float *const *send, float *const *recv, size_t count,
                           hipStream_t const *streams) = 0;
    // recv_root[i] = sum_p send[p][i] on rank 0 only (attention-mpi.c:380).
    virtual int reduce_sum_to_root(float *const *send, float *recv_root, size_t count,
                                   hipStream_t const *streams) = 0;
    // recv[r][i] = sum_p send[p][r * count + i], i < count, on every rank r: rank r keeps the r-th of P equal
    // shares of the sum (send buffers hold P * count elements).  The parallel form of the reference's
    // reduce to the root: every rank then sends ITS rows home.
    virtual int reduce_scatter_sum(float *const *send, float *const *recv, size_t count,
                                   hipStream_t const *streams) = 0;
    // last error text of the underlying library ("" when none)
    virtual const char *last_error() const { return ""; }
    // ranks whose transport passed the known-answer self-test when the object was created (0 = none was run:
    // loopback ranks compute their collectives with this library's own kernels)
    virtual int selftested_ranks() const { return 0; }

protected:
    int P_ = 0;
};

// devs[r] = HIP device ordinal of rank r.  Returns nullptr (message on stderr) on failure.  *hung (optional) is
// set when the self-test did not FINISH within its deadline ($SDPA_RCCL_SELFTEST_TIMEOUT_S, default 60): kernels
// of the transport may then still sit on the devices, and the caller must not fall back to computing on them.
Collectives *make_rccl_collectives(int P, const int *devs, bool *hung = nullptr);
Collectives *make_loopback_collectives(int P, int dev);

}  // namespace sdpa
