// sdpa_rccl_abi_check.cpp -- compile-only: every hand-declared piece of sdpa_rccl_abi.h against the real header.
// Never linked into the library.  `make` compiles it with -fsyntax-only where <rccl/rccl.h> exists.
#include <rccl/rccl.h>

#include <type_traits>

#include "sdpa_rccl_abi.h"

namespace {
using namespace sdpa::rccl_abi;

static_assert((int)ncclSuccess == kNcclSuccess, "ncclSuccess");
static_assert((int)ncclFloat32 == kNcclFloat && (int)ncclFloat == kNcclFloat, "ncclFloat32");
static_assert((int)ncclSum == kNcclSum && (int)ncclMax == kNcclMax, "ncclRedOp_t values");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int),
              "RCCL's enums are int-sized: they travel as int through the hand-declared pointers");
static_assert(sizeof(::ncclComm_t) == sizeof(sdpa::rccl_abi::ncclComm_t) && std::is_pointer<::ncclComm_t>::value, "ncclComm_t is a pointer");

// the calling-convention class of a parameter or result: how it travels in registers
template <class T, class = void> struct abi_class { using type = T; };
template <class T> struct abi_class<T, std::enable_if_t<std::is_enum<T>::value>> {
    static_assert(sizeof(T) == 4, "enum wider than int");
    using type = int;
};
template <class T> struct abi_class<T *, void> { using type = void *; };              // any data pointer (const or not, any pointee)
template <class T> using abi_t = typename abi_class<std::remove_cv_t<T>>::type;

template <class R, class D> struct same_abi : std::false_type {};
template <class R1, class... A1, class R2, class... A2>
struct same_abi<R1 (*)(A1...), R2 (*)(A2...)>
    : std::integral_constant<bool, sizeof...(A1) == sizeof...(A2) && std::is_same<abi_t<R1>, abi_t<R2>>::value &&
                                       std::is_same<void(abi_t<A1>...), void(abi_t<A2>...)>::value> {};

#define SDPA_CHECK_FN(real, declared) \
    static_assert(same_abi<decltype(&real), declared>::value, #real " does not match " #declared " (sdpa_rccl_abi.h)")
SDPA_CHECK_FN(ncclCommInitAll, CommInitAll_t);
SDPA_CHECK_FN(ncclCommDestroy, CommDestroy_t);
SDPA_CHECK_FN(ncclGroupStart, GroupStart_t);
SDPA_CHECK_FN(ncclGroupEnd, GroupEnd_t);
SDPA_CHECK_FN(ncclAllReduce, AllReduce_t);
SDPA_CHECK_FN(ncclAllGather, AllGather_t);
SDPA_CHECK_FN(ncclReduce, Reduce_t);
SDPA_CHECK_FN(ncclReduceScatter, ReduceScatter_t);
SDPA_CHECK_FN(ncclGetErrorString, GetErrorString_t);

// and the check can fail: a signature with one argument too few, or a size_t where an int belongs, is refused
static_assert(!same_abi<decltype(&ncclAllGather), AllReduce_t>::value, "arity is checked");
static_assert(!same_abi<decltype(&ncclReduce), int (*)(const void *, void *, size_t, int, int, size_t, ::ncclComm_t, hipStream_t)>::value,
              "argument classes are checked");
}  // namespace
