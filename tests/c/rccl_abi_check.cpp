// rccl_abi_check.cpp — compile-time comparison of the RCCL ABI subset that csrc/hipadj_comm.hpp declares by hand (the library binds
// RCCL with dlopen and does not include rccl.h) with the toolkit's own <rccl/rccl.h>.  Compiled (not run) by tests/test_abi_and_host.py:
//   hipcc -fsyntax-only -I scimlsensitivity.jl_amd/csrc tests/c/rccl_abi_check.cpp
// A changed enumerator, id size or signature fails the build here instead of all-reducing garbage on an 8-GPU node.
#include <rccl/rccl.h>
#include <type_traits>
#include "hipadj_comm.hpp"

using namespace hipadj;
static_assert(RCCL_DOUBLE == (int)ncclDouble && RCCL_DOUBLE == (int)ncclFloat64, "ncclDouble");
static_assert(RCCL_SUM == (int)ncclSum, "ncclSum");
static_assert(sizeof(RcclUniqueId) == sizeof(ncclUniqueId) && sizeof(RcclUniqueId) == NCCL_UNIQUE_ID_BYTES && alignof(RcclUniqueId) == alignof(ncclUniqueId), "ncclUniqueId");
static_assert(std::is_standard_layout<ncclUniqueId>::value && std::is_trivially_copyable<ncclUniqueId>::value, "ncclUniqueId is passed by value as plain bytes");
static_assert(sizeof(ncclResult_t) == sizeof(int) && (int)ncclSuccess == 0, "ncclResult_t");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "enumerators travel as int");
static_assert(std::is_same<ncclComm_t, struct ncclComm*>::value && sizeof(RcclComm) == sizeof(ncclComm_t), "ncclComm_t is a pointer");
// signatures: same arity, and every parameter has the size / kind the hand-written pointer type assumes
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId*)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclCommCount), ncclResult_t (*)(const ncclComm_t, int*)>::value, "ncclCommCount");
static_assert(std::is_same<decltype(&ncclCommUserRank), ncclResult_t (*)(const ncclComm_t, int*)>::value, "ncclCommUserRank");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value, "ncclAllReduce");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char* (*)(ncclResult_t)>::value, "ncclGetErrorString");
int main() { return 0; }
