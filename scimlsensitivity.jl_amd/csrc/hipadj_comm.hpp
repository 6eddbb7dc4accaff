// hipadj_comm.hpp — the one exchange of the sharded path: all-reduce(sum) of dL/dp over RCCL / xGMI.
//
// The ensemble shards over independent trajectories (the reference's EnsembleDistributed, test/Core4/distributed.jl:
// every worker solves its trajectories, the outer loss sums them), one process per GPU, one handle per process.  A handle
// that carries a communicator all-reduces its dp[np] in-stream at the end of every hipadj_adjoint(_dev) call — no host
// synchronisation, no second stream.  The communicator is either created here from a 128-byte unique id that the host
// ships to the other ranks by its own means (Julia Distributed, MPI, torch.distributed, a file), or an existing
// ncclComm_t handed in by the host (hipadj_comm_attach: not owned).
//
// RCCL is bound with dlopen at first use, like hiprtc (hipadj_user.hpp): a process that has torch loaded already carries
// an RCCL (torch/lib/librccl.so, soname librccl.so.1) and binding by soname reuses it instead of mixing two copies; a
// process without communicators never loads it.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

namespace hipadj {

// the subset of rccl.h this file needs (ABI of NCCL 2.x / RCCL: rccl/rccl.h:40-43, 187, 220, 260, 339, 448, 467, 611)
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
constexpr int RCCL_SUM = 0, RCCL_DOUBLE = 8;

struct RcclApi {
    void* lib = nullptr;
    std::string err;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*CommCount)(RcclComm, int*) = nullptr;
    int (*CommUserRank)(RcclComm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

inline RcclApi& rccl_api() {
    static RcclApi A;
    static std::once_flag once;
    std::call_once(once, [] {
        // an already loaded copy first (torch's), then the ROCm installation
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) { A.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (A.lib) break; }
        if (!A.lib) for (const char* nm : names) { A.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (A.lib) break; }
        if (!A.lib) { A.err = "RCCL not found (librccl.so): multi-GPU dL/dp all-reduce is unavailable"; return; }
        auto S = [&](const char* s) { void* p = dlsym(A.lib, s); if (!p && A.err.empty()) A.err = std::string("RCCL symbol missing: ") + s; return p; };
        A.GetUniqueId = (decltype(A.GetUniqueId))S("ncclGetUniqueId");
        A.CommInitRank = (decltype(A.CommInitRank))S("ncclCommInitRank");
        A.CommDestroy = (decltype(A.CommDestroy))S("ncclCommDestroy");
        A.CommCount = (decltype(A.CommCount))S("ncclCommCount");
        A.CommUserRank = (decltype(A.CommUserRank))S("ncclCommUserRank");
        A.AllReduce = (decltype(A.AllReduce))S("ncclAllReduce");
        A.GetErrorString = (decltype(A.GetErrorString))S("ncclGetErrorString");
    });
    return A;
}

inline std::string rccl_error(const char* what, int rc) {
    RcclApi& A = rccl_api();
    return std::string(what) + " failed: " + ((A.GetErrorString && rc) ? A.GetErrorString(rc) : "RCCL unavailable");
}

}  // namespace hipadj
