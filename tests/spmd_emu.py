"""TEST-ONLY host emulation of the SPMD bodies of wide models (f and the joint VJP of `hipadj_wmodel_register`) with T COOPERATING threads.

The one-thread harness of tests/test_wtrace.py (HIPADJ_W_FOR = a plain loop, wg_sum(x) = x) checks the arithmetic of an emitted body; it cannot see what only exists between
threads: a missing wg_sync() between a producer and a consumer phase, a gradient entry written by two threads, an entry of du / dlam nobody owns, a wg_sum not reached by
every thread.  Here the body runs on T host threads of which exactly ONE is runnable at a time: a thread runs until it reaches wg_sync() / wg_sum() / wg_sum2() (or the end),
then hands the baton to the next thread of a fixed ORDER; when all have arrived the next phase starts.  With the order 0, 1, ... a consumer with a smaller id than its
producer reads stale data if the barrier between them is missing; the reversed order catches the other direction — deterministically, no timing involved.  Threads that
reach a different NUMBER of collective calls dead-lock the device; here the run reports it.  Never imported by the product package."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile
import numpy as np

RUNTIME = r'''
#include <cmath>
#include <cstring>
#include <semaphore>
#include <thread>
#include <vector>
#include <memory>
namespace emu {
struct Sched {
    int T = 1, reverse = 0, arrived = 0, failed = 0;
    std::vector<std::unique_ptr<std::binary_semaphore>> go;
    std::vector<double> slot, slot2;
    std::vector<int> done;
    int order(int k) const { return reverse ? T - 1 - k : k; }
    int rank(int tid) const { return reverse ? T - 1 - tid : tid; }
    void init(int t, int rev) { T = t; reverse = rev; arrived = 0; failed = 0; go.clear(); for (int i = 0; i < T; ++i) go.emplace_back(new std::binary_semaphore(0)); slot.assign(T, 0.0); slot2.assign(T, 0.0); done.assign(T, 0); }
    // the calling thread has finished a phase: pass the baton; returns when its next phase may start
    void barrier(int tid) {
        ++arrived;
        if (arrived == T) { arrived = 0; next_runnable(-1); }      // last arriver: the next phase starts with the first thread of the order
        else next_runnable(rank(tid));
        go[tid]->acquire();
    }
    void next_runnable(int after_rank) {                            // wake the next thread of the order that has not finished the kernel
        for (int k = after_rank + 1; k < T; ++k) { const int t = order(k); if (!done[t]) { go[t]->release(); return; } }
        // nobody left in this phase who is still alive: threads that returned early while others wait in a collective = the device would hang
        failed = 1;
        for (int k = 0; k < T; ++k) go[order(k)]->release();
    }
    void finish(int tid) {                                          // the body returned
        done[tid] = 1;
        int alive = 0; for (int t = 0; t < T; ++t) alive += !done[t];
        if (alive == 0) return;
        if (arrived == alive) { arrived = 0; next_runnable(-1); }   // everybody else is waiting in a collective this thread never reached
        else next_runnable(rank(tid));
    }
};
static Sched S;
static thread_local int TID = 0;
inline void wg_sync_() { S.barrier(TID); }
inline double wg_sum_(double x) {
    S.slot[TID] = x; S.barrier(TID);
    double s = 0.0; for (int t = 0; t < S.T; ++t) s += S.slot[t];
    S.barrier(TID);
    return s;
}
inline void wg_sum2_(double a, double b, double& sa, double& sb) {
    S.slot[TID] = a; S.slot2[TID] = b; S.barrier(TID);
    double x = 0.0, y = 0.0; for (int t = 0; t < S.T; ++t) { x += S.slot[t]; y += S.slot2[t]; }
    S.barrier(TID);
    sa = x; sb = y;
}
template <class F> int run(int T, int reverse, F&& body) {
    S.init(T, reverse);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { TID = t; S.go[t]->acquire(); if (!S.failed) body(t); S.finish(t); });
    S.go[S.order(0)]->release();
    for (auto& x : th) x.join();
    return S.failed;
}
}  // namespace emu
static int T = 1;
#define HIPADJ_W_FOR(i, n) for (int i = tid; i < (n); i += T)
#define wg_sync() emu::wg_sync_()
#define wg_sum(x) emu::wg_sum_(x)
#define wg_sum2(a, b, sa, sb) emu::wg_sum2_(a, b, sa, sb)
static const int N = %(n)d, NP = %(np)d;
static void model_f_t(double* __restrict__ du, const double* __restrict__ u, const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {
    (void)u; (void)p; (void)t; (void)ws; (void)tid;
%(f)s
}
template <bool WP> static void model_vjp_t(double* __restrict__ dlam, double* __restrict__ gp, double (&acc)[%(na)d], double w, const double* __restrict__ lam, const double* __restrict__ u,
                                           const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {
    (void)gp; (void)acc; (void)w; (void)u; (void)p; (void)t; (void)ws; (void)tid;
%(vjp)s
}
template <bool WP> static void model_cost_t(double* __restrict__ dlam, double* __restrict__ gp, double (&acc)[%(na)d], double w, const double* __restrict__ u,
                                            const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {
    (void)dlam; (void)gp; (void)acc; (void)w; (void)u; (void)p; (void)t; (void)ws; (void)tid;
%(cost)s
}
extern "C" int spmd_cost(int threads, int reverse, int wp, double* dlam, double* gp, double* acc_out, double w, const double* u, const double* p, double t, double* ws) {
    T = threads;
    return emu::run(threads, reverse, [&](int tid) {
        double a[%(na)d] = {0};
        if (wp) model_cost_t<true>(dlam, gp, a, w, u, p, t, ws, tid); else model_cost_t<false>(dlam, gp, a, w, u, p, t, ws, tid);
        for (int q = 0; q < %(na)d; ++q) acc_out[tid * %(na)d + q] = a[q];
    });
}
extern "C" int spmd_f(int threads, int reverse, double* du, const double* u, const double* p, double t, double* ws) {
    T = threads;
    return emu::run(threads, reverse, [&](int tid) { model_f_t(du, u, p, t, ws, tid); });
}
extern "C" int spmd_vjp(int threads, int reverse, int wp, double* dlam, double* gp, double* acc_out /* [T][na] */, double w, const double* lam, const double* u, const double* p, double t, double* ws) {
    T = threads;
    return emu::run(threads, reverse, [&](int tid) {
        double a[%(na)d] = {0};
        if (wp) model_vjp_t<true>(dlam, gp, a, w, lam, u, p, t, ws, tid); else model_vjp_t<false>(dlam, gp, a, w, lam, u, p, t, ws, tid);
        for (int q = 0; q < %(na)d; ++q) acc_out[tid * %(na)d + q] = a[q];
    });
}
'''

_cache = {}


class SpmdModel:
    """f / vjp of one pair of bodies under T cooperating threads (threads, reverse order or not)."""

    def __init__(self, f_body, vjp_body, n, npar, lds_doubles=0, nacc=0, acc_first=0, cost_body=""):
        self.n, self.np, self.nw, self.nacc, self.a0 = int(n), int(npar), max(int(lds_doubles), 1), int(nacc), int(acc_first)
        src = RUNTIME % dict(f=f_body, vjp=vjp_body, cost=cost_body, n=self.n, np=self.np, na=max(self.nacc, 1))
        key = hashlib.sha1(src.encode()).hexdigest()
        if key not in _cache:
            d = tempfile.mkdtemp(prefix="spmd_emu_")
            path = os.path.join(d, "m.cpp")
            open(path, "w").write(src)
            so = os.path.join(d, "m.so")
            subprocess.check_call(["g++", "-O1", "-std=c++20", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", "-o", so, path])
            _cache[key] = C.CDLL(so)
        self.L = _cache[key]

    def f(self, u, p, t, threads=64, reverse=False):
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        u, p = np.ascontiguousarray(u, dtype=np.float64), np.ascontiguousarray(p, dtype=np.float64)
        du, ws = np.full(self.n, np.nan), np.zeros(self.nw)
        if self.L.spmd_f(int(threads), int(reverse), P(du), P(u), P(p), C.c_double(t), P(ws)):
            raise RuntimeError("the threads of the workgroup did not reach the same collective calls (the device would hang)")
        return du

    def vjp(self, lam, u, p, t, w=1.0, wp=True, threads=64, reverse=False):
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        lam, u, p = (np.ascontiguousarray(a, dtype=np.float64) for a in (lam, u, p))
        dlam, gp, ws = np.full(self.n, np.nan), np.zeros(self.np), np.zeros(self.nw)
        acc = np.zeros((int(threads), max(self.nacc, 1)))
        if self.L.spmd_vjp(int(threads), int(reverse), int(wp), P(dlam), P(gp), P(acc), C.c_double(w), P(lam), P(u), P(p), C.c_double(t), P(ws)):
            raise RuntimeError("the threads of the workgroup did not reach the same collective calls (the device would hang)")
        for q in range(self.nacc):
            gp[self.a0 + q] += acc[:, q].sum()
        return dlam, gp

    def cost(self, u, p, t, w=1.0, wp=True, threads=64, reverse=False, dlam0=None):
        """The cost body ADDS dg/du into dlam (dlam0, default zeros) and w dg/dp into the gradient: returns (dlam, gp)."""
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        u, p = np.ascontiguousarray(u, dtype=np.float64), np.ascontiguousarray(p, dtype=np.float64)
        dlam = np.zeros(self.n) if dlam0 is None else np.array(dlam0, dtype=np.float64)
        gp, ws = np.zeros(self.np), np.zeros(self.nw)
        acc = np.zeros((int(threads), max(self.nacc, 1)))
        if self.L.spmd_cost(int(threads), int(reverse), int(wp), P(dlam), P(gp), P(acc), C.c_double(w), P(u), P(p), C.c_double(t), P(ws)):
            raise RuntimeError("the threads of the workgroup did not reach the same collective calls (the device would hang)")
        for q in range(self.nacc):
            gp[self.a0 + q] += acc[:, q].sum()
        return dlam, gp
