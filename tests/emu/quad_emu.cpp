// tests/emu/quad_emu.cpp — TEST-ONLY host emulation of the four-lanes-per-trajectory bodies (csrc/hipadj_quad_ts5.hpp).
//
// NOT a product path and NOT a CPU fallback: libhipadj never contains it, the package never loads it.  The quad bodies are SPMD code whose lanes exchange
// operands through DPP quad_perm moves; here FOUR HOST THREADS run the four lanes of one quad in lockstep and meet in hipadj_quad_emu_exchange (every
// lane publishes its value, a barrier, every lane reads the lane the control word selects, a barrier).  A quad whose lanes stop calling quad_perm the same
// number of times — the device's lanes cannot: they share a program counter — deadlocks here and is reported after a time-out.
// Compiled with g++ -ffp-contract=off (fma() calls stay fused, as written); -DHIPADJ_QUAD_GAUSS_NZ=1 builds the one-component Gauss instantiation that
// returned wrong numbers on the device (hipadj_quad_ts5.hpp QuadNZ).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#define HIPADJ_QUAD_EMU 1
#include "../../scimlsensitivity.jl_amd/csrc/hipadj_plan.hpp"
#include "../../scimlsensitivity.jl_amd/csrc/hipadj_quad_ts5.hpp"

using namespace hipadj;

namespace {
struct QuadCtx {
    double slot[4];
    std::atomic<int> arrived{0};
    std::atomic<int> generation{0};
    std::atomic<bool> failed{false};
    int lanes = 4;
    void barrier() {
        const int gen = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == lanes) { arrived.store(0, std::memory_order_relaxed); generation.fetch_add(1, std::memory_order_release); return; }
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        while (generation.load(std::memory_order_acquire) == gen) {
            if (failed.load(std::memory_order_relaxed)) return;
            if (++spins > 64) {                                 // a short spin, then give the core away: four threads must make progress on ANY number of cores
                std::this_thread::yield();
                if ((spins & 0x3FF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { failed.store(true); return; }   // the lanes of the quad diverged
            }
        }
    }
};
thread_local QuadCtx* t_ctx = nullptr;
thread_local int t_lane = 0;
std::string g_err;
}  // namespace

namespace hipadj {
double hipadj_quad_emu_exchange(int ctrl, double x) {
    QuadCtx* q = t_ctx;
    q->slot[t_lane] = x;
    q->barrier();
    const double r = q->slot[(ctrl >> (2 * t_lane)) & 3];
    q->barrier();
    return r;
}
}  // namespace hipadj

template <class F> static bool run_quad(F&& body, int lanes = 4) {     // body(c) on `lanes` threads in lockstep; false when the lanes diverged
    QuadCtx ctx; ctx.lanes = lanes;
    std::thread th[4];
    for (int c = 0; c < lanes; ++c) th[c] = std::thread([&, c] { t_ctx = &ctx; t_lane = c; body(c); });
    for (int c = 0; c < lanes; ++c) th[c].join();
    return !ctx.failed.load();
}

extern "C" const char* quad_emu_last_error() { return g_err.c_str(); }
extern "C" int quad_emu_gauss_nz() { return QuadNZ<2>::value; }

// adaptive Tsit5 of a model with a component form: forward_tsit5_quad + adjoint_tsit5_quad<ALG> exactly as k_forward_tsit5_quad / k_adjoint_tsit5_quad run them, one quad at a time
template <class Mo, int ALG>
static int run_ts5(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu, double* du0, double* dp, double* out, int* nsteps_out) {
    constexpr int N = Mo::N, NP = Mo::NP, RW = 2 + 5 * N;
    AdaptGeom g; g.N = P.N; g.Npad = P.Npad; g.M = P.M; g.Smax = P.Smax; g.maxit = P.Smax; g.nck = P.nck; g.t0 = cfg->t0; g.t1 = cfg->t1; g.dt0 = cfg->dt;
    g.abstol = cfg->abstol; g.reltol = cfg->reltol; g.loss_shift = cfg->loss_shift; g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start;
    { const double lw = cfg->loss_scale != 0.0 ? cfg->loss_scale : 1.0; g.la = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? lw : 0.0; g.lb = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? -lw : 1.0; g.lflags = (cfg->reference_literal && (cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD)) ? 3 : 0; }
    g.p_shared = cfg->p_shared; g.cont_cost = cfg->cont_cost; g.SmaxI = P.SmaxI; g.SmaxA = 0;
    const long Np = P.Npad;
    std::vector<double> rec(ALG != 1 ? (size_t)P.Smax * RW * Np : 0), outT((size_t)P.M * N * Np), yT((size_t)N * Np), ckpt((size_t)P.nck * N * Np);
    std::vector<double> cotT(cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT ? (size_t)P.M * N * Np : 0), dp_traj((size_t)NP * Np, 0.0);
    std::vector<int> nsteps((size_t)Np, 0);
    int flag = 0;
    for (long i = 0; i < P.N; ++i) {
        const bool ok = run_quad([&](int c) {
            forward_tsit5_quad<Mo>(g, i, c, u0, p, rec.empty() ? nullptr : rec.data(), nsteps.data(), P.save_times.data(), outT.data(), P.ck_times.data(),
                                   ckpt.empty() ? nullptr : ckpt.data(), yT.data(), &flag); });
        if (!ok) { g_err = "forward_tsit5_quad: the lanes of a quad diverged"; return HIPADJ_ERR_HIP; }
        if (nsteps[i] > P.Smax) { g_err = "forward solve exceeded max_steps"; return HIPADJ_ERR_MAXITERS; }
    }
    if (nsteps_out) for (long i = 0; i < P.N; ++i) nsteps_out[i] = nsteps[i];
    if (out) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) out[i * P.M * N + c] = outT[(size_t)c * Np + i];
    if (!cotT.empty()) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) cotT[(size_t)c * Np + i] = dLdu[i * P.M * N + c];
    for (long i = 0; i < P.N; ++i) {
        const bool ok = run_quad([&](int c) {
            adjoint_tsit5_quad<Mo, ALG>(g, i, c, p, rec.data(), nsteps.data(), yT.data(), ckpt.empty() ? nullptr : ckpt.data(), P.ck_times.data(), P.save_times.data(),
                                        P.tstops_desc.data(), (int)P.tstops_desc.size(), cotT.empty() ? nullptr : cotT.data(), du0, dp_traj.data(), &flag); });
        if (!ok) { g_err = "adjoint_tsit5_quad: the lanes of a quad diverged"; return HIPADJ_ERR_HIP; }
    }
    if (cfg->p_shared) { for (int j = 0; j < NP; ++j) { double s = 0; for (long i = 0; i < P.N; ++i) s += dp_traj[(size_t)j * Np + i]; dp[j] = s; } }
    else for (long i = 0; i < P.N; ++i) for (int j = 0; j < NP; ++j) dp[i * NP + j] = dp_traj[(size_t)j * Np + i];
    return HIPADJ_OK;
}

template <class Mo>
static int dispatch_ts5(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu, double* du0, double* dp, double* out, int* nsteps) {
    switch (cfg->alg) {
    case HIPADJ_ALG_INTERPOLATING: return run_ts5<Mo, 0>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    case HIPADJ_ALG_BACKSOLVE: return run_ts5<Mo, 1>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    case HIPADJ_ALG_GAUSS: return run_ts5<Mo, 2>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    default: g_err = "quad emulator: Interpolating, Backsolve, Gauss"; return HIPADJ_ERR_UNSUPPORTED;
    }
}
extern "C" int quad_emu_forward_adjoint(const hipadj_config* cfg, const double* u0, const double* p, const double* dLdu, double* du0, double* dp, double* out, int* nsteps) {
    if (cfg->stepper != HIPADJ_STEPPER_TSIT5_ADAPTIVE || cfg->cont_cost != 0) { g_err = "quad emulator: adaptive Tsit5 without a cost"; return HIPADJ_ERR_UNSUPPORTED; }
    Plan P; const int rc = make_plan(cfg, P, g_err); if (rc) return rc;
    if (P.ip_ckpt) { g_err = "quad emulator: no checkpointing = true"; return HIPADJ_ERR_UNSUPPORTED; }
    switch (cfg->model) {
    case HIPADJ_MODEL_LORENZ: return dispatch_ts5<ModelLorenz>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    case HIPADJ_MODEL_LV: return dispatch_ts5<ModelLV>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    case HIPADJ_MODEL_LVT: return dispatch_ts5<ModelLVT>(cfg, P, u0, p, dLdu, du0, dp, out, nsteps);
    default: g_err = "quad emulator: lorenz, lv, lvt"; return HIPADJ_ERR_UNSUPPORTED;
    }
}

// Lorenz, fixed-step RK4 forward solve: forward_quad_ev as k_forward_quad runs it (the lanes c >= n of a quad leave at once: three threads).  knots_out [N][S + 1][2][3]
// = (u_k, f(u_k)) per knot, out [N][M][3] = sol(ts) on the grid, yT [N][3].
extern "C" int quad_emu_forward_rk4(const hipadj_config* cfg, const double* u0, const double* p, double* knots_out, double* out, double* yT_out) {
    using Mo = ModelLorenz;
    constexpr int N = Mo::N;
    if (cfg->model != HIPADJ_MODEL_LORENZ || cfg->stepper != HIPADJ_STEPPER_RK4_FIXED) { g_err = "quad emulator: Lorenz on fixed-step RK4"; return HIPADJ_ERR_UNSUPPORTED; }
    Plan P; const int rc = make_plan(cfg, P, g_err); if (rc) return rc;
    if (P.offgrid) { g_err = "quad emulator: loss times on the step grid"; return HIPADJ_ERR_UNSUPPORTED; }
    Geom g; g.N = P.N; g.Npad = P.Npad; g.S = P.S; g.M = P.M; g.t0 = cfg->t0; g.dt = cfg->dt; g.loss_shift = cfg->loss_shift;
    g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start; g.p_shared = cfg->p_shared; g.kmask = -1; g.h_last = P.h_last;
    { const double lw = cfg->loss_scale != 0.0 ? cfg->loss_scale : 1.0; g.la = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? lw : 0.0; g.lb = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? -lw : 1.0; g.lflags = (cfg->reference_literal && (cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD)) ? 3 : 0; }   // hipadj_create's rule (csrc/hipadj_api.hip)
    const long Np = P.Npad;
    std::vector<int> ek, es, ec;
    forward_events(P, cfg->dt, ek, es, ec);
    const int nev = (int)ek.size();
    ek.push_back(0); es.push_back(-1); ec.push_back(-1);
    std::vector<dbl2> knots((size_t)(P.S + 1) * N * Np);
    std::vector<double> outT((size_t)(P.M > 0 ? P.M : 1) * N * Np), yT((size_t)N * Np);
    const FwdEvents ev{ek.data(), es.data(), ec.data(), nev};
    for (long i = 0; i < P.N; ++i) {
        const bool ok = run_quad([&](int c) { forward_quad_ev<Mo>(g, i, c, u0, p, ev, knots.data(), nullptr, outT.data(), yT.data()); }, N);
        if (!ok) { g_err = "forward_quad_ev: the lanes of a quad diverged"; return HIPADJ_ERR_HIP; }
    }
    for (long i = 0; i < P.N; ++i) {
        for (int k = 0; k <= P.S; ++k) for (int j = 0; j < N; ++j) {
            const dbl2 d = knots[((size_t)k * N + j) * Np + i];
            knots_out[((i * (P.S + 1) + k) * 2 + 0) * N + j] = d.x; knots_out[((i * (P.S + 1) + k) * 2 + 1) * N + j] = d.y; }
        for (int c = 0; c < P.M * N; ++c) out[i * P.M * N + c] = outT[(size_t)c * Np + i];
        for (int j = 0; j < N; ++j) yT_out[i * N + j] = yT[(size_t)j * Np + i];
    }
    return HIPADJ_OK;
}
