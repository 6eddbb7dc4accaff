// hipadj_api.hip — the C-ABI shared library (include/hipadj.h) over the gfx950 kernel family.
// Host side only does validation, workspace ownership, launch sequencing and timing; all arithmetic is in
// hipadj_lane.hpp / hipadj_kernels.hpp.  No CPU fallback exists: without a usable HIP device
// hipadj_create returns HIPADJ_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hipadj.h"
#include "hipadj_kernels.hpp"
#include "hipadj_field.hpp"
#include "hipadj_mlp.hpp"
#include "hipadj_adaptive.hpp"
#include "hipadj_plan.hpp"
#include "hipadj_user.hpp"

using namespace hipadj;

static constexpr int HIPADJ_AUTO_MAXITERS = 100000;   // max_steps == 0: the reference's default maxiters

struct hipadj_handle {
    hipadj_config cfg{};
    int n = 0, np = 0;
    long N = 0, Npad = 0;
    int S = 0, M = 0, nck = 0, nseg = 1, nq = 0;
    Geom g{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};            // forward begin/end
    // adjoint timing: a ring of event sets harvested with hipEventQuery, so that back-to-back asynchronous
    // calls never block the host on the previous call (a blocking harvest serialises launch and execution)
    static constexpr int NSET = 16;
    struct EvSet { hipEvent_t a0 = nullptr, a1 = nullptr, k0 = nullptr, k1 = nullptr; bool pending = false, full = true; } evs[NSET];
    int ev_next = 0;
    std::vector<double> save_times;
    std::vector<int> save_of_knot, ckpt_of_knot, seg_bounds;
    // device workspaces (owned)
    double *d_u0 = nullptr, *d_p = nullptr, *d_outT = nullptr, *d_yT = nullptr, *d_ckpt = nullptr, *d_cotT = nullptr;
    double *d_segbuf = nullptr, *d_dp_traj = nullptr, *d_qres = nullptr, *d_qa = nullptr, *d_qb = nullptr, *d_partial = nullptr;
    double *d_io_a = nullptr, *d_du0 = nullptr, *d_dp = nullptr;   // staging for the host-pointer API
    dbl2 *d_knots = nullptr, *d_adj = nullptr;
    bool field = false;                   // workgroup-per-trajectory family (BRUSS)
    bool ip_ckpt = false;                 // Interpolating/Gauss checkpointing=true
    double *d_fknots = nullptr, *d_fadj = nullptr;
    bool mlp = false; int NQ = 0, ksplit = 1;
    double *d_w2t = nullptr, *d_ax = nullptr, *d_al = nullptr, *d_ah1 = nullptr, *d_ah2 = nullptr, *d_ag1 = nullptr, *d_ag2 = nullptr;
    double *d_c1 = nullptr, *d_c2 = nullptr, *d_c3 = nullptr;
    MlpGeom mg{};
    FieldGeom fg{};
    bool user = false;                    // runtime-compiled right-hand side (hipadj_user.hpp)
    hipModule_t umod = nullptr;
    hipFunction_t uf_forward = nullptr, uf_main = nullptr, uf_tail = nullptr, uf_gk = nullptr;   // tail = k_compose_finish or k_finish
    bool adaptive = false;                // adaptive Tsit5 (hipadj_adaptive.hpp)
    AdaptGeom ag{};
    int cbs = 0;         // k_compose_finish workgroup size: 0 = by ensemble size, 64 / 256 forced (HIPADJ_CBS; tuning study)
    bool wpb4 = false;   // k_interp in 256-thread workgroups (HIPADJ_WPB=4; tuning study)
    double *d_rec = nullptr, *d_save_t = nullptr, *d_ck_t = nullptr, *d_tstops = nullptr, *d_arec = nullptr;
    int *d_nsteps = nullptr, *d_nsteps_adj = nullptr, ntstops = 0, SmaxA = 0;
    bool auto_steps = false;              // max_steps == 0: record capacity sized from a counting pass of the forward solve
    long rec_cap = 0;                     // accepted steps the record buffer(s) currently hold per trajectory
    unsigned* d_ticket = nullptr;
    int *d_prev_ck = nullptr, *d_save_of_knot = nullptr, *d_ckpt_of_knot = nullptr, *d_seg_bounds = nullptr, *d_flag = nullptr;
    const double* p_dev_last = nullptr;  // device p used by the last forward (the adjoint reuses it)
    bool have_forward = false, timing_pending_fwd = false;
    int fused_final = 0;                  // 1: dp reduced in-launch by the last-arriving workgroup (HIPADJ_FUSED_FINAL)
    int timing = 2;                       // 0: no events, 1: dominant-kernel bracket only, 2: + whole-call bracket (HIPADJ_TIMING)
    double ws_bytes = 0;
    hipadj_stats st{};
    std::string err;
};

static thread_local std::string g_create_error;
static int user_prepare(hipadj_handle* h);   // hiprtc compilation of the kernels of a runtime-registered model
static int adaptive_autosize(hipadj_handle* h);   // record capacity from the counting pass (max_steps == 0)

extern "C" int hipadj_version(void) { return HIPADJ_VERSION; }

extern "C" const char* hipadj_status_string(int s) {
    switch (s) {
    case HIPADJ_OK: return "ok";
    case HIPADJ_ERR_INVALID_ARG: return "invalid argument";
    case HIPADJ_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
    case HIPADJ_ERR_HIP: return "HIP runtime error";
    case HIPADJ_ERR_NONFINITE: return "non-finite value in a trajectory's sensitivities";
    case HIPADJ_ERR_STATE: return "invalid call order (forward solve required first)";
    case HIPADJ_ERR_MAXITERS: return "adaptive solve exceeded max_steps";
    case HIPADJ_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
    }
}

extern "C" const char* hipadj_last_error(const hipadj_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int hipadj_model_sizes(int32_t model, const int32_t dims[4], int32_t* n, int32_t* np) {
    if (!n || !np) return HIPADJ_ERR_INVALID_ARG;
    return plan_model_sizes(model, dims, n, np);
}

#define HIPADJ_FAIL(h, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (h)->err = _b; return (code); } while (0)
#define HIP_TRY(h, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    HIPADJ_FAIL(h, HIPADJ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

template <class T> static int dev_alloc(hipadj_handle* h, T** p, size_t count) {
    if (count == 0) { *p = nullptr; return HIPADJ_OK; }
    HIP_TRY(h, hipMalloc((void**)p, count * sizeof(T)));
    h->ws_bytes += (double)(count * sizeof(T));
    return HIPADJ_OK;
}
extern "C" int hipadj_model_register(const char* name, int32_t n, int32_t np, const char* f_body, const char* vjp_u_body,
                                     const char* vjp_p_body, int32_t* model_id) {
    return user_register(name, n, np, f_body, vjp_u_body, vjp_p_body, model_id, g_create_error);
}

extern "C" int hipadj_model_set_cost(int32_t model_id, const char* dgdu_body, const char* dgdp_body) {
    return user_set_cost(model_id, dgdu_body, dgdp_body, g_create_error);
}

extern "C" int hipadj_model_set_cost_function(int32_t model_id, const char* g_body) {
    return user_set_cost_function(model_id, g_body, g_create_error);
}

extern "C" int hipadj_model_check(int32_t model_id) {
    std::vector<char> code; std::map<std::string, std::string> low;
    std::vector<std::string> exprs = {"hipadj::k_forward<hipadj::UserModel>", user_has_cost(model_id) ? "hipadj::k_interp<hipadj::UserModel, 1, 7>" : "hipadj::k_interp<hipadj::UserModel, 1, 1>"};
    if (const char* e = std::getenv("HIPADJ_CHECK_EXPRS")) {   // debugging hook: further ';'-separated kernel instantiations (ISA studies with HIPADJ_RTC_DUMP)
        std::string t(e); size_t a = 0;
        while (a <= t.size()) { const size_t b = t.find(';', a); const std::string x = t.substr(a, b == std::string::npos ? std::string::npos : b - a); if (!x.empty()) exprs.push_back(x); if (b == std::string::npos) break; a = b + 1; }
    }
    return user_compile(model_id, exprs, code, low, g_create_error);
}

// Every kernel a handle of this configuration would launch, compiled now (no device needed): the ahead-of-time form of what
// hipadj_create does lazily.  Built-in models have nothing to compile.
struct UserKernels;
static int user_compile_config(const hipadj_config* cfg, std::string& err);
extern "C" int hipadj_model_check_config(const hipadj_config* cfg) {
    if (!cfg) { g_create_error = "cfg == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->model < HIPADJ_MODEL_USER_BASE) return HIPADJ_OK;
    return user_compile_config(cfg, g_create_error);
}

#define TRY(expr) do { int _rc = (expr); if (_rc != HIPADJ_OK) return _rc; } while (0)

static void free_all(hipadj_handle* h) {
    void* ptrs[] = {h->d_u0, h->d_p, h->d_outT, h->d_yT, h->d_ckpt, h->d_cotT, h->d_segbuf, h->d_dp_traj, h->d_qres, h->d_qa,
                    h->d_qb, h->d_partial, h->d_io_a, h->d_du0, h->d_dp, h->d_knots, h->d_adj, h->d_fknots, h->d_fadj, h->d_w2t, h->d_ax, h->d_al, h->d_ah1, h->d_ah2, h->d_ag1, h->d_ag2, h->d_c1, h->d_c2, h->d_c3, h->d_ticket, h->d_prev_ck, h->d_save_of_knot,
                    h->d_ckpt_of_knot, h->d_seg_bounds, h->d_flag, h->d_rec, h->d_save_t, h->d_ck_t, h->d_tstops, h->d_nsteps, h->d_arec, h->d_nsteps_adj};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->umod) (void)hipModuleUnload(h->umod);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    for (auto& q : h->evs) for (hipEvent_t e : {q.a0, q.a1, q.k0, q.k1}) if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

extern "C" int hipadj_create(const hipadj_config* cfg, hipadj_handle** out) {
    if (!out) { g_create_error = "out == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!cfg) { g_create_error = "cfg == NULL"; return HIPADJ_ERR_INVALID_ARG; }
        auto* h = new hipadj_handle();
    auto fail = [&](int code) { g_create_error = h->err; free_all(h); delete h; return code; };
    h->cfg = *cfg; h->cfg.save_times = nullptr;
    Plan P;
    { const int prc = make_plan(cfg, P, h->err); if (prc != HIPADJ_OK) return fail(prc); }
    const int n = P.n, np = P.np; const long S = P.S;
    h->n = n; h->np = np; h->N = P.N; h->Npad = P.Npad; h->S = P.S; h->M = P.M; h->nck = P.nck; h->nseg = P.nseg; h->nq = P.nq;
    h->save_times = P.save_times; h->save_of_knot = P.save_of_knot; h->ckpt_of_knot = P.ckpt_of_knot; h->seg_bounds = P.seg_bounds;
    const bool bs_ckpt = P.bs_ckpt || P.ip_ckpt;
    h->ip_ckpt = P.ip_ckpt;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { h->err = hipadj_status_string(HIPADJ_ERR_NO_DEVICE); return fail(HIPADJ_ERR_NO_DEVICE); }
    if (cfg->device < 0 || cfg->device >= ndev) { h->err = "device ordinal out of range"; return fail(HIPADJ_ERR_INVALID_ARG); }
    auto HT = [&](hipError_t e, const char* what) { if (e != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; };
    if (!HT(hipSetDevice(cfg->device), "hipSetDevice")) return fail(HIPADJ_ERR_HIP);
    if (!HT(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking), "hipStreamCreate")) return fail(HIPADJ_ERR_HIP);
    h->stream = h->own_stream;
    for (auto& e : h->ev) if (!HT(hipEventCreate(&e), "hipEventCreate")) return fail(HIPADJ_ERR_HIP);
    for (auto& q : h->evs) for (hipEvent_t* e : {&q.a0, &q.a1, &q.k0, &q.k1}) if (!HT(hipEventCreate(e), "hipEventCreate")) return fail(HIPADJ_ERR_HIP);

    const long Np = h->Npad;
    int rc = HIPADJ_OK;
    auto A = [&](int r) { if (rc == HIPADJ_OK) rc = r; };
    A(dev_alloc(h, &h->d_u0, (size_t)h->N * n));
    A(dev_alloc(h, &h->d_p, cfg->p_shared ? (size_t)np : (size_t)h->N * np));
    h->field = P.field;
    h->adaptive = P.adaptive;
    if (P.adaptive) {
        const int RW = 2 + 5 * n;   // record width, hipadj_adaptive.hpp
        A(dev_alloc(h, &h->d_outT, (size_t)h->M * n * Np));
        A(dev_alloc(h, &h->d_yT, (size_t)n * Np));
        h->auto_steps = cfg->max_steps == 0;   // capacity follows the measured step counts (adaptive_autosize); else the caller's bound
        if (cfg->alg != HIPADJ_ALG_BACKSOLVE) {
            h->rec_cap = h->auto_steps ? 128 : (P.ip_ckpt ? P.SmaxI : P.Smax);   // checkpointing=true: one interval per lane
            A(dev_alloc(h, &h->d_rec, (size_t)h->rec_cap * RW * Np));
        }
        A(dev_alloc(h, &h->d_nsteps, (size_t)Np));
        if (cfg->alg == HIPADJ_ALG_QUADRATURE) {   // dense adjoint solution: the reverse solve also stops at every loss time
            A(dev_alloc(h, &h->d_nsteps_adj, (size_t)Np));
            h->SmaxA = 2 * (h->auto_steps ? (int)h->rec_cap : P.Smax) + h->M + 16;
            A(dev_alloc(h, &h->d_arec, (size_t)h->SmaxA * RW * Np));
        }
        if (P.nck > 0) { A(dev_alloc(h, &h->d_ckpt, (size_t)P.nck * n * Np)); A(dev_alloc(h, &h->d_ck_t, (size_t)P.nck)); }
        if (h->M > 0) A(dev_alloc(h, &h->d_save_t, (size_t)h->M));
        h->ntstops = (int)P.tstops_desc.size();
        if (h->ntstops > 0) A(dev_alloc(h, &h->d_tstops, (size_t)h->ntstops));
        if (cfg->loss_kind == HIPADJ_LOSS_COTANGENT) A(dev_alloc(h, &h->d_cotT, (size_t)h->M * n * Np));
        if (rc == HIPADJ_OK) {
            bool ok2 = true;
            if (h->M > 0) ok2 = ok2 && HT(hipMemcpy(h->d_save_t, P.save_times.data(), sizeof(double) * h->M, hipMemcpyHostToDevice), "memcpy");
            if (P.nck > 0) ok2 = ok2 && HT(hipMemcpy(h->d_ck_t, P.ck_times.data(), sizeof(double) * P.nck, hipMemcpyHostToDevice), "memcpy");
            if (h->ntstops > 0) ok2 = ok2 && HT(hipMemcpy(h->d_tstops, P.tstops_desc.data(), sizeof(double) * h->ntstops, hipMemcpyHostToDevice), "memcpy");
            if (!ok2) rc = HIPADJ_ERR_HIP;
        }
        AdaptGeom& ag = h->ag;
        ag.N = h->N; ag.Npad = Np; ag.M = h->M; ag.Smax = (h->auto_steps && cfg->alg != HIPADJ_ALG_BACKSOLVE) ? (int)h->rec_cap : P.Smax; ag.maxit = h->auto_steps ? HIPADJ_AUTO_MAXITERS : P.Smax; ag.nck = P.nck; ag.SmaxI = P.SmaxI; ag.t0 = cfg->t0; ag.t1 = cfg->t1; ag.dt0 = cfg->dt;
        ag.abstol = cfg->abstol; ag.reltol = cfg->reltol; ag.loss_shift = cfg->loss_shift; ag.loss_kind = cfg->loss_kind;
        ag.no_start = cfg->no_start; ag.p_shared = cfg->p_shared; ag.cont_cost = cfg->cont_cost;
    } else if (!P.field && !P.mlp) {
        A(dev_alloc(h, &h->d_outT, (size_t)h->M * n * Np));
        A(dev_alloc(h, &h->d_yT, (size_t)n * Np));
        if (cfg->alg != HIPADJ_ALG_BACKSOLVE && !P.ip_ckpt) A(dev_alloc(h, &h->d_knots, (size_t)(S + 1) * n * Np));
        if (bs_ckpt) A(dev_alloc(h, &h->d_ckpt, (size_t)h->nck * n * Np));
        if (cfg->loss_kind == HIPADJ_LOSS_COTANGENT) A(dev_alloc(h, &h->d_cotT, (size_t)h->M * n * Np));
        A(dev_alloc(h, &h->d_segbuf, (size_t)h->nseg * (1 + n) * (n + np) * Np));
    } else if (P.mlp) {
        h->mlp = true; h->field = false; h->NQ = P.NQ;
        const size_t Hh = cfg->dims[1], Bb = cfg->dims[2], Q = (size_t)h->N * S * P.NQ, HP = Hh + 16;
        const long groups = cfg->p_shared ? 1 : h->N;
        const long nslabs = (long)(Q / groups) * ((Bb + 63) / 64);   // weight-gradient GEMMs: 64-sample slabs, 2 workgroups per CU
        h->ksplit = (int)(nslabs < 512 ? nslabs : 512);
        A(dev_alloc(h, &h->d_fknots, (size_t)h->N * (S + 1) * 2 * n));
        A(dev_alloc(h, &h->d_w2t, (size_t)groups * Hh * Hh));
        A(dev_alloc(h, &h->d_ax, Q * 16 * Bb)); A(dev_alloc(h, &h->d_al, Q * 16 * Bb));
        A(dev_alloc(h, &h->d_ah1, Q * HP * Bb)); A(dev_alloc(h, &h->d_ah2, Q * HP * Bb));
        A(dev_alloc(h, &h->d_ag1, Q * Hh * Bb)); A(dev_alloc(h, &h->d_ag2, Q * Hh * Bb));
        A(dev_alloc(h, &h->d_c1, (size_t)groups * h->ksplit * Hh * HP));
        A(dev_alloc(h, &h->d_c2, (size_t)groups * h->ksplit * Hh * 16));
        A(dev_alloc(h, &h->d_c3, (size_t)groups * h->ksplit * 16 * HP));
        if (rc == HIPADJ_OK) {   // padding rows (zeros) and the ones rows are written once / by the sweep; zero everything first
            if (hipMemset(h->d_ax, 0, Q * 16 * Bb * 8) != hipSuccess || hipMemset(h->d_al, 0, Q * 16 * Bb * 8) != hipSuccess ||
                hipMemset(h->d_ah1, 0, Q * HP * Bb * 8) != hipSuccess || hipMemset(h->d_ah2, 0, Q * HP * Bb * 8) != hipSuccess) { h->err = "hipMemset failed"; rc = HIPADJ_ERR_HIP; }
        }
    } else {
        A(dev_alloc(h, &h->d_fknots, (size_t)h->N * (S + 1) * 2 * n));
        if (cfg->alg == HIPADJ_ALG_QUADRATURE) A(dev_alloc(h, &h->d_fadj, (size_t)h->N * S * 4 * n));
    }
    A(dev_alloc(h, &h->d_dp_traj, (size_t)np * Np));
    A(dev_alloc(h, &h->d_partial, (size_t)((h->N + 15) / 16) * np));   // one row per finishing workgroup (at most N / 16 of them)
    A(dev_alloc(h, &h->d_io_a, (size_t)h->N * (h->M > 0 ? h->M : 1) * n));
    A(dev_alloc(h, &h->d_du0, (size_t)h->N * n));
    A(dev_alloc(h, &h->d_dp, cfg->p_shared ? (size_t)np : (size_t)h->N * np));
    A(dev_alloc(h, &h->d_save_of_knot, (size_t)S + 1));
    A(dev_alloc(h, &h->d_ckpt_of_knot, (size_t)S + 1));
    A(dev_alloc(h, &h->d_prev_ck, (size_t)S + 1));
    A(dev_alloc(h, &h->d_seg_bounds, (size_t)h->nseg + 1));
    A(dev_alloc(h, &h->d_flag, 1));
    A(dev_alloc(h, &h->d_ticket, 1));
    const std::vector<double>&qa = P.qa, &qb = P.qb;
    if (cfg->alg == HIPADJ_ALG_QUADRATURE) {
        if (!h->field) A(dev_alloc(h, &h->d_adj, (size_t)S * 2 * n * Np));
        A(dev_alloc(h, &h->d_qres, (size_t)h->nq * np * Np));
        A(dev_alloc(h, &h->d_qa, (size_t)h->nq)); A(dev_alloc(h, &h->d_qb, (size_t)h->nq));
    }
    if (rc != HIPADJ_OK) return fail(rc);
    bool ok = HT(hipMemcpy(h->d_save_of_knot, h->save_of_knot.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice), "memcpy") &&
              HT(hipMemcpy(h->d_ckpt_of_knot, h->ckpt_of_knot.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice), "memcpy") &&
              HT(hipMemcpy(h->d_prev_ck, P.prev_ck.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice), "memcpy") &&
              HT(hipMemcpy(h->d_seg_bounds, h->seg_bounds.data(), sizeof(int) * (h->nseg + 1), hipMemcpyHostToDevice), "memcpy") &&
              HT(hipMemset(h->d_flag, 0, sizeof(int)), "memset") && HT(hipMemset(h->d_ticket, 0, sizeof(unsigned)), "memset");
    if (ok && h->nq > 0) ok = HT(hipMemcpy(h->d_qa, qa.data(), sizeof(double) * h->nq, hipMemcpyHostToDevice), "memcpy") &&
                              HT(hipMemcpy(h->d_qb, qb.data(), sizeof(double) * h->nq, hipMemcpyHostToDevice), "memcpy");
    if (!ok) return fail(HIPADJ_ERR_HIP);

    Geom& g = h->g;
    g.N = h->N; g.Npad = Np; g.S = (int)S; g.M = h->M; g.t0 = cfg->t0; g.dt = cfg->dt; g.loss_shift = cfg->loss_shift;
    g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start; g.p_shared = cfg->p_shared;
    g.kmask = -1;
    if (const char* e = std::getenv("HIPADJ_TIMING")) h->timing = std::atoi(e);
    if (const char* e = std::getenv("HIPADJ_FUSED_FINAL")) h->fused_final = std::atoi(e);
    if (const char* e = std::getenv("HIPADJ_WPB")) h->wpb4 = std::atoi(e) == 4;
    if (const char* e = std::getenv("HIPADJ_CBS")) h->cbs = std::atoi(e);
    h->fg.N = h->N; h->fg.S = (int)S; h->fg.M = h->M; h->fg.t0 = cfg->t0; h->fg.dt = cfg->dt; h->fg.loss_shift = cfg->loss_shift;
    h->fg.loss_kind = cfg->loss_kind; h->fg.no_start = cfg->no_start; h->fg.p_shared = cfg->p_shared;
    h->mg.N = h->N; h->mg.B = cfg->dims[2]; h->mg.S = (int)S; h->mg.M = h->M; h->mg.t0 = cfg->t0; h->mg.dt = cfg->dt; h->mg.loss_shift = cfg->loss_shift;
    h->mg.loss_kind = cfg->loss_kind; h->mg.no_start = cfg->no_start; h->mg.p_shared = cfg->p_shared; h->mg.NQ = P.NQ;

    h->st.struct_size = sizeof(hipadj_stats); h->st.n = n; h->st.np = np; h->st.ntraj = h->N; h->st.nsteps = S;
    h->st.time_segments = h->nseg; h->st.workspace_bytes = h->ws_bytes;
    // ALGORITHMIC bytes of one reverse pass: knots (u,f) once, cotangents (if read), du0 + dp out
    double bytes = 0.0;
    if (P.adaptive) bytes = 0.0;   // data-dependent (accepted steps per trajectory): not modelled
    else if (cfg->alg == HIPADJ_ALG_BACKSOLVE || P.ip_ckpt) bytes = (double)h->N * ((double)h->nck * 8.0 * n + 8.0 * n);
    else bytes = (double)h->N * (double)(S + 1) * 16.0 * n;
    if (cfg->alg == HIPADJ_ALG_QUADRATURE) bytes += (double)h->N * (double)S * 2.0 * 32.0 * n;   // dense lambda write + read
    if (cfg->loss_kind == HIPADJ_LOSS_COTANGENT) bytes += (double)h->N * h->M * 8.0 * n;
    bytes += (double)h->N * 8.0 * (n + np);
    h->st.adjoint_algorithmic_bytes = bytes;
    h->st.vjp_steps = (double)h->N * (double)S * 4.0;
    h->user = P.user;
    if (cfg->cont_cost == HIPADJ_CCOST_MODEL && !user_has_cost(cfg->model)) { h->err = "cont_cost = HIPADJ_CCOST_MODEL but the model has no cost (hipadj_model_set_cost)"; return fail(HIPADJ_ERR_INVALID_ARG); }
    if (P.user) { const int urc = user_prepare(h); if (urc != HIPADJ_OK) return fail(urc); }
    *out = h;
    return HIPADJ_OK;
}

extern "C" int hipadj_destroy(hipadj_handle* h) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
    return HIPADJ_OK;
}

extern "C" int hipadj_set_timing(hipadj_handle* h, int level) {
    if (!h || level < 0 || level > 2) return HIPADJ_ERR_INVALID_ARG;
    h->timing = level;
    return HIPADJ_OK;
}

extern "C" int hipadj_set_stream(hipadj_handle* h, void* s) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return HIPADJ_OK;
}

static void harvest_set(hipadj_handle* h, hipadj_handle::EvSet& q, bool block) {
    if (!q.pending) return;
    hipEvent_t last = q.full ? q.a1 : q.k1;
    if (block) { if (hipEventSynchronize(last) != hipSuccess) { q.pending = false; return; } }
    else if (hipEventQuery(last) != hipSuccess) return;           // still running: look again later
    float ms = 0.f;
    if (q.full && hipEventElapsedTime(&ms, q.a0, q.a1) == hipSuccess) { h->st.adjoint_ms_last = ms; h->st.adjoint_ms_total += ms; }
    if (hipEventElapsedTime(&ms, q.k0, q.k1) == hipSuccess) { h->st.adjoint_main_kernel_ms_last = ms; h->st.adjoint_main_kernel_ms_total += ms; }
    q.pending = false;
}
static void harvest_timing(hipadj_handle* h, bool block) {
    float ms = 0.f;
    if (h->timing_pending_fwd && (block ? hipEventSynchronize(h->ev[1]) : hipEventQuery(h->ev[1])) == hipSuccess) {
        if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) { h->st.forward_ms_last = ms; h->st.forward_ms_total += ms; }
        h->timing_pending_fwd = false;
    }
    for (int j = 0; j < hipadj_handle::NSET; ++j) harvest_set(h, h->evs[(h->ev_next + j) % hipadj_handle::NSET], block);   // oldest first
}

extern "C" int hipadj_synchronize(hipadj_handle* h) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    harvest_timing(h, true);
    int flag = 0;
    HIP_TRY(h, hipMemcpy(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(h, hipMemset(h->d_flag, 0, sizeof(int)));
        if (flag & 4) HIPADJ_FAIL(h, HIPADJ_ERR_MAXITERS, "adaptive Tsit5 exceeded max_steps = %d accepted steps on at least one trajectory (raise max_steps or loosen tolerances)", h->ag.Smax);
        HIPADJ_FAIL(h, HIPADJ_ERR_NONFINITE, "non-finite sensitivities (flag %d): a trajectory diverged", flag);
    }
    return HIPADJ_OK;
}

extern "C" int hipadj_get_stats(hipadj_handle* h, hipadj_stats* st) {
    if (!h || !st) return HIPADJ_ERR_INVALID_ARG;
    if (st->struct_size != sizeof(hipadj_stats)) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_stats.struct_size mismatch");
    *st = h->st;
    return HIPADJ_OK;
}

// ---- launch helpers --------------------------------------------------------------------------------------
static int launch_transpose_to_soa(hipadj_handle* h, const double* src, double* dst, int C) {
    dim3 blk(32, 8), grd((unsigned)((h->Npad + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_aos_to_soa, grd, blk, 0, h->stream, src, dst, h->N, h->Npad, C);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
static int launch_transpose_to_aos(hipadj_handle* h, const double* src, double* dst, int C) {
    dim3 blk(32, 8), grd((unsigned)((h->N + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_soa_to_aos, grd, blk, 0, h->stream, src, dst, h->N, h->Npad, C);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}

template <class Mo> static int forward_impl(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    dbl2* knots = h->d_knots;
    double* ck = h->d_ckpt;
    hipLaunchKernelGGL((k_forward<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, d_u0, d_p, knots, ck,
                       h->d_ckpt_of_knot, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, h->d_save_of_knot, h->d_yT);
    HIP_TRY(h, hipGetLastError());
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}

template <class Mo, int LOSS> static int adjoint_impl_l(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    // software-prefetch depth, chosen so that each kernel keeps 2 waves per SIMD (<= 256 VGPRs): the cotangent ring
    // and the Gauss-node state cost registers
    constexpr int PF = (LOSS & 1) == 1 ? 8 : 6, PFG = 4;   // LOSS here = MODE = discrete-loss kind | (continuous cost << 1)
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    const unsigned cblocks = (unsigned)((h->N + FIN / 4 - 1) / (FIN / 4));   // composition: 4 lanes per trajectory
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;   // per-trajectory dp rows [N][np]
    double* dp_sum = (h->cfg.p_shared && h->fused_final) ? d_dp : (double*)nullptr;   // in-launch last-arriver reduction (optional)
    // composition workgroups: 64 trajectories each, or 16 each while that still leaves the chip short of workgroups
    // (10^4 trajectories: 625 instead of 157 workgroups, -1.7 us per reverse pass; profiles/README.md)
    const bool small_blocks = h->cbs == 64 || (h->cbs == 0 && cblocks < 1024);
    const unsigned compose_blocks = small_blocks ? (unsigned)((h->N + 15) / 16) : cblocks;
    auto launch_compose = [&]() {
        if (small_blocks)
            hipLaunchKernelGGL((k_compose_finish<Mo, 64>), dim3(compose_blocks), dim3(64), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        else
            hipLaunchKernelGGL((k_compose_finish<Mo>), dim3(compose_blocks), dim3(FIN), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
    };
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);                   // ring full: only now wait for the oldest call
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    hipEvent_t k0 = es.k0, k1 = es.k1;          // dominant-kernel bracket
    bool dispatch_events = false;               // k0/k1 ride on the kernel's dispatch packet instead (k_interp, below)
    if (h->timing >= 1 && !(h->cfg.alg == HIPADJ_ALG_INTERPOLATING && !h->ip_ckpt)) HIP_TRY(h, hipEventRecord(k0, h->stream));
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->ip_ckpt)
            hipLaunchKernelGGL((k_interp_ckpt<Mo, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
        else if (h->wpb4) {
            // 256-thread workgroups, four (wave block, segment) items each: one wave per SIMD by construction (hipadj_kernels.hpp)
            const unsigned items = waves * (unsigned)h->nseg;
            hipExtLaunchKernelGGL((k_interp<Mo, PF, LOSS, true, 4>), dim3((items + 3) / 4), dim3(4 * WAVE), 0, h->stream, h->timing >= 1 ? k0 : (hipEvent_t) nullptr,
                                  h->timing >= 1 ? k1 : (hipEvent_t) nullptr, 0, h->g, sp, p, (const dbl2*)h->d_knots, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
            dispatch_events = true;
        } else if (h->timing >= 1) {
            // the dominant kernel's own begin/end timestamps (events attached to the dispatch packet): what rocprofv3 reports
            // as the kernel's duration.  A hipEventRecord pair around the launch also counts the two marker packets and the
            // dispatch latency (+8-10 us on a 0.12 ms kernel).
            hipExtLaunchKernelGGL((k_interp<Mo, PF, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, k0, k1, 0, h->g, sp, p,
                                  (const dbl2*)h->d_knots, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
            dispatch_events = true;
        } else
        hipLaunchKernelGGL((k_interp<Mo, PF, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p,
                           (const dbl2*)h->d_knots, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1 && !dispatch_events) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_BACKSOLVE: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        hipLaunchKernelGGL((k_backsolve<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_yT,
                           (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot, (const double*)h->d_cotT,
                           (const int*)h->d_save_of_knot, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_GAUSS: if constexpr ((LOSS >> 1) >= 2) { HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "GaussAdjoint with dgdp_continuous is not offered"); } else {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->ip_ckpt)
            hipLaunchKernelGGL((k_gauss_ckpt<Mo, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
        else
            hipLaunchKernelGGL((k_gauss<Mo, PFG, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const dbl2*)h->d_knots,
                               (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_GAUSS_KRONROD: if constexpr ((LOSS >> 1) >= 2) { HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "GaussKronrodAdjoint with dgdp_continuous is not offered"); } else {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        hipLaunchKernelGGL((k_gauss<Mo, PFG, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const dbl2*)h->d_knots,
                           (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_QUADRATURE: {
        hipLaunchKernelGGL((k_quad_adj<Mo, PF, LOSS>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, p, (const dbl2*)h->d_knots,
                           (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_adj, d_du0);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_quad_gk<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nq), dim3(WAVE), 0, h->stream, h->g, p, (const dbl2*)h->d_knots,
                           (const dbl2*)h->d_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        break; }
    }
    // finishing stage: NaN/Inf scan + per-workgroup partial sums of mu (Interpolating fused it with the composition)
    if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                           (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->cfg.p_shared && !h->fused_final) {   // dp = sum over workgroup partials, fixed order
        const unsigned nb = h->cfg.alg == HIPADJ_ALG_QUADRATURE ? fblocks : compose_blocks;
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)nb, h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;   // timings are harvested lazily at the next synchronize / call
}

template <class Mo> static int adjoint_impl(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    // no loss times => no cotangent buffer exists: run the LSQ specialisation (its jump select is never taken)
    const int mode = ((h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0) ? 0 : 1) | (h->cfg.cont_cost << 1);   // MODE = loss | cost << 1
    switch (mode) {
    case 0: return adjoint_impl_l<Mo, 0>(h, d_cot, d_du0, d_dp);
    case 1: return adjoint_impl_l<Mo, 1>(h, d_cot, d_du0, d_dp);
    case 2: return adjoint_impl_l<Mo, 2>(h, d_cot, d_du0, d_dp);
    case 3: return adjoint_impl_l<Mo, 3>(h, d_cot, d_du0, d_dp);
    case 4: return adjoint_impl_l<Mo, 4>(h, d_cot, d_du0, d_dp);
    case 5: return adjoint_impl_l<Mo, 5>(h, d_cot, d_du0, d_dp);
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "cont_cost %d is not available for compiled-in models", h->cfg.cont_cost);
    }
}

// ---- workgroup-per-trajectory family (Brusselator) -----------------------------------------------------
template <int G> static int field_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    hipLaunchKernelGGL((k_bruss_forward<G>), dim3((unsigned)h->N), dim3(Bruss<G>::T), 0, h->stream, h->fg, d_u0, d_p, h->d_fknots,
                       (d_out && h->M > 0) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
template <int G> static int field_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    const dim3 grid((unsigned)h->N), blk(Bruss<G>::T);
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING:
        hipLaunchKernelGGL((k_bruss_adjoint<G, 0>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_of_knot, d_du0, h->d_dp_traj, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        break;
    case HIPADJ_ALG_GAUSS:
        hipLaunchKernelGGL((k_bruss_adjoint<G, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_of_knot, d_du0, h->d_dp_traj, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        break;
    case HIPADJ_ALG_QUADRATURE: {
        hipLaunchKernelGGL((k_bruss_quad_adj<G>), grid, blk, 0, h->stream, h->fg, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_of_knot, h->d_fadj, d_du0, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_bruss_quad_gk<G, 128>), dim3((unsigned)h->N, (unsigned)h->nq), blk, 0, h->stream, h->fg, h->Npad, p,
                           (const double*)h->d_fknots, (const double*)h->d_fadj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        const unsigned waves = (unsigned)(h->Npad / WAVE);
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        break; }
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg not available for the PDE family");
    }
    hipLaunchKernelGGL((k_finish<0, 3>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                       (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, h->cfg.p_shared ? d_dp : (double*)nullptr);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = true;
    return HIPADJ_OK;
}
#define DISPATCH_GRID(h, fn, ...)                                                          \
    switch ((h)->cfg.dims[0]) {                                                            \
    case 8: return fn<8>(__VA_ARGS__);                                                     \
    case 16: return fn<16>(__VA_ARGS__);                                                   \
    case 32: return fn<32>(__VA_ARGS__);                                                   \
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "unsupported Brusselator grid %d", (h)->cfg.dims[0]); }


// ---- FP64-MFMA family (MLP neural ODE) -----------------------------------------------------------------
template <int H> static int mlp_forward_launch(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const int groups = h->cfg.p_shared ? 1 : (int)h->N;
    hipLaunchKernelGGL(k_mlp_transpose_w2, dim3(64, (unsigned)groups), dim3(256), 0, h->stream, H, Mlp<H>::NPAR, H * 2 + H, d_p, h->d_w2t);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL((k_mlp_forward<H>), dim3((unsigned)(h->mg.B / 16), (unsigned)h->N), dim3(Mlp<H>::NT), 0, h->stream, h->mg, d_u0, d_p, (const double*)h->d_w2t,
                       h->d_fknots, (d_out && h->M > 0) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
template <int H> static int mlp_adjoint_launch(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    constexpr int HP = Mlp<H>::HP;
    const double* p = h->p_dev_last;
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    MlpRec<H> R{h->d_ax, h->d_al, h->d_ah1, h->d_ah2, h->d_ag1, h->d_ag2};
    const dim3 grid((unsigned)(h->mg.B / 16), (unsigned)h->N), blk(64), sweep_blk(Mlp<H>::NT);
    if (h->cfg.alg == HIPADJ_ALG_GAUSS)
        hipLaunchKernelGGL((k_mlp_adjoint<H, 2>), grid, sweep_blk, 0, h->stream, h->mg, p, (const double*)h->d_w2t, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_of_knot, R, d_du0, h->d_flag);
    else
        hipLaunchKernelGGL((k_mlp_adjoint<H, 0>), grid, sweep_blk, 0, h->stream, h->mg, p, (const double*)h->d_w2t, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_of_knot, R, d_du0, h->d_flag);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    const long groups = h->cfg.p_shared ? 1 : h->N;
    const long Qper = (h->N * (long)h->S * h->NQ) / groups;
    const int B = h->mg.B, ks = h->ksplit;
    if (B % 64 == 0) {
        const size_t lds1 = (size_t)HP * WG_PITCH * sizeof(double), lds2 = (size_t)16 * WG_PITCH * sizeof(double);
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_wgrad<H / 16 + 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        hipLaunchKernelGGL((k_mlp_wgrad<H / 16 + 1>), dim3(1, (unsigned)ks, (unsigned)groups), dim3(64 * (H / 16)), lds1, h->stream, (const double*)h->d_ag2, (const double*)h->d_ah1, H, HP, Qper, B, ks, h->d_c1);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL((k_mlp_wgrad<1>), dim3(1, (unsigned)ks, (unsigned)groups), dim3(64 * (H / 16)), lds2, h->stream, (const double*)h->d_ag1, (const double*)h->d_ax, H, 16, Qper, B, ks, h->d_c2);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL((k_mlp_wgrad<H / 16 + 1>), dim3(1, (unsigned)ks, (unsigned)groups), dim3(64), lds1, h->stream, (const double*)h->d_al, (const double*)h->d_ah2, 16, HP, Qper, B, ks, h->d_c3);
        HIP_TRY(h, hipGetLastError());
    } else {   // batches that are not a multiple of 64 columns: 16-sample chunks straight from global memory
        hipLaunchKernelGGL((k_mlp_wgrad_small<H / 16 + 1>), dim3(H / 16, (unsigned)ks, (unsigned)groups), blk, 0, h->stream, (const double*)h->d_ag2, (const double*)h->d_ah1, H, HP, Qper, B, ks, h->d_c1);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL((k_mlp_wgrad_small<1>), dim3(H / 16, (unsigned)ks, (unsigned)groups), blk, 0, h->stream, (const double*)h->d_ag1, (const double*)h->d_ax, H, 16, Qper, B, ks, h->d_c2);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL((k_mlp_wgrad_small<H / 16 + 1>), dim3(1, (unsigned)ks, (unsigned)groups), blk, 0, h->stream, (const double*)h->d_al, (const double*)h->d_ah2, 16, HP, Qper, B, ks, h->d_c3);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL((k_mlp_wreduce<H>), dim3((Mlp<H>::NPAR + 255) / 256, (unsigned)groups), dim3(256), 0, h->stream, ks, (const double*)h->d_c1, (const double*)h->d_c2,
                       (const double*)h->d_c3, d_dp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = true;
    return HIPADJ_OK;
}
#define DISPATCH_HIDDEN(h, fn, ...)                                                        \
    switch ((h)->cfg.dims[1]) {                                                            \
    case 32: return fn<32>(__VA_ARGS__);                                                   \
    case 128: return fn<128>(__VA_ARGS__);                                                 \
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "unsupported hidden width %d", (h)->cfg.dims[1]); }

#define DISPATCH_MODEL(h, fn, ...)                                                         \
    switch ((h)->cfg.model) {                                                              \
    case HIPADJ_MODEL_LV: return fn<ModelLV>(__VA_ARGS__);                                 \
    case HIPADJ_MODEL_LVT: return fn<ModelLVT>(__VA_ARGS__);                               \
    case HIPADJ_MODEL_LORENZ: return fn<ModelLorenz>(__VA_ARGS__);                         \
    case HIPADJ_MODEL_LINDIAG: return fn<ModelLinDiag>(__VA_ARGS__);                       \
    case HIPADJ_MODEL_FALLMASS: return fn<ModelFallMass>(__VA_ARGS__);                     \
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "model %d has no device kernels", (h)->cfg.model); }

// ---- runtime-compiled models (hipadj_user.hpp) --------------------------------------------------------------
// Kernel instantiations a handle needs, by configuration.  Prefetch depths shrink with n: the knot ring holds
// PF x 2n doubles in VGPRs.
struct UserKernels { std::string forward, main_k, tail, gk; };
static UserKernels user_kernel_names(const hipadj_handle* h) {
    const std::string U = "hipadj::UserModel";
    const int n = h->n, np = h->np;
    const int mode = ((h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0) ? 0 : 1) | (h->cfg.cont_cost << 1);
    const int cc = mode >> 1;
    int PF = n <= 3 ? ((mode & 1) ? 8 : 6) : 1; const int PFG = n <= 3 ? 4 : 1;   // more than three states: the plain rolled sweep (reverse_sweep, PF == 1)
    if (const char* e = std::getenv("HIPADJ_USER_PF")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) PF = v; }   // tuning hook: prefetch depth of k_interp for runtime models
    auto I = [](int v) { return std::to_string(v); };
    UserKernels k;
    const bool seg = (1 + n) * (n + np) <= 64;          // same rule as the planner: wider models stay sequential in time ...
    const std::string SG = seg ? ", true>" : ", false>";  // ... and compile only the one-column path (k_interp SEG)
    const std::string finish = "hipadj::k_finish<" + I(n) + ", " + I(np) + ">";
    const std::string compose = seg ? "hipadj::k_compose_finish<" + U + ">" : "hipadj::k_finish_map<" + I(n) + ", " + I(np) + ">";
    if (h->adaptive) {
        k.forward = "hipadj::k_forward_tsit5<" + U + ">";
        k.main_k = "hipadj::k_adjoint_tsit5<" + U + ", " + I(h->cfg.alg) + ", " + I(cc) + (h->ip_ckpt ? ", true>" : ", false>");
        if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) k.gk = "hipadj::k_quad_gk_tsit5<" + U + ", " + I(cc) + ">";
        k.tail = finish;
        return k;
    }
    k.forward = "hipadj::k_forward<" + U + ">";
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING: k.main_k = "hipadj::k_interp<" + U + ", " + I(PF) + ", " + I(mode) + SG; k.tail = compose; break;
    case HIPADJ_ALG_BACKSOLVE: k.main_k = "hipadj::k_backsolve<" + U + ", " + I(cc) + SG; k.tail = compose; break;
    case HIPADJ_ALG_GAUSS: k.main_k = "hipadj::k_gauss<" + U + ", " + I(PFG) + ", " + I(mode) + ", false" + SG; k.tail = compose; break;
    case HIPADJ_ALG_GAUSS_KRONROD: k.main_k = "hipadj::k_gauss<" + U + ", " + I(PFG) + ", " + I(mode) + ", true" + SG; k.tail = compose; break;
    default: k.main_k = "hipadj::k_quad_adj<" + U + ", " + I(PF) + ", " + I(mode) + ">"; k.gk = "hipadj::k_quad_gk<" + U + ", " + I(cc) + ">"; k.tail = finish; break;
    }
    return k;
}

static int user_prepare(hipadj_handle* h) {
    const UserKernels k = user_kernel_names(h);
    std::vector<std::string> exprs = {k.forward, k.main_k, k.tail};
    if (!k.gk.empty()) exprs.push_back(k.gk);
    std::vector<char> code; std::map<std::string, std::string> low;
    const int rc = user_compile(h->cfg.model, exprs, code, low, h->err);
    if (rc != HIPADJ_OK) return rc;
    HIP_TRY(h, hipModuleLoadData(&h->umod, code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_forward, h->umod, low[k.forward].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_main, h->umod, low[k.main_k].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_tail, h->umod, low[k.tail].c_str()));
    if (!k.gk.empty()) HIP_TRY(h, hipModuleGetFunction(&h->uf_gk, h->umod, low[k.gk].c_str()));
    return HIPADJ_OK;
}

static int user_compile_config(const hipadj_config* cfg, std::string& err) {
    hipadj_handle h;
    h.cfg = *cfg; h.cfg.save_times = nullptr;
    Plan P;
    { const int prc = make_plan(cfg, P, err); if (prc != HIPADJ_OK) return prc; }
    h.n = P.n; h.np = P.np; h.M = P.M; h.adaptive = P.adaptive; h.ip_ckpt = P.ip_ckpt;
    const UserKernels k = user_kernel_names(&h);
    std::vector<std::string> exprs = {k.forward, k.main_k, k.tail};
    if (!k.gk.empty()) exprs.push_back(k.gk);
    std::vector<char> code; std::map<std::string, std::string> low;
    return user_compile(cfg->model, exprs, code, low, err);
}

// launch of a module kernel.  `sig` is the SAME kernel template instantiated for a compiled-in model: it is never called,
// it only lets the compiler check that the argument list handed to hipModuleLaunchKernel has exactly the kernel's
// parameter types (a mismatch would otherwise be silent memory corruption on the device).
template <class... P, class... A> static int ulaunch(void (*sig)(P...), hipadj_handle* h, hipFunction_t fn, dim3 g, dim3 b, A... args) {
    (void)sig;
    static_assert(sizeof...(P) == sizeof...(A), "argument count differs from the kernel's parameter list");
    static_assert((std::is_same<P, A>::value && ...), "argument types differ from the kernel's parameter list");
    void* ptrs[] = {(void*)&args...};
    HIP_TRY(h, hipModuleLaunchKernel(fn, g.x, g.y, g.z, b.x, b.y, b.z, 0, h->stream, ptrs, nullptr));
    return HIPADJ_OK;
}
static_assert(std::is_same<decltype(&k_interp<ModelLV, 8, 1>), decltype(&k_gauss<ModelLV, 4, 1, false>)>::value, "k_interp / k_gauss share one launch site");

static int user_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    double* outT = (d_out && h->M > 0) ? h->d_outT : (double*)nullptr;
    if (h->adaptive) {
        const bool sized = h->auto_steps && h->cfg.alg != HIPADJ_ALG_BACKSOLVE;
        if (sized && h->ip_ckpt) h->ag.SmaxI = (int)h->rec_cap;
        for (int pass = 0; pass < 2; ++pass) {
            TRY(ulaunch(&k_forward_tsit5<ModelLV>, h, h->uf_forward, dim3(waves), dim3(WAVE), h->ag, d_u0, d_p, h->ip_ckpt ? (double*)nullptr : h->d_rec, h->d_nsteps,
                        (const double*)h->d_save_t, outT, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag));
            if (!sized) break;
            const int again = adaptive_autosize(h);
            if (again < 0) return again;
            if (again == 0) break;
        }
    }
    else
        TRY(ulaunch(&k_forward<ModelLV>, h, h->uf_forward, dim3(waves), dim3(WAVE), h->g, d_u0, d_p, h->d_knots, h->d_ckpt, (const int*)h->d_ckpt_of_knot, outT,
                    (const int*)h->d_save_of_knot, h->d_yT));
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}

static int user_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN), cblocks = (unsigned)((h->N + FIN / 4 - 1) / (FIN / 4));
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    double* no_sum = nullptr;
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    bool composed = false;
    if (h->adaptive) {
        TRY(ulaunch(&k_adjoint_tsit5<ModelLV, 0, 0, false>, h, h->uf_main, dim3(waves), dim3(WAVE), h->ag, p, (const double*)h->d_rec, (const int*)h->d_nsteps, (const double*)h->d_yT,
                    (const double*)h->d_ckpt, (const double*)h->d_ck_t, (const double*)h->d_save_t, (const double*)h->d_tstops, h->ntstops,
                    (const double*)h->d_cotT, d_du0, h->d_dp_traj, h->d_flag, h->d_arec, h->d_nsteps_adj, h->SmaxA));
        if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            TRY(ulaunch(&k_quad_gk_tsit5<ModelLV, 0>, h, h->uf_gk, dim3(waves, (unsigned)h->nq), dim3(WAVE), h->ag, p, (const double*)h->d_rec, (const int*)h->d_nsteps, (const double*)h->d_arec,
                        (const int*)h->d_nsteps_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres));
            hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
            HIP_TRY(h, hipGetLastError());
        }
    } else {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        const dim3 sgrid(waves, (unsigned)h->nseg);
        switch (h->cfg.alg) {
        case HIPADJ_ALG_INTERPOLATING: case HIPADJ_ALG_GAUSS: case HIPADJ_ALG_GAUSS_KRONROD:
            TRY(ulaunch(&k_interp<ModelLV, 8, 1>, h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, p, (const dbl2*)h->d_knots, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf));
            composed = true; break;
        case HIPADJ_ALG_BACKSOLVE:
            TRY(ulaunch(&k_backsolve<ModelLV, 0>, h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, p, (const double*)h->d_yT, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot,
                        (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_segbuf));
            composed = true; break;
        default: {
            TRY(ulaunch(&k_quad_adj<ModelLV, 8, 1>, h, h->uf_main, dim3(waves), dim3(WAVE), h->g, p, (const dbl2*)h->d_knots, (const double*)h->d_cotT, (const int*)h->d_save_of_knot, h->d_adj, d_du0));
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            TRY(ulaunch(&k_quad_gk<ModelLV, 0>, h, h->uf_gk, dim3(waves, (unsigned)h->nq), dim3(WAVE), h->g, p, (const dbl2*)h->d_knots, (const dbl2*)h->d_adj, (const double*)h->d_qa,
                        (const double*)h->d_qb, atol, rtol, h->d_qres));
            hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
            HIP_TRY(h, hipGetLastError());
            break; }
        }
    }
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    const bool seg_kernels = (1 + h->n) * (h->n + h->np) <= 64;
    if (composed && seg_kernels)
        TRY(ulaunch(&k_compose_finish<ModelLV>, h, h->uf_tail, dim3(cblocks), dim3(FIN), h->g, h->nseg, (const double*)h->d_segbuf, d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    else if (composed)
        TRY(ulaunch(&k_finish_map<2, 4>, h, h->uf_tail, dim3(fblocks), dim3(FIN), h->N, h->Npad, (const double*)h->d_segbuf, d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    else
        TRY(ulaunch(&k_finish<2, 4>, h, h->uf_tail, dim3(fblocks), dim3(FIN), h->N, h->Npad, (const double*)d_du0, (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    if (h->cfg.p_shared) {
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)((composed && seg_kernels) ? cblocks : fblocks), h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;
}

// ---- adaptive Tsit5 (hipadj_adaptive.hpp) ------------------------------------------------------------------
// max_steps == 0 ("auto", the reference's maxiters = 1e5 behaviour): the record buffers start small and follow the measured
// step counts.  Every forward pass stores the TRUE number of accepted steps per trajectory (also beyond the capacity, where the
// records are simply not written); after the pass the host takes the maximum: if it fits, the pass stands (steady state: ONE
// pass + one host synchronisation); if not, the buffers are regrown to that maximum (+12 %) and the pass is repeated — the step
// sequence is the same, so the second pass cannot overflow.  Lorenz at the default tolerances takes ~100 steps: 0.2 GB of
// records for 10^4 trajectories instead of the 2.85 GB a fixed 2048-step bound reserves.
// Returns 1 when the forward pass has to be repeated, 0 when it stands, a negative status on error.
static int adaptive_autosize(hipadj_handle* h) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<int> ns((size_t)h->Npad);
    HIP_TRY(h, hipMemcpy(ns.data(), h->d_nsteps, sizeof(int) * (size_t)h->Npad, hipMemcpyDeviceToHost));
    long mx = 1;
    for (long i = 0; i < h->N; ++i) if (ns[i] > mx) mx = ns[i];
    if (mx <= h->rec_cap || mx >= HIPADJ_AUTO_MAXITERS) return 0;     // fits (or ran into maxiters: the flag reports it)
    const int RW = 2 + 5 * h->n;
    const long cap = mx + mx / 8 + 8;
    auto regrow = [&](double** buf, size_t old_count, size_t new_count) -> int {
        if (*buf) { (void)hipFree(*buf); *buf = nullptr; h->ws_bytes -= (double)(old_count * sizeof(double)); }
        return dev_alloc(h, buf, new_count);
    };
    TRY(regrow(&h->d_rec, (size_t)h->rec_cap * RW * h->Npad, (size_t)cap * RW * h->Npad));
    if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        const long capA = 2 * cap + h->M + 16;
        TRY(regrow(&h->d_arec, (size_t)h->SmaxA * RW * h->Npad, (size_t)capA * RW * h->Npad));
        h->SmaxA = (int)capA;
    }
    h->rec_cap = cap;
    h->st.workspace_bytes = h->ws_bytes;
    h->ag.Smax = (int)cap; h->ag.SmaxI = (int)cap;
    HIP_TRY(h, hipMemset(h->d_flag, 0, sizeof(int)));                  // the overflow mark of the pass that is being repeated
    return 1;
}

template <class Mo> static int adaptive_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const bool sized = h->auto_steps && h->cfg.alg != HIPADJ_ALG_BACKSOLVE;
    if (sized && h->ip_ckpt) h->ag.SmaxI = (int)h->rec_cap;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL((k_forward_tsit5<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->ag, d_u0, d_p, h->ip_ckpt ? (double*)nullptr : h->d_rec, h->d_nsteps,
                           (const double*)h->d_save_t, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        if (!sized) break;
        const int again = adaptive_autosize(h);
        if (again < 0) return again;
        if (again == 0) break;
    }
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}
template <class Mo, int ALG, int CC, bool CK = false> static int adaptive_adjoint_l(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    hipLaunchKernelGGL((k_adjoint_tsit5<Mo, ALG, CC, CK>), dim3(waves), dim3(WAVE), 0, h->stream, h->ag, h->p_dev_last, (const double*)h->d_rec,
                       (const int*)h->d_nsteps, (const double*)h->d_yT, (const double*)h->d_ckpt, (const double*)h->d_ck_t, (const double*)h->d_save_t,
                       (const double*)h->d_tstops, h->ntstops, (const double*)h->d_cotT, d_du0, h->d_dp_traj, h->d_flag,
                       h->d_arec, h->d_nsteps_adj, h->SmaxA);
    HIP_TRY(h, hipGetLastError());
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    if constexpr (ALG == 3) {
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_quad_gk_tsit5<Mo, CC>), dim3(waves, (unsigned)h->nq), dim3(WAVE), 0, h->stream, h->ag, h->p_dev_last, (const double*)h->d_rec,
                           (const int*)h->d_nsteps, (const double*)h->d_arec, (const int*)h->d_nsteps_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                       (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, (double*)nullptr);
    HIP_TRY(h, hipGetLastError());
    if (h->cfg.p_shared) {
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)fblocks, h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;
}
template <class Mo> static int adaptive_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    if (h->ip_ckpt) {   // checkpointing=true for Interpolating / Gauss: per-interval re-solve inside the sweep
        switch (h->cfg.alg * 4 + h->cfg.cont_cost) {
        case HIPADJ_ALG_INTERPOLATING * 4 + 0: return adaptive_adjoint_l<Mo, 0, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_INTERPOLATING * 4 + 1: return adaptive_adjoint_l<Mo, 0, 1, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_INTERPOLATING * 4 + 2: return adaptive_adjoint_l<Mo, 0, 2, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS * 4 + 0: return adaptive_adjoint_l<Mo, 2, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS * 4 + 1: return adaptive_adjoint_l<Mo, 2, 1, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return adaptive_adjoint_l<Mo, 4, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return adaptive_adjoint_l<Mo, 4, 1, true>(h, d_cot, d_du0, d_dp);
        default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg %d / cont_cost %d has no checkpointed adaptive device kernel", h->cfg.alg, h->cfg.cont_cost);
        }
    }
    switch (h->cfg.alg * 4 + h->cfg.cont_cost) {
    case HIPADJ_ALG_INTERPOLATING * 4 + 0: return adaptive_adjoint_l<Mo, 0, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_INTERPOLATING * 4 + 1: return adaptive_adjoint_l<Mo, 0, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_INTERPOLATING * 4 + 2: return adaptive_adjoint_l<Mo, 0, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 0: return adaptive_adjoint_l<Mo, 1, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 1: return adaptive_adjoint_l<Mo, 1, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 2: return adaptive_adjoint_l<Mo, 1, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS * 4 + 0: return adaptive_adjoint_l<Mo, 2, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS * 4 + 1: return adaptive_adjoint_l<Mo, 2, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 0: return adaptive_adjoint_l<Mo, 3, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 1: return adaptive_adjoint_l<Mo, 3, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 2: return adaptive_adjoint_l<Mo, 3, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return adaptive_adjoint_l<Mo, 4, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return adaptive_adjoint_l<Mo, 4, 1>(h, d_cot, d_du0, d_dp);
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg %d / cont_cost %d has no adaptive device kernel", h->cfg.alg, h->cfg.cont_cost);
    }
}

static int forward_dispatch(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (h->user) return user_forward(h, d_u0, d_p, d_out);
    if (h->field) { DISPATCH_GRID(h, field_forward, h, d_u0, d_p, d_out); }
    if (h->mlp) { DISPATCH_HIDDEN(h, mlp_forward_launch, h, d_u0, d_p, d_out); }
    if (h->adaptive) { DISPATCH_MODEL(h, adaptive_forward, h, d_u0, d_p, d_out); }
    DISPATCH_MODEL(h, forward_impl, h, d_u0, d_p, d_out);
}
static int adjoint_dispatch(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    if (h->user) return user_adjoint(h, d_cot, d_du0, d_dp);
    if (h->field) { DISPATCH_GRID(h, field_adjoint, h, d_cot, d_du0, d_dp); }
    if (h->mlp) { DISPATCH_HIDDEN(h, mlp_adjoint_launch, h, d_cot, d_du0, d_dp); }
    if (h->adaptive) { DISPATCH_MODEL(h, adaptive_adjoint, h, d_cot, d_du0, d_dp); }
    DISPATCH_MODEL(h, adjoint_impl, h, d_cot, d_du0, d_dp);
}

extern "C" int hipadj_forward_dev(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!d_u0 || !d_p) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "u0 and p must be non-NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    harvest_timing(h, false);
    // keep private copies: the adjoint needs p, and u0 may be released by the caller
    const size_t pb = sizeof(double) * (h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np);
    if (d_p != h->d_p) HIP_TRY(h, hipMemcpyAsync(h->d_p, d_p, pb, hipMemcpyDeviceToDevice, h->stream));
    h->p_dev_last = h->d_p;
    HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    TRY(forward_dispatch(h, d_u0, h->d_p, d_out));
    HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    h->timing_pending_fwd = true; h->have_forward = true; h->st.forward_calls++;
    return HIPADJ_OK;
}

extern "C" int hipadj_adjoint_dev(hipadj_handle* h, const double* d_dLdu, double* d_du0, double* d_dp) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!h->have_forward) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_adjoint called before hipadj_forward (the reverse pass consumes the forward solution)");
    if (!d_du0 || !d_dp) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "du0 and dp must be non-NULL");
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0 && !d_dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    harvest_timing(h, false);
    TRY(adjoint_dispatch(h, d_dLdu, d_du0, d_dp));
    h->st.adjoint_calls++;
    return HIPADJ_OK;
}

extern "C" int hipadj_forward(hipadj_handle* h, const double* u0, const double* p, double* out) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!u0 || !p) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "u0 and p must be non-NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t pb = sizeof(double) * (h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np);
    HIP_TRY(h, hipMemcpyAsync(h->d_u0, u0, sizeof(double) * (size_t)h->N * h->n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_p, p, pb, hipMemcpyHostToDevice, h->stream));
    TRY(hipadj_forward_dev(h, h->d_u0, h->d_p, out ? h->d_io_a : nullptr));
    if (out && h->M > 0) HIP_TRY(h, hipMemcpyAsync(out, h->d_io_a, sizeof(double) * (size_t)h->N * h->M * h->n, hipMemcpyDeviceToHost, h->stream));
    return hipadj_synchronize(h);
}

extern "C" int hipadj_adjoint(hipadj_handle* h, const double* dLdu, double* du0, double* dp) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!du0 || !dp) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "du0 and dp must be non-NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const bool cot = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0;
    if (cot && !dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    if (cot) HIP_TRY(h, hipMemcpyAsync(h->d_io_a, dLdu, sizeof(double) * (size_t)h->N * h->M * h->n, hipMemcpyHostToDevice, h->stream));
    TRY(hipadj_adjoint_dev(h, cot ? h->d_io_a : nullptr, h->d_du0, h->d_dp));
    const size_t pb = sizeof(double) * (h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np);
    HIP_TRY(h, hipMemcpyAsync(du0, h->d_du0, sizeof(double) * (size_t)h->N * h->n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dp, h->d_dp, pb, hipMemcpyDeviceToHost, h->stream));
    return hipadj_synchronize(h);
}
