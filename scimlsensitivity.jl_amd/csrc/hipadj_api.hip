// hipadj_api.hip — the C-ABI shared library (include/hipadj.h) over the gfx950 kernel family.
// Host side only does validation, workspace ownership, launch sequencing and timing; all arithmetic is in
// hipadj_lane.hpp / hipadj_kernels.hpp.  No CPU fallback exists: without a usable HIP device
// hipadj_create returns HIPADJ_ERR_NO_DEVICE.
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "hipadj_host.hpp"
#include <sys/mman.h>
#include "hipadj_plan.hpp"
#include "hipadj_user.hpp"
#include "hipadj_comm.hpp"
#include "hipadj_multi.hpp"
#include "hipadj_route.hpp"

static thread_local std::string g_create_error;
static int user_prepare(hipadj_handle* h);   // hiprtc compilation of the kernels of a runtime-registered model
static bool fused_eligible(const hipadj_config* cfg, const Plan& P);   // which sweeps finish their reverse pass in one launch
static int wide_prepare(hipadj_handle* h);   // ... of a wide model (workgroup-per-trajectory family, hipadj_wide.hpp)
static int wide_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out);
static int wide_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp);

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
    case HIPADJ_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
    }
}

extern "C" int hipadj_device_count(void) { int n = 0; return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? n : 0; }

extern "C" const char* hipadj_last_error(const hipadj_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int hipadj_model_sizes(int32_t model, const int32_t dims[4], int32_t* n, int32_t* np) {
    if (!n || !np) return HIPADJ_ERR_INVALID_ARG;
    return plan_model_sizes(model, dims, n, np);
}

extern "C" int hipadj_model_register(const char* name, int32_t n, int32_t np, const char* f_body, const char* vjp_u_body,
                                     const char* vjp_p_body, int32_t* model_id) {
    return user_register(name, n, np, f_body, vjp_u_body, vjp_p_body, model_id, g_create_error);
}

extern "C" int hipadj_wmodel_register(const char* name, int32_t n, int32_t np, int32_t threads, int32_t lds_doubles, int32_t nacc, int32_t acc_first,
                                      const char* f_body, const char* vjp_body, int32_t* model_id) {
    return user_register_wide(name, n, np, threads, lds_doubles, nacc, acc_first, f_body, vjp_body, model_id, g_create_error);
}

extern "C" int hipadj_model_set_cost(int32_t model_id, const char* dgdu_body, const char* dgdp_body) {
    return user_set_cost(model_id, dgdu_body, dgdp_body, g_create_error);
}

extern "C" int hipadj_model_set_cost_function(int32_t model_id, const char* g_body) {
    return user_set_cost_function(model_id, g_body, g_create_error);
}

extern "C" int hipadj_wmodel_set_cost(int32_t model_id, const char* cost_body) {
    return user_set_wide_cost(model_id, cost_body, g_create_error);
}

extern "C" int hipadj_model_set_discrete_loss(int32_t model_id, const char* dgdu_body, const char* dgdp_body) {
    return user_set_discrete_loss(model_id, dgdu_body, dgdp_body, nullptr, g_create_error);
}
extern "C" int hipadj_model_set_discrete_loss_function(int32_t model_id, const char* l_body) {
    if (!l_body) { g_create_error = "hipadj_model_set_discrete_loss_function: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    return user_set_discrete_loss(model_id, nullptr, nullptr, l_body, g_create_error);
}
extern "C" int hipadj_wmodel_set_discrete_loss(int32_t model_id, const char* dloss_body) {
    return user_set_wide_discrete_loss(model_id, dloss_body, g_create_error);
}

extern "C" int hipadj_wmodel_declare_dense_chain(int32_t model_id, const int32_t* widths, int32_t nwidths, int32_t activation, int32_t input_power) {
    return user_declare_dense_chain(model_id, widths, nwidths, activation, input_power, g_create_error);
}

extern "C" int hipadj_model_set_mass_matrix(int32_t model_id, const double* M) {
    return user_set_mass_matrix(model_id, M, g_create_error);
}

extern "C" int hipadj_wmodel_set_affect(int32_t model_id, const char* affect_body, const char* affect_vjp_body) {
    return user_set_wide_affect(model_id, affect_body, affect_vjp_body, g_create_error);
}
extern "C" int hipadj_model_set_affect(int32_t model_id, const char* affect_body) {
    return user_set_affect(model_id, affect_body, g_create_error);
}
extern "C" int hipadj_model_set_continuous_callback(int32_t model_id, const char* condition_body, const char* affect_body, int32_t max_events) {
    return user_set_continuous_callback(model_id, condition_body, affect_body, max_events, g_create_error);
}
extern "C" int hipadj_model_set_callback_direction(int32_t model_id, int32_t direction) {
    return user_set_callback_direction(model_id, direction, g_create_error);
}
extern "C" int hipadj_model_set_vector_continuous_callback(int32_t model_id, int32_t ncond, const char* condition_body, const char* affect_body, int32_t max_events) {
    return user_set_continuous_callback(model_id, condition_body, affect_body, max_events, g_create_error, ncond);
}

// ---- DiscreteCallback affects applied between solves (host-level composition of event problems, interface.py) --------------------------------
// Both calls are synchronous and take HOST pointers: the data of an event is N x (n + np) doubles.  The kernels are compiled for the model with
// hiprtc on first use (same cache as the solve kernels).
namespace {
struct AffectFns { hipModule_t mod = nullptr; hipFunction_t apply = nullptr, vjp = nullptr; };
// loaded modules are kept per (model, source revision, device): an event chain calls these entry points once per event and reverse callback
// (they reloaded the code object on every call).  Never unloaded: a model's affect lives as long as the process.
struct AffectCache { std::mutex mu; std::map<std::string, AffectFns> m; };
AffectCache& affect_cache() { static AffectCache c; return c; }
int affect_functions_uncached(int32_t model, AffectFns& F, std::string& err);
int affect_functions(int32_t model, int32_t device, AffectFns& F, std::string& err) {
    int rev = 0;
    { UserRegistry& R = user_registry(); std::lock_guard<std::mutex> lk(R.mu); const int idx = model - HIPADJ_MODEL_USER_BASE; if (idx >= 0 && idx < (int)R.models.size()) rev = R.models[idx].rev; }
    const std::string key = std::to_string(model) + "#" + std::to_string(rev) + "@" + std::to_string(device);
    AffectCache& Cc = affect_cache();
    std::lock_guard<std::mutex> lk(Cc.mu);
    auto it = Cc.m.find(key);
    if (it != Cc.m.end()) { F = it->second; return HIPADJ_OK; }
    const int rc = affect_functions_uncached(model, F, err);
    if (rc == HIPADJ_OK) Cc.m[key] = F;
    return rc;
}
int affect_functions_uncached(int32_t model, AffectFns& F, std::string& err) {
    std::vector<char> code; std::map<std::string, std::string> low;
    const std::vector<std::string> exprs = user_model_is_wide(model) ? std::vector<std::string>{"hipadj::k_wide_affect<hipadj::UserW>", "hipadj::k_wide_affect_vjp<hipadj::UserW>"}
                                                                     : std::vector<std::string>{"hipadj::k_user_affect<hipadj::UserModel>", "hipadj::k_user_affect_vjp<hipadj::UserModel>"};
    const int rc = user_compile(model, exprs, code, low, err);
    if (rc != HIPADJ_OK) return rc;
    if (hipModuleLoadData(&F.mod, code.data()) != hipSuccess || hipModuleGetFunction(&F.apply, F.mod, low[exprs[0]].c_str()) != hipSuccess ||
        hipModuleGetFunction(&F.vjp, F.mod, low[exprs[1]].c_str()) != hipSuccess) { err = "hipadj affect: loading the compiled kernels failed"; if (F.mod) (void)hipModuleUnload(F.mod); return HIPADJ_ERR_HIP; }
    return HIPADJ_OK;
}
struct DevBufs {
    std::vector<void*> v;
    ~DevBufs() { for (void* q : v) if (q) (void)hipFree(q); }
    double* get(size_t count, const double* src, bool& ok) {
        void* q = nullptr;
        if (hipMalloc(&q, sizeof(double) * (count ? count : 1)) != hipSuccess) { ok = false; return nullptr; }
        v.push_back(q);
        if (src && hipMemcpy(q, src, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        return (double*)q;
    }
};
int affect_prologue(int32_t model, int32_t device, int64_t N, int32_t& n, int32_t& np, std::string& err) {
    if (N <= 0) { err = "hipadj affect: N must be positive"; return HIPADJ_ERR_INVALID_ARG; }
    if (user_model_sizes(model, &n, &np) != HIPADJ_OK) { err = "hipadj affect: unknown model id (affects are attached to runtime-registered models)"; return HIPADJ_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { err = "no usable HIP device (this library has no CPU fallback)"; return HIPADJ_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return HIPADJ_ERR_HIP; }
    return HIPADJ_OK;
}
}  // namespace

extern "C" int hipadj_affect_apply(int32_t model_id, int32_t device, int64_t N, const double* u, const double* p, int32_t p_shared, double t, double* out, double* p_out) {
    if (!u || !p || !out) { g_create_error = "hipadj_affect_apply: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    int32_t n = 0, np = 0;
    { const int rc = affect_prologue(model_id, device, N, n, np, g_create_error); if (rc != HIPADJ_OK) return rc; }
    AffectFns F;
    { const int rc = affect_functions(model_id, device, F, g_create_error); if (rc != HIPADJ_OK) return rc; }
    bool ok = true; DevBufs B;
    double* d_u = B.get((size_t)N * n, u, ok); double* d_p = B.get(p_shared ? (size_t)np : (size_t)N * np, p, ok); double* d_o = B.get((size_t)N * n, nullptr, ok);
    double* d_po = B.get((size_t)N * np, nullptr, ok);
    long Nl = (long)N, ldp = p_shared ? 0 : np;
    void* args[] = {&Nl, &ldp, &d_u, &d_p, &t, &d_o, &d_po};
    ok = ok && hipModuleLaunchKernel(F.apply, (unsigned)((N + 255) / 256), 1, 1, 256, 1, 1, 0, nullptr, args, nullptr) == hipSuccess;
    ok = ok && hipMemcpy(out, d_o, sizeof(double) * (size_t)N * n, hipMemcpyDeviceToHost) == hipSuccess;
    if (p_out) ok = ok && hipMemcpy(p_out, d_po, sizeof(double) * (size_t)N * np, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { g_create_error = "hipadj_affect_apply: a HIP call failed"; return HIPADJ_ERR_HIP; }
    return HIPADJ_OK;
}

extern "C" int hipadj_affect_vjp(int32_t model_id, int32_t device, int64_t N, const double* u, const double* p, int32_t p_shared, double t, const double* lam,
                                 const double* gp, double* lam_out, double* gp_out) {
    if (!u || !p || !lam || !gp || !lam_out || !gp_out) { g_create_error = "hipadj_affect_vjp: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    int32_t n = 0, np = 0;
    { const int rc = affect_prologue(model_id, device, N, n, np, g_create_error); if (rc != HIPADJ_OK) return rc; }
    AffectFns F;
    { const int rc = affect_functions(model_id, device, F, g_create_error); if (rc != HIPADJ_OK) return rc; }
    bool ok = true; DevBufs B;
    double* d_u = B.get((size_t)N * n, u, ok); double* d_p = B.get(p_shared ? (size_t)np : (size_t)N * np, p, ok); double* d_l = B.get((size_t)N * n, lam, ok);
    double* d_gi = B.get((size_t)N * np, gp, ok); double* d_lo = B.get((size_t)N * n, nullptr, ok); double* d_g = B.get((size_t)N * np, nullptr, ok);
    long Nl = (long)N, ldp = p_shared ? 0 : np;
    void* args[] = {&Nl, &ldp, &d_u, &d_p, &t, &d_l, &d_gi, &d_lo, &d_g};
    ok = ok && hipModuleLaunchKernel(F.vjp, (unsigned)((N + 255) / 256), 1, 1, 256, 1, 1, 0, nullptr, args, nullptr) == hipSuccess;
    ok = ok && hipMemcpy(lam_out, d_lo, sizeof(double) * (size_t)N * n, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(gp_out, d_g, sizeof(double) * (size_t)N * np, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { g_create_error = "hipadj_affect_vjp: a HIP call failed"; return HIPADJ_ERR_HIP; }
    return HIPADJ_OK;
}

static std::vector<std::string> wide_kernel_names(int alg, bool ts5 = false, int cost = 0, bool ip_ckpt = false, bool offgrid = false) {
    // the reverse kernels of a handle with a built-in continuous cost are instantiated for WideWithCost<UserW, kind> (hipadj_wide.hpp); the forward solve never sees the cost
    const std::string U = cost ? "hipadj::WideWithCost<hipadj::UserW, " + std::to_string(cost) + ">" : std::string("hipadj::UserW");
    if (ts5) {
        std::vector<std::string> e = {"hipadj::k_wide_forward_ts5<hipadj::UserW>", alg == HIPADJ_ALG_BACKSOLVE ? "hipadj::k_wide_backsolve_ts5<" + U + ">"
                                      : "hipadj::k_wide_adjoint_ts5<" + U + ", " + (alg == HIPADJ_ALG_INTERPOLATING ? "0" : (alg == HIPADJ_ALG_QUADRATURE ? "3" : (alg == HIPADJ_ALG_GAUSS_KRONROD ? "4" : "2"))) + (ip_ckpt ? ", true>" : ", false>")};
        if (alg == HIPADJ_ALG_QUADRATURE) e.push_back("hipadj::k_wide_quad_gk<" + U + ", " + std::to_string(HIPADJ_WIDE_MAXSEG) + ", true>");
        return e;
    }
    std::vector<std::string> e = {"hipadj::k_wide_forward<hipadj::UserW>"};
    if (offgrid) {   // loss times off the step grid: the sweep over the reverse step list, and out = sol(ts) by interpolation in the `gk` slot
        if (alg == HIPADJ_ALG_BACKSOLVE) e.push_back("hipadj::k_wide_backsolve_og<" + U + ">");
        else if (alg == HIPADJ_ALG_QUADRATURE) e.push_back("hipadj::k_wide_quad_adj_og<" + U + ">");
        else e.push_back("hipadj::k_wide_adjoint_og<" + U + (alg == HIPADJ_ALG_GAUSS ? ", 2>" : (alg == HIPADJ_ALG_GAUSS_KRONROD ? ", 4>" : ", 0>")));
        e.push_back("hipadj::k_wide_out_offgrid<hipadj::UserW>");
        if (alg == HIPADJ_ALG_QUADRATURE) e.push_back("hipadj::k_wide_quad_gk<" + U + ", " + std::to_string(HIPADJ_WIDE_MAXSEG) + ", false, true>");   // fourth name: uf_aux
        return e;
    }
    switch (alg) {
    case HIPADJ_ALG_INTERPOLATING: e.push_back(std::string("hipadj::k_wide_adjoint") + (ip_ckpt ? "_ck<" : "<") + U + ", 0>"); break;
    case HIPADJ_ALG_GAUSS: e.push_back(std::string("hipadj::k_wide_adjoint") + (ip_ckpt ? "_ck<" : "<") + U + ", 2>"); break;
    case HIPADJ_ALG_GAUSS_KRONROD: e.push_back(std::string("hipadj::k_wide_adjoint") + (ip_ckpt ? "_ck<" : "<") + U + ", 4>"); break;
    case HIPADJ_ALG_BACKSOLVE: e.push_back("hipadj::k_wide_backsolve<" + U + ">"); break;
    case HIPADJ_ALG_QUADRATURE: e.push_back("hipadj::k_wide_quad_adj<" + U + ">"); e.push_back("hipadj::k_wide_quad_gk<" + U + ", " + std::to_string(HIPADJ_WIDE_MAXSEG) + ", false>"); break;
    default: break;
    }
    return e;
}

extern "C" int hipadj_model_check(int32_t model_id) {
    std::vector<char> code; std::map<std::string, std::string> low;
    if (user_model_is_wide(model_id)) {   // a wide model: both forward solves and every sweep of the family on both steppers, without a cost (cost variants and the
                                          // 160 KB LDS limit of the adaptive Interpolating / Backsolve sweeps are per configuration: hipadj_model_check_config)
        std::vector<std::string> all;
        for (bool ts5 : {false, true})
            for (int a : {HIPADJ_ALG_INTERPOLATING, HIPADJ_ALG_GAUSS, HIPADJ_ALG_GAUSS_KRONROD, HIPADJ_ALG_BACKSOLVE, HIPADJ_ALG_QUADRATURE}) {
                if (ts5 && (a == HIPADJ_ALG_INTERPOLATING || a == HIPADJ_ALG_BACKSOLVE) && user_wide_ts5_interp_lds(model_id, a == HIPADJ_ALG_BACKSOLVE) * 8 > 160L * 1024) continue;
                for (const auto& x : wide_kernel_names(a, ts5)) if (std::find(all.begin(), all.end(), x) == all.end()) all.push_back(x);
            }
        return user_compile(model_id, all, code, low, g_create_error);
    }
    std::vector<std::string> exprs = {"hipadj::k_forward<hipadj::UserModel>", user_has_cost(model_id) ? "hipadj::k_interp<hipadj::UserModel, 1, 7>" : "hipadj::k_interp<hipadj::UserModel, 1, 1>"};
    if (user_has_affect(model_id)) { exprs.push_back("hipadj::k_user_affect<hipadj::UserModel>"); exprs.push_back("hipadj::k_user_affect_vjp<hipadj::UserModel>"); }
    if (const char* e = std::getenv("HIPADJ_CHECK_EXPRS")) {   // debugging hook: further ';'-separated kernel instantiations (ISA studies with HIPADJ_RTC_DUMP)
        std::string t(e); size_t a = 0;
        while (a <= t.size()) { const size_t b = t.find(';', a); const std::string x = t.substr(a, b == std::string::npos ? std::string::npos : b - a); if (!x.empty()) exprs.push_back(x); if (b == std::string::npos) break; a = b + 1; }
    }
    return user_compile(model_id, exprs, code, low, g_create_error);
}

extern "C" int hipadj_runtime_compiler(char* buf, int32_t cap) {
    if (!buf || cap <= 0) return HIPADJ_ERR_INVALID_ARG;
    const std::string d = rtc_describe();
    std::snprintf(buf, (size_t)cap, "%s", d.c_str());
    RtcApi& A = rtc_api();
    return (A.lib && A.err.empty()) ? HIPADJ_OK : HIPADJ_ERR_UNSUPPORTED;
}

// Every kernel a handle of this configuration would launch, compiled now (no device needed): the ahead-of-time form of what
// hipadj_create does lazily.  Built-in models have nothing to compile.
struct UserKernels;
static int user_compile_config(const hipadj_config* cfg, std::string& err);
extern "C" int hipadj_model_check_config(const hipadj_config* cfg) {
    if (!cfg) { g_create_error = "cfg == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->model < HIPADJ_MODEL_USER_BASE) return HIPADJ_OK;
    if (cfg->cont_cost == HIPADJ_CCOST_MODEL && !user_has_cost(cfg->model)) { g_create_error = "cont_cost = HIPADJ_CCOST_MODEL but the model has no cost (hipadj_model_set_cost / hipadj_wmodel_set_cost)"; return HIPADJ_ERR_INVALID_ARG; }
    return user_compile_config(cfg, g_create_error);
}


#ifdef HIPADJ_WAVE_TRACE
// development builds (-DHIPADJ_WAVE_TRACE, scripts/r6/wave_trace.py): a device buffer of 24 time stamps per wave that the next one-launch reverse passes fill; never in the shipped library
unsigned long long* g_hipadj_wave_trace = nullptr;
extern "C" int hipadj_debug_set_trace(void* dev_ptr) { g_hipadj_wave_trace = (unsigned long long*)dev_ptr; return HIPADJ_OK; }
#endif

static void host_pin_release(hipadj_handle* h);
static void free_all(hipadj_handle* h) {
    void* ptrs[] = {h->d_u0, h->d_p, h->d_outT, h->d_yT, h->d_ckpt, h->d_cotT, h->d_segbuf, h->d_dp_traj, h->d_qres, h->d_qa,
                    h->d_qb, h->d_partial, h->d_io_a, h->d_du0, h->d_dp, h->d_knots, h->d_adj, h->d_fknots, h->d_fadj, h->d_c1, h->d_ticket, h->d_prev_ck, h->d_save_of_knot,
                    h->d_ckpt_of_knot, h->d_seg_bounds, h->d_flag, h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->d_gtile, h->d_rec, h->d_save_t, h->d_ck_t, h->d_tstops, h->d_nsteps, h->d_arec, h->d_nsteps_adj, h->d_mq_pool, h->d_mq_norm, h->d_mq_panels, h->d_mq_ids, h->d_tbuf, h->d_tcnt, h->d_fev_knot, h->d_fev_save, h->d_fev_ckpt, h->d_wscr, h->d_ldata, h->d_lpart, h->d_lval, h->d_og_i, h->d_og_h, h->d_og_tile};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->d_ev_s) (void)hipFree(h->d_ev_s);
    if (h->d_nev) (void)hipFree(h->d_nev);
    if (h->d_ev_k) (void)hipFree(h->d_ev_k);
    if (h->d_ev_t) (void)hipFree(h->d_ev_t);
    if (h->d_ev_ul) (void)hipFree(h->d_ev_ul);
    if (h->d_ev_ur) (void)hipFree(h->d_ev_ur);
    if (h->d_ev_dl) (void)hipFree(h->d_ev_dl);
    if (h->d_ev_dr) (void)hipFree(h->d_ev_dr);
    if (h->d_save_rev && h->d_save_rev != h->d_save_of_knot) (void)hipFree(h->d_save_rev);
    if (h->umod) (void)hipModuleUnload(h->umod);
    if (h->umod_alt) (void)hipModuleUnload(h->umod_alt);
    host_pin_release(h);
    if (h->lmod) (void)hipModuleUnload(h->lmod);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < h->xfer_nev; ++i) if (h->xfer_ev[i]) (void)hipEventDestroy(h->xfer_ev[i]);
    h->xfer_nev = 0;
    for (auto& q : h->evs) for (hipEvent_t e : {q.a0, q.a1, q.k0, q.k1}) if (e) (void)hipEventDestroy(e);
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); if (h->comm_ready) (void)hipEventDestroy(h->comm_ready); for (auto e : h->comm_done) if (e) (void)hipEventDestroy(e); }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

extern "C" int hipadj_create(const hipadj_config* cfg, hipadj_handle** out) {
    if (!out) { g_create_error = "out == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!cfg) { g_create_error = "cfg == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    // before ANY other field is read: a caller built against another ABI hands over a struct of another size (reading sizeof(hipadj_config) bytes from it would run past its end)
    if (cfg->struct_size != sizeof(hipadj_config)) { g_create_error = "hipadj_config.struct_size mismatch (ABI)"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ndevices < 0) { g_create_error = "hipadj_config.ndevices must be >= 0"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ndevices > 1) return multi_create(cfg, out, g_create_error);      // one handle over several devices: G ordinary handles on contiguous trajectory ranges (hipadj_multi.hpp)
    if (cfg->ndevices == 1 && cfg->device_ids) { hipadj_config c1 = *cfg; c1.device = cfg->device_ids[0]; c1.ndevices = 0; c1.device_ids = nullptr; return hipadj_create(&c1, out); }
    if (cfg->family != HIPADJ_FAMILY_AUTO && cfg->family != HIPADJ_FAMILY_AS_REGISTERED) { g_create_error = "hipadj_config.family must be HIPADJ_FAMILY_AUTO or HIPADJ_FAMILY_AS_REGISTERED"; return HIPADJ_ERR_INVALID_ARG; }
    { int H = 0;      // a declared dense chain of the shape the FP64-MFMA family is built for runs there (hipadj_route.hpp); anything that family refuses continues below
      if (route_eligible(cfg, H)) { const int rrc = route_create(cfg, H, out); if (rrc != HIPADJ_OK || *out) return rrc; } }
    auto* h = new hipadj_handle();
    auto fail = [&](int code) { g_create_error = h->err; free_all(h); delete h; return code; };
    h->cfg = *cfg; h->cfg.save_times = nullptr; h->cfg.checkpoints = nullptr;
    Plan P;
    { const int prc = make_plan(cfg, P, h->err); if (prc != HIPADJ_OK) return fail(prc); }
    const int n = P.n, np = P.np; const long S = P.S;
    h->n = n; h->np = np; h->N = P.N; h->Npad = P.Npad; h->S = P.S; h->M = P.M; h->nck = P.nck; h->nseg = P.nseg; h->nq = P.nq;
    h->save_times = P.save_times; h->save_of_knot = P.save_of_knot; h->ckpt_of_knot = P.ckpt_of_knot; h->seg_bounds = P.seg_bounds;
    const bool bs_ckpt = P.bs_ckpt || P.ip_ckpt || P.og_ck;   // (og_ck: GaussKronrod over the reverse step list is not an ip_ckpt configuration of the lane family, its checkpoint states are needed all the same)
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
    h->adaptive = P.adaptive && !P.wide;
    if (P.adaptive && !P.wide) {
        const int RW = 2 + 5 * n;   // record width, hipadj_adaptive.hpp
        A(dev_alloc(h, &h->d_outT, (size_t)h->M * n * Np));
        A(dev_alloc(h, &h->d_yT, (size_t)n * Np));
        h->auto_steps = cfg->max_steps == 0;   // capacity follows the measured step counts (adaptive_autosize); else the caller's bound
        if (cfg->alg != HIPADJ_ALG_BACKSOLVE) {
            // automatic sizing starts from what 1 GiB of records holds (between 128 and 1024 accepted steps per trajectory; 10^4 Lorenz trajectories:
            // 785): the first forward solve of a handle then stands without the regrow-and-repeat round — a stream synchronisation, hipFree + hipMalloc
            // and a second pass, 23 ms against 0.5 ms of steady state for the 10^4-trajectory Lorenz ensemble at the default tolerances
            long cap0 = (1L << 30) / ((long)RW * 8 * Np);
            cap0 = cap0 < 128 ? 128 : (cap0 > 1024 ? 1024 : cap0);
            if (P.ip_ckpt) cap0 = 128;                                            // checkpointing=true: one interval per lane
            if (const char* e = std::getenv("HIPADJ_REC_CAP0")) { const long v = std::atol(e); if (v > 0) cap0 = v; }   // test hook: start the forward record too small
            h->rec_cap = h->auto_steps ? cap0 : (P.ip_ckpt ? P.SmaxI : P.Smax);
            A(dev_alloc(h, &h->d_rec, (size_t)h->rec_cap * RW * Np));
        }
        A(dev_alloc(h, &h->d_nsteps, (size_t)Np));
        h->maxev = plan_user_events(cfg->model);      // a ContinuousCallback: the event lists of the trajectories (written by the forward kernel, read by the reverse kernel)
        if (h->maxev > 0) {
            A(dev_alloc(h, &h->d_ev_s, (size_t)h->maxev * Np)); A(dev_alloc(h, &h->d_nev, (size_t)Np)); A(dev_alloc(h, &h->d_ev_k, (size_t)h->maxev * Np));
            A(dev_alloc(h, &h->d_ev_t, (size_t)h->maxev * Np)); A(dev_alloc(h, &h->d_ev_ul, (size_t)h->maxev * n * Np)); A(dev_alloc(h, &h->d_ev_ur, (size_t)h->maxev * n * Np));
            if (rc == HIPADJ_OK && !HT(hipMemset(h->d_nev, 0, sizeof(int) * (size_t)Np), "memset")) rc = HIPADJ_ERR_HIP;
        }
        if (cfg->alg == HIPADJ_ALG_QUADRATURE) {   // dense adjoint solution: the reverse solve also stops at every loss time
            A(dev_alloc(h, &h->d_nsteps_adj, (size_t)Np));
            h->SmaxA = 2 * (h->auto_steps ? (int)h->rec_cap : P.Smax) + h->M + 16;
            if (const char* e = std::getenv("HIPADJ_SMAXA")) { const int v = std::atoi(e); if (v > 0) h->SmaxA = v; }   // test hook: start the dense adjoint record too small
            h->ag.SmaxA = h->SmaxA;
            A(dev_alloc(h, &h->d_arec, (size_t)h->SmaxA * RW * Np));
        }
        if (P.nck > 0) { A(dev_alloc(h, &h->d_ckpt, (size_t)P.nck * n * Np)); A(dev_alloc(h, &h->d_ck_t, (size_t)P.nck)); }
        if (h->M > 0) A(dev_alloc(h, &h->d_save_t, (size_t)h->M));
        h->ntstops = (int)P.tstops_desc.size();
        if (h->ntstops > 0) A(dev_alloc(h, &h->d_tstops, (size_t)h->ntstops));
        if (cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT && h->M > 0) A(dev_alloc(h, &h->d_cotT, (size_t)h->M * n * Np));   // cotangents per pass, or the data block of a device-resident loss (hipadj_set_loss_data)
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
        ag.maxev = h->maxev; ag.ev_s = h->d_ev_s; ag.nev = h->d_nev; ag.ev_t = h->d_ev_t; ag.ev_ul = h->d_ev_ul; ag.ev_ur = h->d_ev_ur; ag.ev_k = h->d_ev_k;
    } else if (P.wide) {
        // workgroup-per-trajectory family of runtime models (hipadj_wide.hpp): trajectory-major knots, Backsolve checkpoints, Quadrature records
        h->wide = true;
        h->wg.N = h->N; h->wg.S = (int)S; h->wg.M = h->M; h->wg.nck = h->nck; h->wg.t0 = cfg->t0; h->wg.dt = cfg->dt; h->wg.loss_shift = cfg->loss_shift;
        h->wg.loss_kind = cfg->loss_kind; h->wg.no_start = cfg->no_start; h->wg.p_shared = cfg->p_shared;
        if (P.adaptive) {
            // adaptive Tsit5: trajectory-major dense records.  max_steps = 0: the capacity is what 8 GiB of records hold (of 288 GB), between 64 and 8192
            // accepted steps per trajectory; when the budget cut the capacity below 8192 steps the step counts are read back after every forward solve and the record
            // is regrown on overflow (wide_autosize: one stream synchronisation per forward solve, in that mode only)
            h->wide_ts5 = true;
            const long RW = 2 + 5L * n;
            long budget = 8L << 30;
            if (const char* e = std::getenv("HIPADJ_WIDE_REC_BUDGET")) { const long v = std::atol(e); if (v > 0) budget = v; }   // test hook: start the record too small (regrow path)
            long cap = cfg->max_steps > 0 ? cfg->max_steps : budget / (RW * 8 * h->N);
            if (cfg->max_steps == 0) cap = cap < 64 ? 64 : (cap > 8192 ? 8192 : cap);
            h->rec_cap = cap;
            h->wa.t1 = cfg->t1; h->wa.abstol = cfg->abstol; h->wa.reltol = cfg->reltol; h->wa.dt0 = cfg->dt; h->wa.Smax = (int)cap; h->wa.maxit = (int)cap;
            if (cfg->max_steps == 0) h->wa.maxit = HIPADJ_AUTO_MAXITERS;   // auto-sized: only the reference's maxiters bounds the solve (Backsolve holds no records; the others report
                                                                           // the TRUE step count beyond the capacity, from which wide_autosize regrows the record)
            h->wide_auto = cfg->max_steps == 0 && cfg->alg != HIPADJ_ALG_BACKSOLVE && cap < 8192;   // the 8 GiB budget cut the capacity short: overflow is plausible, check after every forward solve
            h->ag.Smax = (int)cap;   // (the overflow message names it)
            if (P.ip_ckpt) {   // checkpointing = true for Interpolating / Gauss / GaussKronrod (k_wide_adjoint_ts5<., ., true>): the forward states at the checkpoint times and ONE
                               // interval's records per trajectory, re-solved inside the sweep (capacity P.SmaxI steps; an interval that needs more is reported, flag bit 4)
                h->wide_auto = false;
                h->wg.nck = P.nck; h->wide_SmaxI = P.SmaxI;
                A(dev_alloc(h, &h->d_rec, (size_t)h->N * P.SmaxI * RW));
                A(dev_alloc(h, &h->d_ckpt, (size_t)h->N * P.nck * n)); A(dev_alloc(h, &h->d_ck_t, (size_t)P.nck));
                if (rc == HIPADJ_OK && !HT(hipMemcpy(h->d_ck_t, P.ck_times.data(), sizeof(double) * P.nck, hipMemcpyHostToDevice), "memcpy")) rc = HIPADJ_ERR_HIP;
            } else if (cfg->alg != HIPADJ_ALG_BACKSOLVE) A(dev_alloc(h, &h->d_rec, (size_t)h->N * cap * RW));
            else {   // Backsolve keeps no records: y(T) and, checkpointing = true, the forward states at the checkpoint times [N][nck][n]
                A(dev_alloc(h, &h->d_yT, (size_t)h->N * n));
                h->wg.nck = P.nck;
                if (P.nck > 0) {
                    A(dev_alloc(h, &h->d_ckpt, (size_t)h->N * P.nck * n)); A(dev_alloc(h, &h->d_ck_t, (size_t)P.nck));
                    if (rc == HIPADJ_OK && !HT(hipMemcpy(h->d_ck_t, P.ck_times.data(), sizeof(double) * P.nck, hipMemcpyHostToDevice), "memcpy")) rc = HIPADJ_ERR_HIP;
                }
            }
            A(dev_alloc(h, &h->d_nsteps, (size_t)h->N));
            if (cfg->alg == HIPADJ_ALG_QUADRATURE) {   // dense adjoint solution (the reverse solve also stops at every loss time) + the quadrature workspaces
                h->SmaxA = 2 * (int)cap + h->M + 16;
                A(dev_alloc(h, &h->d_arec, (size_t)h->N * h->SmaxA * RW));
                A(dev_alloc(h, &h->d_nsteps_adj, (size_t)h->N));
                A(dev_alloc(h, &h->d_qres, (size_t)h->N * (h->nq > 0 ? h->nq : 1) * np));
                A(dev_alloc(h, &h->d_wscr, (size_t)h->N * (h->nq > 0 ? h->nq : 1) * (3 + HIPADJ_WIDE_MAXSEG) * np));
            }
            if (h->M > 0) A(dev_alloc(h, &h->d_save_t, (size_t)h->M));
            h->ntstops = (int)P.tstops_desc.size(); h->wa.ntstops = h->ntstops;
            if (h->ntstops > 0) A(dev_alloc(h, &h->d_tstops, (size_t)h->ntstops));
            if (rc == HIPADJ_OK) {
                bool ok2 = true;
                if (h->M > 0) ok2 = ok2 && HT(hipMemcpy(h->d_save_t, P.save_times.data(), sizeof(double) * h->M, hipMemcpyHostToDevice), "memcpy");
                if (h->ntstops > 0) ok2 = ok2 && HT(hipMemcpy(h->d_tstops, P.tstops_desc.data(), sizeof(double) * h->ntstops, hipMemcpyHostToDevice), "memcpy");
                if (!ok2) rc = HIPADJ_ERR_HIP;
            }
        } else if (P.ip_ckpt) {   // checkpointing = true for Interpolating / Gauss / GaussKronrod (k_wide_adjoint_ck): the states at the checkpoint knots + one re-solve tile per trajectory
            h->wide_KT = P.ck_longest + 1;
            A(dev_alloc(h, &h->d_fknots, (size_t)h->N * h->wide_KT * 2 * n));
            A(dev_alloc(h, &h->d_ckpt, (size_t)h->N * h->nck * n));
        } else if (cfg->alg != HIPADJ_ALG_BACKSOLVE) A(dev_alloc(h, &h->d_fknots, (size_t)h->N * (S + 1) * 2 * n));
        else {
            A(dev_alloc(h, &h->d_yT, (size_t)h->N * n));
            if (h->nck > 0) A(dev_alloc(h, &h->d_ckpt, (size_t)h->N * h->nck * n));
            if (P.offgrid) A(dev_alloc(h, &h->d_fknots, (size_t)h->N * (S + 1) * 2 * n));   // off-grid Backsolve: out = sol(ts) and the checkpoint states are interpolated from the forward knots
        }
        if (cfg->alg == HIPADJ_ALG_GAUSS_KRONROD) A(dev_alloc(h, &h->d_wscr, (size_t)h->N * 3 * np));   // per trajectory: integrand, Kronrod and Gauss rows of one panel
        if (cfg->alg == HIPADJ_ALG_QUADRATURE && !P.adaptive) {
            A(dev_alloc(h, &h->d_fadj, (size_t)h->N * (P.offgrid ? P.rs_t.size() : (size_t)S) * 4 * n));   // one record per (reverse) step
            A(dev_alloc(h, &h->d_qres, (size_t)h->N * (h->nq > 0 ? h->nq : 1) * np));
            A(dev_alloc(h, &h->d_wscr, (size_t)h->N * (h->nq > 0 ? h->nq : 1) * (3 + HIPADJ_WIDE_MAXSEG) * np));
        }
    } else if (!P.field && !P.mlp) {
        A(dev_alloc(h, &h->d_outT, (size_t)h->M * n * Np));
        A(dev_alloc(h, &h->d_yT, (size_t)n * Np));
        if ((cfg->alg != HIPADJ_ALG_BACKSOLVE && !P.ip_ckpt) || P.offgrid) A(dev_alloc(h, &h->d_knots, (size_t)(S + 1) * n * Np));   // off-grid Backsolve: the checkpoint states are interpolated from the knots
        if (bs_ckpt) A(dev_alloc(h, &h->d_ckpt, (size_t)h->nck * n * Np));
        if (cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT && h->M > 0) A(dev_alloc(h, &h->d_cotT, (size_t)h->M * n * Np));   // cotangents per pass, or the data block of a device-resident loss (hipadj_set_loss_data)
        A(dev_alloc(h, &h->d_segbuf, (size_t)h->nseg * (1 + n) * (n + np) * Np));
        if (!P.user) {   // event knots of the forward solve (k_forward_ev): save times, checkpoints, the knot in front of a shortened last step
            if (const char* e = std::getenv("HIPADJ_FWD_EV")) h->fwd_ev = std::atoi(e);
            std::vector<int> ek, es, ec;
            forward_events(P, cfg->dt, ek, es, ec);
            h->nfev = (int)ek.size();
            const size_t ne = ek.size() + 1;   // never a zero-sized allocation
            ek.push_back(0); es.push_back(-1); ec.push_back(-1);
            A(dev_alloc(h, &h->d_fev_knot, ne)); A(dev_alloc(h, &h->d_fev_save, ne)); A(dev_alloc(h, &h->d_fev_ckpt, ne));
            if (rc == HIPADJ_OK && !(HT(hipMemcpy(h->d_fev_knot, ek.data(), sizeof(int) * ne, hipMemcpyHostToDevice), "memcpy") &&
                                     HT(hipMemcpy(h->d_fev_save, es.data(), sizeof(int) * ne, hipMemcpyHostToDevice), "memcpy") &&
                                     HT(hipMemcpy(h->d_fev_ckpt, ec.data(), sizeof(int) * ne, hipMemcpyHostToDevice), "memcpy"))) rc = HIPADJ_ERR_HIP;
        }
        {   // the composition tree of the one-launch reverse pass (hipadj_fused.hpp): map slots per (block, node) and arrival counters
            if (const char* e = std::getenv("HIPADJ_FUSED")) h->fused = std::atoi(e);
            if (!fused_eligible(cfg, P)) h->fused = 0;
            if (P.user && !rtc_trusted() && !std::getenv("HIPADJ_FUSED")) h->fused = 0;   // a runtime model compiled by a toolkit other than the build's: the in-launch hand-offs rest on
                                                                                          // this compiler's code generation (hipadj_fused.hpp tree_arrive_last): three launches instead
            int radix = 4;
            if (const char* e = std::getenv("HIPADJ_TREE_RADIX")) { const int v = std::atoi(e); radix = (v == 8 || v == 16) ? v : 4; }
            // grouped form (k_interp_fused_g): chosen by the planner (plan_group_choice) for the stage-operator sweeps of BASELINE configs[1] and its shards
            h->fgroup = (h->fused && h->nseg > 1) ? P.fgroup : 0;
            if (h->fgroup && !std::getenv("HIPADJ_TREE_RADIX")) radix = P.tree_radix;
            long slots = 0, ctrs = 0;
            tree_plan_shape(h->fgroup ? (h->nseg + h->fgroup - 1) / h->fgroup : h->nseg, radix, (long)(Np / 64), h->tp, &slots, &ctrs);
            h->tcnt_n = ctrs;
            if (h->tp.nlev == 0) slots = 1;    // a single segment: the root wave finishes alone, no slot is ever written
            const size_t slot_doubles = (size_t)(((1 + n) * (n + np) + 1) / 2) * 128;   // rows of 64 x 16 bytes
            if ((double)slots * (double)slot_doubles * 8.0 >= 2147483648.0) h->fused = 0;        // beyond the buffer descriptor's range
            h->tp.tbuf_bytes = (long)(slots * slot_doubles * 8);
            if (h->fused) {
                A(dev_alloc(h, &h->d_tbuf, (size_t)slots * slot_doubles));
                A(dev_alloc(h, &h->d_tcnt, (size_t)ctrs));
                if (rc == HIPADJ_OK && !HT(hipMemset(h->d_tcnt, 0, sizeof(unsigned) * (size_t)ctrs), "memset")) rc = HIPADJ_ERR_HIP;
            }
        }
        if (P.ip_ckpt && P.ck_longest > HIPADJ_CKPT_KMAX) {   // one re-solve tile [longest + 1][n][64] per (wave, segment) in HBM
            h->gtile_stride = (long)(P.ck_longest + 1) * n * 64;
            A(dev_alloc(h, &h->d_gtile, (size_t)h->gtile_stride * (size_t)(Np / 64) * (size_t)h->nseg));
        }
    } else if (P.mlp) {
        h->mlp = true; h->field = false; h->NQ = P.NQ;
        const size_t Bb = cfg->dims[2];
        A(dev_alloc(h, &h->d_fknots, (size_t)h->N * (S + 1) * 2 * n));
        if (cfg->alg == HIPADJ_ALG_QUADRATURE) {
            // dense adjoint record of pass 1 + the buffers of the host-driven adaptive Gauss-Kronrod pass (mlp_quadrature, hipadj_host_impl.hpp)
            const size_t wg = Bb / 16, per_entry = wg * (size_t)np;
            size_t chunk = ((size_t)768 << 20) / (per_entry * sizeof(double)); chunk = chunk < 2 ? 2 : (chunk > 64 ? 64 : chunk); chunk &= ~(size_t)1;
            h->mq_chunk = (int)chunk;
            h->mq_pool_cap = 2 * (long)h->N * P.nq + 64;
            A(dev_alloc(h, &h->d_fadj, (size_t)h->N * S * 4 * n));
            A(dev_alloc(h, &h->d_c1, chunk * per_entry));
            A(dev_alloc(h, &h->d_mq_pool, (size_t)h->mq_pool_cap * np));
            A(dev_alloc(h, &h->d_mq_norm, (size_t)4 * 4096));
            A(dev_alloc(h, &h->d_mq_ids, (size_t)1 << 16));
            { double* tmp = nullptr; A(dev_alloc(h, &tmp, (size_t)4096 * 3)); h->d_mq_panels = tmp; }      // 4096 panels of 24 bytes
            h->qa_host = P.qa; h->qb_host = P.qb;
        } else {
            // in-register parameter gradient (hipadj_mlp_grad.hpp): one partial gradient per workgroup of 16 columns, no activation records
            A(dev_alloc(h, &h->d_c1, (size_t)h->N * (Bb / 16) * (size_t)np));
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
    h->offgrid = P.offgrid; h->ck_long = P.ck_longest > HIPADJ_CKPT_KMAX;
    if (P.offgrid) {   // reverse step list of the off-grid sweep + the save times for out = sol(ts)
        h->nrs = (int)P.rs_t.size(); h->rs_save_at_start = P.rs_save_at_start;
        A(dev_alloc(h, &h->d_rs_t, (size_t)h->nrs)); A(dev_alloc(h, &h->d_rs_h, (size_t)h->nrs)); A(dev_alloc(h, &h->d_rs_te, (size_t)h->nrs));
        A(dev_alloc(h, &h->d_rs_save, (size_t)h->nrs)); A(dev_alloc(h, &h->d_save_t, (size_t)h->M));
        if (P.nck > 0) {   // Backsolve: checkpoint times + the slot that replaces y at the end of each reverse step
            A(dev_alloc(h, &h->d_ck_t, (size_t)P.nck)); A(dev_alloc(h, &h->d_rs_ck, (size_t)h->nrs));
            if (rc == HIPADJ_OK && !(HT(hipMemcpy(h->d_ck_t, P.ck_times.data(), sizeof(double) * P.nck, hipMemcpyHostToDevice), "memcpy") &&
                                     HT(hipMemcpy(h->d_rs_ck, P.rs_ck.data(), sizeof(int) * h->nrs, hipMemcpyHostToDevice), "memcpy"))) rc = HIPADJ_ERR_HIP;
        }
        if (rc == HIPADJ_OK && !(HT(hipMemcpy(h->d_rs_t, P.rs_t.data(), sizeof(double) * h->nrs, hipMemcpyHostToDevice), "memcpy") &&
                                 HT(hipMemcpy(h->d_rs_h, P.rs_h.data(), sizeof(double) * h->nrs, hipMemcpyHostToDevice), "memcpy") &&
                                 HT(hipMemcpy(h->d_rs_te, P.rs_te.data(), sizeof(double) * h->nrs, hipMemcpyHostToDevice), "memcpy") &&
                                 HT(hipMemcpy(h->d_rs_save, P.rs_save.data(), sizeof(int) * h->nrs, hipMemcpyHostToDevice), "memcpy") &&
                                 HT(hipMemcpy(h->d_save_t, P.save_times.data(), sizeof(double) * h->M, hipMemcpyHostToDevice), "memcpy"))) rc = HIPADJ_ERR_HIP;
    }
    if (P.og_ck) {   // checkpointing = true over the reverse step list: the interval table and one knot tile per lane
        h->og_ck = true; h->og_nint = (int)P.og_S.size();
        std::vector<int> tab; tab.insert(tab.end(), P.og_S.begin(), P.og_S.end()); tab.insert(tab.end(), P.og_qlo.begin(), P.og_qlo.end()); tab.insert(tab.end(), P.og_qhi.begin(), P.og_qhi.end());
        A(dev_alloc(h, &h->d_og_i, tab.size())); A(dev_alloc(h, &h->d_og_h, (size_t)h->og_nint)); A(dev_alloc(h, &h->d_og_tile, (size_t)P.og_tile_knots * n * Np));
        if (rc == HIPADJ_OK && !(HT(hipMemcpy(h->d_og_i, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice), "memcpy") &&
                                 HT(hipMemcpy(h->d_og_h, P.og_hlast.data(), sizeof(double) * h->og_nint, hipMemcpyHostToDevice), "memcpy"))) rc = HIPADJ_ERR_HIP;
        if (rc == HIPADJ_OK && (!h->d_ckpt || !h->d_ck_t || !h->d_knots)) { h->err = "internal: the checkpointed sweep over the reverse step list lacks its checkpoint states, times or forward knots"; rc = HIPADJ_ERR_STATE; }
    }
    const std::vector<double>&qa = P.qa, &qb = P.qb;
    if (cfg->alg == HIPADJ_ALG_QUADRATURE) {
        if (!h->field && !h->mlp && !h->wide) A(dev_alloc(h, &h->d_adj, (size_t)(P.offgrid ? P.rs_t.size() : (size_t)S) * 2 * n * Np));   // one record per reverse step
        if (!h->mlp && !h->wide) A(dev_alloc(h, &h->d_qres, (size_t)h->nq * np * Np));
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
    h->d_save_rev = h->d_save_of_knot;
    if (P.save_of_knot_rev != P.save_of_knot) {   // no_start suppresses the jump at T: the reverse kernels get their own map
        h->d_save_rev = nullptr;
        if (dev_alloc(h, &h->d_save_rev, (size_t)S + 1) != HIPADJ_OK) return fail(HIPADJ_ERR_HIP);
        if (!HT(hipMemcpy(h->d_save_rev, P.save_of_knot_rev.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice), "memcpy")) return fail(HIPADJ_ERR_HIP);
    }

    Geom& g = h->g;
    g.N = h->N; g.Npad = Np; g.S = (int)S; g.M = h->M; g.t0 = cfg->t0; g.dt = cfg->dt; g.loss_shift = cfg->loss_shift;
    g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start; g.p_shared = cfg->p_shared;
    g.kmask = -1; g.h_last = P.h_last;
#ifdef HIPADJ_PRIO_TOGGLE
    g.prio_phase = -1;      // only the grouped one-launch pass sets a phase (k_interp_fused_g)
#endif
    // dgdu = la u + lb c for the kinds that stream a column c: cotangents and model bodies take c itself, HIPADJ_LOSS_LSQ_DATA w (u - c)
    const double lw = cfg->loss_scale != 0.0 ? cfg->loss_scale : 1.0;
    g.la = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? lw : 0.0; g.lb = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? -lw : 1.0;
    const bool gauss_like = cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD;
    g.lflags = (cfg->reference_literal && gauss_like) ? 3 : 0;   // bit 0: the reference's GaussAdjoint drops dgdp_discrete (src/adjoint_common.jl:776 `!isq`; nothing in src/gauss_adjoint.jl
                                                                 // adds it); bit 1: its g_p term of a continuous cost as src/gauss_adjoint.jl:753-758 is written
    if (cfg->reference_literal && gauss_like && (P.wide || P.field || P.mlp) && cfg->cont_cost != HIPADJ_CCOST_NONE) { h->err = "reference_literal (the g_p sign of src/gauss_adjoint.jl:753-758) is offered for the lane-per-trajectory models"; return fail(HIPADJ_ERR_UNSUPPORTED); }
    h->ag.la = g.la; h->ag.lb = g.lb; h->ag.lflags = g.lflags;
    h->wg.la = g.la; h->wg.lb = g.lb; h->wg.lflags = g.lflags;
    h->fg.lsq_w = lw; h->mg.lsq_w = lw;
    if (h->d_cotT && (double)n * (double)Np * 8.0 >= 2147483648.0) { h->err = "the streamed cotangent / data column of one loss time (n x Npad doubles) must stay below 2 GiB per handle (buffer-descriptor range): shard the ensemble"; return fail(HIPADJ_ERR_UNSUPPORTED); }
    if (h->d_cotT && hipMemset(h->d_cotT, 0, sizeof(double) * (size_t)h->M * n * Np) != hipSuccess) { h->err = "hipMemset failed"; return fail(HIPADJ_ERR_HIP); }
    if (const char* e = std::getenv("HIPADJ_TIMING")) h->timing = std::atoi(e);
    {
        int cus = 256, mode = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || cus <= 0) cus = 256;
        if (const char* e = std::getenv("HIPADJ_QUAD")) mode = std::atoi(e);
        const long simds = 4L * cus;
        h->quad_fwd = mode == 2 || (mode == 1 && h->N <= 16 * simds);
        h->quad_adj = mode == 2 || (mode == 1 && h->N <= 32 * simds);
    }
    if (const char* e = std::getenv("HIPADJ_FUSED_FINAL")) h->fused_final = std::atoi(e);
    if (const char* e = std::getenv("HIPADJ_WPB")) h->wpb4 = std::atoi(e) == 4;
    if (const char* e = std::getenv("HIPADJ_NO_OPS")) h->no_ops = std::atoi(e) != 0;
    if (const char* e = std::getenv("HIPADJ_CBS")) h->cbs = std::atoi(e);
    h->fg.N = h->N; h->fg.S = (int)S; h->fg.M = h->M; h->fg.t0 = cfg->t0; h->fg.dt = cfg->dt; h->fg.loss_shift = cfg->loss_shift;
    h->fg.loss_kind = cfg->loss_kind; h->fg.no_start = cfg->no_start; h->fg.p_shared = cfg->p_shared; h->fg.cont_cost = cfg->cont_cost;
    h->mg.N = h->N; h->mg.B = cfg->dims[2]; h->mg.S = (int)S; h->mg.M = h->M; h->mg.t0 = cfg->t0; h->mg.dt = cfg->dt; h->mg.loss_shift = cfg->loss_shift;
    h->mg.loss_kind = cfg->loss_kind; h->mg.no_start = cfg->no_start; h->mg.p_shared = cfg->p_shared; h->mg.NQ = P.NQ;

    h->st.struct_size = sizeof(hipadj_stats); h->st.n = n; h->st.np = np; h->st.ntraj = h->N; h->st.nsteps = S;
    h->st.time_segments = h->nseg; h->st.workspace_bytes = h->ws_bytes;
    h->st.launches_per_pass = (h->fused && h->d_tbuf) ? 1 : ((!P.field && !P.mlp && !P.adaptive && !P.wide && cfg->alg != HIPADJ_ALG_QUADRATURE) ? 3 : 0);
    if (P.wide) h->st.launches_per_pass = (cfg->alg == HIPADJ_ALG_QUADRATURE ? 2 : 1) + (cfg->p_shared ? (h->N > 64 ? 2 : 1) : 0);   // the sweep (+ the GK15 pass) + the one- or two-level k_wide_reduce_dp for shared parameters
    // ALGORITHMIC bytes of one reverse pass: knots (u,f) once, cotangents (if read), du0 + dp out
    double bytes = 0.0;
    if (P.adaptive) bytes = 0.0;   // data-dependent (accepted steps per trajectory): not modelled
    else if (cfg->alg == HIPADJ_ALG_BACKSOLVE || P.ip_ckpt) bytes = (double)h->N * ((double)h->nck * 8.0 * n + 8.0 * n);
    else bytes = (double)h->N * (double)(S + 1) * 16.0 * n;
    if (cfg->alg == HIPADJ_ALG_QUADRATURE) bytes += (double)h->N * (double)S * 2.0 * 32.0 * n;   // dense lambda write + read
    if (cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT) bytes += (double)h->N * h->M * 8.0 * n;   // the cotangent / data block is read once
    bytes += (double)h->N * 8.0 * (n + np);
    h->st.adjoint_algorithmic_bytes = bytes;
    h->st.vjp_steps = (double)h->N * (double)S * 4.0;
    h->user = P.user;
    if (cfg->cont_cost == HIPADJ_CCOST_MODEL && !user_has_cost(cfg->model)) { h->err = "cont_cost = HIPADJ_CCOST_MODEL but the model has no cost (hipadj_model_set_cost / hipadj_wmodel_set_cost)"; return fail(HIPADJ_ERR_INVALID_ARG); }
    if (cfg->loss_kind == HIPADJ_LOSS_MODEL && !user_has_dloss(cfg->model)) { h->err = "loss_kind = HIPADJ_LOSS_MODEL but the model has no discrete-loss bodies (hipadj_model_set_discrete_loss / hipadj_wmodel_set_discrete_loss)"; return fail(HIPADJ_ERR_INVALID_ARG); }
    if (P.wide) { const int urc = wide_prepare(h); if (urc != HIPADJ_OK) return fail(urc); }
    else if (P.user) { const int urc = user_prepare(h); if (urc != HIPADJ_OK) return fail(urc); h->has_mm = user_mass_matrix_inverse(cfg->model, h->minv); h->dae = user_model_is_dae(cfg->model); }
    // Everything above — the zero fills of the workspaces (d_cotT, tickets, flags) and the uploads of the plan tables — was issued on the NULL stream, which does not order with
    // the handle's non-blocking stream: a hipMemset of device memory may still be pending when the first call on the handle's stream writes the same buffer.  Seen (round 6): with
    // other processes loading the GPU, the data block of a device-resident loss (hipadj_set_loss_data right after hipadj_create: transposed into d_cotT on the handle's stream) was
    // wiped by the create-time hipMemset(d_cotT) that landed later — gradients of an all-zero data block, 323 times in 12 000 handles under load, never on a quiet device
    // (scripts/r6/loop_lsq_diag.py, profiles/r6_lsq_zero_data_race.jsonl).  Drain the device once here.
    static const bool no_drain = [] { const char* e = std::getenv("HIPADJ_CREATE_NO_DRAIN"); return e && e[0] == '1'; }();      // REPRODUCTION HOOK of the race above (tests / scripts only)
    if (!no_drain && hipDeviceSynchronize() != hipSuccess) { h->err = "hipDeviceSynchronize failed at the end of hipadj_create"; return fail(HIPADJ_ERR_HIP); }
    *out = h;
    return HIPADJ_OK;
}


// ---- dL/dp all-reduce over RCCL (hipadj_comm.hpp) ---------------------------------------------------------------
extern "C" int hipadj_comm_unique_id(char* id) {
    if (!id) { g_create_error = "id == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    RcclApi& A = rccl_api();
    if (!A.err.empty()) { g_create_error = A.err; return HIPADJ_ERR_RCCL; }
    RcclUniqueId u;
    const int rc = A.GetUniqueId(&u);
    if (rc != 0) { g_create_error = rccl_error("ncclGetUniqueId", rc); return HIPADJ_ERR_RCCL; }
    std::memcpy(id, u.internal, sizeof(u.internal));
    return HIPADJ_OK;
}

#define HIPADJ_NO_MULTI(h, what) do { if ((h)->multi) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, what ": not offered on a handle over several devices (hipadj_config.ndevices > 1): it already spans the node; shard across processes with one single-device handle per process") ; } while (0)
extern "C" int hipadj_comm_overlap(hipadj_handle* h, int on) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_overlap(h->inner, on); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    HIPADJ_NO_MULTI(h, "hipadj_comm_overlap");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (on && !h->comm_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->comm_ready, hipEventDisableTiming));
        for (auto& e : h->comm_done) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!on && h->comm_stream) HIP_TRY(h, hipStreamSynchronize(h->comm_stream));
    // switching it on AGAIN while collectives are in flight must not forget them: the sequence number — whose parity picks the comm_done event a pass
    // waits for — restarts only on a real off -> on transition, where the second stream has been drained by the `off` above
    if ((on ? 1 : 0) != h->comm_overlap) h->comm_seq = 0;
    h->comm_overlap = on ? 1 : 0;
    { const char* e = std::getenv("HIPADJ_TEST_COMM_DELAY"); h->comm_test_delay = (on && e) ? std::atol(e) : 0L; }   // test hook, see hipadj_adjoint_dev
    return HIPADJ_OK;
}

extern "C" int hipadj_comm_destroy(hipadj_handle* h) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_destroy(h->inner); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    if (h->multi) return HIPADJ_OK;
    if (h->comm_stream) { (void)hipSetDevice(h->cfg.device); (void)hipStreamSynchronize(h->comm_stream); }
    if (h->comm && h->comm_owned) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
        const int rc = rccl_api().CommDestroy(h->comm);
        h->comm = nullptr; h->comm_owned = false;
        if (rc != 0) { h->err = rccl_error("ncclCommDestroy", rc); return HIPADJ_ERR_RCCL; }
    }
    h->comm = nullptr; h->comm_owned = false;
    return HIPADJ_OK;
}

extern "C" int hipadj_comm_init_rank(hipadj_handle* h, const char* id, int nranks, int rank) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_init_rank(h->inner, id, nranks, rank); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    HIPADJ_NO_MULTI(h, "hipadj_comm_init_rank");
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_comm_init_rank: id != NULL and 0 <= rank < nranks required");
    if (!h->cfg.p_shared) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "per-trajectory parameters (p_shared = 0) have no cross-shard reduction: dp stays sharded like du0");
    RcclApi& A = rccl_api();
    if (!A.err.empty()) { h->err = A.err; return HIPADJ_ERR_RCCL; }
    TRY(hipadj_comm_destroy(h));
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    RcclUniqueId u;
    std::memcpy(u.internal, id, sizeof(u.internal));
    RcclComm c = nullptr;
    const int rc = A.CommInitRank(&c, nranks, u, rank);   // collective over the nranks processes
    if (rc != 0 || !c) { h->err = rccl_error("ncclCommInitRank", rc); return HIPADJ_ERR_RCCL; }
    h->comm = c; h->comm_owned = true;
    return HIPADJ_OK;
}

extern "C" int hipadj_comm_count(hipadj_handle* h, int* nranks) {
    if (!h || !nranks) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_count(h->inner, nranks); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    *nranks = 0;                                   // no communicator: the handle's dp is its shard's own sum
    if (h->multi || !h->comm) return HIPADJ_OK;
    RcclApi& A = rccl_api();
    if (!A.err.empty()) { h->err = A.err; return HIPADJ_ERR_RCCL; }
    const int rc = A.CommCount(h->comm, nranks);
    if (rc != 0) { h->err = rccl_error("ncclCommCount", rc); return HIPADJ_ERR_RCCL; }
    return HIPADJ_OK;
}

// Collective: every rank all-reduces the probe (1, rank + 1, 2^-rank) exactly as hipadj_adjoint(_dev) all-reduces dp (same
// communicator, stream, datatype and operator) and compares with what nranks ranks must produce.  A wrong binding of the RCCL
// ABI (datatype / operator enumerators, the by-value unique id) or a communicator that spans other ranks than the host believes
// shows up here, before a gradient is wrong.  Synchronises the handle's stream.
extern "C" int hipadj_comm_selfcheck(hipadj_handle* h) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_selfcheck(h->inner); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    HIPADJ_NO_MULTI(h, "hipadj_comm_selfcheck");
    if (!h->comm) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_comm_selfcheck: the handle has no communicator");
    RcclApi& A = rccl_api();
    if (!A.err.empty()) { h->err = A.err; return HIPADJ_ERR_RCCL; }
    int nranks = 0, rank = -1;
    int rc = A.CommCount(h->comm, &nranks);
    if (rc == 0 && A.CommUserRank) rc = A.CommUserRank(h->comm, &rank);
    if (rc != 0 || nranks < 1) { h->err = rccl_error("ncclCommCount / ncclCommUserRank", rc); return HIPADJ_ERR_RCCL; }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    double probe[3] = {1.0, (double)(rank + 1), std::ldexp(1.0, -(rank < 0 ? 0 : rank % 50))}, got[3] = {0, 0, 0};
    double* d = nullptr;
    HIP_TRY(h, hipMalloc(&d, sizeof(probe)));
    hipError_t e = hipMemcpyAsync(d, probe, sizeof(probe), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        rc = A.AllReduce(d, d, 3, RCCL_DOUBLE, RCCL_SUM, h->comm, h->stream);
        e = hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(d);
    if (rc != 0) { h->err = rccl_error("ncclAllReduce (self-check)", rc); return HIPADJ_ERR_RCCL; }
    HIP_TRY(h, e);
    double want2 = 0.0;
    for (int r = 0; r < nranks; ++r) want2 += std::ldexp(1.0, -(r % 50));
    const double want[3] = {(double)nranks, 0.5 * nranks * (nranks + 1.0), want2};
    if (rank < 0) { probe[1] = 0; }   // rank unknown (an attached communicator of an RCCL without ncclCommUserRank): only the count and the powers are checked
    const bool ok = got[0] == want[0] && (rank < 0 || got[1] == want[1]) && (rank < 0 || std::fabs(got[2] - want[2]) <= 1e-15 * want[2]);
    if (!ok) {
        char msg[256];
        std::snprintf(msg, sizeof msg, "hipadj_comm_selfcheck: all-reduce over %d ranks returned (%.17g, %.17g, %.17g), expected (%.17g, %.17g, %.17g)",
                      nranks, got[0], got[1], got[2], want[0], want[1], want[2]);
        h->err = msg;
        return HIPADJ_ERR_RCCL;
    }
    return HIPADJ_OK;
}

extern "C" int hipadj_comm_attach(hipadj_handle* h, void* nccl_comm) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { const int rc_ = hipadj_comm_attach(h->inner, nccl_comm); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    HIPADJ_NO_MULTI(h, "hipadj_comm_attach");
    if (nccl_comm && !h->cfg.p_shared) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "per-trajectory parameters (p_shared = 0) have no cross-shard reduction: dp stays sharded like du0");
    if (nccl_comm) { RcclApi& A = rccl_api(); if (!A.err.empty()) { h->err = A.err; return HIPADJ_ERR_RCCL; } }
    TRY(hipadj_comm_destroy(h));
    h->comm = nccl_comm; h->comm_owned = false;
    return HIPADJ_OK;
}

extern "C" int hipadj_destroy(hipadj_handle* h) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->multi) { (void)multi_synchronize(h); multi_free(h); delete h; return HIPADJ_OK; }
    if (h->route) { route_free(h); delete h; return HIPADJ_OK; }
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    // ... and every OTHER stream this handle may have enqueued work on (hipadj_set_stream moves it; a staged upload or a kernel of an earlier stream may still be in flight while
    // its buffers — the registered staging block among them — are released below): destroy is not a hot path, the device-wide wait is the simple guarantee
    (void)hipDeviceSynchronize();
    (void)hipadj_comm_destroy(h);
    free_all(h);
    delete h;
    return HIPADJ_OK;
}

extern "C" int hipadj_set_timing(hipadj_handle* h, int level) {
    if (!h || level < 0 || level > 2) return HIPADJ_ERR_INVALID_ARG;
    for (hipadj_handle* c : h->shards) c->timing = level;
    if (h->route) h->inner->timing = level;
    h->timing = level;
    return HIPADJ_OK;
}

extern "C" int hipadj_set_stream(hipadj_handle* h, void* s) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->route) { (void)hipadj_set_stream(h->inner, s); h->stream = h->inner->stream; return HIPADJ_OK; }
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return HIPADJ_OK;
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
    if (h->route) { const int rc_ = hipadj_synchronize(h->inner); if (rc_ != HIPADJ_OK) h->err = hipadj_last_error(h->inner); return rc_; }
    if (h->multi) return multi_synchronize(h);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIP_TRY(h, hipStreamSynchronize(h->comm_stream));       // an overlapped all-reduce of dp (hipadj_comm_overlap)
    harvest_timing(h, true);
    int flag = 0;
    HIP_TRY(h, hipMemcpy(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
        if (flag & 4) HIPADJ_FAIL(h, HIPADJ_ERR_MAXITERS, "the adaptive %s solve exceeded its step capacity (record capacity %d, step bound %d) on at least one trajectory, or a semi-explicit DAE found no consistent initial state, or a ContinuousCallback fired more often than the max_events it was registered with (raise max_steps or loosen tolerances)", h->cfg.stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE ? "Rosenbrock23" : "Tsit5", h->ag.Smax, h->ag.maxit);
        HIPADJ_FAIL(h, HIPADJ_ERR_NONFINITE, "non-finite sensitivities (flag %d): a trajectory diverged", flag);
    }
    return HIPADJ_OK;
}

// events per trajectory of the last forward solve of a handle whose model carries a ContinuousCallback (host pointer, [ntraj]; synchronous)
extern "C" int hipadj_event_counts(hipadj_handle* h, int32_t* counts) {
    if (!h || !counts) return HIPADJ_ERR_INVALID_ARG;
    if (h->multi || h->maxev <= 0 || !h->d_nev) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_event_counts: the handle's model carries no ContinuousCallback (hipadj_model_set_continuous_callback), or the handle spans several devices");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(counts, h->d_nev, sizeof(int32_t) * (size_t)h->N, hipMemcpyDeviceToHost));
    return HIPADJ_OK;
}

// save_positions = (true, true) of a ContinuousCallback: the event times and the states just before / after the affect of the last forward solve — t [ntraj][max_events],
// ul, ur [ntraj][max_events][n] (host pointers, any may be NULL; entries beyond a trajectory's event count are zero; synchronous)
extern "C" int hipadj_event_states(hipadj_handle* h, double* t, double* ul, double* ur) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->multi || h->maxev <= 0 || !h->d_nev) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_event_states: the handle's model carries no ContinuousCallback (hipadj_model_set_continuous_callback), or the handle spans several devices");
    if (!h->have_forward) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_event_states: no forward solve yet");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long Np = h->Npad; const int n = h->n, me = h->maxev;
    std::vector<int> ne((size_t)Np);
    HIP_TRY(h, hipMemcpy(ne.data(), h->d_nev, sizeof(int) * (size_t)Np, hipMemcpyDeviceToHost));
    std::vector<double> buf((size_t)me * n * Np);
    if (t) {
        HIP_TRY(h, hipMemcpy(buf.data(), h->d_ev_t, sizeof(double) * (size_t)me * Np, hipMemcpyDeviceToHost));
        for (long i = 0; i < h->N; ++i) for (int k = 0; k < me; ++k) t[i * me + k] = k < ne[i] ? buf[(size_t)k * Np + i] : 0.0;
    }
    for (int side = 0; side < 2; ++side) {
        double* dst = side ? ur : ul;
        if (!dst) continue;
        HIP_TRY(h, hipMemcpy(buf.data(), side ? h->d_ev_ur : h->d_ev_ul, sizeof(double) * (size_t)me * n * Np, hipMemcpyDeviceToHost));
        for (long i = 0; i < h->N; ++i) for (int k = 0; k < me; ++k) for (int j = 0; j < n; ++j) dst[((size_t)i * me + k) * n + j] = k < ne[i] ? buf[((size_t)k * n + j) * Np + i] : 0.0;
    }
    return HIPADJ_OK;
}
// which component of a VectorContinuousCallback fired at each event of the last forward solve (0 for a scalar condition; + 256: the event terminated the trajectory's solve):
// idx [ntraj][max_events], host pointer, -1 beyond a trajectory's event count; synchronous
extern "C" int hipadj_event_components(hipadj_handle* h, int32_t* idx) {
    if (!h || !idx) return HIPADJ_ERR_INVALID_ARG;
    if (h->multi || h->maxev <= 0 || !h->d_nev) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_event_components: the handle's model carries no ContinuousCallback (hipadj_model_set_continuous_callback), or the handle spans several devices");
    if (!h->have_forward) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_event_components: no forward solve yet");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long Np = h->Npad; const int me = h->maxev;
    std::vector<int> ne((size_t)Np), buf((size_t)me * Np);
    HIP_TRY(h, hipMemcpy(ne.data(), h->d_nev, sizeof(int) * (size_t)Np, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(buf.data(), h->d_ev_k, sizeof(int) * (size_t)me * Np, hipMemcpyDeviceToHost));
    for (long i = 0; i < h->N; ++i) for (int k = 0; k < me; ++k) idx[i * me + k] = k < ne[i] ? buf[(size_t)k * Np + i] : -1;
    return HIPADJ_OK;
}
// cotangents of a loss on the saved event states for the following reverse passes: dl (at u-), dr (at u+), [ntraj][max_events][n] host pointers (entries beyond a trajectory's
// event count are ignored); either may be NULL (= zero); both NULL removes them
extern "C" int hipadj_set_event_cotangents(hipadj_handle* h, const double* dl, const double* dr) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (h->multi || h->maxev <= 0 || !h->d_nev) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_set_event_cotangents: the handle's model carries no ContinuousCallback (hipadj_model_set_continuous_callback), or the handle spans several devices");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long Np = h->Npad; const int n = h->n, me = h->maxev;
    const size_t cnt = (size_t)me * n * Np;
    std::vector<double> buf(cnt);
    for (int side = 0; side < 2; ++side) {
        const double* src = side ? dr : dl; double** dev = side ? &h->d_ev_dr : &h->d_ev_dl;
        if (!src) { if (side) h->ag.ev_dr = nullptr; else h->ag.ev_dl = nullptr; continue; }
        if (!*dev) TRY(dev_alloc(h, dev, cnt));
        std::fill(buf.begin(), buf.end(), 0.0);
        for (long i = 0; i < h->N; ++i) for (int k = 0; k < me; ++k) for (int j = 0; j < n; ++j) buf[((size_t)k * n + j) * Np + i] = src[((size_t)i * me + k) * n + j];
        HIP_TRY(h, hipMemcpy(*dev, buf.data(), sizeof(double) * cnt, hipMemcpyHostToDevice));
        if (side) h->ag.ev_dr = *dev; else h->ag.ev_dl = *dev;
    }
    return HIPADJ_OK;
}

extern "C" int hipadj_get_stats(hipadj_handle* h, hipadj_stats* st) {
    if (!h || !st) return HIPADJ_ERR_INVALID_ARG;
    if (st->struct_size != sizeof(hipadj_stats)) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_stats.struct_size mismatch");
    if (h->multi) return multi_get_stats(h, st);
    if (h->route) return route_get_stats(h, st);
    *st = h->st;
    return HIPADJ_OK;
}

#define DISPATCH_GRID(h, fn, ...)                                                          \
    switch ((h)->cfg.dims[0]) {                                                            \
    case 8: return fn<8>(__VA_ARGS__);                                                     \
    case 16: return fn<16>(__VA_ARGS__);                                                   \
    case 32: return fn<32>(__VA_ARGS__);                                                   \
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "unsupported Brusselator grid %d", (h)->cfg.dims[0]); }

#define DISPATCH_HIDDEN(h, fn, ...)                                                        \
    switch ((h)->cfg.dims[1]) {                                                            \
    case 32: return fn<32>(__VA_ARGS__);                                                   \
    case 64: return fn<64>(__VA_ARGS__);                                                   \
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
// kernels with a fused tail: the on-grid Interpolating / Gauss / GaussKronrod / Backsolve sweeps of the lane family (compiled-in and runtime models)
static bool fused_eligible(const hipadj_config* cfg, const Plan& P) {
    if (const char* e = std::getenv("HIPADJ_FUSED")) if (std::atoi(e) == 0) return false;
    const bool alg_ok = cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD || cfg->alg == HIPADJ_ALG_BACKSOLVE;
    if (!(alg_ok && !P.ip_ckpt && !P.offgrid && !P.adaptive && !P.wide && !P.field && !P.mlp)) return false;
    if (P.user && plan_seg_fits(P.n, P.np)) {
        // runtime models with segment lanes: the tail holds up to 4 child maps next to the wave's own, so wide maps turn the kernel into one of the
        // heavily spilling ones (8-state ring, Backsolve: 4.9 KB of scratch per lane, and its -O1 build came back wrong and irreproducible on the GPU,
        // profiles/r3_fused_wide_lane_probe.log).  Such kernels take milliseconds, the two saved launches ~10 us: they keep the three-launch sequence.
        int cap = 64;
        if (const char* e = std::getenv("HIPADJ_FUSED_USER_CAP")) { const int v = std::atoi(e); if (v > 0) cap = v; }   // test hook
        if ((1 + P.n) * (P.n + P.np) > cap) return false;
    }
    return true;
}
struct UserKernels { std::string forward, main_k, tail, gk, aux; };
static UserKernels user_kernel_names(const hipadj_handle* h) {
    const std::string U = "hipadj::UserModel";
    const int n = h->n, np = h->np;
    const int mode = (loss_streams(h) ? 0 : 1) | (h->cfg.cont_cost << 1);
    const int cc = mode >> 1;
    // knot prefetch depth of the reverse sweeps: the unrolled PF-deep blocks pay for small step bodies only (n <= 3 and a right-hand side without
    // library math, user_calls_math); everything else runs the rolled sweep with one knot in flight, which measured fastest at every n = 2 ... 8
    const bool deep = n <= 3 && !user_calls_math(h->cfg.model);
    int PF = deep ? ((mode & 1) ? 8 : 6) : 1, PFG = deep ? 4 : 1;
    if (const char* e = std::getenv("HIPADJ_USER_PF")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) PF = v; }   // tuning hook: prefetch depth of k_interp for runtime models
    if (const char* e = std::getenv("HIPADJ_USER_PFG")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) PFG = v; }   // ... and of k_gauss / k_quad_adj
    auto I = [](int v) { return std::to_string(v); };
    UserKernels k;
    const bool seg = plan_seg_fits(n, np);              // same rule as the planner: wider models stay sequential in time ...
    const std::string SG = seg ? ", true>" : ", false>";  // ... and compile only the one-column path (k_interp SEG)
    const std::string finish = "hipadj::k_finish<" + I(n) + ", " + I(np) + ">";
    const std::string compose = seg ? "hipadj::k_compose_finish<" + U + ">" : "hipadj::k_finish_map<" + I(n) + ", " + I(np) + ">";
    if (h->adaptive) {
        const bool ros = h->cfg.stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE;      // same kernels, STEP = 1 (ros23_integrate)
        k.forward = "hipadj::k_forward_tsit5<" + U + (ros ? ", 1>" : ", 0>");
        k.main_k = "hipadj::k_adjoint_tsit5<" + U + ", " + I(h->cfg.alg) + ", " + I(cc) + (h->ip_ckpt ? ", true" : ", false") + (ros ? ", 1>" : ", 0>");
        if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) k.gk = "hipadj::k_quad_gk_tsit5<" + U + ", " + I(cc) + ">";
        k.tail = finish;
        return k;
    }
    k.forward = "hipadj::k_forward<" + U + ">";
    if (h->offgrid) {   // loss times off the step grid (planner: InterpolatingAdjoint only); the `gk` slot carries the out = sol(ts) kernel
        if (h->og_ck) k.main_k = "hipadj::k_offgrid_ckpt<" + U + ", " + I(mode) + ", " + I(h->cfg.alg == HIPADJ_ALG_INTERPOLATING ? 0 : (h->cfg.alg == HIPADJ_ALG_GAUSS ? 2 : 4)) + ">";
        else if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE) k.main_k = "hipadj::k_backsolve_offgrid<" + U + ", " + I(cc) + ">";
        else if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) { k.main_k = "hipadj::k_quad_adj_offgrid<" + U + ", " + I(mode) + ">"; k.aux = "hipadj::k_quad_gk_offgrid<" + U + ", " + I(cc) + ">"; }   // round 5
        else if (h->nseg > 1) k.main_k = "hipadj::k_offgrid_seg<" + U + ", " + I(mode) + (h->cfg.alg == HIPADJ_ALG_GAUSS ? ", true>" : ", false>");   // time-segmented over the reverse step list
        else if (h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD) k.main_k = "hipadj::k_gauss_offgrid<" + U + ", " + I(mode) + ", true>";
        else k.main_k = std::string(h->cfg.alg == HIPADJ_ALG_GAUSS ? "hipadj::k_gauss_offgrid<" : "hipadj::k_interp_offgrid<") + U + ", " + I(mode) + ">";
        k.gk = "hipadj::k_out_offgrid<" + U + ">"; k.tail = (h->nseg > 1 && h->cfg.alg != HIPADJ_ALG_BACKSOLVE) ? compose : finish;
        return k;
    }
    if (h->ip_ckpt) {   // checkpointing=true (Interpolating / Gauss): checkpoint tiles + in-kernel interval re-solve; the planner admits models whose segment columns fit the VGPRs
        k.main_k = std::string(h->cfg.alg == HIPADJ_ALG_INTERPOLATING ? "hipadj::k_interp_ckpt<" : "hipadj::k_gauss_ckpt<") + U + ", " + I(mode) + (h->ck_long ? ", true>" : ", false>"); k.tail = compose;
        return k;
    }
    if (h->fused && h->cfg.alg != HIPADJ_ALG_QUADRATURE) {   // one launch per reverse pass (hipadj_fused.hpp): the sweep kernel finishes the pass; `tail` stays a valid (unused) name
        const std::string SB = seg ? "true" : "false";
        switch (h->cfg.alg) {
        case HIPADJ_ALG_INTERPOLATING: k.main_k = "hipadj::k_interp_fused<" + U + ", " + I(PF) + ", " + I(mode) + ", " + SB + ", false>"; break;
        case HIPADJ_ALG_BACKSOLVE: k.main_k = "hipadj::k_backsolve_fused<" + U + ", " + I(cc) + ", " + SB + ">"; break;
        case HIPADJ_ALG_GAUSS: k.main_k = "hipadj::k_gauss_fused<" + U + ", " + I(PFG) + ", " + I(mode) + ", false, " + SB + ">"; break;
        default: k.main_k = "hipadj::k_gauss_fused<" + U + ", " + I(PFG) + ", " + I(mode) + ", true, " + SB + ">"; break;
        }
        k.tail = finish;
        return k;
    }
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
    if (!k.aux.empty()) exprs.push_back(k.aux);
    std::vector<char> code; std::map<std::string, std::string> low;
    const bool force_o1 = std::getenv("HIPADJ_RTC_FORCE_O1") != nullptr;   // debugging hook: run the -O1 build only (no self-test)
    const int rc = user_compile(h->cfg.model, exprs, code, low, h->err, force_o1);
    if (rc != HIPADJ_OK) return rc;
    HIP_TRY(h, hipModuleLoadData(&h->umod, code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_forward, h->umod, low[k.forward].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_main, h->umod, low[k.main_k].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_tail, h->umod, low[k.tail].c_str()));
    if (!k.gk.empty()) HIP_TRY(h, hipModuleGetFunction(&h->uf_gk, h->umod, low[k.gk].c_str()));
    if (!k.aux.empty()) HIP_TRY(h, hipModuleGetFunction(&h->uf_aux, h->umod, low[k.aux].c_str()));
    // Reverse kernels of wide models can spill thousands of registers (512 registers + KBs of scratch per lane).  One such kernel came back WRONG from
    // the toolkit's compiler at -O3 and right at -O1 / -O0 (8-state ring, dual-number VJPs, GaussAdjoint: 1232 spilled registers, 2860 B of scratch;
    // DESIGN.md 6.8) — no pattern a static check could flag.  So a reverse kernel with >= 1 KB of scratch per lane gets a second build at -O1, and the
    // first adjoint call runs both and compares (user_adjoint): agreement keeps the -O3 build, disagreement the -O1 one.  HIPADJ_RTC_SELFTEST=0 skips it.
    int scratch = 0;
    if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, h->uf_main) != hipSuccess) scratch = 0;
    const char* st = std::getenv("HIPADJ_RTC_SELFTEST");
    const int st_min = st ? std::atoi(st) : (rtc_trusted() ? 1024 : 1);   // 0 = off, otherwise the scratch size (bytes per lane) from which the self-test runs; an untrusted compiler: every kernel
    if (!force_o1 && st_min > 0 && (scratch >= st_min || (!st && !rtc_trusted())) && !std::getenv("HIPADJ_RTC_OVERRIDE")) {
        std::vector<char> code2; std::map<std::string, std::string> low2;
        const int rc2 = user_compile(h->cfg.model, exprs, code2, low2, h->err, true);
        if (rc2 != HIPADJ_OK) return rc2;
        HIP_TRY(h, hipModuleLoadData(&h->umod_alt, code2.data()));
        HIP_TRY(h, hipModuleGetFunction(&h->uf_main_alt, h->umod_alt, low2[k.main_k].c_str()));
        h->rtc_selftest = 1;
    }
    return HIPADJ_OK;
}

static int user_compile_config(const hipadj_config* cfg, std::string& err) {
    hipadj_handle h;
    h.cfg = *cfg; h.cfg.save_times = nullptr; h.cfg.checkpoints = nullptr;
    Plan P;
    { const int prc = make_plan(cfg, P, err); if (prc != HIPADJ_OK) return prc; }
    if (P.wide) { std::vector<char> code; std::map<std::string, std::string> low; return user_compile(cfg->model, wide_kernel_names(cfg->alg, P.adaptive, cfg->cont_cost, P.ip_ckpt, P.offgrid), code, low, err); }
    h.n = P.n; h.np = P.np; h.M = P.M; h.adaptive = P.adaptive; h.ip_ckpt = P.ip_ckpt; h.offgrid = P.offgrid; h.og_ck = P.og_ck; h.ck_long = P.ck_longest > HIPADJ_CKPT_KMAX; h.nseg = P.nseg;
    h.fused = fused_eligible(cfg, P) ? 1 : 0;
    const UserKernels k = user_kernel_names(&h);
    std::vector<std::string> exprs = {k.forward, k.main_k, k.tail};
    if (!k.gk.empty()) exprs.push_back(k.gk);
    if (!k.aux.empty()) exprs.push_back(k.aux);
    std::vector<char> code; std::map<std::string, std::string> low;
    return user_compile(cfg->model, exprs, code, low, err);
}

// launch of a module kernel.  `Sig` is the type of the SAME kernel template instantiated for a compiled-in model
// (`decltype(&k_interp<ModelLV, 8, 1>)`: unevaluated, so nothing is instantiated in this translation unit): it only lets the
// compiler check that the argument list handed to hipModuleLaunchKernel has exactly the kernel's parameter types (a mismatch
// would otherwise be silent memory corruption on the device).
template <class Sig> struct usig;
template <class... P> struct usig<void (*)(P...)> {
    template <class... A> static int launch(hipadj_handle* h, hipFunction_t fn, dim3 g, dim3 b, A... args) {
        static_assert(sizeof...(P) == sizeof...(A), "argument count differs from the kernel's parameter list");
        static_assert((std::is_same<P, A>::value && ...), "argument types differ from the kernel's parameter list");
        void* ptrs[] = {(void*)&args...};
        HIP_TRY(h, hipModuleLaunchKernel(fn, g.x, g.y, g.z, b.x, b.y, b.z, 0, h->stream, ptrs, nullptr));
        return HIPADJ_OK;
    }
};
static_assert(std::is_same<decltype(&k_interp<ModelLV, 8, 1>), decltype(&k_gauss<ModelLV, 4, 1, false>)>::value, "k_interp / k_gauss share one launch site");
static_assert(std::is_same<decltype(&k_interp_fused<ModelLV, 8, 1>), decltype(&k_gauss_fused<ModelLV, 4, 1, false>)>::value, "k_interp_fused / k_gauss_fused share one launch site");
static_assert(std::is_same<decltype(&k_interp_ckpt<ModelLV, 1>), decltype(&k_gauss_ckpt<ModelLV, 1>)>::value, "k_interp_ckpt / k_gauss_ckpt share one launch site");
static_assert(std::is_same<decltype(&k_interp_offgrid<ModelLV, 1>), decltype(&k_gauss_offgrid<ModelLV, 1>)>::value, "k_interp_offgrid / k_gauss_offgrid share one launch site");

static int user_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    double* outT = (d_out && h->M > 0) ? h->d_outT : (double*)nullptr;
    if (h->adaptive) {
        const bool sized = h->auto_steps && h->cfg.alg != HIPADJ_ALG_BACKSOLVE;
        if (sized && h->ip_ckpt) h->ag.SmaxI = (int)h->rec_cap;
        for (int pass = 0; pass < 2; ++pass) {
            TRY(usig<decltype(&k_forward_tsit5<ModelLV>)>::launch(h, h->uf_forward, dim3(waves), dim3(WAVE), h->ag, d_u0, d_p, h->ip_ckpt ? (double*)nullptr : h->d_rec, h->d_nsteps,
                        (const double*)h->d_save_t, outT, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag));
            if (!sized) break;
            const int again = adaptive_autosize(h);
            if (again < 0) return again;
            if (again == 0) break;
        }
    }
    else
        TRY(usig<decltype(&k_forward<ModelLV>)>::launch(h, h->uf_forward, dim3(waves), dim3(WAVE), h->g, d_u0, d_p, h->d_knots, h->d_ckpt, (const int*)h->d_ckpt_of_knot, outT,
                    (const int*)h->d_save_of_knot, h->d_yT));
    if (h->offgrid && outT)
        TRY(usig<decltype(&k_out_offgrid<ModelLV>)>::launch(h, h->uf_gk, dim3(waves), dim3(WAVE), h->g, (const dbl2*)h->d_knots, (const double*)h->d_save_t, h->M, h->d_outT));
    if (h->offgrid && h->d_ckpt)   // Backsolve: the checkpoint states at the (off-grid) checkpoint times
        TRY(usig<decltype(&k_out_offgrid<ModelLV>)>::launch(h, h->uf_gk, dim3(waves), dim3(WAVE), h->g, (const dbl2*)h->d_knots, (const double*)h->d_ck_t, h->nck, h->d_ckpt));
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}

// Mass matrix: the sweep leaves nu(t0) = M^T lam(t0) in du0 [N][n]; the reference returns lam(t0) (src/sensitivity_interface.jl:500),
// so every row is multiplied by M^{-T} in place.  One thread per trajectory; n <= 8.
struct MassInv { double a[64]; };
__global__ void k_mass_du0(long N, int n, MassInv mi, double* __restrict__ du0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double v[8], r[8];
    for (int j = 0; j < n; ++j) v[j] = du0[i * n + j];
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += mi.a[k * n + j] * v[k]; r[j] = s; }
    for (int j = 0; j < n; ++j) du0[i * n + j] = r[j];
}

static int user_adjoint_run(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN), cblocks = (unsigned)((h->N + FIN / 4 - 1) / (FIN / 4));
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    double* no_sum = nullptr;
    const double* cotT = h->cot_soa ? h->cot_soa : h->d_cotT;   // hipadj_adjoint_dev_soa: the caller's block, already in the streaming layout
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0 && !h->cot_soa) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));   // runtime models keep the transposition launch
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    bool composed = false;
    if (h->adaptive) {
        for (int pass = 0; pass < 2; ++pass) {
            TRY(usig<decltype(&k_adjoint_tsit5<ModelLV, 0, 0, false>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->ag, p, (const double*)h->d_rec, (const int*)h->d_nsteps, (const double*)h->d_yT,
                        (const double*)h->d_ckpt, (const double*)h->d_ck_t, (const double*)h->d_save_t, (const double*)h->d_tstops, h->ntstops,
                        cotT, d_du0, h->d_dp_traj, h->d_flag, h->d_arec, h->d_nsteps_adj, h->SmaxA));
            if (!(h->cfg.alg == HIPADJ_ALG_QUADRATURE && h->auto_steps)) break;
            const int again = adaptive_adjoint_autosize(h);
            if (again < 0) return again;
            if (again == 0) break;
        }
        if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            TRY(usig<decltype(&k_quad_gk_tsit5<ModelLV, 0>)>::launch(h, h->uf_gk, dim3(waves, (unsigned)h->nq), dim3(WAVE), h->ag, p, (const double*)h->d_rec, (const int*)h->d_nsteps, (const double*)h->d_arec,
                        (const int*)h->d_nsteps_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres));
            hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj, (h->cfg.loss_kind == HIPADJ_LOSS_MODEL || h->dae || h->maxev > 0) ? 1 : 0);      // (a ContinuousCallback: the event jumps' parameter terms are already in dp_traj)
            HIP_TRY(h, hipGetLastError());
        }
    } else if (h->offgrid) {
        RevSteps R{h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->nrs, h->rs_save_at_start, h->cfg.t1};
        if (h->og_ck) {
            const OgIntervals I{h->d_og_i, h->d_og_i + h->og_nint, h->d_og_i + 2 * h->og_nint, h->d_og_h, h->d_ck_t, h->og_nint};
            TRY(usig<decltype(&k_offgrid_ckpt<ModelLV, 1, 0>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->g, R, I, p, (const double*)h->d_ckpt, h->d_og_tile, cotT, d_du0, h->d_dp_traj));
        } else
        if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE)
            TRY(usig<decltype(&k_backsolve_offgrid<ModelLV, 0>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->g, R, p, (const double*)h->d_yT, (const double*)h->d_ckpt, cotT, d_du0, h->d_dp_traj));
        else if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {   // dense lambda over the reverse step list, then adaptive GK15 per (trajectory, loss interval): adjoint_impl's sequence for the compiled-in models
            const bool dl = h->cfg.loss_kind == HIPADJ_LOSS_MODEL;
            TRY(usig<decltype(&k_quad_adj_offgrid<ModelLV, 1>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->g, R, p, (const dbl2*)h->d_knots, cotT, h->d_adj, d_du0, dl ? h->d_dp_traj : (double*)nullptr));
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            TRY(usig<decltype(&k_quad_gk_offgrid<ModelLV, 0>)>::launch(h, h->uf_aux, dim3(waves, (unsigned)h->nq), dim3(WAVE), h->g, R, p, (const dbl2*)h->d_knots, (const dbl2*)h->d_adj, (const double*)h->d_qa,
                        (const double*)h->d_qb, atol, rtol, h->d_qres));
            hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj, dl ? 1 : 0);
            HIP_TRY(h, hipGetLastError());
        } else if (h->nseg > 1) {
            SegPlan sp{h->nseg, h->d_seg_bounds};
            TRY(usig<decltype(&k_offgrid_seg<ModelLV, 1, false>)>::launch(h, h->uf_main, dim3(waves, (unsigned)h->nseg), dim3(WAVE), h->g, R, sp, p, (const dbl2*)h->d_knots, cotT, h->d_segbuf));
            composed = true;
        } else
        TRY(usig<decltype(&k_interp_offgrid<ModelLV, 1>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->g, R, p, (const dbl2*)h->d_knots, cotT, d_du0, h->d_dp_traj));
    } else {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        const dim3 sgrid(waves, (unsigned)h->nseg);
        if (h->fused && h->d_tbuf && h->cfg.alg != HIPADJ_ALG_QUADRATURE) {
            // one launch per reverse pass: the sweep kernel composes the segment maps, writes du0 / the dp rows and reduces dp (hipadj_fused.hpp)
            TreePlan tp = h->tp; tp.tbuf = h->d_tbuf; tp.cnt = h->d_tcnt; tp.partial = h->d_partial; tp.ticket = h->d_ticket;
            double* dps = h->cfg.p_shared ? d_dp : (double*)nullptr;
            if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE)
                TRY(usig<decltype(&k_backsolve_fused<ModelLV, 0>)>::launch(h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, tp, p, (const double*)h->d_yT, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot,
                            cotT, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag));
            else
                TRY(usig<decltype(&k_interp_fused<ModelLV, 8, 1>)>::launch(h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, tp, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev,
                            d_du0, dp_rows, dps, h->d_flag));
            if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
            if (h->has_mm) {
                MassInv mi; std::memcpy(mi.a, h->minv, sizeof(mi.a));
                hipLaunchKernelGGL(k_mass_du0, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->N, h->n, mi, d_du0);
                HIP_TRY(h, hipGetLastError());
            }
            if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
            es.pending = h->timing >= 1; es.full = h->timing >= 2;
            return HIPADJ_OK;
        }
        if (h->ip_ckpt) {
            TRY(usig<decltype(&k_interp_ckpt<ModelLV, 1>)>::launch(h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, p, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck,
                                                                   cotT, (const int*)h->d_save_rev, h->d_segbuf, h->d_gtile, h->gtile_stride));
            composed = true;
        } else
        switch (h->cfg.alg) {
        case HIPADJ_ALG_INTERPOLATING: case HIPADJ_ALG_GAUSS: case HIPADJ_ALG_GAUSS_KRONROD:
            TRY(usig<decltype(&k_interp<ModelLV, 8, 1>)>::launch(h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, h->d_segbuf));
            composed = true; break;
        case HIPADJ_ALG_BACKSOLVE:
            TRY(usig<decltype(&k_backsolve<ModelLV, 0>)>::launch(h, h->uf_main, sgrid, dim3(WAVE), h->g, sp, p, (const double*)h->d_yT, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot,
                        cotT, (const int*)h->d_save_rev, h->d_segbuf));
            composed = true; break;
        default: {
            // a model with discrete-loss bodies (HIPADJ_LOSS_MODEL): the sweep leaves the sum of dgdp_discrete over the loss times in dp_traj, to which k_quad_sum adds the quadrature
            const bool dl = h->cfg.loss_kind == HIPADJ_LOSS_MODEL;
            TRY(usig<decltype(&k_quad_adj<ModelLV, 8, 1>)>::launch(h, h->uf_main, dim3(waves), dim3(WAVE), h->g, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, h->d_adj, d_du0,
                        dl ? h->d_dp_traj : (double*)nullptr));
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            TRY(usig<decltype(&k_quad_gk<ModelLV, 0>)>::launch(h, h->uf_gk, dim3(waves, (unsigned)h->nq), dim3(WAVE), h->g, p, (const dbl2*)h->d_knots, (const dbl2*)h->d_adj, (const double*)h->d_qa,
                        (const double*)h->d_qb, atol, rtol, h->d_qres));
            hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj, dl ? 1 : 0);
            HIP_TRY(h, hipGetLastError());
            break; }
        }
    }
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    const bool seg_kernels = plan_seg_fits(h->n, h->np);
    if (composed && seg_kernels)
        TRY(usig<decltype(&k_compose_finish<ModelLV>)>::launch(h, h->uf_tail, dim3(cblocks), dim3(FIN), h->g, h->nseg, (const double*)h->d_segbuf, d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    else if (composed)
        TRY(usig<decltype(&k_finish_map<2, 4>)>::launch(h, h->uf_tail, dim3(fblocks), dim3(FIN), h->N, h->Npad, (const double*)h->d_segbuf, d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    else
        TRY(usig<decltype(&k_finish<2, 4>)>::launch(h, h->uf_tail, dim3(fblocks), dim3(FIN), h->N, h->Npad, (const double*)d_du0, (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, no_sum));
    if (h->cfg.p_shared) {
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)((composed && seg_kernels) ? cblocks : fblocks), h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->has_mm) {
        MassInv mi; std::memcpy(mi.a, h->minv, sizeof(mi.a));
        hipLaunchKernelGGL(k_mass_du0, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->N, h->n, mi, d_du0);
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
int adaptive_autosize(hipadj_handle* h) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<int> ns((size_t)h->Npad);
    HIP_TRY(h, hipMemcpy(ns.data(), h->d_nsteps, sizeof(int) * (size_t)h->Npad, hipMemcpyDeviceToHost));
    long mx = 1;
    for (long i = 0; i < h->N; ++i) if (ns[i] > mx) mx = ns[i];
    if (mx <= h->rec_cap || mx >= HIPADJ_AUTO_MAXITERS) return 0;     // fits (or ran into maxiters: the flag reports it)
    const int RW = 2 + 5 * h->n;
    const long cap = mx + mx / 8 + 8;
    // the larger buffer first, the old one released only once it exists: a failed allocation leaves the handle with its old (consistent) record and capacities, and without
    // a forward solution — a later hipadj_adjoint_dev then reports HIPADJ_ERR_STATE instead of sweeping a null record
    auto regrow = [&](double** buf, size_t old_count, size_t new_count) -> int {
        double* nb = nullptr;
        const int rc = dev_alloc(h, &nb, new_count);
        if (rc != HIPADJ_OK) { h->have_forward = false; return rc; }
        if (*buf) { (void)hipFree(*buf); h->ws_bytes -= (double)(old_count * sizeof(double)); }
        *buf = nb;
        return HIPADJ_OK;
    };
    TRY(regrow(&h->d_rec, (size_t)h->rec_cap * RW * h->Npad, (size_t)cap * RW * h->Npad));
    if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        const long capA = 2 * cap + h->M + 16;
        TRY(regrow(&h->d_arec, (size_t)h->SmaxA * RW * h->Npad, (size_t)capA * RW * h->Npad));
        h->SmaxA = (int)capA; h->ag.SmaxA = h->SmaxA;
    }
    h->rec_cap = cap;
    h->st.workspace_bytes = h->ws_bytes;
    h->ag.Smax = (int)cap; h->ag.SmaxI = (int)cap;
    if (h->ip_ckpt) { HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream)); return 0; }   // (the pass marked "more steps than the old capacity": not an error here)
                                // checkpointing=true: the forward pass writes no records (only the checkpoint states), so nothing has to be
                                // repeated; the buffer just regrown is the ONE-interval record buffer of the reverse sweep, sized by the
                                // whole-trajectory step count — a safe bound for any single interval
    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));                  // the overflow mark of the pass that is being repeated
    return 1;
}


// The dense ADJOINT record of QuadratureAdjoint on the adaptive path with max_steps == 0: its capacity starts as a guess (twice the
// forward capacity + the loss times); the sweep stores the TRUE step count per trajectory, the host reads the maximum after the sweep and,
// when the record did not fit, regrows it and asks for the sweep to be repeated (same step sequence: it cannot overflow again) — what
// adaptive_autosize does for the forward records.  One stream synchronisation per adjoint call, in this mode only.
// Returns 1 = repeat the sweep, 0 = it stands, negative = error.
int adaptive_adjoint_autosize(hipadj_handle* h) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<int> ns((size_t)h->Npad);
    HIP_TRY(h, hipMemcpy(ns.data(), h->d_nsteps_adj, sizeof(int) * (size_t)h->Npad, hipMemcpyDeviceToHost));
    long mx = 1;
    for (long i = 0; i < h->N; ++i) if (ns[i] > mx) mx = ns[i];
    if (mx <= h->SmaxA || mx >= 8L * HIPADJ_AUTO_MAXITERS) return 0;
    const int RW = 2 + 5 * h->n;
    const long capA = mx + mx / 8 + 8;
    { double* nb = nullptr;     // the larger record first (a failed allocation keeps the old one and its capacity)
      TRY(dev_alloc(h, &nb, (size_t)capA * RW * h->Npad));
      if (h->d_arec) { (void)hipFree(h->d_arec); h->ws_bytes -= (double)((size_t)h->SmaxA * RW * h->Npad * sizeof(double)); }
      h->d_arec = nb; }
    h->SmaxA = (int)capA; h->ag.SmaxA = h->SmaxA;
    h->st.workspace_bytes = h->ws_bytes;
    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
    return 1;
}

// ---- wide runtime models: workgroup-per-trajectory family (hipadj_wide.hpp) ----------------------------------------------------------------
static int wide_prepare(hipadj_handle* h) {
    if (h->wide_ts5 && (h->cfg.alg == HIPADJ_ALG_INTERPOLATING || h->cfg.alg == HIPADJ_ALG_BACKSOLVE)) {
        const long lds = user_wide_ts5_interp_lds(h->cfg.model, h->cfg.alg == HIPADJ_ALG_BACKSOLVE) * 8;
        if (lds > 160L * 1024)
            HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "wide model %d: Interpolating- / BacksolveAdjoint on the adaptive solution need %ld KB of LDS (state tiles + scratch + 5 np + the parameter copy), a workgroup has 160 KB — use GaussAdjoint, whose sweep integrates lam only",
                        h->cfg.model, lds / 1024);
    }
    const std::vector<std::string> exprs = wide_kernel_names(h->cfg.alg, h->wide_ts5, h->cfg.cont_cost, h->ip_ckpt, h->offgrid);
    std::vector<char> code; std::map<std::string, std::string> low;
    const int rc = user_compile(h->cfg.model, exprs, code, low, h->err);
    if (rc != HIPADJ_OK) return rc;
    HIP_TRY(h, hipModuleLoadData(&h->umod, code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_forward, h->umod, low[exprs[0]].c_str()));
    HIP_TRY(h, hipModuleGetFunction(&h->uf_main, h->umod, low[exprs[1]].c_str()));
    if (exprs.size() > 2) HIP_TRY(h, hipModuleGetFunction(&h->uf_gk, h->umod, low[exprs[2]].c_str()));
    if (exprs.size() > 3) HIP_TRY(h, hipModuleGetFunction(&h->uf_aux, h->umod, low[exprs[3]].c_str()));   // off-grid Quadrature: uf_gk is the out = sol(ts) kernel, the GK pass sits here
    h->wide_T = user_wide_threads(h->cfg.model);
    return HIPADJ_OK;
}

// The dense record of a wide model's adaptive forward solve with max_steps = 0 when the 8 GiB budget cut its capacity below 8192 steps (large states): the kernel stores the
// TRUE step count per trajectory; the host reads the maximum and, when some trajectory did not fit, regrows the record (and QuadratureAdjoint's adjoint record) and asks for the
// solve to be repeated — what adaptive_autosize does for the lane family.  One stream synchronisation per forward solve, in this mode only.  1 = repeat, 0 = it stands.
static int wide_autosize(hipadj_handle* h) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<int> ns((size_t)h->N);
    HIP_TRY(h, hipMemcpy(ns.data(), h->d_nsteps, sizeof(int) * (size_t)h->N, hipMemcpyDeviceToHost));
    long mx = 1;
    for (long i = 0; i < h->N; ++i) if (ns[i] > mx) mx = ns[i];
    if (mx <= h->rec_cap || mx >= HIPADJ_AUTO_MAXITERS) return 0;     // fits (or ran into maxiters: the flag reports it)
    const long RW = 2 + 5L * h->n, cap = mx + mx / 8 + 8;
    // the larger buffer first, the old one released only once it exists: a failed allocation leaves the handle with its old (consistent) record and capacities, and without
    // a forward solution — a later hipadj_adjoint_dev then reports HIPADJ_ERR_STATE instead of sweeping a null record
    auto regrow = [&](double** buf, size_t old_count, size_t new_count) -> int {
        double* nb = nullptr;
        const int rc = dev_alloc(h, &nb, new_count);
        if (rc != HIPADJ_OK) { h->have_forward = false; return rc; }
        if (*buf) { (void)hipFree(*buf); h->ws_bytes -= (double)(old_count * sizeof(double)); }
        *buf = nb;
        return HIPADJ_OK;
    };
    TRY(regrow(&h->d_rec, (size_t)h->N * h->rec_cap * RW, (size_t)h->N * cap * RW));
    if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        const long capA = 2 * cap + h->M + 16;
        TRY(regrow(&h->d_arec, (size_t)h->N * h->SmaxA * RW, (size_t)h->N * capA * RW));
        h->SmaxA = (int)capA;
    }
    h->rec_cap = cap; h->wa.Smax = (int)cap; h->ag.Smax = (int)cap;
    h->st.workspace_bytes = h->ws_bytes;
    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));                  // the overflow mark of the pass that is being repeated
    return 1;
}

static int wide_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (h->wide_ts5) {
        const bool bs = h->cfg.alg == HIPADJ_ALG_BACKSOLVE, ck = h->ip_ckpt;      // ck: the checkpoint states only, no dense record
        for (int pass = 0; pass < 2; ++pass) {
            TRY(usig<decltype(&k_wide_forward_ts5<WideProbe>)>::launch(h, h->uf_forward, dim3((unsigned)h->N), dim3((unsigned)h->wide_T), h->wg, h->wa, d_u0, d_p, (bs || ck) ? (double*)nullptr : h->d_rec, h->d_nsteps,
                        (const double*)h->d_save_t, (d_out && h->M > 0) ? d_out : (double*)nullptr, (const double*)h->d_ck_t, ((bs || ck) && h->wg.nck > 0) ? h->d_ckpt : (double*)nullptr,
                        bs ? h->d_yT : (double*)nullptr, h->d_flag));
            if (!h->wide_auto) break;
            const int again = wide_autosize(h);
            if (again < 0) return again;
            if (again == 0) break;
        }
        return HIPADJ_OK;
    }
    const bool bs = h->cfg.alg == HIPADJ_ALG_BACKSOLVE, ck = h->ip_ckpt, og = h->offgrid;
    TRY(usig<decltype(&k_wide_forward<WideProbe>)>::launch(h, h->uf_forward, dim3((unsigned)h->N), dim3((unsigned)h->wide_T), h->wg, d_u0, d_p,
                ((bs && !og) || ck) ? (double*)nullptr : h->d_fknots, (d_out && h->M > 0 && !og) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot,
                ((bs || ck) && !og && h->nck > 0) ? h->d_ckpt : (double*)nullptr, (const int*)h->d_ckpt_of_knot, bs ? h->d_yT : (double*)nullptr));
    if (og && d_out && h->M > 0)     // the save times are not knots: out = sol(ts) from the forward Hermite interpolant
        TRY(usig<decltype(&k_wide_out_offgrid<WideProbe>)>::launch(h, h->uf_gk, dim3((unsigned)h->N), dim3((unsigned)h->wide_T), h->wg, (const double*)h->d_fknots, (const double*)h->d_save_t, h->M, d_out));
    if (og && bs && h->nck > 0)      // Backsolve: the checkpoint states at the (off-grid) checkpoint times
        TRY(usig<decltype(&k_wide_out_offgrid<WideProbe>)>::launch(h, h->uf_gk, dim3((unsigned)h->N), dim3((unsigned)h->wide_T), h->wg, (const double*)h->d_fknots, (const double*)h->d_ck_t, h->nck, h->d_ckpt));
    return HIPADJ_OK;
}

static int wide_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const double* p = h->p_dev_last;
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    const dim3 grid((unsigned)h->N), blk((unsigned)h->wide_T);
    // per-trajectory gradient rows: straight into the caller's dp when the parameters are per trajectory, else a workspace that k_wide_reduce_dp sums
    double* rows = h->cfg.p_shared ? h->d_dp_traj : d_dp;
    if (h->wide_ts5 && h->cfg.alg == HIPADJ_ALG_BACKSOLVE) {
        TRY(usig<decltype(&k_wide_backsolve_ts5<WideProbe>)>::launch(h, h->uf_main, grid, blk, h->wg, h->wa, p, (const double*)h->d_yT, (const double*)(h->wg.nck > 0 ? h->d_ckpt : nullptr),
                    (const double*)h->d_ck_t, (const double*)h->d_save_t, (const double*)h->d_tstops, d_cot, d_du0, rows, h->d_flag));
    } else if (h->wide_ts5) {
        const bool quad = h->cfg.alg == HIPADJ_ALG_QUADRATURE;
        const bool ck = h->ip_ckpt;   // checkpointing = true: d_rec holds one interval per trajectory, written by the sweep itself
        TRY(usig<decltype(&k_wide_adjoint_ts5<WideProbe, 2>)>::launch(h, h->uf_main, grid, blk, h->wg, h->wa, p, ck ? (const double*)nullptr : (const double*)h->d_rec, (const int*)h->d_nsteps, (const double*)h->d_save_t,
                    (const double*)h->d_tstops, d_cot, d_du0, rows, h->d_flag, quad ? h->d_arec : (double*)nullptr, quad ? h->d_nsteps_adj : (int*)nullptr, quad ? h->SmaxA : 0,
                    h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD ? h->d_wscr : (double*)nullptr, ck ? h->d_rec : (double*)nullptr, (const double*)(ck ? h->d_ckpt : nullptr), (const double*)(ck ? h->d_ck_t : nullptr), h->wide_SmaxI));
        if (quad) {
            if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            const WideQuadSrc src{nullptr, nullptr, h->d_rec, h->d_nsteps, h->d_arec, h->d_nsteps_adj, h->wa.Smax, h->SmaxA, nullptr, nullptr, 0};
            TRY(usig<decltype(&k_wide_quad_gk<WideProbe, HIPADJ_WIDE_MAXSEG, true>)>::launch(h, h->uf_gk, dim3((unsigned)h->N, (unsigned)h->nq), blk, h->wg, p, src,
                        (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_wscr, h->d_qres));
            hipLaunchKernelGGL(k_wide_quad_sum, grid, dim3(256), 0, h->stream, h->N, h->np, h->nq, (const double*)h->d_qres, rows, h->cfg.loss_kind == HIPADJ_LOSS_MODEL ? 1 : 0);
            HIP_TRY(h, hipGetLastError());
        }
    } else if (h->offgrid) {
        const RevSteps R{h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->nrs, h->rs_save_at_start, h->cfg.t1};
        if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE)
            TRY(usig<decltype(&k_wide_backsolve_og<WideProbe>)>::launch(h, h->uf_main, grid, blk, h->wg, R, p, (const double*)h->d_yT, (const double*)(h->nck > 0 ? h->d_ckpt : nullptr), d_cot, d_du0, rows, h->d_flag));
        else if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
            TRY(usig<decltype(&k_wide_quad_adj_og<WideProbe>)>::launch(h, h->uf_main, grid, blk, h->wg, R, p, (const double*)h->d_fknots, d_cot, h->d_fadj, d_du0, h->d_flag, rows));
            if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
            const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
            const WideQuadSrc src{h->d_fknots, h->d_fadj, nullptr, nullptr, nullptr, nullptr, 0, 0, h->d_rs_t, h->d_rs_te, h->nrs};
            TRY(usig<decltype(&k_wide_quad_gk<WideProbe, HIPADJ_WIDE_MAXSEG, false, true>)>::launch(h, h->uf_aux, dim3((unsigned)h->N, (unsigned)h->nq), blk, h->wg, p, src,
                        (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_wscr, h->d_qres));
            hipLaunchKernelGGL(k_wide_quad_sum, grid, dim3(256), 0, h->stream, h->N, h->np, h->nq, (const double*)h->d_qres, rows, h->cfg.loss_kind == HIPADJ_LOSS_MODEL ? 1 : 0);
            HIP_TRY(h, hipGetLastError());
        } else
        TRY(usig<decltype(&k_wide_adjoint_og<WideProbe, 0>)>::launch(h, h->uf_main, grid, blk, h->wg, R, p, (const double*)h->d_fknots, d_cot, d_du0, rows, h->d_flag,
                    h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD ? h->d_wscr : (double*)nullptr));
    } else
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING: case HIPADJ_ALG_GAUSS: case HIPADJ_ALG_GAUSS_KRONROD:
        if (h->ip_ckpt)
            TRY(usig<decltype(&k_wide_adjoint_ck<WideProbe, 0>)>::launch(h, h->uf_main, grid, blk, h->wg, p, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck,
                        h->d_fknots, h->wide_KT, d_cot, (const int*)h->d_save_rev, d_du0, rows, h->d_flag, h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD ? h->d_wscr : (double*)nullptr));
        else
        TRY(usig<decltype(&k_wide_adjoint<WideProbe, 0>)>::launch(h, h->uf_main, grid, blk, h->wg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, rows, h->d_flag,
                    h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD ? h->d_wscr : (double*)nullptr));
        break;
    case HIPADJ_ALG_BACKSOLVE:
        TRY(usig<decltype(&k_wide_backsolve<WideProbe>)>::launch(h, h->uf_main, grid, blk, h->wg, p, (const double*)h->d_yT, (const double*)(h->nck > 0 ? h->d_ckpt : nullptr),
                    (const int*)h->d_ckpt_of_knot, d_cot, (const int*)h->d_save_rev, d_du0, rows, h->d_flag));
        break;
    case HIPADJ_ALG_QUADRATURE: {
        TRY(usig<decltype(&k_wide_quad_adj<WideProbe>)>::launch(h, h->uf_main, grid, blk, h->wg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, h->d_fadj, d_du0, h->d_flag, rows));
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        const WideQuadSrc src{h->d_fknots, h->d_fadj, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
        TRY(usig<decltype(&k_wide_quad_gk<WideProbe, HIPADJ_WIDE_MAXSEG, false>)>::launch(h, h->uf_gk, dim3((unsigned)h->N, (unsigned)h->nq), blk, h->wg, p, src,
                    (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_wscr, h->d_qres));
        hipLaunchKernelGGL(k_wide_quad_sum, grid, dim3(256), 0, h->stream, h->N, h->np, h->nq, (const double*)h->d_qres, rows, h->cfg.loss_kind == HIPADJ_LOSS_MODEL ? 1 : 0);
        HIP_TRY(h, hipGetLastError());
        break; }
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg not available for wide models");
    }
    if (h->timing >= 1 && h->cfg.alg != HIPADJ_ALG_QUADRATURE) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    if (h->cfg.p_shared) {
        const unsigned jb = (unsigned)((h->np + 255) / 256);
        if (h->N > 64) {   // two levels: chunk partials into d_partial (ceil(N / 16) rows are allocated, ceil(N / C) <= that are used), then the partials in chunk order
            int C = (int)std::ceil(std::sqrt((double)h->N)); if (C < 16) C = 16;
            const long nchunk = (h->N + C - 1) / C;
            hipLaunchKernelGGL(k_wide_reduce_dp_chunks, dim3(jb, (unsigned)nchunk), dim3(256), 0, h->stream, h->N, h->np, C, (const double*)h->d_dp_traj, h->d_partial);
            hipLaunchKernelGGL(k_wide_reduce_dp, dim3(jb), dim3(256), 0, h->stream, nchunk, h->np, (const double*)h->d_partial, d_dp, h->d_flag);
        } else
            hipLaunchKernelGGL(k_wide_reduce_dp, dim3(jb), dim3(256), 0, h->stream, h->N, h->np, (const double*)h->d_dp_traj, d_dp, h->d_flag);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;
}

static int forward_dispatch(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (h->wide) return wide_forward(h, d_u0, d_p, d_out);
    if (h->user) return user_forward(h, d_u0, d_p, d_out);
    if (h->field) { DISPATCH_GRID(h, field_forward, h, d_u0, d_p, d_out); }
    if (h->mlp) { DISPATCH_HIDDEN(h, mlp_forward_launch, h, d_u0, d_p, d_out); }
    if (h->adaptive) { DISPATCH_MODEL(h, adaptive_forward, h, d_u0, d_p, d_out); }
    DISPATCH_MODEL(h, forward_impl, h, d_u0, d_p, d_out);
}

// Ground truth for a disagreement between the -O3 and the -O1 build of a reverse kernel ("which build reproduces itself" cannot tell when a
// build is wrong in a reproducible way).  Neither build is asked: the directional derivative of the loss along random directions (d in u0, e in p) is taken by
// central differences THROUGH THE FORWARD KERNEL (small, never one of the heavily spilling kernels) on the handle's own inputs,
//     D = [L(u0 + eps d, p + eps e) - L(u0 - eps d, p - eps e)] / (2 eps),   L = sum_i <Delta_i, u(t_i)>  or  sum_i |u(t_i) - shift|^2 / 2,
// and compared with  A = <du0, d> + <dp, e>  of each candidate.  Returns in `err[c]` the worst relative error |A - D| / |D| over the directions.
// Available on the fixed step without a continuous cost (differences through an adaptive solve carry its step-size noise; the cost integral is not formed by the
// forward kernel); `avail` says so.  The forward solution of the handle is restored before returning (one more forward solve on the original inputs).
static int user_ground_truth(hipadj_handle* h, const double* d_cot, const std::vector<double>* cand_du0[2], const std::vector<double>* cand_dp[2], double (&err)[2], bool& avail) {
    avail = false; err[0] = err[1] = 1e300;
    // mass-matrix models: du0 handed back is lam(t0) = M^{-T} nu(t0) (the reference's convention), NOT dL/du0 = M' du0 — the differences would contradict both builds;
    // device-resident losses other than the shifted least squares: the arbiter below knows only the cotangent and the shift form of the loss
    if (h->adaptive || h->cfg.cont_cost != 0 || h->M <= 0 || h->has_mm || h->cfg.loss_kind > HIPADJ_LOSS_LSQ_SHIFT || std::getenv("HIPADJ_RTC_NO_GROUND_TRUTH")) return HIPADJ_OK;
    const size_t N = (size_t)h->N, n = (size_t)h->n, np = (size_t)h->np, M = (size_t)h->M;
    const size_t n0 = N * n, n1 = h->cfg.p_shared ? np : N * np, no = N * M * n;
    std::vector<double> u0(n0), p(n1), cot, outp(no), outm(no), up(n0), pp(n1), d(n0), e(n1);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(u0.data(), h->d_u0, sizeof(double) * n0, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(p.data(), h->d_p, sizeof(double) * n1, hipMemcpyDeviceToHost));
    const bool cotl = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT;
    if (cotl) { cot.resize(no); HIP_TRY(h, hipMemcpy(cot.data(), d_cot, sizeof(double) * no, hipMemcpyDeviceToHost)); }
    double *d_u = nullptr, *d_pp = nullptr, *d_o = nullptr;
    auto release = [&]() { if (d_u) (void)hipFree(d_u); if (d_pp) (void)hipFree(d_pp); if (d_o) (void)hipFree(d_o); };
    if (hipMalloc(&d_u, sizeof(double) * n0) != hipSuccess || hipMalloc(&d_pp, sizeof(double) * n1) != hipSuccess || hipMalloc(&d_o, sizeof(double) * no) != hipSuccess) { release(); return HIPADJ_OK; }
    double us = 1.0, ps = 1.0;
    for (double v : u0) us = std::max(us, std::fabs(v));
    for (double v : p) ps = std::max(ps, std::fabs(v));
    const double eps = 1e-6;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return ((double)(lcg >> 11) / 9007199254740992.0) * 2.0 - 1.0; };
    // the jump at t0 is suppressed under no_start (src/adjoint_common.jl:761): that save time does not enter the loss
    const bool skip0 = h->cfg.no_start && h->save_times.size() > 0 && std::fabs(h->save_times[0] - h->cfg.t0) <= 1e-12 * std::max(1.0, std::fabs(h->cfg.t0));
    auto loss_diff = [&]() {      // L(+) - L(-)
        double s = 0.0;
        for (size_t i = 0; i < N; ++i) for (size_t m = skip0 ? 1 : 0; m < M; ++m) for (size_t j = 0; j < n; ++j) {
            const size_t o = (i * M + m) * n + j;
            if (cotl) s += cot[o] * (outp[o] - outm[o]);
            else { const double a = outp[o] - h->cfg.loss_shift, b = outm[o] - h->cfg.loss_shift; s += 0.5 * (a * a - b * b); }
        }
        return s;
    };
    int rc = HIPADJ_OK;
    double worst[2] = {0.0, 0.0};
    for (int dir = 0; dir < 2 && rc == HIPADJ_OK; ++dir) {
        for (auto& v : d) v = rnd() * us;
        for (auto& v : e) v = rnd() * ps;
        for (int sgn = 0; sgn < 2 && rc == HIPADJ_OK; ++sgn) {
            const double sg = sgn == 0 ? eps : -eps;
            for (size_t k = 0; k < n0; ++k) up[k] = u0[k] + sg * d[k];
            for (size_t k = 0; k < n1; ++k) pp[k] = p[k] + sg * e[k];
            // (on the handle's stream: a null-stream copy does not order with the forward kernels that follow on a non-blocking stream; up / pp stay untouched until the synchronize below)
            if (hipMemcpyAsync(d_u, up.data(), sizeof(double) * n0, hipMemcpyHostToDevice, h->stream) != hipSuccess || hipMemcpyAsync(d_pp, pp.data(), sizeof(double) * n1, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rc = HIPADJ_ERR_HIP; break; }
            rc = forward_dispatch(h, d_u, d_pp, d_o);
            if (rc == HIPADJ_OK && (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy((sgn == 0 ? outp : outm).data(), d_o, sizeof(double) * no, hipMemcpyDeviceToHost) != hipSuccess)) rc = HIPADJ_ERR_HIP;
        }
        if (rc != HIPADJ_OK) break;
        const double D = loss_diff() / (2.0 * eps);
        for (int c = 0; c < 2; ++c) {
            double A = 0.0;
            for (size_t k = 0; k < n0; ++k) A += (*cand_du0[c])[k] * d[k];
            for (size_t k = 0; k < n1; ++k) A += (*cand_dp[c])[k] * e[k];
            const double r = std::isfinite(A) ? std::fabs(A - D) / std::max(std::fabs(D), 1e-300) : 1e300;
            worst[c] = std::max(worst[c], r);
        }
    }
    // the handle's forward solution back on the original inputs (the reverse pass that follows reads it)
    if (hipMemcpy(d_u, u0.data(), sizeof(double) * n0, hipMemcpyHostToDevice) == hipSuccess) {
        const int rr = forward_dispatch(h, d_u, h->d_p, nullptr);
        if (rr != HIPADJ_OK) rc = rr;
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = HIPADJ_ERR_HIP;
    } else rc = HIPADJ_ERR_HIP;
    (void)hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream);
    release();
    if (rc != HIPADJ_OK) return rc;
    err[0] = worst[0]; err[1] = worst[1]; avail = true;
    return HIPADJ_OK;
}
// The first reverse pass of a runtime model whose reverse kernel spills heavily (user_prepare): run the -O3 build, then the -O1 build, compare
// du0 and dp on the host.  Agreement (1e-9 relative, and no non-finite flag from the -O3 run): the -O3 build stays.  Otherwise the -O1 build is
// used from here on and a note goes to stderr.  The outputs handed back are those of the build that stays.  This FIRST reverse pass of such a handle
// synchronises the stream and copies du0 / dp to the host (include/hipadj.h says so at hipadj_adjoint_dev): it cannot be part of a stream capture.
static int user_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    if (h->rtc_selftest != 1) return user_adjoint_run(h, d_cot, d_du0, d_dp);
    h->rtc_selftest = 2;
    const size_t n0 = (size_t)h->N * h->n, n1 = h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np;
    std::vector<double> a0(n0), a1(n1), b0(n0), b1(n1);
    auto fetch = [&](std::vector<double>& x0, std::vector<double>& x1, int& flag) -> int {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemcpy(x0.data(), d_du0, sizeof(double) * n0, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(x1.data(), d_dp, sizeof(double) * n1, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost));
        return HIPADJ_OK;
    };
    // -O1 first, -O3 second: on agreement the outputs left in du0 / dp are those of the build that stays
    int flag_a = 0, flag_b = 0;
    std::swap(h->uf_main, h->uf_main_alt);                 // the -O1 build
    TRY(user_adjoint_run(h, d_cot, d_du0, d_dp));
    TRY(fetch(a0, a1, flag_a));
    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));   // in stream order with the second run
    std::swap(h->uf_main, h->uf_main_alt);                 // the -O3 build
    TRY(user_adjoint_run(h, d_cot, d_du0, d_dp));
    TRY(fetch(b0, b1, flag_b));
    // fixed step: the two builds do the same arithmetic up to contraction / reassociation (1e-9).  Adaptive Tsit5: they may legitimately accept different step
    // sequences, so they only have to agree at the level of the solver's own tolerances
    const double tol = h->adaptive ? std::max(1e-9, 100.0 * std::max(h->cfg.abstol, h->cfg.reltol)) : 1e-9;
    auto differ = [tol](const std::vector<double>& x, const std::vector<double>& y) {   // x: the build under test, y: the reference build
        double scale = 0.0, d = 0.0;
        for (size_t i = 0; i < x.size(); ++i) { if (!(std::fabs(x[i]) <= 1.79e308)) return true; scale = std::max(scale, std::fabs(y[i])); d = std::max(d, std::fabs(x[i] - y[i])); }
        return d > tol * (scale > 0.0 ? scale : 1.0);
    };
    const bool bad = (flag_b & 1) || differ(b0, a0) || differ(b1, a1);
    if (bad && !(flag_a & 1)) {
        // A disagreement says that ONE of the builds is wrong.  Ask neither: central differences of the loss through the forward kernel (user_ground_truth).
        {
            const std::vector<double>* c0[2] = {&a0, &b0}; const std::vector<double>* c1[2] = {&a1, &b1};     // 0: the -O1 build, 1: the -O3 build
            double gerr[2]; bool avail = false;
            TRY(user_ground_truth(h, d_cot, c0, c1, gerr, avail));
            if (avail) {
                const double GT_TOL = 1e-4;      // central differences at eps = 1e-6 are good to ~1e-8 on smooth problems; a miscompiled kernel is off by O(1)
                const bool ok1 = gerr[0] < GT_TOL, ok3 = gerr[1] < GT_TOL && !(flag_b & 1);
                if (ok3 && (!ok1 || gerr[1] <= gerr[0])) {          // the -O3 build is right (its outputs are in place: it ran last... before the ground truth's forward solves; re-run for the outputs)
                    h->rtc_selftest = 4;
                    std::fprintf(stderr, "hipadj: the -O1 build of the reverse kernel of runtime model %d disagrees with its -O3 build; finite differences of the forward solve side with the -O3 build (relative error %.1e vs %.1e): keeping it (DESIGN.md 6.8)\n", h->cfg.model, gerr[1], gerr[0]);
                    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
                    return user_adjoint_run(h, d_cot, d_du0, d_dp);
                }
                if (ok1) {
                    h->rtc_selftest = 3;
                    std::swap(h->uf_main, h->uf_main_alt);          // the -O1 build from here on
                    std::fprintf(stderr, "hipadj: the -O3 build of the reverse kernel of runtime model %d disagrees with its -O1 build on the first reverse pass; finite differences of the forward solve side with the -O1 build (relative error %.1e vs %.1e): using the -O1 build (DESIGN.md 6.8)\n", h->cfg.model, gerr[0], gerr[1]);
                    HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
                    return user_adjoint_run(h, d_cot, d_du0, d_dp);
                }
                // neither candidate is within GT_TOL of the differences: on a stiff or chaotic long-horizon problem the differences themselves (eps = 1e-6) can be off by more than
                // that, so this is no verdict — fall through to the reproducibility tie-break below instead of refusing
                std::fprintf(stderr, "hipadj: runtime model %d: the -O3 and -O1 builds of the reverse kernel disagree and finite differences of the forward solve match neither (relative errors %.1e / %.1e): deciding by reproducibility (DESIGN.md 6.8)\n", h->cfg.model, gerr[1], gerr[0]);
            }
        }
        // No ground truth for this configuration (adaptive stepper, continuous cost): the older tie-break.  Every reverse kernel is bit-reproducible on fixed inputs (fixed summation orders, no
        // arrival-order dependence), so the tie-break is a second run of each: the -O1 build is used if it reproduces itself (the case this test was
        // written for); if it does not, but the -O3 build does, the -O3 build stays (seen once: a one-launch Backsolve kernel of an 8-state model whose
        // -O1 build gave different wrong answers on every run); if neither does, the handle refuses to hand out gradients.
        auto same = [](const std::vector<double>& x, const std::vector<double>& y) { return std::memcmp(x.data(), y.data(), sizeof(double) * x.size()) == 0; };
        std::vector<double> c0(n0), c1(n1); int flag_c = 0;
        std::swap(h->uf_main, h->uf_main_alt);             // the -O1 build, twice more: three identical results before it is trusted over the -O3 build
        bool o1_repro = true;
        for (int rep = 0; rep < 2 && o1_repro; ++rep) {
            HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
            TRY(user_adjoint_run(h, d_cot, d_du0, d_dp));
            TRY(fetch(c0, c1, flag_c));
            o1_repro = same(c0, a0) && same(c1, a1);
        }
        if (o1_repro) {
            h->rtc_selftest = 3;                            // ... from here on, and its outputs for this call
            std::fprintf(stderr, "hipadj: the -O3 build of the reverse kernel of runtime model %d disagrees with its -O1 build on the first reverse pass; using the -O1 build (DESIGN.md 6.8)\n", h->cfg.model);
            return HIPADJ_OK;
        }
        std::swap(h->uf_main, h->uf_main_alt);             // the -O3 build
        HIP_TRY(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
        TRY(user_adjoint_run(h, d_cot, d_du0, d_dp));
        TRY(fetch(c0, c1, flag_c));
        if (same(c0, b0) && same(c1, b1) && !(flag_b & 1)) {
            h->rtc_selftest = 4;
            std::fprintf(stderr, "hipadj: the -O1 build of the reverse kernel of runtime model %d disagrees with its -O3 build and does not reproduce itself; keeping the -O3 build (DESIGN.md 6.8)\n", h->cfg.model);
            return HIPADJ_OK;
        }
        h->rtc_selftest = 1;                                // nothing settled: the next call tests again
        HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "the -O3 and -O1 builds of the reverse kernel of runtime model %d disagree and neither reproduces itself: no trustworthy build (DESIGN.md 6.8)", h->cfg.model);
    }
    // agreement (or both non-finite: a diverged trajectory — the flag of the last run stands for the caller's check): the -O3 build stays, its outputs are in place
    return HIPADJ_OK;
}

static int adjoint_dispatch(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    if (h->wide) return wide_adjoint(h, d_cot, d_du0, d_dp);
    if (h->user) return user_adjoint(h, d_cot, d_du0, d_dp);
    if (h->field) { DISPATCH_GRID(h, field_adjoint, h, d_cot, d_du0, d_dp); }
    if (h->mlp) { DISPATCH_HIDDEN(h, mlp_adjoint_launch, h, d_cot, d_du0, d_dp); }
    if (h->adaptive) { DISPATCH_MODEL(h, adaptive_adjoint, h, d_cot, d_du0, d_dp); }
    DISPATCH_MODEL(h, adjoint_impl, h, d_cot, d_du0, d_dp);
}

// test hook of the overlapped all-reduce (HIPADJ_TEST_COMM_DELAY, hipadj_adjoint_dev): spin `ticks` of the 100 MHz wall clock, then dp *= 2
__global__ void k_test_delay_scale(double* __restrict__ dp, int np, long ticks) {
    const long t0 = (long)wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) {}
    for (int j = threadIdx.x; j < np; j += blockDim.x) dp[j] *= 2.0;
}

extern "C" int hipadj_forward_dev(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!d_u0 || !d_p) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "u0 and p must be non-NULL");
    if (h->route) return route_forward_dev(h, d_u0, d_p, d_out);
    if (h->multi) return multi_forward_dev(h, d_u0, d_p, d_out);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    harvest_timing(h, false);
    // keep private copies: the adjoint needs p, and u0 may be released by the caller
    const size_t pb = sizeof(double) * (h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np);
    if (d_p != h->d_p) HIP_TRY(h, hipMemcpyAsync(h->d_p, d_p, pb, hipMemcpyDeviceToDevice, h->stream));
    h->p_dev_last = h->d_p;
    // a runtime model whose two builds still have to be compared: the arbiter of a disagreement (user_ground_truth) differentiates the forward solve around THESE inputs
    if (h->rtc_selftest == 1 && d_u0 != h->d_u0) HIP_TRY(h, hipMemcpyAsync(h->d_u0, d_u0, sizeof(double) * (size_t)h->N * h->n, hipMemcpyDeviceToDevice, h->stream));
    if (h->d_tcnt) {   // arrival counters of the one-launch reverse pass: each is reset by its last arriver; a pass that died half-way (a faulting
                       // user model) must not poison the next solve, so every forward solve starts from zeros (in stream order, off the reverse path)
        HIP_TRY(h, hipMemsetAsync(h->d_tcnt, 0, sizeof(unsigned) * (size_t)h->tcnt_n, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_ticket, 0, sizeof(unsigned), h->stream));
    }
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
    if (h->route) return route_adjoint_dev(h, d_dLdu, d_du0, d_dp);
    if (h->multi) return multi_adjoint_dev(h, d_dLdu, d_du0, d_dp);
    if (h->cfg.loss_kind == HIPADJ_LOSS_LSQ_DATA && h->M > 0 && !h->have_ldata) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "loss_kind = HIPADJ_LOSS_LSQ_DATA: hand the data block over first (hipadj_set_loss_data / hipadj_set_loss_data_dev)");
    // device-resident losses: the workgroup families read the handle's data block in the cotangents' place (the lane family streams its transposed copy in d_cotT)
    if (h->cfg.loss_kind == HIPADJ_LOSS_LSQ_DATA || h->cfg.loss_kind == HIPADJ_LOSS_MODEL) d_dLdu = h->have_ldata ? h->d_ldata : nullptr;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    harvest_timing(h, false);
    const bool overlap = h->comm && h->comm_overlap && h->comm_stream;
    // overlap (hipadj_comm_overlap): the caller alternates between TWO dp buffers; this pass may only overwrite its buffer once the all-reduce of the pass before last,
    // which used the same one, is done
    if (overlap && h->comm_seq >= 2) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->comm_done[h->comm_seq & 1], 0));
    TRY(adjoint_dispatch(h, d_dLdu, d_du0, d_dp));
    if (overlap) {   // the all-reduce on its own stream, after this pass; the NEXT pass on h->stream does not wait for it (a 24-byte all-reduce is ~10-20 us of latency
                     // against a 30 us shard pass: in-stream it is a third of the step)
        HIP_TRY(h, hipEventRecord(h->comm_ready, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
        const int rc = rccl_api().AllReduce(d_dp, d_dp, (size_t)h->np, RCCL_DOUBLE, RCCL_SUM, h->comm, h->comm_stream);
        if (rc != 0) { h->err = rccl_error("ncclAllReduce", rc); return HIPADJ_ERR_RCCL; }
        // test hook (tests/test_gpu_parity.py): a one-rank all-reduce is the identity, so nothing on a 1-GPU box could tell whether a consumer of dp waited for the second
        // stream.  HIPADJ_TEST_COMM_DELAY=<microseconds> makes the collective SLOW and VISIBLE: a spin of that length, then dp *= 2, both on the second stream in front of comm_done.
        if (h->comm_test_delay > 0) { hipLaunchKernelGGL(k_test_delay_scale, dim3(1), dim3(64), 0, h->comm_stream, d_dp, h->np, h->comm_test_delay * 100L); HIP_TRY(h, hipGetLastError()); }
        HIP_TRY(h, hipEventRecord(h->comm_done[h->comm_seq & 1], h->comm_stream));
        ++h->comm_seq;
    } else if (h->comm) {   // the one exchange of the sharded ensemble: dp = sum over the ranks' shards, in-stream
        const int rc = rccl_api().AllReduce(d_dp, d_dp, (size_t)h->np, RCCL_DOUBLE, RCCL_SUM, h->comm, h->stream);
        if (rc != 0) { h->err = rccl_error("ncclAllReduce", rc); return HIPADJ_ERR_RCCL; }
    }
    h->st.adjoint_calls++;
    return HIPADJ_OK;
}

static int upload_block(hipadj_handle* h, double* d_dst, const double* src, size_t count);
static int download_block(hipadj_handle* h, double* dst, const double* d_src, size_t count);
// ---- device-resident discrete losses: the data block, the loss value, cotangents in the streaming layout -----------------------------------------------
static int loss_data_install(hipadj_handle* h) {   // h->d_ldata [N][M][n] is in place (stream order): the lane family's transposed copy
    const bool lane = !h->wide && !h->field && !h->mlp;
    if (lane && h->d_cotT && h->M > 0) TRY(launch_transpose_to_soa(h, h->d_ldata, h->d_cotT, h->M * h->n));
    h->have_ldata = true;
    return HIPADJ_OK;
}
static int loss_data_buffer(hipadj_handle* h) {
    if (h->cfg.loss_kind != HIPADJ_LOSS_LSQ_DATA && h->cfg.loss_kind != HIPADJ_LOSS_MODEL) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_set_loss_data: the handle's loss_kind is neither HIPADJ_LOSS_LSQ_DATA nor HIPADJ_LOSS_MODEL");
    if (h->M <= 0) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_set_loss_data: the handle has no loss times");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->d_ldata) { TRY(dev_alloc(h, &h->d_ldata, (size_t)h->N * h->M * h->n)); h->st.workspace_bytes = h->ws_bytes; }
    return HIPADJ_OK;
}
extern "C" int hipadj_set_loss_data_dev(hipadj_handle* h, const double* d_data) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!d_data) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "data must be non-NULL");
    if (h->route) return route_set_loss_data(h, d_data, true);
    if (h->multi) return multi_set_loss_data(h, d_data, true);
    TRY(loss_data_buffer(h));
    HIP_TRY(h, hipMemcpyAsync(h->d_ldata, d_data, sizeof(double) * (size_t)h->N * h->M * h->n, hipMemcpyDeviceToDevice, h->stream));
    return loss_data_install(h);
}
extern "C" int hipadj_set_loss_data(hipadj_handle* h, const double* data) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!data) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "data must be non-NULL");
    if (h->route) return route_set_loss_data(h, data, false);
    if (h->multi) return multi_set_loss_data(h, data, false);
    TRY(loss_data_buffer(h));
    TRY(upload_block(h, h->d_ldata, data, (size_t)h->N * h->M * h->n));
    TRY(loss_data_install(h));
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // the caller's buffer may go away
    return HIPADJ_OK;
}

extern "C" int hipadj_loss_value_dev(hipadj_handle* h, const double* d_out, double* d_loss) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!d_out || !d_loss) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "out and loss must be non-NULL");
    if (h->route) return route_loss_value(h, d_out, d_loss, true);
    if (h->multi) return multi_loss_value(h, d_out, d_loss, true);
    const int kind = h->cfg.loss_kind;
    if (kind == HIPADJ_LOSS_COTANGENT) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_loss_value: a cotangent handle does not know the loss (its gradient comes from the caller's AD)");
    if (kind == HIPADJ_LOSS_LSQ_DATA && !h->have_ldata) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_loss_value: hand the data block over first (hipadj_set_loss_data)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const long total = (long)h->N * h->M * h->n;
    // the loss time at t0 does not enter the loss under no_start (src/adjoint_common.jl:761)
    const int m0 = (h->cfg.no_start && !h->save_times.empty() && std::fabs(h->save_times[0] - h->cfg.t0) <= 1e-12 * std::max(1.0, std::fabs(h->cfg.t0))) ? 1 : 0;
    if (total == 0) { HIP_TRY(h, hipMemsetAsync(d_loss, 0, sizeof(double), h->stream)); return HIPADJ_OK; }
    if (kind == HIPADJ_LOSS_MODEL) {
        bool has_value = false;
        if (h->wide || !user_has_dloss(h->cfg.model, &has_value) || !has_value)
            HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_loss_value: the model's discrete loss was given as gradient bodies; register the loss itself with hipadj_model_set_discrete_loss_function (lane models)");
        if (!h->lf_value) {
            const std::vector<std::string> ex = {"hipadj::k_user_loss_value<hipadj::UserModel>"};
            std::vector<char> code; std::map<std::string, std::string> low;
            TRY(user_compile(h->cfg.model, ex, code, low, h->err));
            HIP_TRY(h, hipModuleLoadData(&h->lmod, code.data()));
            HIP_TRY(h, hipModuleGetFunction(&h->lf_value, h->lmod, low[ex[0]].c_str()));
        }
        if (!h->d_save_t) {   // (fixed-step handles on the grid keep no device copy of the save times)
            TRY(dev_alloc(h, &h->d_save_t, (size_t)h->M));
            HIP_TRY(h, hipMemcpyAsync(h->d_save_t, h->save_times.data(), sizeof(double) * h->M, hipMemcpyHostToDevice, h->stream));
        }
        if (!h->d_lpart) TRY(dev_alloc(h, &h->d_lpart, (size_t)std::max<long>(h->N, (total + LV_CHUNK - 1) / LV_CHUNK)));
        long Nl = h->N, ldp = h->cfg.p_shared ? 0 : h->np; int M = h->M, m0_ = m0;
        const double* dd = h->have_ldata ? h->d_ldata : nullptr; const double* pp = h->p_dev_last ? h->p_dev_last : h->d_p; const double* st = h->d_save_t;
        void* args[] = {&Nl, &M, &m0_, &ldp, &d_out, &dd, &pp, &st, &h->d_lpart};
        HIP_TRY(h, hipModuleLaunchKernel(h->lf_value, (unsigned)((h->N + 255) / 256), 1, 1, 256, 1, 1, 0, h->stream, args, nullptr));
        hipLaunchKernelGGL(k_sum_fixed, dim3(1), dim3(256), 0, h->stream, h->N, (const double*)h->d_lpart, d_loss);
        HIP_TRY(h, hipGetLastError());
        return HIPADJ_OK;
    }
    const long nb = (total + LV_CHUNK - 1) / LV_CHUNK;
    if (!h->d_lpart) TRY(dev_alloc(h, &h->d_lpart, (size_t)std::max<long>(h->N, nb)));
    hipLaunchKernelGGL(k_loss_value, dim3((unsigned)nb), dim3(256), 0, h->stream, total, h->M, h->n, m0, kind, h->cfg.loss_shift, h->cfg.loss_scale != 0.0 ? h->cfg.loss_scale : 1.0,
                       d_out, (const double*)h->d_ldata, h->d_lpart);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL(k_sum_fixed, dim3(1), dim3(256), 0, h->stream, nb, (const double*)h->d_lpart, d_loss);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
extern "C" int hipadj_loss_value(hipadj_handle* h, const double* out, double* loss) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!out || !loss) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "out and loss must be non-NULL");
    if (h->route) return route_loss_value(h, out, loss, false);
    if (h->multi) return multi_loss_value(h, out, loss, false);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // d_io_a is the host API's staging block of [N][M][n]; the value lands in a word of its own (it used to overwrite du0's first entry in the staging buffer)
    if (!h->d_lval) TRY(dev_alloc(h, &h->d_lval, 1));
    TRY(upload_block(h, h->d_io_a, out, (size_t)h->N * h->M * h->n));
    TRY(hipadj_loss_value_dev(h, h->d_io_a, h->d_lval));
    return download_block(h, loss, h->d_lval, 1);
}

extern "C" int hipadj_soa_stride(hipadj_handle* h, int64_t* ld) {
    if (!h || !ld) return HIPADJ_ERR_INVALID_ARG;
    HIPADJ_NO_MULTI(h, "hipadj_soa_stride (the streaming layout is padded per shard)");
    if (h->route || h->wide || h->field || h->mlp) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "the streaming cotangent layout belongs to the lane-per-trajectory family; this handle's family reads [N][M][n] in place");
    *ld = h->Npad;
    return HIPADJ_OK;
}
extern "C" int hipadj_adjoint_dev_soa(hipadj_handle* h, const double* d_dLdu_soa, double* d_du0, double* d_dp) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    HIPADJ_NO_MULTI(h, "hipadj_adjoint_dev_soa (the streaming layout is padded per shard)");
    if (h->route || h->wide || h->field || h->mlp) HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "hipadj_adjoint_dev_soa: the streaming cotangent layout belongs to the lane-per-trajectory family; this handle's family reads [N][M][n] in place");
    if (h->cfg.loss_kind != HIPADJ_LOSS_COTANGENT || h->M <= 0) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "hipadj_adjoint_dev_soa: the handle takes no cotangents (loss_kind / no loss times)");
    if (!d_dLdu_soa) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    if (h->rtc_selftest == 1) HIPADJ_FAIL(h, HIPADJ_ERR_STATE, "hipadj_adjoint_dev_soa: the first reverse pass of this handle cross-checks two builds of a runtime-compiled kernel; run hipadj_adjoint_dev once first");
    h->cot_soa = d_dLdu_soa;
    const int rc = hipadj_adjoint_dev(h, d_dLdu_soa, d_du0, d_dp);
    h->cot_soa = nullptr;
    return rc;
}

// The copier threads of the host-pointer calls: a process-wide pool, started at the first block of 4 MB and more and parked on a condition variable in between (creating eight
// threads per call cost 0.2-0.3 ms of a 0.9 ms call: profiles/r6_host_api_ab.jsonl).  run(nt, job) hands job(t), t = 0 .. nt-1, to the pool and returns at once; wait() blocks until
// every job(t) has returned.  One block at a time (the mutex of the host-pointer transfer that owns it).
struct XferPool {
    std::mutex own;                      // held by the transfer using the pool
    std::mutex m; std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    std::function<void(unsigned)> job; unsigned gen = 0, active = 0, njob = 0; bool stop = false;
    void ensure(unsigned nt) {
        while (th.size() < nt) { const unsigned t = (unsigned)th.size(); th.emplace_back([this, t] { loop(t); }); }
    }
    void loop(unsigned t) {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (t >= njob) continue;
            auto f = job;
            lk.unlock(); f(t); lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    void run(unsigned nt, std::function<void(unsigned)> f) {
        ensure(nt);
        { std::lock_guard<std::mutex> lk(m); job = std::move(f); njob = nt; active = nt; ++gen; }
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return active == 0; }); }
    ~XferPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto& x : th) if (x.joinable()) x.join(); }
};
static XferPool& xfer_pool() { static XferPool P; return P; }
// Pinned staging of the host-pointer calls.  host_copy_par: memcpy by a few host threads (one thread moves ~10 GB/s, the link 50-60).
static void host_copy_par(double* dst, const double* src, size_t count) {
    const size_t bytes = count * sizeof(double);
    unsigned nt = std::thread::hardware_concurrency(); nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
    if (const char* e = std::getenv("HIPADJ_HOST_COPY_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) nt = (unsigned)v; }
    if (bytes < ((size_t)4 << 20) || nt == 1) { std::memcpy(dst, src, bytes); return; }
    XferPool& P = xfer_pool();
    std::lock_guard<std::mutex> own(P.own);
    const size_t per = (count + nt - 1) / nt;
    P.run(nt - 1, [=](unsigned t) { const size_t a = per * (t + 1), b = std::min(count, a + per); if (a < b) std::memcpy(dst + a, src + a, (b - a) * sizeof(double)); });
    std::memcpy(dst, src, std::min(per, count) * sizeof(double));
    P.wait();
}
// Pipelined staging (round 6): a block of 4 MB and more crosses in chunks, the CPU copy of one chunk running while the DMA of its neighbour is in flight —
// the host-pointer forward / reverse calls pay max(CPU copy, DMA) + one chunk instead of their sum (24 MB: ~0.87 -> ~0.55 ms).  The copier threads (the pool above) each
// copy their slice of every chunk in order and counts the chunk's arrivals; the calling thread alone talks to the runtime (issues the uploads / waits for the download events).
static constexpr size_t HIPADJ_XFER_MIN_BYTES = (size_t)4 << 20;
static unsigned xfer_threads() {
    unsigned nt = std::thread::hardware_concurrency(); nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
    if (const char* e = std::getenv("HIPADJ_HOST_COPY_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) nt = (unsigned)v; }
    return nt;
}
static int xfer_chunks(size_t count) {      // chunks of ~3 MB, at most 16 (the handle's events)
    static const bool off = [] { const char* e = std::getenv("HIPADJ_HOST_PIPELINE"); return e && e[0] == '0'; }();      // A/B hook
    if (off || count * sizeof(double) < HIPADJ_XFER_MIN_BYTES) return 1;
    static const size_t chunk = [] { const char* e = std::getenv("HIPADJ_HOST_CHUNK_MB"); const int v = e ? std::atoi(e) : 0; return (size_t)(v >= 1 && v <= 64 ? v : 3) << 20; }();      // A/B hook: no size between 1 and 12 MB stands out of the run-to-run spread (profiles/r6_host_api_chunk_ab.jsonl)
    const size_t c = (count * sizeof(double) + chunk - 1) / chunk;
    return (int)(c > 16 ? 16 : c);
}
// host -> pinned block -> device
static int stage_up_pipelined(hipadj_handle* h, double* d_dst, double* pin, const double* src, size_t count, int C) {
    const unsigned nt = xfer_threads();
    const size_t per = (count + C - 1) / C;
    std::vector<std::atomic<unsigned>> done(C);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    auto work = [&](unsigned t) {
        for (int c = 0; c < C; ++c) {
            const size_t a = per * c, b = std::min(count, a + per), len = b - a, sl = (len + nt - 1) / nt, x = std::min(len, sl * t), y = std::min(len, x + sl);
            if (x < y) std::memcpy(pin + a + x, src + a + x, (y - x) * sizeof(double));
            done[c].fetch_add(1, std::memory_order_release);
        }
    };
    XferPool& P = xfer_pool();
    std::lock_guard<std::mutex> own(P.own);
    P.run(nt, work);
    hipError_t e = hipSuccess;
    for (int c = 0; c < C; ++c) {
        while (done[c].load(std::memory_order_acquire) < nt) std::this_thread::yield();
        const size_t a = per * c, b = std::min(count, a + per);
        if (e == hipSuccess && a < b) e = hipMemcpyAsync(d_dst + a, pin + a, (b - a) * sizeof(double), hipMemcpyHostToDevice, h->stream);
    }
    P.wait();
    HIP_TRY(h, e);
    return HIPADJ_OK;
}
// device -> pinned block, enqueue: one copy + one event per chunk (in-stream; nothing waits)
static int stage_down_enqueue(hipadj_handle* h, double* pin, const double* d_src, size_t count) {
    const int C = xfer_chunks(count);
    h->xfer_chunks = C;
    if (C == 1) { HIP_TRY(h, hipMemcpyAsync(pin, d_src, count * sizeof(double), hipMemcpyDeviceToHost, h->stream)); return HIPADJ_OK; }
    while (h->xfer_nev < C) { HIP_TRY(h, hipEventCreateWithFlags(&h->xfer_ev[h->xfer_nev], hipEventDisableTiming)); ++h->xfer_nev; }
    const size_t per = (count + C - 1) / C;
    for (int c = 0; c < C; ++c) {
        const size_t a = per * c, b = std::min(count, a + per);
        if (a < b) HIP_TRY(h, hipMemcpyAsync(pin + a, d_src + a, (b - a) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipEventRecord(h->xfer_ev[c], h->stream));
    }
    return HIPADJ_OK;
}
// ... finish: pinned block -> the caller's array, chunk by chunk as the events complete
static int stage_down_finish(hipadj_handle* h, double* dst, const double* pin, size_t count) {
    const int C = h->xfer_chunks;
    if (C <= 1) { HIP_TRY(h, hipStreamSynchronize(h->stream)); host_copy_par(dst, pin, count); return HIPADJ_OK; }
    const unsigned nt = xfer_threads();
    const size_t per = (count + C - 1) / C;
    std::atomic<int> ready(0);      // chunks [0, ready) have arrived; -1: a runtime error, the copiers stop
    auto work = [&](unsigned t) {
        for (int c = 0; c < C; ++c) {
            int r;
            while ((r = ready.load(std::memory_order_acquire)) <= c) { if (r < 0) return; std::this_thread::yield(); }
            const size_t a = per * c, b = std::min(count, a + per), len = b - a, sl = (len + nt - 1) / nt, x = std::min(len, sl * t), y = std::min(len, x + sl);
            if (x < y) std::memcpy(dst + a + x, pin + a + x, (y - x) * sizeof(double));
        }
    };
    XferPool& P = xfer_pool();
    std::lock_guard<std::mutex> own(P.own);
    P.run(nt, work);
    hipError_t e = hipSuccess;
    for (int c = 0; c < C && e == hipSuccess; ++c) { e = hipEventSynchronize(h->xfer_ev[c]); if (e == hipSuccess) ready.store(c + 1, std::memory_order_release); }
    if (e != hipSuccess) ready.store(-1, std::memory_order_release);
    P.wait();
    HIP_TRY(h, e);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return HIPADJ_OK;
}
// The staging block lives in its OWN anonymous mapping with an inaccessible guard page on either side — never in the brk heap.  Round 5's block was posix_memalign memory: once
// glibc's dynamic mmap threshold has risen (a larger block was freed earlier), such a block comes from the heap, page-adjacent to the caller's small arrays, and registering it
// there made the next pageable device-to-host copy into a fresh heap array just below it die with "Memory access fault by GPU ... Write access to a read-only page" (the one-in-
// a-dozen abort of round 5's bench; reproduced in seconds by scripts/r6/fault_stress.py, reduced in scripts/r6/repro_readonly_fault.hip; profiles/r6_fault_*).  The guard pages
// keep the kernel from merging the mapping with a neighbour's and the registration from sharing a page with anything else.
static void host_pin_release(hipadj_handle* h) {
    if (!h->h_pin) return;
    (void)hipHostUnregister(h->h_pin);
    if (h->pin_map_len) (void)munmap((char*)h->h_pin - 4096, h->pin_map_len); else std::free(h->h_pin);      // (else: the reproduction hook's heap block)
    h->h_pin = nullptr; h->pin_count = 0; h->pin_map_len = 0;
}
static double* host_pin(hipadj_handle* h, size_t count) {      // the handle's pinned block, grown on demand; nullptr: no pinned memory to be had (the pageable copy still works)
    if (h->pin_count >= count) return h->h_pin;
    if (std::getenv("HIPADJ_NO_PINNED")) return nullptr;
    if (h->h_pin) { (void)hipStreamSynchronize(h->stream); (void)hipDeviceSynchronize(); host_pin_release(h); }   // (device-wide: the block may have been read on a stream the handle has since left)
    // ordinary (cached) pages, registered with the runtime: hipHostMalloc's default block is fine-grained coherent memory, which the host WRITES at a fraction of its memcpy rate
    // (measured: the staged upload took 12.5 ms against 7.7 ms for the plain pageable copy, profiles/r5_visit2_bench.json)
    const size_t bytes = count * sizeof(double), len = (bytes + 4095) / 4096 * 4096 + 2 * 4096;
    if (std::getenv("HIPADJ_PIN_HEAP")) {      // REPRODUCTION HOOK (scripts/r6/fault_ab.py): round 5's block, posix_memalign memory registered in place — never use otherwise
        void* q = nullptr;
        if (posix_memalign(&q, 4096, bytes) != 0 || !q) return nullptr;
        std::memset(q, 0, bytes);
        if (hipHostRegister(q, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); std::free(q); return nullptr; }
        h->h_pin = (double*)q; h->pin_count = count; h->pin_map_len = 0;
        if (std::getenv("HIPADJ_TRACE_PIN")) std::fprintf(stderr, "hipadj host_pin: handle %p registered [%p, %p) (HEAP block: reproduction hook)\n", (void*)h, q, (void*)((char*)q + bytes));
        return h->h_pin;
    }
    char* m = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == (char*)MAP_FAILED) return nullptr;
    (void)mprotect(m, 4096, PROT_NONE); (void)mprotect(m + len - 4096, 4096, PROT_NONE);
    void* q = m + 4096;
    std::memset(q, 0, bytes);                                                    // touch the pages before they are pinned
    if (hipHostRegister(q, len - 2 * 4096, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); (void)munmap(m, len); return nullptr; }
    h->h_pin = (double*)q; h->pin_count = count; h->pin_map_len = len;
    if (std::getenv("HIPADJ_TRACE_PIN")) std::fprintf(stderr, "hipadj host_pin: handle %p registered [%p, %p) (own mapping, guard pages)\n", (void*)h, q, (void*)((char*)q + len - 2 * 4096));
    return h->h_pin;
}
// ---- host-pointer transfers --------------------------------------------------------------------------------------------------------------------------------------------
// EVERY transfer between a caller's host array and the device goes through the handle's staging block (round 6): the host side of each DMA is memory this library owns,
// registered once, in its own mapping — the caller's arrays are only ever touched by the CPU.  Why: a pageable hipMemcpy makes the runtime pin the caller's pages on the fly and
// keep those pins cached; with the arrays of a numpy / Julia host (allocated, freed and re-allocated at the same heap addresses from call to call) a device-to-host copy of du0
// ended in "Memory access fault by GPU ... Write access to a read-only page" and took the host process with it (round 5: one bench run in a dozen; round 6: reproduced in the
// first seconds of scripts/r6/fault_stress.py, profiles/r6_fault_stress_reproduced_visit3.log; A/B of the three transfer modes: scripts/r6/fault_ab.py).  Staging costs one
// CPU copy per block (host_copy_par: a few threads at memcpy rate — 24 MB of out / Delta in ~0.5 ms) and removes the runtime's on-the-fly pinning from the path altogether.
// HIPADJ_HOST_DIRECT=1 restores the direct pageable copies (A/B); HIPADJ_NO_PINNED=1 or a failed registration fall back to them as well.
static bool host_direct() { static const bool v = [] { const char* e = std::getenv("HIPADJ_HOST_DIRECT"); return e && e[0] == '1'; }(); return v; }
// the handle's staging block with room for `count` doubles (grown on demand: a growth drains the device first, so it may only be asked for while nothing staged is pending —
// every call site asks ONCE, up front, for the most it will hold); nullptr: direct pageable copies
static double* stage_block(hipadj_handle* h, size_t count) { return host_direct() ? nullptr : host_pin(h, count + 8); }

// host -> device: count doubles of `src` into d_dst (asynchronous once the host copy is done; the stream is drained first: an earlier transfer may still be using the block)
static int upload_block(hipadj_handle* h, double* d_dst, const double* src, size_t count) {
    // (HIPADJ_HOST_DIRECT=1 keeps round 5's rule for this one: blocks of 1 MB and more through the registered block, smaller ones pageable)
    double* pin = (host_direct() && count * sizeof(double) >= ((size_t)1 << 20)) ? host_pin(h, count) : stage_block(h, count);
    if (pin) {
        static const bool trace = std::getenv("HIPADJ_HOST_TIMING") != nullptr;      // diagnosis: where a host-pointer call spends its time (stderr)
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const auto t0 = std::chrono::steady_clock::now();
        const int C = xfer_chunks(count);
        if (C > 1) TRY(stage_up_pipelined(h, d_dst, pin, src, count, C)); else host_copy_par(pin, src, count);
        const auto t1 = std::chrono::steady_clock::now();
        if (C == 1) HIP_TRY(h, hipMemcpyAsync(d_dst, pin, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (trace) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            const auto t2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, C > 1 ? "hipadj upload_block: %.1f MB in chunks, host copies + issue of the DMAs %.3f ms, rest of the DMAs %.3f ms\n" : "hipadj upload_block: %.1f MB, host copy into the pinned block %.3f ms, DMA %.3f ms\n", count * 8.0 / 1e6,
                         std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    } else HIP_TRY(h, hipMemcpyAsync(d_dst, src, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return HIPADJ_OK;
}
// device -> host, synchronous: count doubles of d_src into `dst`
static int download_block(hipadj_handle* h, double* dst, const double* d_src, size_t count) {
    double* pin = stage_block(h, count);
    if (pin) {
        TRY(stage_down_enqueue(h, pin, d_src, count));
        TRY(stage_down_finish(h, dst, pin, count));
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemcpyAsync(dst, d_src, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return HIPADJ_OK;
}

// host-pointer calls = enqueue (copies in, the device call, the copy out into the staging block: nothing waits for the device beyond the block's hand-over) + finish (drain,
// CPU copy into the caller's array); a handle over several devices enqueues on every shard before it finishes any (hipadj_multi.hpp)
static int forward_host_enqueue(hipadj_handle* h, const double* u0, const double* p, double* out) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t nu = (size_t)h->N * h->n, pc = h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np, no = (size_t)h->N * h->M * h->n;
    double* pin = stage_block(h, std::max(nu + pc, (out && h->M > 0) ? no : (size_t)0));
    if (pin) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(pin, u0, sizeof(double) * nu); std::memcpy(pin + nu, p, sizeof(double) * pc);
        HIP_TRY(h, hipMemcpyAsync(h->d_u0, pin, sizeof(double) * nu, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_p, pin + nu, sizeof(double) * pc, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->d_u0, u0, sizeof(double) * nu, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_p, p, sizeof(double) * pc, hipMemcpyHostToDevice, h->stream));
    }
    TRY(hipadj_forward_dev(h, h->d_u0, h->d_p, out ? h->d_io_a : nullptr));
    if (out && h->M > 0) {      // (in-stream behind the uploads that read the block)
        if (pin) TRY(stage_down_enqueue(h, pin, h->d_io_a, no)); else HIP_TRY(h, hipMemcpyAsync(out, h->d_io_a, sizeof(double) * no, hipMemcpyDeviceToHost, h->stream));
    }
    return HIPADJ_OK;
}
static int forward_host_finish(hipadj_handle* h, double* out) {
    if (!out || h->M <= 0 || !h->h_pin || host_direct()) return HIPADJ_OK;      // direct mode: the copy went into `out` itself
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    TRY(stage_down_finish(h, out, h->h_pin, (size_t)h->N * h->M * h->n));
    return HIPADJ_OK;
}
extern "C" int hipadj_forward(hipadj_handle* h, const double* u0, const double* p, double* out) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!u0 || !p) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "u0 and p must be non-NULL");
    if (h->route) return route_forward(h, u0, p, out);
    if (h->multi) return multi_forward(h, u0, p, out);
    TRY(forward_host_enqueue(h, u0, p, out));
    TRY(forward_host_finish(h, out));
    return hipadj_synchronize(h);
}

// The host-pointer reverse call in two phases: adjoint_host_run (upload of the cotangents + the reverse pass, nothing waits) and adjoint_host_download (du0 / dp through the
// staging block once the stream is drained).  A handle over several devices runs phase 1 on every shard before phase 2 on any, so the devices still work concurrently
// (hipadj_multi.hpp).
static int adjoint_host_run(hipadj_handle* h, const double* dLdu) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const bool cot = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0;
    if (cot && !dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    if (cot) TRY(upload_block(h, h->d_io_a, dLdu, (size_t)h->N * h->M * h->n));
    TRY(hipadj_adjoint_dev(h, cot ? h->d_io_a : nullptr, h->d_du0, h->d_dp));
    // an overlapped all-reduce (hipadj_comm_overlap) runs on the handle's SECOND stream: the copy of dp is enqueued on the first one and has to wait for the collective
    // that was just recorded, not only for the reverse pass (with more than one rank the host could otherwise receive the un-reduced shard sum)
    if (h->comm && h->comm_overlap && h->comm_stream && h->comm_seq > 0) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->comm_done[(h->comm_seq - 1) & 1], 0));
    return HIPADJ_OK;
}
static int adjoint_host_download(hipadj_handle* h, double* du0, double* dp) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t nu = (size_t)h->N * h->n, pc = h->cfg.p_shared ? (size_t)h->np : (size_t)h->N * h->np;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double* pin = stage_block(h, nu + pc);
    HIP_TRY(h, hipMemcpyAsync(pin ? pin : du0, h->d_du0, sizeof(double) * nu, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(pin ? pin + nu : dp, h->d_dp, sizeof(double) * pc, hipMemcpyDeviceToHost, h->stream));
    if (pin) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(du0, pin, sizeof(double) * nu); std::memcpy(dp, pin + nu, sizeof(double) * pc);
    }
    return HIPADJ_OK;
}
extern "C" int hipadj_adjoint(hipadj_handle* h, const double* dLdu, double* du0, double* dp) {
    if (!h) return HIPADJ_ERR_INVALID_ARG;
    if (!du0 || !dp) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "du0 and dp must be non-NULL");
    if (h->route) return route_adjoint(h, dLdu, du0, dp);
    if (h->multi) {
        if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0 && !dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
        return multi_adjoint(h, dLdu, du0, dp);
    }
    static const bool trace = std::getenv("HIPADJ_HOST_TIMING") != nullptr;      // diagnosis (stderr): enqueue (upload + launches + download requests) and drain of one host-pointer call
    const auto t0 = std::chrono::steady_clock::now();
    TRY(adjoint_host_run(h, dLdu));
    TRY(adjoint_host_download(h, du0, dp));
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = hipadj_synchronize(h);
    if (trace) std::fprintf(stderr, "hipadj_adjoint: enqueue %.3f ms, drain %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return rc;
}
