// hipadj_user.hpp — runtime model ingestion for the lane-per-trajectory kernel family (host side of the library).
//
// The reference lets the caller hand in the right-hand side and its two vector-Jacobian products as plain functions,
//   ODEFunction(f!; vjp = (dlam, lam, u, p, t), vjp_p = (dgrad, lam, u, p, t))       src/derivative_wrappers.jl:284-359
// (priority vjp_p > paramjac > AD; exercised by test/Core3/user_vjp.jl:14-38, 77).  On the device the same seam is a
// struct with three inlined static functions (hipadj_models.hpp).  hipadj_model_register stores the three BODIES as HIP
// C++ text; at hipadj_create the kernels the configuration needs (forward solve, reverse sweep, segment composition /
// finishing) are instantiated for that struct and compiled for gfx950 with hiprtc from the library's own headers
// (read from csrc/ next to libhipadj.so), loaded with hipModuleLoadData and launched with hipModuleLaunchKernel.
// Nothing here is a CPU path: a model that fails to compile fails hipadj_create with the compiler log.
//
// hiprtc is bound with dlopen / dlmopen at first use, not at link time (rtc_api below: the toolkit's compiler, also inside a torch process).
#pragma once

#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <elf.h>
#include <limits.h>
#include <link.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <unistd.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "hipadj_plan.hpp"

namespace hipadj {

struct UserModelSrc {
    std::string name, f, vjp_u, vjp_p;
    std::string dgdu, dgdp;   // optional continuous cost (hipadj_model_set_cost)
    std::string gfun;         // ... or the cost itself (hipadj_model_set_cost_function): gradients by dual numbers
    bool has_cost = false;
    // optional discrete loss ON THE DEVICE (hipadj_model_set_discrete_loss[_function], hipadj_wmodel_set_discrete_loss): dgdu_discrete / dgdp_discrete of ReverseLossCallback
    std::string dl_du, dl_dp, dl_fun, wdloss;
    std::string waffect_vjp;           // wide models: the reverse callback of `affect` as text (hipadj_wmodel_set_affect)
    bool has_dloss = false;
    bool auto_vjp = false;    // only f was given: vjp_u / vjp_p by forward-mode dual numbers (hipadj_dual.hpp)
    std::string affect;       // DiscreteCallback affect body (hipadj_model_set_affect): modifies un (pre-set to u) from u, p, t; empty = identity
    std::string cc_cond, cc_affect;   // ContinuousCallback (hipadj_model_set_continuous_callback): the condition body assigns `c` from u, p, t; the affect body edits un (pre-set to u)
    int cc_dir = 0;                   // which crossings fire: 0 both, +1 upcrossings only, -1 downcrossings only (hipadj_model_set_callback_direction)
    int cc_maxev = 0, cc_ncond = 1;   // ... the capacity of the per-trajectory event list, and the components of the condition (VectorContinuousCallback: `out[k]`, `idx`)
    bool cols = true;         // the VJP bodies compile for Cols<G> (column bundles); cleared by user_compile when they do not
    bool has_mm = false;      // constant non-singular mass matrix (hipadj_model_set_mass_matrix): minv = M^{-1}, row-major n x n
    double minv[64] = {0};
    bool dae = false;         // constant SINGULAR mass matrix of a semi-explicit DAE (round 6): mm = M itself (row-major), zero rows = algebraic variables; Rosenbrock23 only
    double mm[64] = {0};
    int n = 0, np = 0, rev = 0;   // rev: bumped when the sources change, part of the code-cache key
    // wide model (hipadj_wmodel_register; hipadj_wide.hpp): SPMD bodies f / vjp for a workgroup of `threads` per trajectory
    bool wide = false;
    std::string wvjp;             // the joint VJP body (the reference's vecjacobian! contract)
    std::string wcost;            // continuous cost of a wide model (hipadj_wmodel_set_cost): SPMD body adding g_u into dlam and, WP, w g_p into gp / acc
    int threads = 0, nw = 0, nacc = 0, acc0 = 0;
    // hipadj_wmodel_declare_dense_chain (ABI 109): the model IS the dense chain widths[0] -> ... -> widths[L] (Lux's parameter order) with `chain_act` on every layer but the
    // last and the input map x -> x^chain_power: hipadj_create may then pick the kernel family itself (hipadj_route.hpp)
    std::vector<int> chain; int chain_power = 1, chain_act = 0;
};

struct UserRegistry {
    std::mutex mu;
    std::vector<UserModelSrc> models;                       // id = HIPADJ_MODEL_USER_BASE + index
    std::map<std::string, std::vector<char>> code_cache;    // (id | name expressions) -> gfx950 code object
    std::map<std::string, std::map<std::string, std::string>> lowered_cache;   // same key -> expression -> mangled name
};
inline UserRegistry& user_registry() { static UserRegistry r; return r; }

inline int user_model_sizes(int32_t model, int32_t* n, int32_t* np) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) return HIPADJ_ERR_INVALID_ARG;
    *n = R.models[idx].n; *np = R.models[idx].np;
    return HIPADJ_OK;
}

struct RtcApi {
    void* lib = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    hiprtcResult (*AddNameExpression)(hiprtcProgram, const char*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetLoweredName)(hiprtcProgram, const char*, const char**) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    std::string err, path;     // path: the bound libhiprtc
    bool isolated = false;     // loaded with dlmopen next to another hiprtc / comgr copy
    bool by_path = false;      // bound from the toolkit installation by path ($HIPADJ_HIPRTC with a path, $ROCM_PATH/lib, the build-time ROCm root) — not by soname
    char*** ns_environ = nullptr;   // `environ` of the C library copy inside that namespace
    // A link-map namespace holds its own libc, whose `environ` was copied when it was loaded.  setenv in the process (python: os.environ[...] = ...)
    // may move the array; the copy then dangles and the next getenv inside hiprtc / comgr walks freed memory.  Every entry into the namespace
    // first points its `environ` at the live array.
    void enter() const { if (ns_environ) *ns_environ = ::environ; }
};

// Which hiprtc compiles the runtime models.  A process that has torch loaded carries torch's own HIP runtime, hiprtc and comgr (the wheel
// bundles the ROCm it was built against), and binding hiprtc by soname picks THAT compiler — an older LLVM than the toolkit that built
// libhipadj.so.  The bundled ROCm 7.0 compiler miscompiles wide runtime models (5-state ring with a dense mass matrix, GaussAdjoint: parameter
// gradient wrong in the 6th digit and worse, the same translation unit is exact with the 7.2 toolkit) — so all device code of
// this library, static and runtime-compiled, comes from ONE compiler: the toolkit's libhiprtc ($HIPADJ_HIPRTC, else $ROCM_PATH/lib, else the
// build-time ROCm root).  When another hiprtc / comgr copy is already mapped, the toolkit's pair is loaded into its own link-map namespace
// (dlmopen): libhiprtc resolves libamd_comgr by soname, and in the global namespace that would be the older copy again.  The code object is
// loaded by whatever HIP runtime the process uses (code-object ABI v6 on both sides).  Without a toolkit installation: the soname binding.
#ifndef HIPADJ_ROCM_PATH
#define HIPADJ_ROCM_PATH "/opt/rocm"
#endif
inline std::string rtc_realpath(const std::string& p) { char b[PATH_MAX]; return realpath(p.c_str(), b) ? std::string(b) : std::string(); }
inline std::string rtc_dirname(const std::string& p) { const size_t k = p.rfind('/'); return k == std::string::npos ? std::string() : p.substr(0, k); }
struct RtcForeign { std::string dir; bool found; };
inline int rtc_phdr_cb(struct dl_phdr_info* info, size_t, void* u) {
    RtcForeign* f = (RtcForeign*)u;
    if (!info->dlpi_name || !info->dlpi_name[0]) return 0;
    const std::string rp = rtc_realpath(info->dlpi_name);
    const size_t k = rp.rfind('/');
    const std::string base = k == std::string::npos ? rp : rp.substr(k + 1);
    if ((base.compare(0, 9, "libhiprtc") == 0 || base.compare(0, 12, "libamd_comgr") == 0) && rtc_dirname(rp) != f->dir) f->found = true;
    return 0;
}

inline RtcApi& rtc_api() {
    static RtcApi A;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string want;
        if (const char* e = std::getenv("HIPADJ_HIPRTC")) want = e;                       // a path, or a bare soname = "whatever the process has"
        else {
            std::vector<std::string> roots;
            if (const char* r = std::getenv("ROCM_PATH")) roots.push_back(r);
            roots.push_back(HIPADJ_ROCM_PATH);
            for (const auto& r : roots) { const std::string c = rtc_realpath(r + "/lib/libhiprtc.so"); if (!c.empty()) { want = c; break; } }
        }
        if (!want.empty() && want.find('/') != std::string::npos) {
            const std::string rp = rtc_realpath(want);
            if (!rp.empty()) {
                RtcForeign f{rtc_dirname(rp), false};
                dl_iterate_phdr(rtc_phdr_cb, &f);
                A.lib = f.found ? dlmopen(LM_ID_NEWLM, rp.c_str(), RTLD_NOW | RTLD_LOCAL) : dlopen(rp.c_str(), RTLD_NOW | RTLD_LOCAL);
                if (A.lib) { A.path = rp; A.isolated = f.found; A.by_path = true; if (f.found) A.ns_environ = (char***)dlsym(A.lib, "environ"); }
            }
        } else if (!want.empty()) { A.lib = dlopen(want.c_str(), RTLD_NOW | RTLD_LOCAL); if (A.lib) A.path = want; }
        if (!A.lib) {
            const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
            for (const char* nm : names) { A.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (A.lib) { A.path = nm; break; } }
        }
        if (!A.lib) { A.err = "hiprtc not found (libhiprtc.so): runtime-compiled models are unavailable"; return; }
        auto S = [&](const char* s) { void* p = dlsym(A.lib, s); if (!p && A.err.empty()) A.err = std::string("hiprtc symbol missing: ") + s; return p; };
        A.CreateProgram = (decltype(A.CreateProgram))S("hiprtcCreateProgram");
        A.AddNameExpression = (decltype(A.AddNameExpression))S("hiprtcAddNameExpression");
        A.CompileProgram = (decltype(A.CompileProgram))S("hiprtcCompileProgram");
        A.GetProgramLogSize = (decltype(A.GetProgramLogSize))S("hiprtcGetProgramLogSize");
        A.GetProgramLog = (decltype(A.GetProgramLog))S("hiprtcGetProgramLog");
        A.GetLoweredName = (decltype(A.GetLoweredName))S("hiprtcGetLoweredName");
        A.GetCodeSize = (decltype(A.GetCodeSize))S("hiprtcGetCodeSize");
        A.GetCode = (decltype(A.GetCode))S("hiprtcGetCode");
        A.DestroyProgram = (decltype(A.DestroyProgram))S("hiprtcDestroyProgram");
    });
    return A;
}

// "path [own link-map namespace]; HIP x.y.z" of the bound hiprtc: the HIP version is the one the compiler itself reports (hiprtc defines
// HIP_VERSION_* for every program; a one-line probe prints them through #pragma message).
inline std::string rtc_describe() {
    RtcApi& A = rtc_api();
    if (!A.lib || !A.err.empty()) return A.err.empty() ? "hiprtc unavailable" : A.err;
    static std::string ver;
    static std::once_flag once;
    std::call_once(once, [&] {
        const char* src = "#define HIPADJ_S2(x) #x\n#define HIPADJ_S(x) HIPADJ_S2(x)\n"
                          "#pragma message(\"hipadj-probe \" HIPADJ_S(HIP_VERSION_MAJOR) \".\" HIPADJ_S(HIP_VERSION_MINOR) \".\" HIPADJ_S(HIP_VERSION_PATCH) \" end\")\n";
        hiprtcProgram prog = nullptr;
        A.enter();
        if (A.CreateProgram(&prog, src, "hipadj_probe.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return;
        const char* opts[] = {"--offload-arch=gfx950"};
        (void)A.CompileProgram(prog, 1, opts);
        size_t ls = 0; A.GetProgramLogSize(prog, &ls);
        std::string log(ls, '\0'); if (ls) A.GetProgramLog(prog, &log[0]);
        A.DestroyProgram(&prog);
        const size_t a = log.find("hipadj-probe "), b = log.find(" end", a == std::string::npos ? 0 : a);
        if (a != std::string::npos && b != std::string::npos) ver = log.substr(a + 13, b - a - 13);
    });
    return A.path + (A.isolated ? " [own link-map namespace]" : "") + "; HIP " + (ver.empty() ? "?" : ver);
}

// Is the bound compiler the one this library was built and tested with?  No when the toolkit's libhiprtc could not be bound by path (a torch wheel alone,
// no ROCm installation: the soname then resolves to the wheel's bundled pair) or when the compiler reports an older HIP than the build toolkit.  The
// bundled ROCm 7.0 pair miscompiled WIDE runtime models, so an untrusted compiler gets the round-1 limits back: segment lanes up to 64
// doubles of state (plan_seg_cap), and EVERY reverse kernel is cross-checked against its -O1 build on the first reverse pass (user_prepare), not only the
// heavily spilling ones.  HIPADJ_RTC_TRUST=1 / 0 overrides.  One note on stderr.  
inline bool rtc_trusted() {
    static const bool ok = [] {
        if (const char* e = std::getenv("HIPADJ_RTC_TRUST")) return e[0] != '0';
        RtcApi& A = rtc_api();
        if (!A.lib || !A.err.empty()) return true;                 // nothing will be compiled at all
        bool good = A.by_path;
        const std::string d = rtc_describe();
        const size_t k = d.rfind("HIP ");
        int maj = 0, min = 0;
        if (k != std::string::npos && std::sscanf(d.c_str() + k + 4, "%d.%d", &maj, &min) == 2) {
#if defined(HIP_VERSION_MAJOR) && defined(HIP_VERSION_MINOR)
            if (maj < HIP_VERSION_MAJOR || (maj == HIP_VERSION_MAJOR && min < HIP_VERSION_MINOR)) good = false;
#endif
        } else good = false;
        if (!good) std::fprintf(stderr, "hipadj: runtime models are compiled by %s, not by the toolkit this library was built with: segment lanes limited to 64 doubles, "
                                        "every reverse kernel cross-checked against its -O1 build (set HIPADJ_HIPRTC or ROCM_PATH; HIPADJ_RTC_TRUST=1 overrides)\n", d.c_str());
        return good;
    }();
    return ok;
}
inline int user_seg_cap() { return rtc_trusted() ? 0 : 64; }

// directory holding the library's kernel headers: $HIPADJ_CSRC_DIR, else csrc/ next to libhipadj.so
inline std::string user_csrc_dir() {
    if (const char* e = std::getenv("HIPADJ_CSRC_DIR")) return e;
    Dl_info info;
    if (dladdr((const void*)&user_csrc_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
    }
    return "csrc";
}

inline bool user_read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss; ss << f.rdbuf(); out = ss.str();
    return true;
}

inline std::string user_wide_struct(const UserModelSrc& m) {
    std::ostringstream o;
    const int na = m.nacc > 0 ? m.nacc : 1;
    o << "#include \"hipadj_wide.hpp\"\n"
      << "namespace hipadj {\n// runtime-registered wide model '" << m.name << "' (workgroup-per-trajectory family, hipadj_wide.hpp)\n"
      << "#define HIPADJ_W_FOR(i, n) for (int i = tid; i < (n); i += T)\n#define wg_sync() hipadj::wide_sync<T>()\n#define wg_sum(x) hipadj::wide_sum_all<T>(x)\n#define wg_sum2(a, b, sa, sb) hipadj::wide_sum2_all<T>(a, b, sa, sb)\n"
      // tanh of a model body = the 31-instruction form of the MFMA family (|difference| to libm 2.2e-16: hipadj_wide.hpp wide_tanh); the library tanh is ~150
      // instructions and dominated a joint VJP of the 2-50-2 neural ODE
      << "#define tanh(x) hipadj::wide_tanh(x)\n"
      << "struct UserW {\n    static constexpr int N = " << m.n << ", NP = " << m.np << ", T = " << m.threads << ", NW = " << m.nw << ", NACC = " << m.nacc
      << ", ACC0 = " << m.acc0 << ";\n"
      << "    static __device__ __forceinline__ void f(double* __restrict__ du, const double* __restrict__ u, const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {\n"
      << "        (void)u; (void)p; (void)t; (void)ws; (void)tid;\n" << m.f << "\n    }\n"
      << "    template <bool WP> static __device__ __forceinline__ void vjp(double* __restrict__ dlam, double* __restrict__ gp, double (&acc)[" << na << "], double w,\n"
      << "            const double* __restrict__ lam, const double* __restrict__ u, const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {\n"
      << "        (void)gp; (void)acc; (void)w; (void)u; (void)p; (void)t; (void)ws; (void)tid;\n" << m.wvjp << "\n    }\n"
      << "    template <bool WP> static __device__ __forceinline__ void cost(double* __restrict__ dlam, double* __restrict__ gp, double (&acc)[" << na << "], double w,\n"
      << "            const double* __restrict__ u, const double* __restrict__ p, double t, double* __restrict__ ws, int tid) {\n"
      << "        (void)dlam; (void)gp; (void)acc; (void)w; (void)u; (void)p; (void)t; (void)ws; (void)tid;\n" << m.wcost << "\n    }\n";
    if (m.has_dloss)   // the model's discrete loss (hipadj_wmodel_set_discrete_loss; HIPADJ_LOSS_MODEL): ADDS dl_i/du into dlam and, WP, dl_i/dp into gp / acc (hipadj_wide.hpp wide_jump)
        o << "    static constexpr bool HAS_DLOSS = true;\n"
          << "    template <bool WP> static __device__ __forceinline__ void dloss(double* __restrict__ dlam, double* __restrict__ gp, double (&acc)[" << na << "],\n"
          << "            const double* __restrict__ u, const double* __restrict__ p, double t, int i, const double* __restrict__ d, double* __restrict__ ws, int tid) {\n"
          << "        (void)dlam; (void)gp; (void)acc; (void)u; (void)p; (void)t; (void)i; (void)d; (void)ws; (void)tid;\n" << m.wdloss << "\n    }\n";
    if (!m.affect.empty())   // DiscreteCallback affect of a wide model and its reverse callback (hipadj_wmodel_set_affect): serial bodies run by ONE thread per trajectory
                             // between two pieces of an event chain (k_wide_affect / k_wide_affect_vjp, hipadj_wide.hpp) — an event happens a handful of times per solve
        o << "    static __device__ void affect(double* __restrict__ un, double* __restrict__ pn, const double* __restrict__ u, const double* __restrict__ p, double t) {\n"
          << "        (void)un; (void)pn; (void)u; (void)p; (void)t;\n" << m.affect << "\n    }\n"
          << "    static __device__ void affect_vjp(double* __restrict__ lo, double* __restrict__ go, const double* __restrict__ lam, const double* __restrict__ gp,\n"
          << "            const double* __restrict__ u, const double* __restrict__ p, double t) {\n"
          << "        (void)lo; (void)go; (void)lam; (void)gp; (void)u; (void)p; (void)t;\n" << m.waffect_vjp << "\n    }\n";
    o << "};\n#undef tanh\n}  // namespace hipadj\n";
    return o.str();
}

inline std::string user_model_struct(const UserModelSrc& m) {
    if (m.wide) return user_wide_struct(m);
    std::ostringstream o;
    o << "#include \"hipadj_kernels.hpp\"\n#include \"hipadj_adaptive.hpp\"\n#include \"hipadj_dual.hpp\"\n"
      << "namespace hipadj {\n// runtime-registered model '" << m.name << "'\nstruct UserModel {\n"
      << "    static constexpr int N = " << m.n << ", NP = " << m.np << ";\n    static constexpr bool TIME_DEP = true;\n"
      // column bundles through the VJP bodies (hipadj_models.hpp).  Dual-number VJPs: only up to three states — the bundle keeps the whole dual Jacobian live next
      // to all columns, and measured slower than the per-column form from n = 4 on (ring n = 3: 5.0 -> 1.45 ms, n = 4: 1.6 -> 3.3 ms; profiles/r2_user_auto_cols_ab.log)
      << "    static constexpr bool HAS_COLS = " << ((m.cols && (!m.auto_vjp || m.n <= 3)) ? "true" : "false") << ";\n";
    if (m.dae) {
        // M u' = f with a singular M = [Md 0; 0 0] (src/adjoint_common.jl:117-135): the model stays f; the Rosenbrock23 lanes integrate it — and M' lam' = -J' lam — in
        // mass-matrix form (hipadj_adaptive.hpp, model_dae): they ask for the entries of M and for which variables are algebraic (zero rows of M, :116-122)
        char num[40];
        o << "    static constexpr bool DAE = true;\n    HIPADJ_HD static double mass(int i, int j) { const int k = i * N + j; return";
        for (int e = 0; e < m.n * m.n; ++e) if (m.mm[e] != 0.0) { snprintf(num, sizeof(num), "%.17g", m.mm[e]); o << " k == " << e << " ? " << num << " :"; }
        o << " 0.0; }\n    HIPADJ_HD static bool isalg(int i) { return";
        bool any = false;
        for (int i = 0; i < m.n; ++i) { bool z = true; for (int j = 0; j < m.n; ++j) z = z && m.mm[i * m.n + j] == 0.0; if (z) { o << (any ? " || " : " ") << "i == " << i; any = true; } }
        o << (any ? "" : " false") << "; }\n";
    }
    if (m.has_mm) {
        // constant mass matrix M u' = f (ODEFunction(f; mass_matrix = M), src/adjoint_common.jl:110-135): the kernels integrate
        // u' = F(u) = M^{-1} f(u) and the adjoint of THAT system, nu' = -F_u^T nu with the plain jumps nu += g_u.  nu = M^T lam, where lam is
        // the reference's adjoint (M^T lam' = -f_u^T lam, jumps M^{-T} g_u :805-807), mu' = -F_p^T nu = -f_p^T lam is the same
        // parameter integrand; the host maps du0 = lam(t0) = M^{-T} nu(t0) after the sweep (k_mass_du0).  Zero entries of M^{-1} are dropped.
        char num[40];
        o << "    template <class real> HIPADJ_HD static void mm_inv(real (&v)[N]) {\n        real r[N];\n";
        for (int i = 0; i < m.n; ++i) {
            o << "        r[" << i << "] = real(0.0)";
            for (int j = 0; j < m.n; ++j) if (m.minv[i * m.n + j] != 0.0) { snprintf(num, sizeof(num), "%.17g", m.minv[i * m.n + j]); o << " + real(" << num << ") * v[" << j << "]"; }
            o << ";\n";
        }
        o << "        for (int i = 0; i < N; ++i) v[i] = r[i];\n    }\n"
          << "    template <class LT> HIPADJ_HD static void mm_invT(LT (&w)[N], const LT (&lam)[N]) {\n";
        for (int i = 0; i < m.n; ++i) {
            o << "        w[" << i << "] = LT(0.0)";
            for (int j = 0; j < m.n; ++j) if (m.minv[j * m.n + i] != 0.0) { snprintf(num, sizeof(num), "%.17g", m.minv[j * m.n + i]); o << " + (" << num << ") * lam[" << j << "]"; }
            o << ";\n";
        }
        o << "    }\n";
    }
    if (m.auto_vjp) {
        // f is compiled for real = double and real = Dual<K>; the VJPs are lam-weighted column sums of the dual partials
        o << "    template <class real> HIPADJ_HD static void " << (m.has_mm ? "f_raw_t" : "f_t") << "(real (&du)[N], const real (&u)[N], const real (&p)[NP], real t) {\n"
          << "        (void)u; (void)p; (void)t;\n" << m.f << "\n    }\n";
        if (m.has_mm)
            o << "    template <class real> HIPADJ_HD static void f_t(real (&du)[N], const real (&u)[N], const real (&p)[NP], real t) { f_raw_t<real>(du, u, p, t); mm_inv<real>(du); }\n";
        o << ""
          << "    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) { f_t<double>(du, u, p, t); }\n"
          // LT = double: one adjoint column; LT = Cols<G>: a bundle of G columns behind ONE dual-number evaluation of f (hipadj_models.hpp)
          << "    template <class LT> HIPADJ_HD static void vjp_u_t(LT (&out)[N], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        Dual<N> uu[N], pp[NP], dd[N];\n"
          << "        for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n"
          << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n"
          << "        f_t<Dual<N>>(dd, uu, pp, Dual<N>(t));\n"
          << "        for (int j = 0; j < N; ++j) { LT s = LT(0.0); for (int i = 0; i < N; ++i) s += lam[i] * dd[i].d[j]; out[j] = s; }\n    }\n"
          << "    template <class LT> HIPADJ_HD static void vjp_p_t(LT (&out)[NP], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        Dual<NP> uu[N], pp[NP], dd[N];\n"
          << "        for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n"
          << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n"
          << "        f_t<Dual<NP>>(dd, uu, pp, Dual<NP>(t));\n"
          << "        for (int j = 0; j < NP; ++j) { LT s = LT(0.0); for (int i = 0; i < N; ++i) s += lam[i] * dd[i].d[j]; out[j] = s; }\n    }\n"
          << "    HIPADJ_HD static void vjp_u(double (&out)[N], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_u_t<double>(out, lam, u, p, t); }\n"
          << "    HIPADJ_HD static void vjp_p(double (&out)[NP], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_p_t<double>(out, lam, u, p, t); }\n"
          // df/du from ONE dual evaluation (Rosenbrock23's W; hipadj_models.hpp model_jacobian): n unit-vector vjp_u calls would be n of them
          << "    static constexpr bool HAS_JAC = true;\n"
          << "    HIPADJ_HD static void jac(double (&J)[N][N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        Dual<N> uu[N], pp[NP], dd[N];\n"
          << "        for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n"
          << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n"
          << "        f_t<Dual<N>>(dd, uu, pp, Dual<N>(t));\n"
          << "        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) J[i][j] = dd[i].d[j];\n    }\n";
    } else {
        // the VJP bodies are templates on the type LT of `lam` / `out`: double = one adjoint column, Cols<G> = a bundle of G columns that shares
        // everything depending on (u, p, t) only (hipadj_models.hpp).  A body that is not linear in lam does not compile for Cols: HAS_COLS = false then.
        const char* sfx = m.has_mm ? "_raw" : "";
        o << "    HIPADJ_HD static void f" << sfx << "(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        (void)u; (void)p; (void)t;\n" << m.f << "\n    }\n"
          << "    template <class LT> HIPADJ_HD static void vjp_u_raw_t(LT (&out)[N], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        (void)lam; (void)u; (void)p; (void)t;\n" << m.vjp_u << "\n    }\n"
          << "    template <class LT> HIPADJ_HD static void vjp_p_raw_t(LT (&out)[NP], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        (void)lam; (void)u; (void)p; (void)t;\n" << m.vjp_p << "\n    }\n";
        if (m.has_mm)   // F = M^{-1} f:  F_u^T lam = f_u^T (M^{-T} lam),  F_p^T lam = f_p^T (M^{-T} lam)
            o << "    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) { f_raw(du, u, p, t); mm_inv<double>(du); }\n"
              << "    template <class LT> HIPADJ_HD static void vjp_u_t(LT (&out)[N], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { LT w[N]; mm_invT<LT>(w, lam); vjp_u_raw_t<LT>(out, w, u, p, t); }\n"
              << "    template <class LT> HIPADJ_HD static void vjp_p_t(LT (&out)[NP], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { LT w[N]; mm_invT<LT>(w, lam); vjp_p_raw_t<LT>(out, w, u, p, t); }\n";
        else
            o << "    template <class LT> HIPADJ_HD static void vjp_u_t(LT (&out)[N], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_u_raw_t<LT>(out, lam, u, p, t); }\n"
              << "    template <class LT> HIPADJ_HD static void vjp_p_t(LT (&out)[NP], const LT (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_p_raw_t<LT>(out, lam, u, p, t); }\n";
        o << "    HIPADJ_HD static void vjp_u(double (&out)[N], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_u_t<double>(out, lam, u, p, t); }\n"
          << "    HIPADJ_HD static void vjp_p(double (&out)[NP], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_p_t<double>(out, lam, u, p, t); }\n";
    }
    // DiscreteCallback affect (u, p) <- a(u, p, t) (hipadj_model_set_affect): the body edits `un` and / or `pn`, which start as copies of u and p;
    // the reverse callback's products with both Jacobians by forward-mode dual numbers (one pass seeded on u, one on p)
    o << "    template <class real> HIPADJ_HD static void affect_t(real (&un)[N], real (&pn)[NP], const real (&u)[N], const real (&p)[NP], real t) {\n"
      << "        (void)u; (void)p; (void)t;\n        for (int i = 0; i < N; ++i) un[i] = u[i];\n        for (int i = 0; i < NP; ++i) pn[i] = p[i];\n" << m.affect << "\n    }\n"
      << "    HIPADJ_HD static void affect(double (&un)[N], double (&pn)[NP], const double (&u)[N], const double (&p)[NP], double t) { affect_t<double>(un, pn, u, p, t); }\n"
      << "    HIPADJ_HD static void affect_vjp(double (&lo)[N], double (&go)[NP], const double (&lam)[N], const double (&gp)[NP], const double (&u)[N], const double (&p)[NP], double t) {\n"
      << "        {   Dual<N> uu[N], pp[NP], du_[N], dp_[NP];\n"
      << "            for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n"
      << "            for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n"
      << "            affect_t<Dual<N>>(du_, dp_, uu, pp, Dual<N>(t));\n"
      << "            for (int j = 0; j < N; ++j) { double s = 0.0; for (int i = 0; i < N; ++i) s += lam[i] * du_[i].d[j]; for (int k = 0; k < NP; ++k) s += gp[k] * dp_[k].d[j]; lo[j] = s; } }\n"
      << "        {   Dual<NP> uu[N], pp[NP], du_[N], dp_[NP];\n"
      << "            for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n"
      << "            for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n"
      << "            affect_t<Dual<NP>>(du_, dp_, uu, pp, Dual<NP>(t));\n"
      << "            for (int j = 0; j < NP; ++j) { double s = 0.0; for (int i = 0; i < N; ++i) s += lam[i] * du_[i].d[j]; for (int k = 0; k < NP; ++k) s += gp[k] * dp_[k].d[j]; go[j] = s; } }\n    }\n";
    if (!m.cc_cond.empty()) {
        // ContinuousCallback(condition, affect!) (hipadj_model_set_continuous_callback; hipadj_adaptive.hpp "events", src/callback_tracking.jl:232-479): the condition and the affect
        // compile for double and for dual numbers — c_u, c_p, c_t, the directional derivative a_u v + a_t and the products a_u' lam, a_p' lam of the reverse jump
        // (a VectorContinuousCallback: NCOND components `out[k]`, the affect sees the index `idx` of the one that fired; a scalar condition is NCOND = 1 with `c` an alias of out[0])
        o << "    static constexpr bool HAS_COND = true;\n    static constexpr int NCOND = " << (m.cc_ncond > 0 ? m.cc_ncond : 1) << ", CDIR = " << m.cc_dir << ";\n"
          << "    template <class real> HIPADJ_HD static void cond_t(real (&out)[NCOND], const real (&u)[N], const real (&p)[NP], real t) {\n        (void)u; (void)p; (void)t;\n"
          << "        for (int k = 0; k < NCOND; ++k) out[k] = real(0.0);\n        real& c = out[0]; (void)c;\n" << m.cc_cond << "\n    }\n"
          << "    HIPADJ_HD static void cond(double (&out)[NCOND], const double (&u)[N], const double (&p)[NP], double t) { cond_t<double>(out, u, p, t); }\n"
          << "    HIPADJ_HD static void cond_grad(double (&gu)[N], double (&gp)[NP], double& gt, int k, const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        {   Dual<N> uu[N], pp[NP], o[NCOND];\n            for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n            for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n"
          << "            cond_t<Dual<N>>(o, uu, pp, Dual<N>(t));\n            for (int j = 0; j < N; ++j) { double v = 0.0; for (int q = 0; q < NCOND; ++q) v = (q == k) ? o[q].d[j] : v; gu[j] = v; } }\n"
          << "        {   Dual<NP> uu[N], pp[NP], o[NCOND];\n            for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n            for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n"
          << "            cond_t<Dual<NP>>(o, uu, pp, Dual<NP>(t));\n            for (int j = 0; j < NP; ++j) { double v = 0.0; for (int q = 0; q < NCOND; ++q) v = (q == k) ? o[q].d[j] : v; gp[j] = v; } }\n"
          << "        {   Dual<1> uu[N], pp[NP], o[NCOND];\n            for (int j = 0; j < N; ++j) uu[j] = Dual<1>(u[j]);\n            for (int j = 0; j < NP; ++j) pp[j] = Dual<1>(p[j]);\n"
          << "            cond_t<Dual<1>>(o, uu, pp, Dual<1>::seed(t, 0));\n            double v = 0.0; for (int q = 0; q < NCOND; ++q) v = (q == k) ? o[q].d[0] : v; gt = v; }\n    }\n"
          // (`terminate = true;` in the affect body: terminate!(integrator) — the trajectory's solve ends at this event)
          << "    template <class real> HIPADJ_HD static bool cc_affect_t(real (&un)[N], const real (&u)[N], const real (&p)[NP], real t, int idx) {\n"
          << "        (void)u; (void)p; (void)t; (void)idx; bool terminate = false;\n        for (int i = 0; i < N; ++i) un[i] = u[i];\n" << m.cc_affect << "\n        return terminate;\n    }\n"
          << "    HIPADJ_HD static bool cc_affect(double (&un)[N], const double (&u)[N], const double (&p)[NP], double t, int idx) { return cc_affect_t<double>(un, u, p, t, idx); }\n"
          << "    HIPADJ_HD static void cc_affect_jvp(double (&out)[N], const double (&u)[N], const double (&v)[N], const double (&p)[NP], double t, int idx) {\n"
          << "        Dual<1> uu[N], pp[NP], un[N];\n        for (int j = 0; j < N; ++j) { uu[j] = Dual<1>(u[j]); uu[j].d[0] = v[j]; }\n        for (int j = 0; j < NP; ++j) pp[j] = Dual<1>(p[j]);\n"
          << "        cc_affect_t<Dual<1>>(un, uu, pp, Dual<1>::seed(t, 0), idx);\n        for (int j = 0; j < N; ++j) out[j] = un[j].d[0];\n    }\n"
          << "    HIPADJ_HD static void cc_affect_vjp(double (&lo)[N], double (&go)[NP], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double t, int idx) {\n"
          << "        {   Dual<N> uu[N], pp[NP], un[N];\n            for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n            for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n"
          << "            cc_affect_t<Dual<N>>(un, uu, pp, Dual<N>(t), idx);\n            for (int j = 0; j < N; ++j) { double s = 0.0; for (int i = 0; i < N; ++i) s += lam[i] * un[i].d[j]; lo[j] = s; } }\n"
          << "        {   Dual<NP> uu[N], pp[NP], un[N];\n            for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n            for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n"
          << "            cc_affect_t<Dual<NP>>(un, uu, pp, Dual<NP>(t), idx);\n            for (int j = 0; j < NP; ++j) { double s = 0.0; for (int i = 0; i < N; ++i) s += lam[i] * un[i].d[j]; go[j] = s; } }\n    }\n";
    }
    if (m.has_dloss) {
        // discrete loss on the device (hipadj_model_set_discrete_loss[_function]; HIPADJ_LOSS_MODEL): dgdu_discrete(out, u, p, t_i, i) / dgdp_discrete(out, u, p, t_i, i) of
        // ReverseLossCallback (src/adjoint_common.jl:771-779) with d = the data column of this (trajectory, loss time)
        o << "    static constexpr bool HAS_DLOSS = true;\n";
        if (!m.dl_fun.empty()) {
            o << "    template <class real> HIPADJ_HD static real l_disc_t(const real (&u)[N], const real (&p)[NP], real t, int i, const double (&d)[N]) {\n"
              << "        (void)u; (void)p; (void)t; (void)i; (void)d; real l = 0.0;\n" << m.dl_fun << "\n        return l;\n    }\n"
              << "    static constexpr bool HAS_LVALUE = true;\n"
              << "    HIPADJ_HD static double l_disc(const double (&u)[N], const double (&p)[NP], double t, int i, const double (&d)[N]) { return l_disc_t<double>(u, p, t, i, d); }\n"
              << "    HIPADJ_HD static void dgdu_disc(double (&out)[N], const double (&u)[N], const double (&p)[NP], double t, int i, const double (&d)[N]) {\n"
              << "        Dual<N> uu[N], pp[NP];\n        for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n"
              << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n        const Dual<N> lv = l_disc_t<Dual<N>>(uu, pp, Dual<N>(t), i, d);\n"
              << "        for (int j = 0; j < N; ++j) out[j] = lv.d[j];\n    }\n"
              << "    HIPADJ_HD static void dgdp_disc(double (&out)[NP], const double (&u)[N], const double (&p)[NP], double t, int i, const double (&d)[N]) {\n"
              << "        Dual<NP> uu[N], pp[NP];\n        for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n"
              << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n        const Dual<NP> lv = l_disc_t<Dual<NP>>(uu, pp, Dual<NP>(t), i, d);\n"
              << "        for (int j = 0; j < NP; ++j) out[j] = lv.d[j];\n    }\n";
        } else {
            o << "    HIPADJ_HD static void dgdu_disc(double (&out)[N], const double (&u)[N], const double (&p)[NP], double t, int i, const double (&d)[N]) {\n"
              << "        (void)u; (void)p; (void)t; (void)i; (void)d;\n" << m.dl_du << "\n    }\n"
              << "    HIPADJ_HD static void dgdp_disc(double (&out)[NP], const double (&u)[N], const double (&p)[NP], double t, int i, const double (&d)[N]) {\n"
              << "        (void)u; (void)p; (void)t; (void)i; (void)d;\n" << (m.dl_dp.empty() ? std::string("for (int j = 0; j < NP; ++j) out[j] = 0.0;") : m.dl_dp) << "\n    }\n";
        }
    }
    o << "    // continuous cost attached with hipadj_model_set_cost[_function] (dgdu_continuous / dgdp_continuous); zero when absent\n";
    if (m.has_cost && !m.gfun.empty()) {
        // only g was given: its gradients by forward-mode dual numbers (the reference differentiates `g` with ForwardDiff when
        // no dgdu/dgdp are supplied, src/derivative_wrappers.jl:1428-1441)
        o << "    template <class real> HIPADJ_HD static real g_t(const real (&u)[N], const real (&p)[NP], real t) {\n"
          << "        (void)u; (void)p; (void)t; real g = 0.0;\n" << m.gfun << "\n        return g;\n    }\n"
          << "    HIPADJ_HD static void dgdu(double (&out)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        Dual<N> uu[N], pp[NP];\n        for (int j = 0; j < N; ++j) uu[j] = Dual<N>::seed(u[j], j);\n"
          << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<N>(p[j]);\n        const Dual<N> gv = g_t<Dual<N>>(uu, pp, Dual<N>(t));\n"
          << "        for (int j = 0; j < N; ++j) out[j] = gv.d[j];\n    }\n"
          << "    HIPADJ_HD static void dgdp(double (&out)[NP], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        Dual<NP> uu[N], pp[NP];\n        for (int j = 0; j < N; ++j) uu[j] = Dual<NP>(u[j]);\n"
          << "        for (int j = 0; j < NP; ++j) pp[j] = Dual<NP>::seed(p[j], j);\n        const Dual<NP> gv = g_t<Dual<NP>>(uu, pp, Dual<NP>(t));\n"
          << "        for (int j = 0; j < NP; ++j) out[j] = gv.d[j];\n    }\n};\n}  // namespace hipadj\n";
    } else {
        o << "    HIPADJ_HD static void dgdu(double (&out)[N], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        (void)u; (void)p; (void)t;\n" << (m.has_cost ? m.dgdu : std::string("for (int i = 0; i < N; ++i) out[i] = 0.0;")) << "\n    }\n"
          << "    HIPADJ_HD static void dgdp(double (&out)[NP], const double (&u)[N], const double (&p)[NP], double t) {\n"
          << "        (void)u; (void)p; (void)t;\n" << (m.has_cost ? m.dgdp : std::string("for (int i = 0; i < NP; ++i) out[i] = 0.0;")) << "\n    }\n};\n}  // namespace hipadj\n";
    }
    return o.str();
}

// ---- check of freshly compiled code objects for a known miscompile -------------------------------
// The ROCm 7.2 compiler can place register-spill copies (v_accvgpr_write/read, scratch_store/load) at the top of a
// control-flow join block AHEAD of the `s_or_b64 exec, exec, s[..]` that re-enables the lanes which skipped the region; when
// the join is reached through the region's s_cbranch_execz the copies run with an empty exec mask and a loop-carried value
// is lost (found as a GPU fault of k_adjoint_tsit5 with an empty tstop list).  Every runtime model gets its own register
// allocation, so the pattern is looked for in each new code object — IN PROCESS: the .text section of the ELF is walked with the
// disassembler of libamd_comgr (the code-object manager hiprtc itself sits on, so it is already mapped; bound with dlopen like
// hiprtc), no child process, no temporary file.  Without comgr the check is skipped; HIPADJ_RTC_VERIFY=0 turns it off.
// Same walk as tests/tools/isa_lint.py (which uses llvm-objdump and is the cross-check of this scan in the CPU suite).
struct ComgrDis {
    typedef struct { uint64_t handle; } info_t;
    void* lib = nullptr;
    int (*create)(const char*, uint64_t (*)(uint64_t, char*, uint64_t, void*), void (*)(const char*, void*), void (*)(uint64_t, void*), info_t*) = nullptr;
    int (*disasm)(info_t, uint64_t, void*, uint64_t*) = nullptr;
    int (*destroy)(info_t) = nullptr;
    bool ok = false;
    ComgrDis() {
        for (const char* nm : {"libamd_comgr.so.3", "libamd_comgr.so", "/opt/rocm/lib/libamd_comgr.so"}) if ((lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return;
        create = (decltype(create))dlsym(lib, "amd_comgr_create_disassembly_info");
        disasm = (decltype(disasm))dlsym(lib, "amd_comgr_disassemble_instruction");
        destroy = (decltype(destroy))dlsym(lib, "amd_comgr_destroy_disassembly_info");
        ok = create && disasm && destroy;
    }
};
inline ComgrDis& comgr_dis() { static ComgrDis D; return D; }

struct IsaScanCtx { const char* base; uint64_t size; std::string text; long target; };
inline uint64_t isa_scan_read(uint64_t from, char* to, uint64_t size, void* u) {
    const IsaScanCtx* c = (const IsaScanCtx*)u;
    if (from >= c->size) return 0;
    const uint64_t n = size < c->size - from ? size : c->size - from;
    std::memcpy(to, c->base + from, n);
    return n;
}
inline void isa_scan_print(const char* s, void* u) { IsaScanCtx* c = (IsaScanCtx*)u; while (*s == ' ' || *s == '\t') ++s; c->text = s; }
inline void isa_scan_addr(uint64_t a, void* u) { ((IsaScanCtx*)u)->target = (long)a; }

// returns the number of flagged sites (0 = clean), -1 when the check could not run; `what` names the first site (.text offset)
inline int user_isa_check(const std::vector<char>& code, std::string& what) {
    if (const char* e = std::getenv("HIPADJ_RTC_VERIFY")) if (e[0] == '0') return -1;
    ComgrDis& D = comgr_dis();
    if (!D.ok || code.size() < sizeof(Elf64_Ehdr) || std::memcmp(code.data(), ELFMAG, SELFMAG) != 0) return -1;
    const Elf64_Ehdr* eh = (const Elf64_Ehdr*)code.data();
    if (eh->e_shoff == 0 || eh->e_shoff + (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr) > code.size() || eh->e_shstrndx >= eh->e_shnum) return -1;
    const Elf64_Shdr* sh = (const Elf64_Shdr*)(code.data() + eh->e_shoff);
    const char* names = code.data() + sh[eh->e_shstrndx].sh_offset;
    struct Insn { unsigned long addr; std::string op; long target; };   // target: .text offset of a branch target, -1 = none
    auto starts = [](const std::string& s, const char* pre) { return s.compare(0, std::strlen(pre), pre) == 0; };
    static const char* STOP[] = {"s_cbranch", "s_branch", "s_endpgm", "s_or_b64 exec", "s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec",
                                 "s_mov_b64 exec", "s_andn2_b64 exec", "s_xor_b64 exec"};
    static const char* SPILL[] = {"v_accvgpr_write", "v_accvgpr_read", "scratch_store", "scratch_load", "buffer_store", "buffer_load"};
    int found = 0; bool scanned = false;
    for (int si = 0; si < eh->e_shnum; ++si) {
        if (std::strcmp(names + sh[si].sh_name, ".text") != 0 || sh[si].sh_offset + sh[si].sh_size > code.size()) continue;
        IsaScanCtx c{code.data() + sh[si].sh_offset, sh[si].sh_size, std::string(), -1};
        ComgrDis::info_t info;
        if (D.create("amdgcn-amd-amdhsa--gfx950", isa_scan_read, isa_scan_print, isa_scan_addr, &info) != 0) return -1;
        std::vector<Insn> ins;
        for (uint64_t a = 0; a < c.size;) {
            uint64_t sz = 0; c.target = -1; c.text.clear();
            if (D.disasm(info, a, &c, &sz) != 0 || sz == 0) { a += 4; continue; }     // padding / data words between kernels
            const bool br = starts(c.text, "s_cbranch") || starts(c.text, "s_branch");
            ins.push_back({(unsigned long)a, c.text, br ? c.target : -1});
            a += sz;
        }
        D.destroy(info);
        scanned = true;
        std::map<unsigned long, int> tgt;   // address -> 1 = branch target, 2 = target of an s_cbranch_execz
        for (const auto& i : ins) if (i.target >= 0) { int& t = tgt[(unsigned long)i.target]; t = std::max(t, starts(i.op, "s_cbranch_execz") ? 2 : 1); }
        for (size_t k = 0; k < ins.size(); ++k) {
            if (!starts(ins[k].op, "s_or_b64 exec, exec")) continue;
            int spills = 0;
            for (long j = (long)k - 1; j >= 0; --j) {
                bool stop = false;
                for (const char* sname : STOP) if (starts(ins[j].op, sname)) { stop = true; break; }
                if (stop) break;
                for (const char* sp : SPILL) if (starts(ins[j].op, sp)) { ++spills; break; }
                const auto it = tgt.find(ins[j].addr);
                if (it != tgt.end()) {
                    if (spills > 0 && it->second == 2) { if (!found) { char b[64]; snprintf(b, sizeof(b), ".text + 0x%lx", ins[k].addr); what = b; } ++found; }
                    break;
                }
            }
        }
    }
    return scanned ? found : -1;
}

// Compiles (or fetches from the process-wide cache) the code object holding `exprs` for user model `model`.
// Returns HIPADJ_OK and fills code + lowered names; on failure err carries the hiprtc log.
inline int user_compile(int32_t model, const std::vector<std::string>& exprs, std::vector<char>& code,
                        std::map<std::string, std::string>& lowered, std::string& err, bool low_opt = false) {
    UserRegistry& R = user_registry();
    UserModelSrc src;
    std::string key;
    {
        std::lock_guard<std::mutex> lk(R.mu);
        const int idx = model - HIPADJ_MODEL_USER_BASE;
        if (idx < 0 || idx >= (int)R.models.size()) { err = "unknown user model id"; return HIPADJ_ERR_INVALID_ARG; }
        src = R.models[idx];
        key = std::to_string(model) + "#" + std::to_string(src.rev);
        for (const auto& e : exprs) key += "|" + e;
        if (const char* fl = std::getenv("HIPADJ_RTC_FLAGS")) key += std::string("|flags:") + fl;   // a debugging run with other flags must not be served from the cache
        if (const char* e = std::getenv("HIPADJ_TS5_REGS_USER")) key += std::string("|ts5regs:") + e;
        if (const char* e = std::getenv("HIPADJ_USER_COLS")) key += std::string("|cols:") + e;
        if (low_opt) key += "|O1";                          // the second opinion of the heavy-kernel self-test (user_prepare)
        auto it = R.code_cache.find(key);
        if (it != R.code_cache.end()) { code = it->second; lowered = R.lowered_cache[key]; return HIPADJ_OK; }
    }
    RtcApi& A = rtc_api();
    if (!A.lib || !A.err.empty()) { err = A.err.empty() ? "hiprtc unavailable" : A.err; return HIPADJ_ERR_UNSUPPORTED; }
    constexpr int NH = 8;
    const char* hnames[NH] = {"hipadj_models.hpp", "hipadj_lane.hpp", "hipadj_kernels.hpp", "hipadj_adaptive.hpp", "hipadj_dual.hpp", "hipadj_fused.hpp", "hipadj_wide.hpp", "hipadj_quad.hpp"};
    std::string htext[NH];
    const std::string dir = user_csrc_dir();
    for (int i = 0; i < NH; ++i)
        if (!user_read_file(dir + "/" + hnames[i], htext[i])) { err = "cannot read kernel header " + dir + "/" + hnames[i] + " (set HIPADJ_CSRC_DIR)"; return HIPADJ_ERR_UNSUPPORTED; }
    const char* hptr[NH];
    for (int i = 0; i < NH; ++i) hptr[i] = htext[i].c_str();
    if (const char* e = std::getenv("HIPADJ_USER_COLS")) if (e[0] == '0') src.cols = false;   // A/B hook: the per-column form of the segment lanes
    std::string tu = user_model_struct(src);
    // attempt 0: -O3.  If the ISA check (user_isa_check) flags the code object: attempt 1 at -O1 — a different schedule and
    // register allocation (the one known case is clean there); still flagged => refuse rather than run lanes on garbage.
    for (int attempt = 0; attempt < 2; ++attempt) {
        hiprtcProgram prog = nullptr;
        A.enter();
        if (A.CreateProgram(&prog, tu.c_str(), "hipadj_user_model.hip", NH, hptr, hnames) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return HIPADJ_ERR_HIP; }
        for (const auto& e : exprs) A.AddNameExpression(prog, e.c_str());
        std::vector<std::string> optv = {"--offload-arch=gfx950", (attempt == 0 && !low_opt) ? "-O3" : "-O1", "-std=c++17"};
        {   // adaptive Tsit5 of runtime models: the stage rows stay lane-private LDS columns.  The register form (hipadj_adaptive.hpp: KRegs) is 20-35 % faster here too
            // (runtime LV, 10^4 trajectories: Interpolating 0.51 -> 0.40 ms, Backsolve 0.47 -> 0.31; profiles/r3_tsit5_user_regs_ab.log), and every code object passed the
            // ISA check — but ONE of the suite's kernels came back with wrong parameter gradients (3-state ring, Interpolating, model cost, no tstops: GPU visit 22), the
            // kind of wrong code DESIGN.md 6.8 is about.  HIPADJ_TS5_REGS_USER=1 opts in for experiments; it is not the default under any compiler.
            const char* e = std::getenv("HIPADJ_TS5_REGS_USER");
            optv.push_back((e && e[0] == '1') ? "-DHIPADJ_TS5_REGS_USER=1" : "-DHIPADJ_TS5_REGS_USER=0");
        }
        if (const char* e = std::getenv("HIPADJ_RTC_FLAGS")) { std::istringstream is(e); std::string w; while (is >> w) optv.push_back(w); }   // tuning / debugging hook
        std::vector<const char*> opts; for (const auto& o : optv) opts.push_back(o.c_str());
        A.enter();
        const hiprtcResult cr = A.CompileProgram(prog, (int)opts.size(), opts.data());
        if (cr != HIPRTC_SUCCESS) {
            size_t ls = 0; A.GetProgramLogSize(prog, &ls);
            std::string log(ls, '\0'); if (ls) A.GetProgramLog(prog, &log[0]);
            if (log.size() > 4000) log.resize(4000);
            err = "model '" + src.name + "' failed to compile:\n" + log;
            A.DestroyProgram(&prog);
            if (src.cols && !src.wide) {   // a VJP body that is not written linearly in `lam` (a double temporary holding a lam term, ...) does not compile for column bundles:
                src.cols = false;   // once more in the per-column form; a genuine error fails again and is reported from that (plainer) build
                { std::lock_guard<std::mutex> lk(R.mu); const int idx = model - HIPADJ_MODEL_USER_BASE; if (R.models[idx].rev == src.rev) R.models[idx].cols = false; }
                tu = user_model_struct(src);
                attempt = -1;
                continue;
            }
            return HIPADJ_ERR_INVALID_ARG;
        }
        if (std::getenv("HIPADJ_RTC_SHOWLOG")) {   // debugging hook: the compiler's log of a successful build (remarks)
            size_t ls = 0; A.GetProgramLogSize(prog, &ls);
            std::string log(ls, '\0'); if (ls) A.GetProgramLog(prog, &log[0]);
            std::fprintf(stderr, "%s\n", log.c_str());
        }
        lowered.clear();
        for (const auto& e : exprs) {
            const char* low = nullptr;
            if (A.GetLoweredName(prog, e.c_str(), &low) != HIPRTC_SUCCESS || !low) { err = "no lowered name for " + e; A.DestroyProgram(&prog); return HIPADJ_ERR_HIP; }
            lowered[e] = low;
        }
        size_t cs = 0; A.GetCodeSize(prog, &cs);
        code.assign(cs, 0); A.GetCode(prog, code.data());
        A.DestroyProgram(&prog);
        if (const char* ov = std::getenv("HIPADJ_RTC_OVERRIDE")) {   // debugging hook: run a code object built offline (hipcc) from the translation unit HIPADJ_RTC_DUMP wrote
            if (FILE* fp = std::fopen(ov, "rb")) { std::fseek(fp, 0, SEEK_END); const long sz = std::ftell(fp); std::fseek(fp, 0, SEEK_SET); code.assign((size_t)sz, 0); if (std::fread(code.data(), 1, (size_t)sz, fp) != (size_t)sz) code.clear(); std::fclose(fp); }
        }
        if (const char* d = std::getenv("HIPADJ_RTC_DUMP")) {   // debugging hook: keep the code object (llvm-objdump -d, llvm-readelf --notes)
            static int serial = 0;
            const std::string fn = std::string(d) + "/" + src.name + "_" + std::to_string(serial++) + ".hsaco";
            if (FILE* fp = std::fopen(fn.c_str(), "wb")) { std::fwrite(code.data(), 1, code.size(), fp); std::fclose(fp); }
            if (FILE* fp = std::fopen((fn + ".hip").c_str(), "wb")) {   // ... and the translation unit, with the instantiations spelled out (hipcc -S / -emit-llvm studies)
                std::fwrite(tu.data(), 1, tu.size(), fp);
                std::fprintf(fp, "void* hipadj_name_expressions[] = {");
                for (const auto& e : exprs) std::fprintf(fp, "(void*)&%s, ", e.c_str());
                std::fprintf(fp, "nullptr};\n");
                std::fclose(fp);
            }
        }
        std::string where;
        const int flagged = user_isa_check(code, where);
        if (std::getenv("HIPADJ_RTC_SHOWLOG")) std::fprintf(stderr, "hipadj: ISA check of '%s' (attempt %d): %d %s\n", src.name.c_str(), attempt, flagged, where.c_str());
        if (flagged <= 0) break;                           // clean, or the check could not run (no libamd_comgr: documented)
        if (attempt == 1) {
            err = "model '" + src.name + "': the compiler placed register-spill copies ahead of an exec restore (" + where +
                  ") at -O3 and at -O1; lanes could read stale values, so this configuration is refused (fewer states/parameters, "
                  "another stepper, or HIPADJ_RTC_FLAGS may help; HIPADJ_RTC_VERIFY=0 overrides)";
            return HIPADJ_ERR_UNSUPPORTED;
        }
    }
    {
        std::lock_guard<std::mutex> lk(R.mu);
        R.code_cache[key] = code; R.lowered_cache[key] = lowered;
    }
    return HIPADJ_OK;
}

inline int user_set_cost_function(int32_t model, const char* g, std::string& err) {
    if (!g) { err = "hipadj_model_set_cost_function: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_cost_function: unknown model id"; return HIPADJ_ERR_INVALID_ARG; }
    if (R.models[idx].wide) { err = "hipadj_model_set_cost_function: the cost of a wide model (hipadj_wmodel_register) is one SPMD body: hipadj_wmodel_set_cost"; return HIPADJ_ERR_UNSUPPORTED; }
    R.models[idx].gfun = g; R.models[idx].has_cost = true; R.models[idx].rev++;
    return HIPADJ_OK;
}
inline int user_set_cost(int32_t model, const char* dgdu, const char* dgdp, std::string& err) {
    if (!dgdu || !dgdp) { err = "hipadj_model_set_cost: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_cost: unknown model id"; return HIPADJ_ERR_INVALID_ARG; }
    if (R.models[idx].wide) { err = "hipadj_model_set_cost: the cost of a wide model (hipadj_wmodel_register) is one SPMD body: hipadj_wmodel_set_cost"; return HIPADJ_ERR_UNSUPPORTED; }
    R.models[idx].dgdu = dgdu; R.models[idx].dgdp = dgdp; R.models[idx].gfun.clear(); R.models[idx].has_cost = true; R.models[idx].rev++;
    return HIPADJ_OK;
}
// continuous cost of a WIDE model as one SPMD body (hipadj_wmodel_set_cost); NULL / empty removes it
inline int user_set_wide_cost(int32_t model, const char* body, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || !R.models[idx].wide) { err = "hipadj_wmodel_set_cost: not a wide model id (hipadj_wmodel_register)"; return HIPADJ_ERR_INVALID_ARG; }
    R.models[idx].wcost = body ? body : ""; R.models[idx].has_cost = !R.models[idx].wcost.empty(); R.models[idx].rev++;
    return HIPADJ_OK;
}
// DiscreteCallback affect of a runtime model: `body` edits un[0..n) (a copy of u) from u, p, t with `real` locals; NULL / empty removes it
inline int user_set_affect(int32_t model, const char* body, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_affect: unknown model id (affects are attached to runtime-registered models)"; return HIPADJ_ERR_INVALID_ARG; }
    if (R.models[idx].wide && body && *body) { err = "hipadj_model_set_affect: a wide model (hipadj_wmodel_register) takes its affect together with the reverse callback: hipadj_wmodel_set_affect"; return HIPADJ_ERR_UNSUPPORTED; }
    R.models[idx].affect = body ? body : ""; R.models[idx].rev++;
    return HIPADJ_OK;
}
// ContinuousCallback(condition, affect!; save_positions = (false, false)) of a runtime lane model (src/callback_tracking.jl:232-479; test/Callbacks2/continuous_callbacks.jl):
// condition_body assigns `c` from u, p, t; affect_body edits un[0..n) (a copy of u) from u, p, t; both NULL removes the callback
inline int user_set_continuous_callback(int32_t model, const char* cond, const char* affect, int32_t max_events, std::string& err, int32_t ncond = 1) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_continuous_callback: unknown model id (callbacks are attached to runtime-registered models)"; return HIPADJ_ERR_INVALID_ARG; }
    UserModelSrc& m = R.models[idx];
    const bool hc = cond && *cond, ha = affect && *affect;
    if (!hc && !ha) { m.cc_cond.clear(); m.cc_affect.clear(); m.cc_maxev = 0; m.cc_dir = 0; m.rev++; return HIPADJ_OK; }
    if (!hc) { err = "hipadj_model_set_continuous_callback: a condition body is needed (it assigns `c`; the event is its zero crossing)"; return HIPADJ_ERR_INVALID_ARG; }
    if (m.wide) { err = "hipadj_model_set_continuous_callback: offered for the lane-per-trajectory models (hipadj_model_register); a wide model takes preset-time events (hipadj_wmodel_set_affect)"; return HIPADJ_ERR_UNSUPPORTED; }
    if (m.has_mm || m.dae) { err = "hipadj_model_set_continuous_callback: not offered on a model with a mass matrix"; return HIPADJ_ERR_UNSUPPORTED; }
    if (max_events < 0 || max_events > 4096) { err = "hipadj_model_set_continuous_callback: max_events in 0 .. 4096 (0 = 64)"; return HIPADJ_ERR_INVALID_ARG; }
    if (ha && std::string(affect).find("pn[") != std::string::npos) { err = "hipadj_model_set_continuous_callback: the affect of a ContinuousCallback edits the state (un); parameter-changing affects are offered at preset times (hipadj_model_set_affect)"; return HIPADJ_ERR_UNSUPPORTED; }
    if (ncond < 1 || ncond > 8) { err = "hipadj_model_set_vector_continuous_callback: 1 .. 8 condition components"; return HIPADJ_ERR_INVALID_ARG; }
    m.cc_cond = cond; m.cc_affect = ha ? affect : ""; m.cc_maxev = max_events > 0 ? max_events : 64; m.cc_ncond = ncond; m.rev++;
    return HIPADJ_OK;
}
// ContinuousCallback(condition, affect!, affect_neg!) with one of the two affects `nothing`: +1 = only upcrossings fire (affect_neg! = nothing), -1 = only downcrossings, 0 = both
inline int user_set_callback_direction(int32_t model, int32_t direction, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || R.models[idx].cc_cond.empty()) { err = "hipadj_model_set_callback_direction: the model carries no ContinuousCallback (hipadj_model_set_continuous_callback first)"; return HIPADJ_ERR_INVALID_ARG; }
    if (direction < -1 || direction > 1) { err = "hipadj_model_set_callback_direction: direction is -1 (downcrossings), 0 (both) or +1 (upcrossings)"; return HIPADJ_ERR_INVALID_ARG; }
    R.models[idx].cc_dir = direction; R.models[idx].rev++;
    return HIPADJ_OK;
}
inline int user_model_events(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return (idx >= 0 && idx < (int)R.models.size() && !R.models[idx].cc_cond.empty()) ? R.models[idx].cc_maxev : 0;
}
// wide models: the affect and its reverse callback as text (no dual numbers at n up to 4096); both NULL removes them
inline int user_set_wide_affect(int32_t model, const char* body, const char* vjp_body, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || !R.models[idx].wide) { err = "hipadj_wmodel_set_affect: not a wide model id (hipadj_wmodel_register)"; return HIPADJ_ERR_INVALID_ARG; }
    if ((body && *body) != (vjp_body && *vjp_body) && !(body && *body && vjp_body)) { err = "hipadj_wmodel_set_affect: affect_body and affect_vjp_body come together (an empty vjp body = the identity reverse callback, e.g. a constant dose)"; return HIPADJ_ERR_INVALID_ARG; }
    R.models[idx].affect = body ? body : ""; R.models[idx].waffect_vjp = vjp_body ? vjp_body : ""; R.models[idx].rev++;
    return HIPADJ_OK;
}
inline bool user_has_affect(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return idx >= 0 && idx < (int)R.models.size() && !R.models[idx].affect.empty();
}
// Does the model text call library math (sin, exp, pow, ...)?  Such step bodies are large once inlined; the PF-deep unrolled prefetch blocks then
// multiply them past the instruction cache and the sweep gets SLOWER with depth (ring n = 2 / 3: 0.44 / 0.78 ms at depth 1, 0.92 / 1.25 ms at 6),
// while a polynomial right-hand side gains from it (Lotka-Volterra 0.133 -> 0.076 ms at depth 4): profiles/r2_user_prefetch_depth.log.
inline bool user_calls_math(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) return false;
    const UserModelSrc& m = R.models[idx];
    const std::string text = m.f + " " + m.vjp_u + " " + m.vjp_p;
    for (const char* fn : {"sin", "cos", "tan", "exp", "log", "pow", "sqrt", "cbrt", "erf", "gamma", "atan", "hypot"}) {
        for (size_t at = text.find(fn); at != std::string::npos; at = text.find(fn, at + 1)) {
            size_t e = at + std::strlen(fn);
            while (e < text.size() && (std::isalnum((unsigned char)text[e]) || text[e] == '_')) ++e;    // sinh, expm1, log1p, sqrtf, ...
            while (e < text.size() && text[e] == ' ') ++e;
            if (e < text.size() && text[e] == '(') return true;
        }
    }
    return false;
}
// ODEFunction(f; mass_matrix = M) for a runtime model: M row-major n x n, constant and non-singular; NULL removes it.
// Singular M (semi-explicit DAE, src/adjoint_common.jl:117-135, 790-803) needs an implicit stepper: refused.
inline int user_set_mass_matrix(int32_t model, const double* M, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_mass_matrix: unknown model id"; return HIPADJ_ERR_INVALID_ARG; }
    UserModelSrc& m = R.models[idx];
    if (!M) { if (m.has_mm || m.dae) { m.has_mm = false; m.dae = false; m.rev++; } return HIPADJ_OK; }
    if (!m.cc_cond.empty()) { err = "hipadj_model_set_mass_matrix: the model carries a ContinuousCallback, which is not offered together with a mass matrix"; return HIPADJ_ERR_UNSUPPORTED; }
    if (m.wide || m.n > 8) {   // minv[64] / a[8][16] below are sized for the lane family (n <= 8); the wide kernels never consult has_mm
        err = "hipadj_model_set_mass_matrix: mass matrices are implemented for lane-family runtime models (n <= 8) only, not for wide models (hipadj_wmodel_register)";
        return HIPADJ_ERR_UNSUPPORTED;
    }
    const int n = m.n;
    {   // semi-explicit DAE?  zero rows of M that are also zero columns, the block of the other variables non-singular ("The submatrix corresponding to the differential
        // variables of the mass matrix must be nonsingular!", src/adjoint_common.jl:131-133)
        bool alg[8]; int nalg = 0; bool fin = true;
        for (int i = 0; i < n; ++i) { bool z = true; for (int j = 0; j < n; ++j) { fin = fin && std::isfinite(M[i * n + j]); z = z && M[i * n + j] == 0.0; } alg[i] = z; nalg += z; }
        if (fin && nalg > 0 && nalg < n) {
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) for (int j = 0; j < n; ++j) if (alg[j] && M[i * n + j] != 0.0) { ok = false; break; }
            if (ok) {
                int id[8], nd = 0; double b[8][8], sc = 0.0;
                for (int i = 0; i < n; ++i) if (!alg[i]) id[nd++] = i;
                for (int i = 0; i < nd; ++i) for (int j = 0; j < nd; ++j) { b[i][j] = M[id[i] * n + id[j]]; sc = std::fmax(sc, std::fabs(b[i][j])); }
                for (int c = 0; c < nd && ok; ++c) {
                    int piv = c; for (int r = c + 1; r < nd; ++r) if (std::fabs(b[r][c]) > std::fabs(b[piv][c])) piv = r;
                    if (!(std::fabs(b[piv][c]) > 1e-13 * sc)) { ok = false; break; }
                    if (piv != c) for (int j = 0; j < nd; ++j) std::swap(b[c][j], b[piv][j]);
                    for (int r = c + 1; r < nd; ++r) { const double f = b[r][c] / b[c][c]; for (int j = c; j < nd; ++j) b[r][j] -= f * b[c][j]; }
                }
                if (ok) { std::memcpy(m.mm, M, sizeof(double) * (size_t)n * n); m.dae = true; m.has_mm = false; m.rev++; return HIPADJ_OK; }
            }
        }
    }
    double a[8][16]; double scale = 0.0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        if (!std::isfinite(M[i * n + j])) { err = "hipadj_model_set_mass_matrix: non-finite entry"; return HIPADJ_ERR_INVALID_ARG; }
        a[i][j] = M[i * n + j]; a[i][n + j] = i == j ? 1.0 : 0.0; scale = std::fmax(scale, std::fabs(M[i * n + j]));
    }
    for (int c = 0; c < n; ++c) {   // Gauss-Jordan, partial pivoting
        int piv = c; for (int r = c + 1; r < n; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (!(std::fabs(a[piv][c]) > 1e-13 * scale)) {
            err = "hipadj_model_set_mass_matrix: the mass matrix is singular and not of the semi-explicit form [Md 0; 0 0] (zero rows that are also zero columns, Md non-singular: "
                  "src/adjoint_common.jl:116-135) - that form is integrated by HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE, a non-singular M by every stepper";
            return HIPADJ_ERR_UNSUPPORTED;
        }
        if (piv != c) for (int j = 0; j < 2 * n; ++j) std::swap(a[c][j], a[piv][j]);
        const double d = a[c][c]; for (int j = 0; j < 2 * n; ++j) a[c][j] /= d;
        for (int r = 0; r < n; ++r) if (r != c && a[r][c] != 0.0) { const double f = a[r][c]; for (int j = 0; j < 2 * n; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) m.minv[i * n + j] = a[i][n + j];
    m.has_mm = true; m.dae = false; m.rev++;
    return HIPADJ_OK;
}
inline bool user_model_is_dae(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return idx >= 0 && idx < (int)R.models.size() && R.models[idx].dae;
}
// M^{-1} of the model's mass matrix (row-major into out[n*n]); false when it has none
inline bool user_mass_matrix_inverse(int32_t model, double* out) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || !R.models[idx].has_mm) return false;
    std::memcpy(out, R.models[idx].minv, sizeof(double) * 64);
    return true;
}
// discrete loss bodies of a lane model (hipadj_model_set_discrete_loss): dgdu required, dgdp may be NULL; both NULL removes the loss
inline int user_set_discrete_loss(int32_t model, const char* dgdu, const char* dgdp, const char* lfun, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) { err = "hipadj_model_set_discrete_loss: unknown model id (discrete losses are attached to runtime-registered models)"; return HIPADJ_ERR_INVALID_ARG; }
    UserModelSrc& m = R.models[idx];
    if (m.wide) { err = "hipadj_model_set_discrete_loss: the discrete loss of a wide model (hipadj_wmodel_register) is one SPMD body: hipadj_wmodel_set_discrete_loss"; return HIPADJ_ERR_UNSUPPORTED; }
    if (!dgdu && dgdp) { err = "hipadj_model_set_discrete_loss: dgdp_body without dgdu_body"; return HIPADJ_ERR_INVALID_ARG; }
    m.dl_du = dgdu ? dgdu : ""; m.dl_dp = dgdp ? dgdp : ""; m.dl_fun = lfun ? lfun : "";
    m.has_dloss = !m.dl_du.empty() || !m.dl_fun.empty(); m.rev++;
    return HIPADJ_OK;
}
inline int user_set_wide_discrete_loss(int32_t model, const char* body, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || !R.models[idx].wide) { err = "hipadj_wmodel_set_discrete_loss: not a wide model id (hipadj_wmodel_register)"; return HIPADJ_ERR_INVALID_ARG; }
    R.models[idx].wdloss = body ? body : ""; R.models[idx].has_dloss = !R.models[idx].wdloss.empty(); R.models[idx].rev++;
    return HIPADJ_OK;
}
inline bool user_has_dloss(int32_t model, bool* has_value = nullptr) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) return false;
    if (has_value) *has_value = !R.models[idx].dl_fun.empty();
    return R.models[idx].has_dloss;
}
inline bool user_has_cost(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return idx >= 0 && idx < (int)R.models.size() && R.models[idx].has_cost;
}

inline int user_register(const char* name, int32_t n, int32_t np, const char* f, const char* vu, const char* vp, int32_t* id, std::string& err) {
    if (!name || !f || !id) { err = "hipadj_model_register: NULL argument"; return HIPADJ_ERR_INVALID_ARG; }
    if ((vu == nullptr) != (vp == nullptr)) { err = "hipadj_model_register: give both vjp_u_body and vjp_p_body, or neither (automatic forward-mode VJPs)"; return HIPADJ_ERR_INVALID_ARG; }
    if (n < 1 || n > 8 || np < 1 || np > 32) { err = "hipadj_model_register: need 1 <= n <= 8 and 1 <= np <= 32 (state and parameters live in VGPRs); larger models: hipadj_wmodel_register (workgroup per trajectory)"; return HIPADJ_ERR_INVALID_ARG; }
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    UserModelSrc m; m.name = name; m.n = n; m.np = np; m.f = f; m.auto_vjp = vu == nullptr;
    if (!m.auto_vjp) { m.vjp_u = vu; m.vjp_p = vp; }
    R.models.push_back(m);
    *id = HIPADJ_MODEL_USER_BASE + (int32_t)R.models.size() - 1;
    plan_user_sizes_hook() = &user_model_sizes;
    plan_user_segcap_hook() = &user_seg_cap;
    plan_user_dae_hook() = &user_model_is_dae;
    plan_user_events_hook() = &user_model_events;
    return HIPADJ_OK;
}


inline bool user_model_is_wide(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return idx >= 0 && idx < (int)R.models.size() && R.models[idx].wide;
}
// LDS doubles of the adaptive InterpolatingAdjoint sweep of a wide model (k_wide_adjoint_ts5<., 0>): three state tiles, the model's scratch, five
// parameter-sized rows (mu, inc, est, k0, kc), the parameter copy, reduction rows
inline long user_wide_ts5_interp_lds(int32_t model, bool backsolve = false) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) return 0;
    const UserModelSrc& m = R.models[idx];
    return (backsolve ? 4L : 3L) * m.n + m.nw + 5L * m.np + (m.np <= 4096 ? m.np : 1) + (m.threads / 64) * 34 + 96;
}
inline int user_wide_threads(int32_t model) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    return (idx >= 0 && idx < (int)R.models.size()) ? R.models[idx].threads : 0;
}

// hipadj_wmodel_register: a model for the workgroup-per-trajectory family.  threads = 0 picks the workgroup size: max(n / 2, min(np, 4096) / 4)
// rounded up to whole wavefronts, between 64 (one wavefront per trajectory) and 1024.
// declares (or, widths == NULL, withdraws) the dense-chain structure of a wide model; the widths must reproduce the model's n and np
inline int user_declare_dense_chain(int32_t model, const int32_t* widths, int32_t nwidths, int32_t activation, int32_t input_power, std::string& err) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size() || !R.models[idx].wide) { err = "hipadj_wmodel_declare_dense_chain: not a model of hipadj_wmodel_register"; return HIPADJ_ERR_INVALID_ARG; }
    UserModelSrc& m = R.models[idx];
    if (!widths || nwidths == 0) { m.chain.clear(); return HIPADJ_OK; }
    if (nwidths < 2 || nwidths > 16 || activation != HIPADJ_ACT_TANH || input_power < 1 || input_power > 8) {
        err = "hipadj_wmodel_declare_dense_chain: need 2 <= nwidths <= 16, activation = HIPADJ_ACT_TANH and 1 <= input_power <= 8"; return HIPADJ_ERR_INVALID_ARG; }
    long np = 0;
    for (int l = 1; l < nwidths; ++l) { if (widths[l] < 1 || widths[l - 1] < 1) { err = "hipadj_wmodel_declare_dense_chain: widths must be positive"; return HIPADJ_ERR_INVALID_ARG; } np += (long)widths[l] * widths[l - 1] + widths[l]; }
    if (widths[0] != m.n || widths[nwidths - 1] != m.n || np != m.np) {
        err = "hipadj_wmodel_declare_dense_chain: the widths do not reproduce the registered model (n = widths[0] = widths[last], np = sum of out * in + out per layer)"; return HIPADJ_ERR_INVALID_ARG; }
    m.chain.assign(widths, widths + nwidths); m.chain_power = input_power; m.chain_act = activation;
    return HIPADJ_OK;
}
// the declared chain of a model (false: none, or not a wide model, or the model carries a cost / loss body / reverse callback the matrix-core family has no counterpart for)
inline bool user_dense_chain(int32_t model, std::vector<int>& widths, int& power) {
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    const int idx = model - HIPADJ_MODEL_USER_BASE;
    if (idx < 0 || idx >= (int)R.models.size()) return false;
    const UserModelSrc& m = R.models[idx];
    if (!m.wide || m.chain.empty()) return false;
    widths = m.chain; power = m.chain_power;
    return true;
}

inline int user_register_wide(const char* name, int32_t n, int32_t np, int32_t threads, int32_t lds_doubles, int32_t nacc, int32_t acc_first,
                              const char* f, const char* vjp, int32_t* id, std::string& err) {
    if (!name || !f || !vjp || !id) { err = "hipadj_wmodel_register: NULL argument (f_body and vjp_body are both required)"; return HIPADJ_ERR_INVALID_ARG; }
    if (n < 1 || n > 4096 || np < 1 || np > (1 << 20)) { err = "hipadj_wmodel_register: need 1 <= n <= 4096 and 1 <= np <= 2^20"; return HIPADJ_ERR_INVALID_ARG; }
    if (lds_doubles < 0 || nacc < 0 || nacc > 16 || acc_first < 0 || (nacc > 0 && acc_first + nacc > np)) {
        err = "hipadj_wmodel_register: need lds_doubles >= 0, 0 <= nacc <= 16 and the reduced parameters [acc_first, acc_first + nacc) inside [0, np)"; return HIPADJ_ERR_INVALID_ARG; }
    if (threads == 0) {
        // two state components per thread (measured on the HBM-bound 30 x 50 matrix state, 512 / 2048 trajectories: 384 threads = 4 components each 0.38 / 0.35 of
        // the HBM peak, 768 threads 0.45 / 0.49; profiles/r3_wide_threads_sweep.log), or four parameter entries per thread where the parameters dominate
        const int by_n = ((n + 1) / 2 + 63) / 64 * 64, by_p = (((np < 4096 ? np : 4096) + 3) / 4 + 63) / 64 * 64;
        threads = by_n > by_p ? by_n : by_p;
        if (threads < 64) threads = 64;
        if (threads > 1024) threads = 1024;
    }
    if (threads % 64 != 0 || threads < 64 || threads > 1024) { err = "hipadj_wmodel_register: threads must be a multiple of 64 between 64 and 1024 (0 = automatic)"; return HIPADJ_ERR_INVALID_ARG; }
    if ((n + threads - 1) / threads > 16) { err = "hipadj_wmodel_register: more than 16 state components per thread (raise threads)"; return HIPADJ_ERR_INVALID_ARG; }
    // LDS of the heaviest kernel (Backsolve: four state-sized tiles) + the model's scratch + the gradient accumulator (np <= 8192) + reduction rows: 160 KB per workgroup
    const long lds = 4L * n + lds_doubles + (np <= 8192 ? np : 1) + (np <= 4096 ? np : 1) + (threads / 64) * 34 + 64;   // tiles + scratch + gradient accumulator + parameter copy
    if (lds * 8 > 160L * 1024) { err = "hipadj_wmodel_register: 4 n + lds_doubles + the gradient accumulator (np <= 8192) + the parameter copy (np <= 4096) exceed the 160 KB of LDS of a workgroup"; return HIPADJ_ERR_INVALID_ARG; }
    UserRegistry& R = user_registry();
    std::lock_guard<std::mutex> lk(R.mu);
    UserModelSrc m; m.name = name; m.n = n; m.np = np; m.f = f; m.wvjp = vjp; m.wide = true; m.cols = false;
    m.threads = threads; m.nw = lds_doubles; m.nacc = nacc; m.acc0 = acc_first;
    R.models.push_back(m);
    *id = HIPADJ_MODEL_USER_BASE + (int32_t)R.models.size() - 1;
    plan_user_sizes_hook() = &user_model_sizes;
    plan_user_segcap_hook() = &user_seg_cap;
    plan_user_wide_hook() = &user_model_is_wide;
    return HIPADJ_OK;
}

}  // namespace hipadj
