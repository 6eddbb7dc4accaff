// tests/emu/lane_emu.cpp — TEST-ONLY host emulation of the gfx950 lane bodies.
//
// This is NOT a product path and NOT a CPU fallback: libhipadj never contains it, the package never loads
// it.  It compiles scimlsensitivity.jl_amd/csrc/hipadj_lane.hpp with g++ (HIPADJ_HD -> inline) and loops the
// per-lane bodies over trajectories exactly as the kernels in hipadj_kernels.hpp do, so that the device
// arithmetic (RK4 staging, Hermite midpoints, segment composition, Gauss nodes, GK15 bisection) can be
// compared with the oracle inside the GPU-less build container (`pytest -m "not gpu"`).  The real parity
// tests (`-m gpu`) go through the C ABI on an MI355X.
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include "../../scimlsensitivity.jl_amd/csrc/hipadj_lane.hpp"
#include "../../scimlsensitivity.jl_amd/csrc/hipadj_plan.hpp"
#include "../../scimlsensitivity.jl_amd/csrc/hipadj_adaptive.hpp"

using namespace hipadj;

// TEST-ONLY model: the n-state ring of tests/user_models.py (the device receives it as text through hipadj_model_register;
// the oracle has it as ORC_MODEL_RING).  Its Backsolve state [lam; mu; y] is 3n + 1 wide: n = 4 puts the host emulation on the
// wide-state branch of tsit5_integrate (NZ > TS5_WIDE), which no compiled-in model reaches.  Model id = HIPADJ_MODEL_USER_BASE + n.
template <int NN> struct EmuRing {
    static constexpr int N = NN, NP = NN + 1;
    static constexpr bool TIME_DEP = false;
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        for (int i = 0; i < N; ++i) du[i] = p[i] * (u[(i + 1) % N] - u[i]) + p[N] * std::sin(u[(i + N - 1) % N]);
    }
    // the VJP bodies as templates on the type of lam / out, like the generated runtime models: LT = Cols<G> runs a bundle of G segment columns
    // through them at once (hipadj_models.hpp), which the segmented emulator cases (time_segments > 1) exercise on the host
    static constexpr bool HAS_COLS = true;
    template <class LT> static void vjp_u_t(LT (&dl)[N], const LT (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        for (int j = 0; j < N; ++j) dl[j] = -p[j] * l[j] + p[(j + N - 1) % N] * l[(j + N - 1) % N] + p[N] * std::cos(u[j]) * l[(j + 1) % N];
    }
    template <class LT> static void vjp_p_t(LT (&dg)[NP], const LT (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        LT s = LT(0.0);
        for (int k = 0; k < N; ++k) { dg[k] = l[k] * (u[(k + 1) % N] - u[k]); s += l[k] * std::sin(u[(k + N - 1) % N]); }
        dg[N] = s;
    }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_u_t<double>(dl, l, u, p, t); }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_p_t<double>(dg, l, u, p, t); }
};
// TEST-ONLY: the ring behind a constant mass matrix, wrapped the way the generator of hipadj_user.hpp wraps a runtime model that has one
// (F = M^{-1} f, F_u^T lam = f_u^T (M^{-T} lam), likewise F_p).  M^{-1}[i][j] = 0.8 [i == j] + 0.15 sin(1 + 3 i + 7 j) (tests/emu.py holds the
// same formula); n = 5 is the first ring whose Gauss pass is NOT time-segmented ((1 + n)(n + np) > 64).  Model id = HIPADJ_MODEL_USER_BASE + 100 + n.
template <int NN> struct EmuRingMM {
    static constexpr int N = NN, NP = NN + 1;
    static constexpr bool TIME_DEP = true;
    static double minv(int i, int j) { return (i == j ? 0.8 : 0.0) + 0.15 * std::sin(1.0 + 3.0 * i + 7.0 * j); }
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) {
        double r[N]; EmuRing<NN>::f(r, u, p, t);
        for (int i = 0; i < N; ++i) { double s = 0.0; for (int j = 0; j < N; ++j) s += minv(i, j) * r[j]; du[i] = s; }
    }
    static constexpr bool HAS_COLS = true;
    template <class LT> static void vjp_u_t(LT (&dl)[N], const LT (&l)[N], const double (&u)[N], const double (&p)[NP], double t) {
        LT w[N];
        for (int i = 0; i < N; ++i) { LT s = LT(0.0); for (int j = 0; j < N; ++j) s += minv(j, i) * l[j]; w[i] = s; }
        EmuRing<NN>::template vjp_u_t<LT>(dl, w, u, p, t);
    }
    template <class LT> static void vjp_p_t(LT (&dg)[NP], const LT (&l)[N], const double (&u)[N], const double (&p)[NP], double t) {
        LT w[N];
        for (int i = 0; i < N; ++i) { LT s = LT(0.0); for (int j = 0; j < N; ++j) s += minv(j, i) * l[j]; w[i] = s; }
        EmuRing<NN>::template vjp_p_t<LT>(dg, w, u, p, t);
    }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_u_t<double>(dl, l, u, p, t); }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) { vjp_p_t<double>(dg, l, u, p, t); }
};
// TEST-ONLY: Robertson kinetics in ODE form (tests/user_models.py ROBER, ORC_MODEL_ROBER): the stiff problem of the Rosenbrock23 cases — at p = (0.04, 3e7, 1e4) the W = I - d h J
// solves of the lane bodies run at |h J| ~ 1e4 .. 1e9 with the pivoting LU doing real work.  Model id = HIPADJ_MODEL_USER_BASE + 203.
struct EmuRober {
    static constexpr int N = 3, NP = 3;
    static constexpr bool TIME_DEP = false;
    static constexpr bool HAS_COLS = false;
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = -p[0] * u[0] + p[2] * u[1] * u[2];
        du[1] = p[0] * u[0] - p[1] * u[1] * u[1] - p[2] * u[1] * u[2];
        du[2] = p[1] * u[1] * u[1];
    }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        dl[0] = -p[0] * l[0] + p[0] * l[1];
        dl[1] = p[2] * u[2] * l[0] + (-2.0 * p[1] * u[1] - p[2] * u[2]) * l[1] + 2.0 * p[1] * u[1] * l[2];
        dl[2] = p[2] * u[1] * l[0] - p[2] * u[1] * l[1];
    }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        dg[0] = -u[0] * l[0] + u[0] * l[1];
        dg[1] = -u[1] * u[1] * l[1] + u[1] * u[1] * l[2];
        dg[2] = u[1] * u[2] * l[0] - u[1] * u[2] * l[1];
    }
};
// TEST-ONLY: `rober` exactly as test/Core3/adjoint.jl:1434-1441 writes it (third row: the conservation constraint) with the mass matrix diag(1, 1, 0) of :1450-1454 — a
// semi-explicit DAE, as hipadj_user.hpp generates a runtime model whose mass matrix is singular (DAE / mass / isalg).  Oracle: ORC_MODEL_ROBERDAE under orc_set_mass_matrix.
// Model id = HIPADJ_MODEL_USER_BASE + 204 (KAPPA = 0: the reference's constraint) and + 205 (KAPPA = 5: y1 + y2 + y3 = 1 + 5 (p1 - 0.04), a constraint that depends on a
// parameter — without one the loss jumps' parameter term f_p' [0; dlam_a] is identically zero and nothing would notice it missing).
template <int KAPPA, int MIX = 0> struct EmuRoberDAE {      // MIX = 1 (id + 206): the differential rows mixed by Md = [2 0.3; 0.1 0.5], mass matrix [Md 0; 0 0] — the same trajectory, a non-trivial M'[diff, diff]
    static constexpr int N = 3, NP = 3;
    static constexpr bool TIME_DEP = false;
    static constexpr bool HAS_COLS = false;
    static constexpr bool DAE = true;
    static double mass(int i, int j) { return (i == 2 || j == 2) ? 0.0 : (MIX ? (i == 0 ? (j == 0 ? 2.0 : 0.3) : (j == 0 ? 0.1 : 0.5)) : (i == j ? 1.0 : 0.0)); }
    static bool isalg(int i) { return i == 2; }
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        const double a = -p[0] * u[0] + p[2] * u[1] * u[2], b = p[0] * u[0] - p[1] * u[1] * u[1] - p[2] * u[1] * u[2];
        du[0] = MIX ? 2.0 * a + 0.3 * b : a;
        du[1] = MIX ? 0.1 * a + 0.5 * b : b;
        du[2] = u[0] + u[1] + u[2] - 1.0 - KAPPA * (p[0] - 0.04);
    }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        const double l0 = MIX ? 2.0 * l[0] + 0.1 * l[1] : l[0], l1 = MIX ? 0.3 * l[0] + 0.5 * l[1] : l[1];
        dl[0] = -p[0] * l0 + p[0] * l1 + l[2];
        dl[1] = p[2] * u[2] * l0 + (-2.0 * p[1] * u[1] - p[2] * u[2]) * l1 + l[2];
        dl[2] = p[2] * u[1] * l0 - p[2] * u[1] * l1 + l[2];
    }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        const double l0 = MIX ? 2.0 * l[0] + 0.1 * l[1] : l[0], l1 = MIX ? 0.3 * l[0] + 0.5 * l[1] : l[1];
        dg[0] = -u[0] * l0 + u[0] * l1 - KAPPA * l[2];
        dg[1] = -u[1] * u[1] * l1;
        dg[2] = u[1] * u[2] * l0 - u[1] * u[2] * l1;
    }
};
// TEST-ONLY: the ContinuousCallback problems of the oracle (adjoint_oracle.h event_kind; test/Callbacks2/continuous_callbacks.jl): the falling mass with a condition and an affect,
// written out by hand the way the generator of hipadj_user.hpp derives them from the registered bodies.  Model id = HIPADJ_MODEL_USER_BASE + 300 + KIND (KIND 1: the bouncing
// ball; 4: the moving floor, condition and affect with explicit t) and + 303 (EmuRelax, kind 3: the condition depends on a parameter).
template <int KIND> struct EmuBall {
    static constexpr int N = 2, NP = 2, NCOND = 1;
    static constexpr bool TIME_DEP = false, HAS_COLS = false, HAS_COND = true;
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) { du[0] = u[1]; du[1] = -p[0]; }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dl[0] = 0.0; dl[1] = l[0]; }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dg[0] = -l[1]; dg[1] = 0.0; }
    static void cond(double (&out)[NCOND], const double (&u)[N], const double (&)[NP], double t) { out[0] = KIND == 4 ? u[0] - 0.3 * t : u[0]; }
    static void cond_grad(double (&gu)[N], double (&gp)[NP], double& gt, int, const double (&)[N], const double (&)[NP], double) { gu[0] = 1.0; gu[1] = 0.0; gp[0] = 0.0; gp[1] = 0.0; gt = KIND == 4 ? -0.3 : 0.0; }
    static bool cc_affect(double (&un)[N], const double (&u)[N], const double (&p)[NP], double t, int) { un[0] = u[0]; un[1] = KIND == 4 ? -p[1] * (u[1] - 0.3) + 0.3 + 0.1 * t : -p[1] * u[1]; return KIND == 7; }      // KIND 7: kind 1 with terminate!
    static void cc_affect_jvp(double (&out)[N], const double (&)[N], const double (&v)[N], const double (&p)[NP], double, int) { out[0] = v[0]; out[1] = -p[1] * v[1] + (KIND == 4 ? 0.1 : 0.0); }
    static void cc_affect_vjp(double (&lo)[N], double (&go)[NP], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double, int) {
        lo[0] = lam[0]; lo[1] = -p[1] * lam[1]; go[0] = 0.0; go[1] = -(u[1] - (KIND == 4 ? 0.3 : 0.0)) * lam[1];
    }
};
struct EmuRelax {
    static constexpr int N = 1, NP = 2, NCOND = 1;
    static constexpr bool TIME_DEP = false, HAS_COLS = false, HAS_COND = true;
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) { du[0] = p[0] - u[0]; }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dl[0] = -l[0]; }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dg[0] = l[0]; dg[1] = 0.0; }
    static void cond(double (&out)[NCOND], const double (&u)[N], const double (&p)[NP], double) { out[0] = u[0] - 0.75 * p[0]; }
    static void cond_grad(double (&gu)[N], double (&gp)[NP], double& gt, int, const double (&)[N], const double (&)[NP], double) { gu[0] = 1.0; gp[0] = -0.75; gp[1] = 0.0; gt = 0.0; }
    static bool cc_affect(double (&un)[N], const double (&u)[N], const double (&p)[NP], double, int) { un[0] = u[0] + p[1]; return false; }
    static void cc_affect_jvp(double (&out)[N], const double (&)[N], const double (&v)[N], const double (&)[NP], double, int) { out[0] = v[0]; }
    static void cc_affect_vjp(double (&lo)[N], double (&go)[NP], const double (&lam)[N], const double (&)[N], const double (&)[NP], double, int) { lo[0] = lam[0]; go[0] = 0.0; go[1] = lam[0]; }
};
// TEST-ONLY: the VectorContinuousCallback problem of test/Callbacks2/vector_continuous_callbacks.jl:10-16, 80-96 (oracle: ORC_MODEL_BALL2D, event_kind 5): a ball that falls in
// x and drifts in y between two walls; condition out = [u1, (u3 - 10) u3]; component 0 reflects u2, component 1 reflects u4, both with restitution p2.  Model id = USER_BASE + 305.
struct EmuBall2D {
    static constexpr int N = 4, NP = 2, NCOND = 2;
    static constexpr bool TIME_DEP = false, HAS_COLS = false, HAS_COND = true;
    static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) { du[0] = u[1]; du[1] = -p[0]; du[2] = u[3]; du[3] = 0.0; }
    static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dl[0] = 0.0; dl[1] = l[0]; dl[2] = 0.0; dl[3] = l[2]; }
    static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&)[N], const double (&)[NP], double) { dg[0] = -l[1]; dg[1] = 0.0; }
    static void cond(double (&out)[NCOND], const double (&u)[N], const double (&)[NP], double) { out[0] = u[0]; out[1] = (u[2] - 10.0) * u[2]; }
    static void cond_grad(double (&gu)[N], double (&gp)[NP], double& gt, int k, const double (&u)[N], const double (&)[NP], double) {
        for (int j = 0; j < N; ++j) gu[j] = 0.0;
        gp[0] = 0.0; gp[1] = 0.0; gt = 0.0;
        if (k == 0) gu[0] = 1.0; else gu[2] = 2.0 * u[2] - 10.0;
    }
    static bool cc_affect(double (&un)[N], const double (&u)[N], const double (&p)[NP], double, int k) { for (int j = 0; j < N; ++j) un[j] = u[j]; if (k == 0) un[1] = -p[1] * u[1]; else un[3] = -p[1] * u[3]; return false; }
    static void cc_affect_jvp(double (&out)[N], const double (&)[N], const double (&v)[N], const double (&p)[NP], double, int k) { for (int j = 0; j < N; ++j) out[j] = v[j]; if (k == 0) out[1] = -p[1] * v[1]; else out[3] = -p[1] * v[3]; }
    static void cc_affect_vjp(double (&lo)[N], double (&go)[NP], const double (&lam)[N], const double (&u)[N], const double (&p)[NP], double, int k) {
        for (int j = 0; j < N; ++j) lo[j] = lam[j];
        go[0] = 0.0;
        if (k == 0) { lo[1] = -p[1] * lam[1]; go[1] = -u[1] * lam[1]; } else { lo[3] = -p[1] * lam[3]; go[1] = -u[3] * lam[3]; }
    }
};
static int emu_user_sizes(int32_t model, int32_t* n, int32_t* np) {
    const int nn = model - HIPADJ_MODEL_USER_BASE;
    if (nn >= 203 && nn <= 206) { *n = 3; *np = 3; return HIPADJ_OK; }
    if (nn == 301 || nn == 304 || nn == 307) { *n = 2; *np = 2; return HIPADJ_OK; }
    if (nn == 303) { *n = 1; *np = 2; return HIPADJ_OK; }
    if (nn == 305) { *n = 4; *np = 2; return HIPADJ_OK; }
    if (nn != 4 && nn != 105) return HIPADJ_ERR_INVALID_ARG;
    *n = nn % 100; *np = nn % 100 + 1;
    return HIPADJ_OK;
}
static bool emu_user_dae(int32_t model) { return model >= HIPADJ_MODEL_USER_BASE + 204 && model <= HIPADJ_MODEL_USER_BASE + 206; }
static constexpr int EMU_MAXEV = 16;
// cotangents of a loss on the saved event states ([EMU_MAXEV][n][Npad] in the device layout, or nullptr): ONE definition, in the unit with the entry points
#if !defined(EMU_UNIT) || EMU_UNIT <= 0
const double *g_emu_ev_dl = nullptr, *g_emu_ev_dr = nullptr; double* g_emu_ev_out = nullptr;
#else
extern const double *g_emu_ev_dl, *g_emu_ev_dr; extern double* g_emu_ev_out;
#endif
static int emu_user_events(int32_t model) { const int nn = model - HIPADJ_MODEL_USER_BASE; return (nn == 301 || nn == 303 || nn == 304 || nn == 305 || nn == 307) ? EMU_MAXEV : 0; }
static const bool g_hook_set = (plan_user_sizes_hook() = &emu_user_sizes, plan_user_dae_hook() = &emu_user_dae, plan_user_events_hook() = &emu_user_events, true);

template <class Mo>
static void compose(const Plan& P, const std::vector<double>& segbuf, double* du0, std::vector<double>& dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long Np = P.Npad;
        for (long i = 0; i < P.N; ++i) {       // k_compose
            double lam[N], mu[NP];
            const double* src = segbuf.data() + (size_t)(P.nseg - 1) * NC * R * Np + i;
            for (int j = 0; j < N; ++j) lam[j] = src[(size_t)j * Np];
            for (int j = 0; j < NP; ++j) mu[j] = src[(size_t)(N + j) * Np];
            for (int s = P.nseg - 2; s >= 0; --s) {
                src = segbuf.data() + (size_t)s * NC * R * Np + i;
                double nl[N], nm[NP];
                for (int j = 0; j < N; ++j) nl[j] = src[(size_t)j * Np];
                for (int j = 0; j < NP; ++j) nm[j] = mu[j] + src[(size_t)(N + j) * Np];
                for (int c = 0; c < N; ++c) { for (int j = 0; j < N; ++j) nl[j] += src[((size_t)(c + 1) * R + j) * Np] * lam[c];
                                              for (int j = 0; j < NP; ++j) nm[j] += src[((size_t)(c + 1) * R + N + j) * Np] * lam[c]; }
                for (int j = 0; j < N; ++j) lam[j] = nl[j];
                for (int j = 0; j < NP; ++j) mu[j] = nm[j];
            }
            for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
            for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[j];
        }
}

template <class Mo, int LOSS>   // LOSS = MODE = discrete-loss kind | (continuous cost << 1)
static int run(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu,
               double* du0, double* dp, double* out) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
#ifndef EMU_PF
#define EMU_PF 8
#endif
    constexpr int PF = EMU_PF;   // -DEMU_PF=2|4 exercises the shallow prefetch rings the runtime-compiled models use
    Geom g; g.N = P.N; g.Npad = P.Npad; g.S = P.S; g.M = P.M; g.t0 = cfg->t0; g.dt = cfg->dt; g.loss_shift = cfg->loss_shift;
    g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start; g.p_shared = cfg->p_shared; g.kmask = -1; g.h_last = P.h_last;
    { const double lw = cfg->loss_scale != 0.0 ? cfg->loss_scale : 1.0; g.la = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? lw : 0.0; g.lb = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? -lw : 1.0; g.lflags = (cfg->reference_literal && (cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD)) ? 3 : 0; }   // hipadj_create's rule (csrc/hipadj_api.hip)
    const long Np = P.Npad;
    std::vector<dbl2> knots(((cfg->alg != HIPADJ_ALG_BACKSOLVE && !P.ip_ckpt) || P.offgrid) ? (size_t)(P.S + 1) * N * Np : 0);
    std::vector<double> tile((size_t)((P.ck_longest > HIPADJ_CKPT_KMAX ? P.ck_longest : HIPADJ_CKPT_KMAX) + 1) * N);   // LDS tile, or the HBM slice of k_*_ckpt<..., GT = true>
    std::vector<double> ckpt((P.bs_ckpt || P.ip_ckpt || (P.offgrid && P.nck > 0)) ? (size_t)P.nck * N * Np : 0), outT((size_t)P.M * N * Np), yT((size_t)N * Np);
    std::vector<double> cotT(cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT ? (size_t)P.M * N * Np : 0);
    std::vector<double> dp_traj((size_t)NP * Np, 0.0);
    for (long i = 0; i < P.N; ++i)
        forward_lane<Mo>(g, i, u0, p, knots.empty() ? nullptr : knots.data(), ckpt.empty() ? nullptr : ckpt.data(),
                         P.ckpt_of_knot.data(), outT.data(), P.save_of_knot.data(), yT.data());
    if (P.offgrid) for (long i = 0; i < P.N; ++i) {   // k_out_offgrid: primal output and (Backsolve) checkpoint states by interpolation
        out_offgrid_lane<Mo>(g, i, knots.data(), P.save_times.data(), P.M, outT.data());
        if (P.nck > 0) out_offgrid_lane<Mo>(g, i, knots.data(), P.ck_times.data(), P.nck, ckpt.data());
    }
    if (out) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) out[i * P.M * N + c] = outT[(size_t)c * Np + i];
    if (!cotT.empty()) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) cotT[(size_t)c * Np + i] = dLdu[i * P.M * N + c];
    const double* cot = cotT.empty() ? nullptr : cotT.data();
    const CkptSrc CK{ckpt.empty() ? nullptr : ckpt.data(), P.ckpt_of_knot.data(), P.prev_ck.data(), tile.data(), 1, 0};
    constexpr int KM = HIPADJ_CKPT_KMAX;
    if (P.og_ck) {   // k_offgrid_ckpt<Mo, LOSS, ALG>: checkpointing = true over the reverse step list
        const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
        const OgIntervals I{P.og_S.data(), P.og_qlo.data(), P.og_qhi.data(), P.og_hlast.data(), P.ck_times.data(), (int)P.og_S.size()};
        std::vector<dbl2> og_tile((size_t)P.og_tile_knots * N * Np);
        for (long i = 0; i < P.N; ++i) {
            double lam[1][N], mu[1][NP];
            if (cfg->alg == HIPADJ_ALG_INTERPOLATING) offgrid_ckpt_lane<Mo, LOSS, 0>(g, i, p, ckpt.data(), og_tile.data(), cot, RS, I, lam, mu);
            else if (cfg->alg == HIPADJ_ALG_GAUSS) offgrid_ckpt_lane<Mo, LOSS, 2>(g, i, p, ckpt.data(), og_tile.data(), cot, RS, I, lam, mu);
            else offgrid_ckpt_lane<Mo, LOSS, 4>(g, i, p, ckpt.data(), og_tile.data(), cot, RS, I, lam, mu);
            for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
            for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[0][j];
        }
    } else
    switch (cfg->alg) {
    case HIPADJ_ALG_INTERPOLATING: {
        if (P.offgrid && P.nseg > 1) {   // k_offgrid_seg<..., false> + k_compose_finish
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
            for (long i = 0; i < P.N; ++i) for (int seg = 0; seg < P.nseg; ++seg) {
                const int q_lo = RS.n - P.seg_bounds[seg + 1], q_hi = RS.n - P.seg_bounds[seg];
                double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
                if (seg == P.nseg - 1) { double lam[1][N], mu[1][NP];
                    interp_offgrid_lane<Mo, LOSS, 1>(g, i, p, knots.data(), cot, RS, lam, mu, q_lo, q_hi);
                    for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                    for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
                } else { double lam[NC][N], mu[NC][NP];
                    interp_offgrid_lane<Mo, LOSS, NC>(g, i, p, knots.data(), cot, RS, lam, mu, q_lo, q_hi);
                    for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                                   for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; } }
            }
            compose<Mo>(P, segbuf, du0, dp_traj);
            break;
        }
        if (P.offgrid) {   // k_interp_offgrid: loss times off the step grid, the planner's reverse step list
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), P.nck > 0 ? P.rs_ck.data() : nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            for (long i = 0; i < P.N; ++i) {
                double lam[1][N], mu[1][NP];
                interp_offgrid_lane<Mo, LOSS>(g, i, p, knots.data(), cot, RS, lam, mu);
                for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
                for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[0][j];
            }
            break;
        }
        std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
        for (int seg = 0; seg < P.nseg; ++seg) for (long i = 0; i < P.N; ++i) {
            double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
            if (seg == P.nseg - 1) {
                double lam[1][N], mu[1][NP];
                if (P.ip_ckpt) interp_lane<Mo, 1, PF, LOSS, KM>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else interp_lane<Mo, 1, PF, LOSS>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
            } else {
                double lam[NC][N], mu[NC][NP];
                if (P.ip_ckpt) interp_lane<Mo, NC, PF, LOSS, KM>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else if (g.p_shared) interp_lane<Mo, NC, PF, LOSS, 0, true>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);   // as the library: stage-operator step for models that carry one
                else interp_lane<Mo, NC, PF, LOSS>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                               for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; }
            }
        }
        compose<Mo>(P, segbuf, du0, dp_traj);
        break; }
    case HIPADJ_ALG_BACKSOLVE: {
        if (P.offgrid) {   // k_backsolve_offgrid
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), P.nck > 0 ? P.rs_ck.data() : nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            for (long i = 0; i < P.N; ++i) {
                double lam[1][N], mu[1][NP];
                backsolve_offgrid_lane<Mo, (LOSS >> 1)>(g, i, p, yT.data(), ckpt.empty() ? nullptr : ckpt.data(), cot, RS, lam, mu);
                for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
                for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[0][j];
            }
            break;
        }
        std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
        const double* ck = ckpt.empty() ? nullptr : ckpt.data();
        for (int seg = 0; seg < P.nseg; ++seg) for (long i = 0; i < P.N; ++i) {
            double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
            if (seg == P.nseg - 1) {
                double lam[1][N], mu[1][NP];
                backsolve_lane<Mo, 1, (LOSS >> 1)>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, yT.data(), ck, P.ckpt_of_knot.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
            } else {
                double lam[NC][N], mu[NC][NP];
                backsolve_lane<Mo, NC, (LOSS >> 1)>(g, i, P.seg_bounds[seg], P.seg_bounds[seg + 1], p, yT.data(), ck, P.ckpt_of_knot.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                               for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; }
            }
        }
        compose<Mo>(P, segbuf, du0, dp_traj);
        break; }
    case HIPADJ_ALG_GAUSS: {
        if (P.offgrid && P.nseg > 1) {   // k_offgrid_seg<..., true> + k_compose_finish
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
            for (long i = 0; i < P.N; ++i) for (int seg = 0; seg < P.nseg; ++seg) {
                const int q_lo = RS.n - P.seg_bounds[seg + 1], q_hi = RS.n - P.seg_bounds[seg];
                double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
                if (seg == P.nseg - 1) { double lam[1][N], mu[1][NP];
                    gauss_offgrid_lane<Mo, LOSS, 1>(g, i, p, knots.data(), cot, RS, lam, mu, q_lo, q_hi);
                    for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                    for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
                } else { double lam[NC][N], mu[NC][NP];
                    gauss_offgrid_lane<Mo, LOSS, NC>(g, i, p, knots.data(), cot, RS, lam, mu, q_lo, q_hi);
                    for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                                   for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; } }
            }
            compose<Mo>(P, segbuf, du0, dp_traj);
            break;
        }
        if (P.offgrid) {   // k_gauss_offgrid
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), P.nck > 0 ? P.rs_ck.data() : nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            for (long i = 0; i < P.N; ++i) {
                double lam[1][N], mu[1][NP];
                gauss_offgrid_lane<Mo, LOSS>(g, i, p, knots.data(), cot, RS, lam, mu);
                for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
                for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[0][j];
            }
            break;
        }
        std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
        for (int seg = 0; seg < P.nseg; ++seg) for (long i = 0; i < P.N; ++i) {
            double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
            const int kl = P.seg_bounds[seg], kh = P.seg_bounds[seg + 1];
            if (seg == P.nseg - 1) {
                double lam[1][N], mu[1][NP];
                if (P.ip_ckpt) gauss_lane<Mo, 1, PF, LOSS, KM>(g, i, kl, kh, p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else gauss_lane<Mo, 1, PF, LOSS>(g, i, kl, kh, p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
            } else {
                double lam[NC][N], mu[NC][NP];
                if (P.ip_ckpt) gauss_lane<Mo, NC, PF, LOSS, KM>(g, i, kl, kh, p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else gauss_lane<Mo, NC, PF, LOSS>(g, i, kl, kh, p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                               for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; }
            }
        }
        compose<Mo>(P, segbuf, du0, dp_traj);
        break; }
    case HIPADJ_ALG_GAUSS_KRONROD: {
        if (P.offgrid) {   // k_gauss_offgrid<..., true>: sequential in time
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            for (long i = 0; i < P.N; ++i) {
                double lam[1][N], mu[1][NP];
                gauss_offgrid_lane<Mo, LOSS, 1, true>(g, i, p, knots.data(), cot, RS, lam, mu);
                for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
                for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[0][j];
            }
            break;
        }
        std::vector<double> segbuf((size_t)P.nseg * NC * R * Np, 0.0);
        for (int seg = 0; seg < P.nseg; ++seg) for (long i = 0; i < P.N; ++i) {
            double* dst = segbuf.data() + (size_t)seg * NC * R * Np + i;
            const int kl = P.seg_bounds[seg], kh = P.seg_bounds[seg + 1];
            if (seg == P.nseg - 1) {
                double lam[1][N], mu[1][NP];
                if (P.ip_ckpt) gauss_lane<Mo, 1, PF, LOSS, KM, true>(g, i, kl, kh, p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else gauss_lane<Mo, 1, PF, LOSS, 0, true>(g, i, kl, kh, p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int j = 0; j < N; ++j) dst[(size_t)j * Np] = lam[0][j];
                for (int j = 0; j < NP; ++j) dst[(size_t)(N + j) * Np] = mu[0][j];
            } else {
                double lam[NC][N], mu[NC][NP];
                if (P.ip_ckpt) gauss_lane<Mo, NC, PF, LOSS, KM, true>(g, i, kl, kh, p, nullptr, cot, P.save_of_knot_rev.data(), lam, mu, &CK);
                else gauss_lane<Mo, NC, PF, LOSS, 0, true>(g, i, kl, kh, p, knots.data(), cot, P.save_of_knot_rev.data(), lam, mu);
                for (int c = 0; c < NC; ++c) { for (int j = 0; j < N; ++j) dst[((size_t)c * R + j) * Np] = lam[c][j];
                                               for (int j = 0; j < NP; ++j) dst[((size_t)c * R + N + j) * Np] = mu[c][j]; }
            }
        }
        compose<Mo>(P, segbuf, du0, dp_traj);
        break; }
    case HIPADJ_ALG_QUADRATURE: {
        std::vector<dbl2> adj((size_t)(P.offgrid ? P.rs_t.size() : (size_t)P.S) * 2 * N * Np);
        const double atol = cfg->quad_abstol > 0 ? cfg->quad_abstol : 1e-6, rtol = cfg->quad_reltol > 0 ? cfg->quad_reltol : 1e-3;
        if (P.offgrid) {   // k_quad_adj_offgrid + k_quad_gk_offgrid + k_quad_sum
            const RevSteps RS{P.rs_t.data(), P.rs_h.data(), P.rs_te.data(), P.rs_save.data(), nullptr, (int)P.rs_t.size(), P.rs_save_at_start, cfg->t1};
            for (long i = 0; i < P.N; ++i) {
                double lam[N];
                { double gpo[NP]; quad_adj_offgrid_lane<Mo, LOSS>(g, i, p, knots.data(), cot, RS, adj.data(), lam, gpo); }
                for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
                double acc[NP]; for (int j = 0; j < NP; ++j) acc[j] = 0.0;
                for (int q = 0; q < P.nq; ++q) {
                    double res[NP];
                    quad_gk_offgrid_lane<Mo, 128, (LOSS >> 1)>(g, i, p, knots.data(), adj.data(), RS, P.qa[q], P.qb[q], atol, rtol, res);
                    for (int j = 0; j < NP; ++j) acc[j] += res[j];
                }
                for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = acc[j];
            }
            break;
        }
        for (long i = 0; i < P.N; ++i) {
            double lam[N];
            { double gpo[NP]; quad_adj_lane<Mo, PF, LOSS>(g, i, p, knots.data(), cot, P.save_of_knot_rev.data(), adj.data(), lam, gpo); }
            for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
            double acc[NP]; for (int j = 0; j < NP; ++j) acc[j] = 0.0;
            for (int q = 0; q < P.nq; ++q) {
                double res[NP];
                quad_gk_lane<Mo, 128, (LOSS >> 1)>(g, i, p, knots.data(), adj.data(), P.qa[q], P.qb[q], atol, rtol, res);
                for (int j = 0; j < NP; ++j) acc[j] += res[j];
            }
            for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = acc[j];
        }
        break; }
    default: return HIPADJ_ERR_INVALID_ARG;
    }
    if (cfg->p_shared) { for (int j = 0; j < NP; ++j) { double s = 0; for (long i = 0; i < P.N; ++i) s += dp_traj[(size_t)j * Np + i]; dp[j] = s; } }
    else for (long i = 0; i < P.N; ++i) for (int j = 0; j < NP; ++j) dp[i * NP + j] = dp_traj[(size_t)j * Np + i];
    return HIPADJ_OK;
}

// adaptive Tsit5: loops the lane bodies of hipadj_adaptive.hpp exactly as k_forward_tsit5 / k_adjoint_tsit5 + k_finish do
template <class Mo, int ALG, int CC, bool CK = false, int STEP = 0>      // STEP 1: Rosenbrock23 (hipadj_adaptive.hpp ros23_integrate)
static int run_adaptive(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu,
                        double* du0, double* dp, double* out, int* nsteps_out) {
    constexpr int N = Mo::N, NP = Mo::NP, RW = 2 + 5 * N;
    AdaptGeom g; g.N = P.N; g.Npad = P.Npad; g.M = P.M; g.Smax = P.Smax; g.maxit = P.Smax; g.nck = P.nck; g.t0 = cfg->t0; g.t1 = cfg->t1; g.dt0 = cfg->dt;
    g.abstol = cfg->abstol; g.reltol = cfg->reltol; g.loss_shift = cfg->loss_shift; g.loss_kind = cfg->loss_kind; g.no_start = cfg->no_start;
    { const double lw = cfg->loss_scale != 0.0 ? cfg->loss_scale : 1.0; g.la = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? lw : 0.0; g.lb = cfg->loss_kind == HIPADJ_LOSS_LSQ_DATA ? -lw : 1.0; g.lflags = (cfg->reference_literal && (cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD)) ? 3 : 0; }
    g.p_shared = cfg->p_shared; g.cont_cost = cfg->cont_cost; g.SmaxI = P.SmaxI;
    const long Np = P.Npad;
    std::vector<double> rec(ALG != 1 ? (size_t)(CK ? P.SmaxI : P.Smax) * RW * Np : 0), outT((size_t)P.M * N * Np), yT((size_t)N * Np), ckpt((size_t)P.nck * N * Np);
    std::vector<double> cotT(cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT ? (size_t)P.M * N * Np : 0), dp_traj((size_t)NP * Np, 0.0);
    std::vector<int> nsteps((size_t)Np, 0);
    std::vector<int> ev_s(model_has_cond<Mo>::value ? (size_t)EMU_MAXEV * Np : 0), nev(model_has_cond<Mo>::value ? (size_t)Np : 0), ev_k(ev_s.size());      // ContinuousCallback: the event lists (hipadj_api.hip d_ev_s / d_nev)
    std::vector<double> ev_t(model_has_cond<Mo>::value ? (size_t)EMU_MAXEV * Np : 0), ev_ul(model_has_cond<Mo>::value ? (size_t)EMU_MAXEV * N * Np : 0);
    std::vector<double> ev_ur(ev_ul.size());
    if (model_has_cond<Mo>::value) { g.maxev = EMU_MAXEV; g.ev_s = ev_s.data(); g.nev = nev.data(); g.ev_t = ev_t.data(); g.ev_ul = ev_ul.data(); g.ev_ur = ev_ur.data(); g.ev_k = ev_k.data();
                                   }
    std::vector<double> ev_dl, ev_dr;      // test hook (emu_set_event_cotangents): cotangents at the saved event states, [N][EMU_MAXEV][n] as hipadj_set_event_cotangents takes them
    if (model_has_cond<Mo>::value) for (int side = 0; side < 2; ++side) {
        const double* src = side ? g_emu_ev_dr : g_emu_ev_dl; if (!src) continue;
        std::vector<double>& dst = side ? ev_dr : ev_dl; dst.assign((size_t)EMU_MAXEV * N * Np, 0.0);
        for (long i = 0; i < P.N; ++i) for (int k = 0; k < EMU_MAXEV; ++k) for (int j = 0; j < N; ++j) dst[((size_t)k * N + j) * Np + i] = src[((size_t)i * EMU_MAXEV + k) * N + j];
        if (side) g.ev_dr = dst.data(); else g.ev_dl = dst.data();
    }
    // (... and emu_set_event_output: the event times and states handed back, [N][EMU_MAXEV][1 + 2 n] = t, u-, u+, filled after the forward loop below)
    int flag = 0;
    std::vector<double> kbuf((size_t)KS_ROWS * (2 * N + NP)), kfbuf((size_t)KS_ROWS * N);   // stage storage of one lane (LDS columns on the device), stride 1 here
    for (long i = 0; i < P.N; ++i)
        forward_tsit5_lane<Mo, STEP>(g, i, u0, p, (rec.empty() || CK) ? nullptr : rec.data(), nsteps.data(), P.save_times.data(), outT.data(),
                               P.ck_times.data(), ckpt.empty() ? nullptr : ckpt.data(), yT.data(), &flag, kbuf.data(), 1);
    if (nsteps_out) for (long i = 0; i < P.N; ++i) nsteps_out[i] = model_has_cond<Mo>::value ? nev[i] : nsteps[i];      // (a model with events reports its event counts there)
    if (model_has_cond<Mo>::value && g_emu_ev_out)
        for (long i = 0; i < P.N; ++i) for (int k = 0; k < EMU_MAXEV; ++k) {
            double* o = g_emu_ev_out + ((size_t)i * EMU_MAXEV + k) * (1 + 2 * N);
            const bool live = k < nev[i];
            o[0] = live ? ev_t[(size_t)k * Np + i] : 0.0;
            for (int j = 0; j < N; ++j) { o[1 + j] = live ? ev_ul[((size_t)k * N + j) * Np + i] : 0.0; o[1 + N + j] = live ? ev_ur[((size_t)k * N + j) * Np + i] : 0.0; }
        }
    if (flag & 4) return HIPADJ_ERR_MAXITERS;
    if (out) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) out[i * P.M * N + c] = outT[(size_t)c * Np + i];
    if (!cotT.empty()) for (long i = 0; i < P.N; ++i) for (int c = 0; c < P.M * N; ++c) cotT[(size_t)c * Np + i] = dLdu[i * P.M * N + c];
    const int SmaxA = 2 * P.Smax + P.M + 16;
    g.SmaxA = SmaxA;
    std::vector<double> arec(ALG == 3 ? (size_t)SmaxA * RW * Np : 0);
    std::vector<int> nsteps_adj((size_t)Np, 0);
    const double qatol = cfg->quad_abstol > 0 ? cfg->quad_abstol : 1e-6, qrtol = cfg->quad_reltol > 0 ? cfg->quad_reltol : 1e-3;
    for (long i = 0; i < P.N; ++i) {
        double lam[N], mu[NP];
        adjoint_tsit5_lane<Mo, ALG, CC, CK, STEP>(g, i, p, rec.data(), nsteps.data(), yT.data(), ckpt.empty() ? nullptr : ckpt.data(), P.ck_times.data(),
                                            P.save_times.data(), P.tstops_desc.data(), (int)P.tstops_desc.size(), cotT.empty() ? nullptr : cotT.data(), lam, mu, &flag, kbuf.data(), 1,
                                            arec.empty() ? nullptr : arec.data(), nsteps_adj.data(), SmaxA, kfbuf.data(), CK ? rec.data() : nullptr);
        for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
        if (ALG == 3) {   // k_quad_gk_tsit5 + k_quad_sum
            if (flag & 4) return HIPADJ_ERR_MAXITERS;
            if (!model_dae<Mo>::value && !model_has_cond<Mo>::value) for (int j = 0; j < NP; ++j) mu[j] = 0.0;      // (k_quad_sum with add = 1 for a semi-explicit DAE: the loss jumps' parameter term is already there)
            for (int q = 0; q < P.nq; ++q) {
                double res[NP];
                quad_gk_tsit5_lane<Mo, 128, CC>(g, i, p, rec.data(), nsteps.data(), arec.data(), nsteps_adj.data(), P.qa[q], P.qb[q], qatol, qrtol, res);
                for (int j = 0; j < NP; ++j) mu[j] += res[j];
            }
        }
        for (int j = 0; j < NP; ++j) dp_traj[(size_t)j * Np + i] = mu[j];
    }
    if (flag & 4) return HIPADJ_ERR_MAXITERS;
    if (cfg->p_shared) { for (int j = 0; j < NP; ++j) { double s = 0; for (long i = 0; i < P.N; ++i) s += dp_traj[(size_t)j * Np + i]; dp[j] = s; } }
    else for (long i = 0; i < P.N; ++i) for (int j = 0; j < NP; ++j) dp[i * NP + j] = dp_traj[(size_t)j * Np + i];
    return HIPADJ_OK;
}

template <class Mo>
static int dispatch_adaptive(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu, double* du0, double* dp, double* out, int* ns) {
    if (cfg->stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) {
#define EMU_ROS_CASE(A, C, K) case A * 4 + C: if constexpr (!model_dae<Mo>::value || (C == 0 && A != 1)) return run_adaptive<Mo, A, C, K, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns); else break;
        if (P.ip_ckpt) switch (cfg->alg * 4 + cfg->cont_cost) {
        EMU_ROS_CASE(0, 0, true) EMU_ROS_CASE(0, 1, true) EMU_ROS_CASE(0, 2, true) EMU_ROS_CASE(2, 0, true) EMU_ROS_CASE(2, 1, true) EMU_ROS_CASE(2, 2, true)
        EMU_ROS_CASE(4, 0, true) EMU_ROS_CASE(4, 1, true) EMU_ROS_CASE(4, 2, true)
        default: break;
        }
        else switch (cfg->alg * 4 + cfg->cont_cost) {
        EMU_ROS_CASE(0, 0, false) EMU_ROS_CASE(0, 1, false) EMU_ROS_CASE(0, 2, false) EMU_ROS_CASE(1, 0, false) EMU_ROS_CASE(1, 1, false) EMU_ROS_CASE(1, 2, false)
        EMU_ROS_CASE(2, 0, false) EMU_ROS_CASE(2, 1, false) EMU_ROS_CASE(2, 2, false) EMU_ROS_CASE(3, 0, false) EMU_ROS_CASE(3, 1, false) EMU_ROS_CASE(3, 2, false)
        EMU_ROS_CASE(4, 0, false) EMU_ROS_CASE(4, 1, false) EMU_ROS_CASE(4, 2, false)
        default: break;
        }
#undef EMU_ROS_CASE
        return HIPADJ_ERR_UNSUPPORTED;
    }
    if (P.ip_ckpt) {
        switch (cfg->alg * 4 + cfg->cont_cost) {
        case HIPADJ_ALG_INTERPOLATING * 4 + 0: return run_adaptive<Mo, 0, 0, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_INTERPOLATING * 4 + 1: return run_adaptive<Mo, 0, 1, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_INTERPOLATING * 4 + 2: return run_adaptive<Mo, 0, 2, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS * 4 + 0: return run_adaptive<Mo, 2, 0, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS * 4 + 1: return run_adaptive<Mo, 2, 1, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS * 4 + 2: return run_adaptive<Mo, 2, 2, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return run_adaptive<Mo, 4, 0, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return run_adaptive<Mo, 4, 1, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 2: return run_adaptive<Mo, 4, 2, true>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
        default: return HIPADJ_ERR_UNSUPPORTED;
        }
    }
    switch (cfg->alg * 4 + cfg->cont_cost) {
    case HIPADJ_ALG_INTERPOLATING * 4 + 0: return run_adaptive<Mo, 0, 0>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_INTERPOLATING * 4 + 1: return run_adaptive<Mo, 0, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_INTERPOLATING * 4 + 2: return run_adaptive<Mo, 0, 2>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_BACKSOLVE * 4 + 0: return run_adaptive<Mo, 1, 0>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_BACKSOLVE * 4 + 1: return run_adaptive<Mo, 1, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_BACKSOLVE * 4 + 2: return run_adaptive<Mo, 1, 2>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS * 4 + 0: return run_adaptive<Mo, 2, 0>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS * 4 + 1: return run_adaptive<Mo, 2, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS * 4 + 2: return run_adaptive<Mo, 2, 2>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_QUADRATURE * 4 + 0: return run_adaptive<Mo, 3, 0>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_QUADRATURE * 4 + 1: return run_adaptive<Mo, 3, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_QUADRATURE * 4 + 2: return run_adaptive<Mo, 3, 2>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return run_adaptive<Mo, 4, 0>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return run_adaptive<Mo, 4, 1>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 2: return run_adaptive<Mo, 4, 2>(cfg, P, u0, p, dLdu, du0, dp, out, ns);
    default: return HIPADJ_ERR_UNSUPPORTED;
    }
}

// Build units (tests/emu.py compiles them in parallel): EMU_UNIT undefined = everything in one translation unit (the variant builds
// of test_emu_parity.py); EMU_UNIT = 0 = the C entry points, the per-model dispatchers declared `extern template`;
// EMU_UNIT = 1..16 = the explicit instantiation of ONE model's dispatcher (all lane bodies of that model).
#ifndef EMU_UNIT
#define EMU_UNIT -1
#endif
template <class Mo>
int dispatch_mode(const hipadj_config* cfg, const Plan& P, const double* u0, const double* p, const double* dLdu, double* du0, double* dp, double* out) {
    if (P.adaptive) return dispatch_adaptive<Mo>(cfg, P, u0, p, dLdu, du0, dp, out, nullptr);
    const int mode = ((cfg->loss_kind != HIPADJ_LOSS_LSQ_SHIFT && P.M > 0) ? 0 : 1) | (cfg->cont_cost << 1);
    switch (mode) {
    case 0: return run<Mo, 0>(cfg, P, u0, p, dLdu, du0, dp, out);
    case 1: return run<Mo, 1>(cfg, P, u0, p, dLdu, du0, dp, out);
    case 2: return run<Mo, 2>(cfg, P, u0, p, dLdu, du0, dp, out);
    case 3: return run<Mo, 3>(cfg, P, u0, p, dLdu, du0, dp, out);
    case 4: return run<Mo, 4>(cfg, P, u0, p, dLdu, du0, dp, out);
    case 5: return run<Mo, 5>(cfg, P, u0, p, dLdu, du0, dp, out);
    default: return HIPADJ_ERR_UNSUPPORTED;
    }
}

#if EMU_UNIT == 0
extern template int dispatch_mode<ModelLV>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<ModelLVT>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<ModelLorenz>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<ModelLinDiag>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<ModelFallMass>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRing<4>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRingMM<5>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRober>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRoberDAE<0>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRoberDAE<5>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRoberDAE<5, 1>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuBall<1>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuBall<4>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuRelax>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuBall2D>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
extern template int dispatch_mode<EmuBall<7>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 1
template int dispatch_mode<ModelLV>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 2
template int dispatch_mode<ModelLVT>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 3
template int dispatch_mode<ModelLorenz>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 4
template int dispatch_mode<ModelLinDiag>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 5
template int dispatch_mode<ModelFallMass>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 6
template int dispatch_mode<EmuRing<4>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 7
template int dispatch_mode<EmuRingMM<5>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 8
template int dispatch_mode<EmuRober>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 9
template int dispatch_mode<EmuRoberDAE<0>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 10
template int dispatch_mode<EmuRoberDAE<5>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 11
template int dispatch_mode<EmuRoberDAE<5, 1>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 12
template int dispatch_mode<EmuBall<1>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 13
template int dispatch_mode<EmuBall<4>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 14
template int dispatch_mode<EmuRelax>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 15
template int dispatch_mode<EmuBall2D>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#elif EMU_UNIT == 16
template int dispatch_mode<EmuBall<7>>(const hipadj_config*, const Plan&, const double*, const double*, const double*, double*, double*, double*);
#endif

#if EMU_UNIT <= 0
static std::string g_err;
extern "C" const char* emu_last_error() { return g_err.c_str(); }

extern "C" void emu_set_event_cotangents(const double* dl, const double* dr) { g_emu_ev_dl = dl; g_emu_ev_dr = dr; }
extern "C" void emu_set_event_output(double* out) { g_emu_ev_out = out; }
extern "C" int emu_plan(const hipadj_config* cfg, int* nseg, int* seg_bounds /*[cap]*/, int cap, int* nck, int* nq) {
    Plan P; const int rc = make_plan(cfg, P, g_err); if (rc) return rc;
    *nseg = P.nseg; *nck = P.nck; *nq = P.nq;
    for (int i = 0; i <= P.nseg && i < cap; ++i) seg_bounds[i] = P.seg_bounds[i];
    return 0;
}

extern "C" int emu_forward_adjoint(const hipadj_config* cfg, const double* u0, const double* p, const double* dLdu,
                                   double* du0, double* dp, double* out) {
    Plan P; const int rc = make_plan(cfg, P, g_err); if (rc) return rc;
    switch (cfg->model) {
    case HIPADJ_MODEL_LV: return dispatch_mode<ModelLV>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_LVT: return dispatch_mode<ModelLVT>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_LORENZ: return dispatch_mode<ModelLorenz>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_LINDIAG: return dispatch_mode<ModelLinDiag>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_FALLMASS: return dispatch_mode<ModelFallMass>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 4: return dispatch_mode<EmuRing<4>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 105: return dispatch_mode<EmuRingMM<5>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 203: return dispatch_mode<EmuRober>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 204: return dispatch_mode<EmuRoberDAE<0>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 205: return dispatch_mode<EmuRoberDAE<5>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 206: return dispatch_mode<EmuRoberDAE<5, 1>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 301: return dispatch_mode<EmuBall<1>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 304: return dispatch_mode<EmuBall<4>>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 303: return dispatch_mode<EmuRelax>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 305: return dispatch_mode<EmuBall2D>(cfg, P, u0, p, dLdu, du0, dp, out);
    case HIPADJ_MODEL_USER_BASE + 307: return dispatch_mode<EmuBall<7>>(cfg, P, u0, p, dLdu, du0, dp, out);
    default: g_err = "no emulation for this model"; return HIPADJ_ERR_UNSUPPORTED;
    }
}
#endif   // EMU_UNIT <= 0
