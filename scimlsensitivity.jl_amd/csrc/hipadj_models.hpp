// hipadj_models.hpp — compile-time model registry for the lane-per-trajectory kernel family.
//
// Each model provides device-inlined f, (df/du)^T lam and (df/dp)^T lam with the argument meaning of the
// reference's user-VJP seam  ODEFunction(f!; vjp = (dlam, lam, u, p, t), vjp_p = (dgrad, lam, u, p, t))
// (src/derivative_wrappers.jl:284-359; exercised by test/Core3/user_vjp.jl:14-38): both VJPs are
// UN-negated — the adjoint RHS negates afterwards (src/interpolating_adjoint.jl:169-170).
// Hand-derived; no runtime AD on the device.
#pragma once

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define HIPADJ_HD __host__ __device__ __forceinline__
#else
#define HIPADJ_HD inline
#endif
// Scheduling fence between unrolled RK4 steps (device code only): stops hipcc from hoisting the lambda-independent
// part of LATER steps (Hermite midpoints of knots that are still in flight) above the current step, which would
// turn the counted `s_waitcnt vmcnt(N)` of the software prefetch into a full drain once per block.
#if defined(__HIP_DEVICE_COMPILE__)
#define HIPADJ_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HIPADJ_STEP_FENCE() ((void)0)
#endif

namespace hipadj {

// does a model carry the stage operators (JT / PT) of the multi-column sweep?  (runtime-compiled user models and the small
// registry models do not: they run the generic vjp_u / vjp_p form)
template <class Mo, class = void> struct model_has_ops { static constexpr bool value = false; };
template <class Mo, class = void> struct model_ops_count { static constexpr int value = 1; };
#if !defined(HIPADJ_DISABLE_OPS)     // development A/B switch (scripts/kbench.hip): the generic form for every model
template <class Mo> struct model_has_ops<Mo, decltype((void)Mo::HAS_OPS)> { static constexpr bool value = Mo::HAS_OPS; };
template <class Mo> struct model_ops_count<Mo, decltype((void)Mo::NOC)> { static constexpr int value = Mo::NOC; };
#endif
// A wave-uniform double as the compiler should see it: both halves through v_readfirstlane, i.e. an SGPR pair (device code only).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double hipadj_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
#else
inline double hipadj_uniform(double v) { return v; }
#endif

// ---- column bundles -------------------------------------------------------------------------------------------------------------------
// A time-segment lane carries 1 + n adjoint columns (hipadj_lane.hpp) and the model's VJPs are LINEAR in lam, so G columns can go through a
// VJP body as ONE scalar of type Cols<G>: everything that depends on (u, p, t) only — transcendentals, products of states — is then evaluated once
// per bundle by construction instead of once per column if the optimiser happens to merge the copies (it did not for runtime models with
// n <= 4 states: sin/cos inlined per column, 2.2x the instructions of the n = 5 kernel).  A model opts in with HAS_COLS and
// templated bodies vjp_u_t<LT> / vjp_p_t<LT>; only linear operations exist on the type, so a body that is not linear in lam does not compile
// with it (the runtime-model compiler then falls back to the per-column form).
template <int G> struct Cols {
    double v[G];
    HIPADJ_HD Cols() {}
    HIPADJ_HD Cols(double x) {
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = x; }
    HIPADJ_HD Cols(int x) {
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = (double)x; }
};
#define HIPADJ_COLS_BIN(OP)                                                                                                      \
    template <int G> HIPADJ_HD Cols<G> operator OP(const Cols<G>& a, const Cols<G>& b) { Cols<G> r;                              \
        _Pragma("unroll") for (int g = 0; g < G; ++g) r.v[g] = a.v[g] OP b.v[g];                                                 \
        return r; }
HIPADJ_COLS_BIN(+)
HIPADJ_COLS_BIN(-)
#undef HIPADJ_COLS_BIN
template <int G> HIPADJ_HD Cols<G> operator-(const Cols<G>& a) { Cols<G> r;
#pragma unroll
    for (int g = 0; g < G; ++g) r.v[g] = -a.v[g];
    return r; }
template <int G> HIPADJ_HD Cols<G> operator+(const Cols<G>& a) { return a; }
template <int G> HIPADJ_HD Cols<G> operator*(double s, const Cols<G>& a) { Cols<G> r;
#pragma unroll
    for (int g = 0; g < G; ++g) r.v[g] = s * a.v[g];
    return r; }
template <int G> HIPADJ_HD Cols<G> operator*(const Cols<G>& a, double s) { return s * a; }
template <int G> HIPADJ_HD Cols<G> operator/(const Cols<G>& a, double s) { Cols<G> r;
#pragma unroll
    for (int g = 0; g < G; ++g) r.v[g] = a.v[g] / s;
    return r; }
template <int G> HIPADJ_HD Cols<G>& operator+=(Cols<G>& a, const Cols<G>& b) { a = a + b; return a; }
template <int G> HIPADJ_HD Cols<G>& operator-=(Cols<G>& a, const Cols<G>& b) { a = a - b; return a; }
template <int G> HIPADJ_HD Cols<G>& operator*=(Cols<G>& a, double s) { a = s * a; return a; }
template <int G> HIPADJ_HD Cols<G>& operator/=(Cols<G>& a, double s) { a = a / s; return a; }
// the affine column (index 0 of the first bundle) alone receives the cost terms g_u / g_p
template <int G> HIPADJ_HD void cols_add_first(Cols<G>& a, double x) { a.v[0] += x; }
HIPADJ_HD void cols_add_first(double& a, double x) { a += x; }
template <class Mo, class = void> struct model_has_cols { static constexpr bool value = false; };
template <class Mo> struct model_has_cols<Mo, decltype((void)Mo::HAS_COLS)> { static constexpr bool value = Mo::HAS_COLS; };
// Semi-explicit DAE (round 6): a model with `static constexpr bool DAE = true` carries its constant SINGULAR mass matrix — mass(i, j), row-major, and isalg(i) = row i of M
// is zero (src/adjoint_common.jl:116-122) — and is integrated in mass-matrix form by the Rosenbrock23 lanes (hipadj_adaptive.hpp); every other stepper refuses it at plan time.
// The Jacobian df/du for the W = M - d h J of Rosenbrock23 (hipadj_adaptive.hpp).  A model with dual-number VJPs hands it over from ONE dual evaluation of f (`HAS_JAC`, jac);
// every other model gets it row by row from its VJP, (df/du)' e_r = row r — with hand-written bodies the unit vector folds into them.  (Measured, ring n = 6, 8192 trajectories:
// n unit-vector calls of a dual-number vjp_u are n dual Jacobians — forward 4.5 ms, reverse 32 ms; with jac 2.1 / 15.9 ms; hand-written bodies 2.8 / 12.2 ms: profiles/r6_ros23_auto_vs_hand.jsonl.)
template <class Mo, class = void> struct model_has_jac { static constexpr bool value = false; };
template <class Mo> struct model_has_jac<Mo, decltype((void)Mo::HAS_JAC)> { static constexpr bool value = Mo::HAS_JAC; };
template <class Mo> HIPADJ_HD void model_jacobian(double (&J)[Mo::N][Mo::N], const double (&u)[Mo::N], const double (&p)[Mo::NP], double t) {
    if constexpr (model_has_jac<Mo>::value) Mo::jac(J, u, p, t);
    else {
#pragma unroll
        for (int r = 0; r < Mo::N; ++r) {
            double e[Mo::N], row[Mo::N];
#pragma unroll
            for (int j = 0; j < Mo::N; ++j) e[j] = j == r ? 1.0 : 0.0;
            Mo::vjp_u(row, e, u, p, t);
#pragma unroll
            for (int c = 0; c < Mo::N; ++c) J[r][c] = row[c];
        }
    }
}
template <class Mo, class = void> struct model_dae { static constexpr bool value = false; };
template <class Mo> struct model_dae<Mo, decltype((void)Mo::DAE)> { static constexpr bool value = Mo::DAE; };
// a runtime model with a ContinuousCallback (hipadj_model_set_continuous_callback): cond / cond_grad / cc_affect / cc_affect_jvp / cc_affect_vjp, hipadj_adaptive.hpp "events"
template <class Mo, class = void> struct model_has_cond { static constexpr bool value = false; };
template <class Mo> struct model_has_cond<Mo, decltype((void)Mo::HAS_COND)> { static constexpr bool value = Mo::HAS_COND; };
template <class Mo, class = void> struct model_ncond { static constexpr int value = 1; };      // components of the condition (VectorContinuousCallback: > 1)
template <class Mo> struct model_ncond<Mo, decltype((void)Mo::NCOND)> { static constexpr int value = Mo::NCOND; };
template <class Mo, class = void> struct model_cdir { static constexpr int value = 0; };       // which crossings fire: 0 both, +1 upcrossings only (affect_neg! = nothing), -1 downcrossings only
template <class Mo> struct model_cdir<Mo, decltype((void)Mo::CDIR)> { static constexpr int value = Mo::CDIR; };
// Bundle width for n states and NC columns: at most ELEMS doubles per bundle vector (seven such vectors are live in an RK4 step), the columns spread
// evenly over the fewest bundles.  The sweeps bundle only when ONE bundle holds all columns (NB == 1: n <= 4 for InterpolatingAdjoint at 24 elements,
// n <= 5 for the lambda-only GaussAdjoint step at 32): with two or more bundles the (u, p, t)-dependent work is repeated per bundle and the measured
// kernels are slower than the per-column form (ring n = 5 / 6 / 8: 4.5 / 7.7 / 19.9 ms against 1.4 / 1.9 / 4.6 ms, profiles/r2_user_cols_ab.log).
#ifndef HIPADJ_COLS_ELEMS
#define HIPADJ_COLS_ELEMS 24
#endif
#ifndef HIPADJ_COLS_ELEMS_GAUSS
#define HIPADJ_COLS_ELEMS_GAUSS 32
#endif
template <int N, int NC, int ELEMS = HIPADJ_COLS_ELEMS> struct cols_bundle {
    static constexpr int GMAX = (ELEMS / N) < 2 ? 2 : (ELEMS / N);
    static constexpr int NB = (NC + GMAX - 1) / GMAX;
    static constexpr int G = (NC + NB - 1) / NB;
};
// models whose VJPs cost one evaluation per CALL whatever the number of columns (dual-number models: the whole Jacobian per evaluation of f) bundle
// in several groups as well; set by the generator (COLS_MULTI)
template <class Mo, class = void> struct model_cols_multi { static constexpr bool value = false; };
template <class Mo> struct model_cols_multi<Mo, decltype((void)Mo::COLS_MULTI)> { static constexpr bool value = Mo::COLS_MULTI; };
template <class Mo, class LT> struct model_vjp {   // LT = double: the model's plain entry points; LT = Cols<G>: its templated bodies
    HIPADJ_HD static void u(LT (&out)[Mo::N], const LT (&lam)[Mo::N], const double (&y)[Mo::N], const double (&p)[Mo::NP], double t) { Mo::template vjp_u_t<LT>(out, lam, y, p, t); }
    HIPADJ_HD static void p_(LT (&out)[Mo::NP], const LT (&lam)[Mo::N], const double (&y)[Mo::N], const double (&p)[Mo::NP], double t) { Mo::template vjp_p_t<LT>(out, lam, y, p, t); }
};
template <class Mo> struct model_vjp<Mo, double> {
    HIPADJ_HD static void u(double (&out)[Mo::N], const double (&lam)[Mo::N], const double (&y)[Mo::N], const double (&p)[Mo::NP], double t) { Mo::vjp_u(out, lam, y, p, t); }
    HIPADJ_HD static void p_(double (&out)[Mo::NP], const double (&lam)[Mo::N], const double (&y)[Mo::N], const double (&p)[Mo::NP], double t) { Mo::vjp_p(out, lam, y, p, t); }
};

constexpr int HIPADJ_CKPT_KMAX = 16;   // longest checkpoint interval (steps) the in-kernel re-solve tile holds

// Lotka-Volterra (test/Core3/user_vjp.jl:6-10): du1 = p1 u1 - p2 u1 u2 ; du2 = -p3 u2 + p4 u1 u2
struct ModelLV {
    static constexpr int N = 2, NP = 4;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1];
        du[1] = -p[2] * u[1] + p[3] * u[0] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        dl[0] = (p[0] - p[1] * u[1]) * l[0] + p[3] * u[1] * l[1];
        dl[1] = -p[1] * u[0] * l[0] + (-p[2] + p[3] * u[0]) * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        const double xy = u[0] * u[1];
        dg[0] = u[0] * l[0]; dg[1] = -xy * l[0]; dg[2] = -u[1] * l[1]; dg[3] = xy * l[1];
    }
};

// time-dependent LV `fb` (test/Core3/adjoint.jl:8-12, Jacobian :18-25)
struct ModelLVT {
    static constexpr int N = 2, NP = 4;
    static constexpr bool TIME_DEP = true;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) {
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1] * t;
        du[1] = -p[2] * u[1] + t * p[3] * u[0] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) {
        dl[0] = (p[0] - p[1] * u[1] * t) * l[0] + t * u[1] * p[3] * l[1];
        dl[1] = -p[1] * u[0] * t * l[0] + (-p[2] + t * u[0] * p[3]) * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double t) {
        const double xyt = u[0] * u[1] * t;
        dg[0] = u[0] * l[0]; dg[1] = -xyt * l[0]; dg[2] = -u[1] * l[1]; dg[3] = xyt * l[1];
    }
};

// Lorenz-63 (test/Core3/adjoint.jl:1160-1166), p = (sigma, rho, beta)
struct ModelLorenz {
    static constexpr int N = 3, NP = 3;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * (u[1] - u[0]);
        du[1] = u[0] * (p[1] - u[2]) - u[1];
        du[2] = u[0] * u[1] - p[2] * u[2];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        dl[0] = -p[0] * l[0] + (p[1] - u[2]) * l[1] + u[1] * l[2];
        dl[1] = p[0] * l[0] - l[1] + u[0] * l[2];
        dl[2] = -u[0] * l[1] - p[2] * l[2];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        dg[0] = (u[1] - u[0]) * l[0]; dg[1] = u[0] * l[1]; dg[2] = -u[2] * l[2];
    }
    // Stage operators for the multi-column (time-segmented) reverse sweep — adj_rk4_step_ops, hipadj_lane.hpp.  The same two
    // products as vjp_u / vjp_p, split three ways:
    //   ops_const   what depends on (p, dt) only: formed ONCE per lane; with shared parameters the sweep moves these values to
    //               SGPRs (wave-uniform), so they cost neither VGPRs nor instructions inside the time loop
    //   jt_prep / pt_prep   the y-dependent entries of  JT = shift I + c (df/du)^T  and  PT = c (df/dp)^T : once per stage, shared
    //               by the 1 + n columns of the lane
    //   jt_mul / jt_mul_add / pt_acc   per column: out = JT l , out = base + JT l , mu += PT l — pure FMA chains
    // (df/du)^T = [-s  r-z  y ;  s  -1  x ;  0  -x  -b]  at u = (x, y, z), p = (s, r, b); (df/dp)^T = diag(y - x, x, -z).
    // The four JT of a step (which = 0..3): M1 = I + h/2 J(y_hi)^T, Lm = h/2 J(y_mid)^T, Lm2 = h J(y_mid)^T, M4 = I/3 + h/6 J(y_lo)^T;
    // the two PT scales (which = 0, 1): h/6, h/3.
    static constexpr bool HAS_OPS = true;
    static constexpr int NOC = 26;       // 4 x {c, d0, d1, d2, a10, c r} + {h/6, h/3}
    struct JT { double d0, d1, d2, a10, a01, a02, a12; };
    struct PT { double q0, q1, q2; };
    HIPADJ_HD static void ops_const(double (&oc)[NOC], const double (&p)[NP], double h) {
        const double cs[4] = {0.5 * h, 0.5 * h, h, h / 6.0}, sh[4] = {1.0, 0.0, 0.0, 1.0 / 3.0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            oc[6 * w + 0] = cs[w]; oc[6 * w + 1] = sh[w] - cs[w] * p[0]; oc[6 * w + 2] = sh[w] - cs[w]; oc[6 * w + 3] = sh[w] - cs[w] * p[2];
            oc[6 * w + 4] = cs[w] * p[0]; oc[6 * w + 5] = cs[w] * p[1];
        }
        oc[24] = h / 6.0; oc[25] = h / 3.0;
    }
    template <int W> HIPADJ_HD static void jt_prep(JT& J, const double (&oc)[NOC], const double (&u)[N], double) {
        J.d0 = oc[6 * W + 1]; J.d1 = oc[6 * W + 2]; J.d2 = oc[6 * W + 3]; J.a10 = oc[6 * W + 4];
        J.a01 = oc[6 * W + 5] - oc[6 * W] * u[2]; J.a02 = oc[6 * W] * u[1]; J.a12 = oc[6 * W] * u[0];
    }
    HIPADJ_HD static void jt_mul(double (&o)[N], const JT& J, const double (&l)[N]) {
        o[0] = J.d0 * l[0] + J.a01 * l[1] + J.a02 * l[2];
        o[1] = J.a10 * l[0] + J.d1 * l[1] + J.a12 * l[2];
        o[2] = J.d2 * l[2] - J.a12 * l[1];
    }
    HIPADJ_HD static void jt_mul_add(double (&o)[N], const JT& J, const double (&b)[N], const double (&l)[N]) {
        o[0] = b[0] + J.d0 * l[0] + J.a01 * l[1] + J.a02 * l[2];
        o[1] = b[1] + J.a10 * l[0] + J.d1 * l[1] + J.a12 * l[2];
        o[2] = b[2] + J.d2 * l[2] - J.a12 * l[1];
    }
    template <int W> HIPADJ_HD static void pt_prep(PT& Q, const double (&oc)[NOC], const double (&u)[N], double) {
        Q.q0 = oc[24 + W] * (u[1] - u[0]); Q.q1 = oc[24 + W] * u[0]; Q.q2 = -oc[24 + W] * u[2];
    }
    HIPADJ_HD static void pt_acc(double (&mu)[NP], const PT& Q, const double (&l)[N]) {
        mu[0] += Q.q0 * l[0]; mu[1] += Q.q1 * l[1]; mu[2] += Q.q2 * l[2];
    }
};

// u' = p .* u (test/Core1/sparse_adjoint.jl:6-8: jac = diag(p), paramjac = diag(u))
struct ModelLinDiag {
    static constexpr int N = 2, NP = 2;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * u[0]; du[1] = p[1] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&p)[NP], double) {
        dl[0] = p[0] * l[0]; dl[1] = p[1] * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        dg[0] = u[0] * l[0]; dg[1] = u[1] * l[1];
    }
};

// falling mass (test/Core7/physical_ode_regression.jl:20-23): u' = [u2, -g], p = (g, m)
struct ModelFallMass {
    static constexpr int N = 2, NP = 2;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = u[1]; du[1] = -p[0];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&)[NP], double) {
        dl[0] = 0.0; dl[1] = l[0];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&)[N], const double (&)[NP], double) {
        dg[0] = -l[1]; dg[1] = 0.0;
    }
};

}  // namespace hipadj
