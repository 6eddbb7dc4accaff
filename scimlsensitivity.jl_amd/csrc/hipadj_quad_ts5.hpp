// hipadj_quad_ts5.hpp — adaptive Tsit5 with four lanes per trajectory (round 4; the fixed-step forward solve of this mapping is hipadj_quad.hpp).
//
// The lane-per-trajectory adaptive kernels are bound by ONE wave's instruction stream (157 lone wavefronts at 10^4 trajectories, ~1200 VALU instructions per
// step attempt of the Interpolating sweep, profiles/r3_tsit5_counters.txt): the stage sums, the interpolation of the forward solution, the error estimate
// and the record / gradient stores all cost n (or n + np) instructions per lane.  Here lane c of a quad owns component c of every vector — state y_c,
// adjoint lam_c, parameter sum mu_c — and runs hipadj_adaptive.hpp's tsit5_integrate ITSELF on its one to three components: same controller, tstop clipping,
// FSAL and callback protocol, with the controller's norms summed over the quad by two DPP moves (QuadNorm).  The right-hand sides are the models' COMPONENT
// FORMS (QuadForm / QuadAdj: uniform FMA chains on operands rotated through the quad with quad_perm).  Every lane of a quad carries the same t, dt and
// error estimate bit for bit (the butterfly adds the same numbers in every lane), so a quad never diverges; quads of one wavefront take their own step
// sequences under the exec mask like the lanes of the lane family do.  Lanes beyond the model's components stay alive with zero state and zero constants
// (they contribute exact zeros to the norms) and skip the stores.
//
// Data layout = the lane family's (records [Smax][2 + 5 n][Npad], outT / ckpt / yT / cotT component-major, dp_traj [np][Npad]): every kernel of either
// mapping reads what the other wrote.  Dispatched for models with QuadAdj (compiled-in Lorenz, Lotka-Volterra and its time-dependent form): the forward solve and the Interpolating, Backsolve and Gauss sweeps
// without a continuous cost and without checkpointing=true; everything else keeps the lane kernels.
#pragma once
#include "hipadj_quad.hpp"
#include "hipadj_adaptive.hpp"

namespace hipadj {

// the controller's norms over a quad: this lane's partial sum + the three others', identical bits in all four lanes
struct QuadNorm {
    int ncomp;
    HIPADJ_HD double sum(double x, int, double) const {
        x += quad_perm<HIPADJ_QP(1, 0, 3, 2)>(x);
        x += quad_perm<HIPADJ_QP(2, 3, 0, 1)>(x);
        return x;
    }
    HIPADJ_HD double count(int) const { return (double)ncomp; }
    HIPADJ_HD void begin_attempt() const {}
    HIPADJ_HD void after_k0() const {}
    HIPADJ_HD void after_stage(int) const {}
    HIPADJ_HD void accept(double) const {}
    HIPADJ_HD void fsal() const {}
};

// Component form of the joint VJP and of f on operands rotated through the quad: y1 = y_{c+1}, y2 = y_{c+2}, l1 = lam_{c+1}, l2 = lam_{c+2} (indices mod 3).
template <class Mo> struct QuadAdj { static constexpr bool value = false; };
// Lorenz-63, (df/du)^T = [-s  r-z  y ;  s  -1  x ;  0  -x  -b],  (df/dp)^T lam = ((y - x) lam_0, x lam_1, -z lam_2):
//   (J^T lam)_c = A lam_c + (B0 + B2 y2) l1 + (C0 + C1 y1) l2        c = 0: -s, r - z, y     c = 1: -1, x, s      c = 2: -b, 0, -x
//   (f_p^T lam)_c = (M0 y_c + M1 y1 + M2 y2) lam_c                   c = 0: y - x            c = 1: x             c = 2: -z
//   f_c = (S1 y1 + S2 y2) (Q0 + Q1 y1 + Q2 y2) + D y_c               c = 0: s y1 - s y_c     c = 1: y2 (r - y1) - y_c    c = 2: y1 y2 - b y_c
template <> struct QuadAdj<ModelLorenz> {
    static constexpr bool value = true;
    struct K { double A, B0, B2, C0, C1, M0, M1, M2, S1, S2, Q0, Q1, Q2, D; };
    HIPADJ_HD static K consts(int c, const double (&p)[3]) {
        K k = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (c == 0) { k.A = -p[0]; k.B0 = p[1]; k.B2 = -1.0; k.C1 = 1.0; k.M0 = -1.0; k.M1 = 1.0; k.S1 = 1.0; k.Q0 = p[0]; k.D = -p[0]; }
        else if (c == 1) { k.A = -1.0; k.B2 = 1.0; k.C0 = p[0]; k.M2 = 1.0; k.S2 = 1.0; k.Q0 = p[1]; k.Q1 = -1.0; k.D = -1.0; }
        else if (c == 2) { k.A = -p[2]; k.C1 = -1.0; k.M0 = -1.0; k.S1 = 1.0; k.Q2 = 1.0; k.D = -p[2]; }
        return k;
    }
    static constexpr int R1 = HIPADJ_QP(1, 2, 0, 3), R2 = HIPADJ_QP(2, 0, 1, 3);
    // un-negated products at (y_c, lam_c); WP: also the parameter part
    template <bool WP> HIPADJ_HD static void vjp(const K& k, double yc, double lc, double, double& dl, double& dm) {
        const double y1 = quad_perm<R1>(yc), y2 = quad_perm<R2>(yc), l1 = quad_perm<R1>(lc), l2 = quad_perm<R2>(lc);
        const double b = fma(k.B2, y2, k.B0), cc = fma(k.C1, y1, k.C0);
        dl = fma(cc, l2, fma(b, l1, k.A * lc));
        if (WP) dm = fma(k.M2, y2, fma(k.M1, y1, k.M0 * yc)) * lc; else dm = 0.0;
    }
    // ... and f_c next to them (Backsolve integrates y backwards): one more pair of rotations is not needed, y1 / y2 are shared
    HIPADJ_HD static void vjp_f(const K& k, double yc, double lc, double, double& dl, double& dm, double& fc) {
        const double y1 = quad_perm<R1>(yc), y2 = quad_perm<R2>(yc), l1 = quad_perm<R1>(lc), l2 = quad_perm<R2>(lc);
        const double b = fma(k.B2, y2, k.B0), cc = fma(k.C1, y1, k.C0);
        dl = fma(cc, l2, fma(b, l1, k.A * lc));
        dm = fma(k.M2, y2, fma(k.M1, y1, k.M0 * yc)) * lc;
        fc = fma(fma(k.S2, y2, k.S1 * y1), fma(k.Q2, y2, fma(k.Q1, y1, k.Q0)), k.D * yc);
    }
};

// Lotka-Volterra (n = 2, np = 4: lanes 0, 1 own y_c, lam_c; all four lanes own mu_c), plain and time-dependent (`fb`, test/Core3/adjoint.jl:8-12: the product terms carry t).
// The four operands every lane needs are BROADCASTS of lanes 0 and 1 (X = y_0, Y = y_1, LX = lam_0, LY = lam_1); "the other component" is picked by constants:
//   (J^T lam)_c = (A + t B yo) lam_c + (t C yo) lo,   yo = G0 X + G1 Y, lo = G0 LX + G1 LY       c = 0: p1, -p2, p4, other = 1      c = 1: -p3, p4, -p2, other = 0
//   (f_p^T lam)_c = (E1 X + E2 Y + t E3 X Y)(F1 LX + F2 LY)                                       x lam_0,  -x y t lam_0,  -y lam_1,  x y t lam_1
//   f_c = y_c (A + t B yo)
template <bool TD> struct QuadAdjLV {
    static constexpr bool value = true;
    struct K { double A, B, C, G0, G1, E1, E2, E3, F1, F2; };
    HIPADJ_HD static K consts(int c, const double (&p)[4]) {
        K k = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (c == 0) { k.A = p[0]; k.B = -p[1]; k.C = p[3]; k.G1 = 1.0; k.E1 = 1.0; k.F1 = 1.0; }
        else if (c == 1) { k.A = -p[2]; k.B = p[3]; k.C = -p[1]; k.G0 = 1.0; k.E3 = -1.0; k.F1 = 1.0; }
        else if (c == 2) { k.E2 = -1.0; k.F2 = 1.0; }
        else { k.E3 = 1.0; k.F2 = 1.0; }
        return k;
    }
    static constexpr int B0 = HIPADJ_QP(0, 0, 0, 0), B1 = HIPADJ_QP(1, 1, 1, 1);
    template <bool WP> HIPADJ_HD static void vjp(const K& k, double yc, double lc, double t, double& dl, double& dm) {
        const double X = quad_perm<B0>(yc), Y = quad_perm<B1>(yc), LX = quad_perm<B0>(lc), LY = quad_perm<B1>(lc);
        const double tt = TD ? t : 1.0;
        const double yo = fma(k.G1, Y, k.G0 * X), lo = fma(k.G1, LY, k.G0 * LX);
        dl = fma(k.C * tt * yo, lo, fma(k.B * tt, yo, k.A) * lc);
        if (WP) dm = fma(k.E3 * tt, X * Y, fma(k.E2, Y, k.E1 * X)) * fma(k.F2, LY, k.F1 * LX); else dm = 0.0;
    }
    HIPADJ_HD static void vjp_f(const K& k, double yc, double lc, double t, double& dl, double& dm, double& fc) {
        const double X = quad_perm<B0>(yc), Y = quad_perm<B1>(yc), LX = quad_perm<B0>(lc), LY = quad_perm<B1>(lc);
        const double tt = TD ? t : 1.0;
        const double yo = fma(k.G1, Y, k.G0 * X), lo = fma(k.G1, LY, k.G0 * LX);
        const double g = fma(k.B * tt, yo, k.A);
        dl = fma(k.C * tt * yo, lo, g * lc);
        dm = fma(k.E3 * tt, X * Y, fma(k.E2, Y, k.E1 * X)) * fma(k.F2, LY, k.F1 * LX);
        fc = yc * g;
    }
};
template <> struct QuadAdj<ModelLV> : QuadAdjLV<false> {};
template <> struct QuadAdj<ModelLVT> : QuadAdjLV<true> {};

// ---- forward dense solve (the quad counterpart of forward_tsit5_lane): lane (i, c) integrates component c ------------------------------------------------
template <class Mo>
HIPADJ_HD void forward_tsit5_quad(const AdaptGeom& g, long i, int c, const double* __restrict__ u0, const double* __restrict__ p,
                                  double* __restrict__ rec, int* __restrict__ nsteps, const double* __restrict__ save_t,
                                  double* __restrict__ outT, const double* __restrict__ ck_t, double* __restrict__ ckpt,
                                  double* __restrict__ yT, int* __restrict__ flag) {
    constexpr int N = Mo::N, RW = 2 + 5 * N;
    using Q = QuadForm<Mo>;
    const bool own = c < N;
    const int cc = own ? c : 0;
    double pv[Mo::NP];
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) pv[j] = own ? (g.p_shared ? p[j] : p[i * Mo::NP + j]) : 0.0;     // a spare lane: zero constants, zero state
    const typename Q::K kc = Q::consts(cc, pv);
    KRegs<1> K;
    double u[1] = {own ? u0[i * N + c] : 0.0};
    int s = 0, ms = 0, mc = 0;
    bool overflow = false;
    while (outT && ms < g.M && save_t[ms] <= g.t0) { if (own) outT[((long)ms * N + c) * g.Npad + i] = u[0]; ++ms; }
    while (ckpt && mc < g.nck && ck_t[mc] <= g.t0) { if (own) ckpt[((long)mc * N + c) * g.Npad + i] = u[0]; ++mc; }
    const double TINF = 1.7976931348623157e308;
    double ts_next = (outT && ms < g.M) ? save_t[ms] : TINF, tc_next = (ckpt && mc < g.nck) ? ck_t[mc] : TINF;
    const int na = tsit5_integrate<1>(u, g.t0, g.t1, g.dt0, g.abstol, g.reltol, nullptr, 0, false, g.maxit, K,
        [&](double (&du)[1], const double (&uu)[1], double t) { du[0] = Q::f(kc, uu[0], t); },
        [&](double t, double tprev, double (&un)[1], const auto& KK) -> bool {
            const double h = t - tprev;
            double cf[5][1]; tsit5_poly<1>(KK, h, cf);
            if (s < g.Smax) {
                if (rec && own) {
                    if (c == 0) { rec[((long)s * RW + 0) * g.Npad + i] = tprev; rec[((long)s * RW + 1) * g.Npad + i] = t; }
#pragma unroll
                    for (int m = 0; m < 5; ++m) rec[((long)s * RW + 2 + m * N + c) * g.Npad + i] = cf[m][0];
                }
            } else overflow = true;
            ++s;
            while (ts_next <= t || time_hits(ts_next, t)) {
                double y[1]; poly_eval<1>((ts_next - tprev) / h, cf, y);
                if (own) outT[((long)ms * N + c) * g.Npad + i] = y[0];
                ++ms; ts_next = ms < g.M ? save_t[ms] : TINF; }
            while (tc_next <= t || time_hits(tc_next, t)) {
                double y[1]; poly_eval<1>((tc_next - tprev) / h, cf, y);
                if (own) ckpt[((long)mc * N + c) * g.Npad + i] = y[0];
                ++mc; tc_next = mc < g.nck ? ck_t[mc] : TINF; }
            (void)un;
            return false;
        }, NoPre(), QuadNorm{N});
    if (c == 0) nsteps[i] = s;
    if (yT && own) yT[(long)c * g.Npad + i] = u[0];
    if ((na < 0 || overflow) && c == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(flag, 4);
#endif
    }
}

// one component of the forward dense solution: the step containing t, this lane's five coefficients cached in registers (FwdCursor with one component).
// The record BELOW the current one is requested when the cursor arrives on a record and consumed when it walks down: the sweep changes record ~3 times per attempt
// of a wavefront (16 trajectories, each at its own abscissa), and a change used to cost two dependent round trips (the end point, then the coefficients) — 37 % of the
// wavefront's cycles were s_waitcnt (profiles/r4_tsit5_quad_counters.txt).  Seven extra registers per lane here; the lane family's 5 n + 2 did not pay (hipadj_adaptive.hpp).
// A SECOND look-ahead level (record sc - 2 as well) measured slower: Interpolating 1.19 -> 1.32 ms, Gauss 1.76 -> 2.02 (the copies and the wider live range cost more than the
// remaining wait).
template <class Mo> struct QuadCursor {
    static constexpr int N = Mo::N, RW = 2 + 5 * Mo::N;
    const double* rec; long Npad, i; int ns, sc, c; bool own;
    double ta, tb, cf[5];
    double pta, pcf[5];            // record sc - 1 (requested; valid when sc > 0)
    HIPADJ_HD void fetch_coef(double (&d)[5], int s) const {
        const long base = ((long)s * RW + 2 + c) * Npad + i;
#pragma unroll
        for (int m = 0; m < 5; ++m) d[m] = own ? rec[base + (long)(m * N) * Npad] : 0.0;
    }
    HIPADJ_HD void fetch_below() {
        if (sc > 0) { pta = rec[((long)(sc - 1) * RW + 0) * Npad + i]; fetch_coef(pcf, sc - 1); }
    }
    HIPADJ_HD void init(const double* r, long np, long ii, int nsteps, int comp, bool o) {
        rec = r; Npad = np; i = ii; ns = nsteps; sc = nsteps - 1; c = comp; own = o;
        ta = rec[((long)sc * RW + 0) * Npad + i]; tb = rec[((long)sc * RW + 1) * Npad + i];
        fetch_coef(cf, sc);
        pta = ta;
#pragma unroll
        for (int m = 0; m < 5; ++m) pcf[m] = 0.0;
        fetch_below();
    }
    HIPADJ_HD double eval(double t) {
        if (t < ta && sc > 0) {
            --sc; tb = ta; ta = pta;                       // one record down: requested when the cursor arrived on the record above
#pragma unroll
            for (int m = 0; m < 5; ++m) cf[m] = pcf[m];
            if (t < ta && sc > 0) {                         // further down (a step longer than a forward step): walk, then the coefficients
                while (t < ta && sc > 0) { --sc; tb = ta; ta = rec[((long)sc * RW + 0) * Npad + i]; }
                fetch_coef(cf, sc);
            }
            fetch_below();
        } else if (t > tb && sc < ns - 1) {                 // up: after a rejected attempt
            while (t > tb && sc < ns - 1) { ++sc; ta = tb; tb = rec[((long)sc * RW + 1) * Npad + i]; }
            fetch_coef(cf, sc);
            fetch_below();
        }
        const double th = (t - ta) / (tb - ta);
        return cf[0] + th * (cf[1] + th * (cf[2] + th * (cf[3] + th * cf[4])));
    }
};

// ---- reverse sweeps.  ALG: 0 Interpolating (lane: lam_c, mu_c), 1 Backsolve (lam_c, mu_c, y_c), 2 Gauss (lam_c; mu_c by 3-node Gauss-Legendre per step) ----
// Gauss: lam_c plus a DUMMY second component with zero derivative.  The one-component instantiation of tsit5_integrate returned a wrong lam from this kernel — deterministic,
// growing with the step count on problems with interior loss times (1.8e-5 at tol 1e-11; scripts/r4/ts5_dbg2.py), identical with or without the quadrature nodes — while the
// SAME source with the dummy component agrees with the oracle at 7e-14 (and the forward solve's one-component instantiation is right, and so is the lane family's with the rows in
// LDS).  The SOURCE is right: compiled for the host with one component (tests/emu/quad_emu.cpp: four threads per quad in lockstep, -DHIPADJ_QUAD_GAUSS_NZ=1) it agrees with
// the oracle at 1e-14 and with the two-component build bit for bit (tests/test_quad_emu.py) — the defect is in the device code generation of that instantiation.  The
// dummy adds exact zeros to the controller's norms (ncomp stays n) and two instructions per stage.
#ifndef HIPADJ_QUAD_GAUSS_NZ
#define HIPADJ_QUAD_GAUSS_NZ 2      // 1 = the one-component instantiation (tests/emu/quad_emu.cpp builds it on the host to tell a source defect from a code-generation one)
#endif
template <int ALG> struct QuadNZ { static constexpr int value = ALG == 1 ? 3 : (ALG == 2 ? HIPADJ_QUAD_GAUSS_NZ : 2); };

template <class Mo, int ALG>
HIPADJ_HD void adjoint_tsit5_quad(const AdaptGeom& g, long i, int c, const double* __restrict__ p, const double* __restrict__ rec,
                                  const int* __restrict__ nsteps, const double* __restrict__ yT, const double* __restrict__ ckpt,
                                  const double* __restrict__ ck_t, const double* __restrict__ save_t, const double* __restrict__ tstops_desc,
                                  int ntstops, const double* __restrict__ cotT, double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag) {
    constexpr int N = Mo::N, NP = Mo::NP, NZ = QuadNZ<ALG>::value;
    static_assert(N <= 4 && NP <= 4, "one component of each vector per lane of a quad");
    using QA = QuadAdj<Mo>;
    const bool ownl = c < N, ownm = c < NP;
    double pv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) pv[j] = (ownl || ownm) ? (g.p_shared ? p[j] : p[i * NP + j]) : 0.0;
    const typename QA::K kc = QA::consts((ownl || ownm) ? c : 3, pv);
    KRegs<NZ> K;
    QuadCursor<Mo> cur;
    if (ALG != 1) cur.init(rec, g.Npad, i, nsteps[i] < g.Smax ? nsteps[i] : g.Smax, ownl ? c : 0, ownl);
    QuadCursor<Mo>& cur2 = cur;
    double z[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) z[j] = 0.0;
    if constexpr (ALG == 1) z[2] = ownl ? yT[(long)c * g.Npad + i] : 0.0;
    double gacc = 0.0;
    int cur_time = g.M, bs_cur = g.nck;
    double t_loss = g.M > 0 ? save_t[g.M - 1] : 0.0;
    if (ALG == 1 && bs_cur >= 1 && time_hits(g.t1, ck_t[bs_cur - 1])) --bs_cur;
    double t_ck = (ALG == 1 && ckpt && bs_cur >= 1) ? ck_t[bs_cur - 1] : 0.0;

    auto rhs = [&](double (&dz)[NZ], const double (&zz)[NZ], double t) {
        double dl, dm;
        if constexpr (ALG == 1) {
            double fc;
            QA::vjp_f(kc, zz[2], zz[0], t, dl, dm, fc);
            dz[0] = -dl; dz[1] = -dm; dz[2] = fc;
        } else {
            const double yc = cur.eval(t);
            QA::template vjp<ALG == 0>(kc, yc, zz[0], t, dl, dm);
            dz[0] = -dl;
            if constexpr (ALG == 0) dz[1] = -dm;
            else if constexpr (NZ > 1) dz[1] = 0.0;            // Gauss: the dummy component (QuadNZ)
        }
    };
    auto cb = [&](double t, double tprev, double (&zz)[NZ], const auto& KK) -> bool {
        bool mod = false;
        if (ALG == 2 && t != tprev) {   // IntegratingSumCallback: 3-point Gauss-Legendre of -(df/dp)^T lam on [tprev, t], this lane's parameter
            const double half = 0.5 * (t - tprev), mid = 0.5 * (t + tprev), h = t - tprev;
            // lam at the three nodes from the MONOMIAL form of the step's continuous extension (tsit5_poly, the form the forward records use), built once per step for this
            // lane's component: 25 instructions + 4 per node instead of the seven b_j(theta) polynomials (~40) and a division per node; theta_q = (1 + x_q) / 2 is a constant
            double c1 = h * KK.get(0, 0), c2 = 0.0, c3 = 0.0, c4 = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) { const double kj = KK.get(j, 0); c2 = fma(TS5::r(j, 1), kj, c2); c3 = fma(TS5::r(j, 2), kj, c3); c4 = fma(TS5::r(j, 3), kj, c4); }
            c2 *= h; c3 *= h; c4 *= h;
            const double c0 = KK.get(KS_UPREV, 0);
#pragma unroll 1
            for (int q = 0; q < 3; ++q) {
                const double xq = q == 0 ? -0.7745966692414833770 : (q == 1 ? 0.0 : 0.7745966692414833770);
                const double wq = q == 1 ? 8.0 / 9.0 : 5.0 / 9.0;
                const double tt = half * xq + mid, th = fma(0.5, xq, 0.5);
                const double lamq = fma(th, fma(th, fma(th, fma(th, c4, c3), c2), c1), c0);
                const double yc = cur2.eval(tt);
                double dl, dm;
                QA::template vjp<true>(kc, yc, lamq, tt, dl, dm);
                gacc += half * wq * (-dm);
            }
        }
        if (ALG == 1 && ckpt && bs_cur >= 1 && time_hits(t, t_ck)) {               // backsolve_checkpoint_callbacks
            zz[NZ - 1] = ownl ? ckpt[((long)(bs_cur - 1) * N + c) * g.Npad + i] : 0.0;
            --bs_cur; mod = true;
            t_ck = bs_cur >= 1 ? ck_t[bs_cur - 1] : 0.0;
        }
        if (cur_time >= 1 && time_hits(t, t_loss)) {                                  // ReverseLossCallback
            if (!(g.no_start && ALG != 1 && cur_time == 1)) {
                double yc;
                if (ALG == 1) yc = zz[NZ - 1]; else yc = cur.eval(t);
                if (ownl) zz[0] += (g.loss_kind == 1) ? (yc - g.loss_shift) : __builtin_fma(g.la, yc, g.lb * cotT[((long)(cur_time - 1) * N + c) * g.Npad + i]);
                mod = true;
            }
            --cur_time;
            t_loss = cur_time >= 1 ? save_t[cur_time - 1] : 0.0;
        }
        return mod;
    };
    const bool cb_at_init = g.M > 0 && time_hits(g.t1, save_t[g.M - 1]);
    const int ncomp = ALG == 0 ? N + NP : (ALG == 1 ? 2 * N + NP : N);
    const int na = tsit5_integrate<NZ>(z, g.t1, g.t0, g.dt0, g.abstol, g.reltol, tstops_desc, ntstops, cb_at_init, 8 * g.maxit, K, rhs, cb, NoPre(), QuadNorm{ncomp});
    if (ownl) du0[i * N + c] = z[0];
    if (ownm) dp_traj[(long)c * g.Npad + i] = (ALG == 2) ? gacc : z[NZ > 1 ? 1 : 0];
    if (na < 0 && c == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(flag, 4);
#endif
    }
}

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
template <class Mo>
__global__ void __launch_bounds__(64) k_forward_tsit5_quad(AdaptGeom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                           double* __restrict__ rec, int* __restrict__ nsteps, const double* __restrict__ save_t,
                                                           double* __restrict__ outT, const double* __restrict__ ck_t, double* __restrict__ ckpt,
                                                           double* __restrict__ yT, int* __restrict__ flag) {
    const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 2);
    if (i >= g.N) return;                      // whole quads leave together
    if constexpr (QuadForm<Mo>::value) forward_tsit5_quad<Mo>(g, i, threadIdx.x & 3, u0, p, rec, nsteps, save_t, outT, ck_t, ckpt, yT, flag);
}
template <class Mo, int ALG>
__global__ void __launch_bounds__(64) k_adjoint_tsit5_quad(AdaptGeom g, const double* __restrict__ p, const double* __restrict__ rec,
                                                           const int* __restrict__ nsteps, const double* __restrict__ yT, const double* __restrict__ ckpt,
                                                           const double* __restrict__ ck_t, const double* __restrict__ save_t,
                                                           const double* __restrict__ tstops_desc, int ntstops, const double* __restrict__ cotT,
                                                           double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag) {
    const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 2);
    if (i >= g.N) return;
    if constexpr (QuadAdj<Mo>::value) adjoint_tsit5_quad<Mo, ALG>(g, i, threadIdx.x & 3, p, rec, nsteps, yT, ckpt, ck_t, save_t, tstops_desc, ntstops, cotT, du0, dp_traj, flag);
}
#endif

}  // namespace hipadj
