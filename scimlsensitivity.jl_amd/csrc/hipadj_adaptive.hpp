// hipadj_adaptive.hpp — adaptive Tsit5 (5(4) pair, PI step control, own 4th-order interpolant) for the
// lane-per-trajectory family: per-lane step control, forward dense output in HBM, reverse adjoint sweeps that
// interpolate the forward solution at arbitrary times.  This is the stepper every reference test uses
// (`Tsit5()`, test/Core3/adjoint.jl:31-43, 1167; test/Core1/concrete_solve_derivatives.jl) and SURVEY.md §8f rank 1.
//
// What is restated (reference = SciMLSensitivity.jl; the stepper itself lives in OrdinaryDiffEq [upstream-recall]):
//   forward dense solve, out = sol(ts) by interpolation            src/concrete_solve.jl:701-727
//   Interpolating RHS with y = sol(t) at arbitrary t               src/interpolating_adjoint.jl:150-174, 190-204
//   Backsolve RHS + checkpoint / loss callbacks at tstops          src/backsolve_adjoint.jl:32-61, 523-546; src/adjoint_common.jl:754-821
//   Gauss: lambda-only RHS + IntegratingSumCallback with div(order+1, 2) = 3 Gauss-Legendre nodes per accepted step
//                                                                  src/gauss_adjoint.jl:118-128, 745-759, 809-851
//   reverse solve reuses alg and tolerances, tstops = loss times   src/sensitivity_interface.jl:484-491
// The controller arithmetic follows oracle/adjoint_oracle.c `integrate` line by line (Hairer initial step, error norm
// RMS of err/(abstol + max(|u0|,|u1|) reltol), PI exponents 7/50 and 2/25, gamma 0.9, q in [1/10, 5], tstop clipping
// with a 100-eps snap), so that device and oracle take the same step sequences up to roundoff.
//
// Register / LDS plan (gfx950).  The seven stage derivatives of a step plus the step's start value (8 x NZ doubles per
// lane) live in LDS as lane-private columns  ks[(row * NZ + i) * 64 + lane]  (consecutive lanes -> consecutive 8-byte
// words: conflict-free ds_read_b64/ds_write_b64), NOT in VGPRs.  That lets the stage loop stay ROLLED: one instance of
// the right-hand side (and of the forward-solution cursor inside it) instead of seven, tableau coefficients fetched with
// scalar loads instead of ~60 FP64 literals parked in SGPR pairs.  The earlier fully unrolled register version needed
// 256 VGPRs + up to 256 AGPRs + scratch for NZ >= 9 and produced wrong step sequences on the device for one size
// (n = 4 Backsolve) although the same source is exact on the host; the rolled version needs far fewer registers.
//
// Layout of the forward dense solution (per trajectory a ragged list of accepted steps, trajectory-minor):
//   rec[(s * RW + w) * Npad + i],  RW = 2 + 5 n :  w = 0 t_start, 1 t_end, then c_0..c_4 [5][n], the monomial form of the
//   Tsit5 continuous extension on the step:  y(theta) = c_0 + theta (c_1 + theta (c_2 + theta (c_3 + theta c_4))),
//   c_0 = u_start, c_1 = h k_1, c_m = h sum_j r_{j,m-1} k_j.  The 7 stage derivatives are folded into 4 coefficient
//   vectors when the step is accepted: 35 % fewer bytes per record than (u, k_1..k_7) and a Horner evaluation (4 n FMAs)
//   per reverse-pass stage instead of the 7-term b_j(theta) sum.
//   nsteps[i] accepted steps (the true count; beyond Smax the records are not written and flag bit 4 => HIPADJ_ERR_MAXITERS,
//   or, with an auto-sized capacity, the host regrows the buffers and repeats the pass)
#pragma once

#include "hipadj_lane.hpp"

namespace hipadj {

struct AdaptGeom {
    long N, Npad;
    int M, Smax, nck, SmaxI;   // Smax: record capacity; SmaxI: capacity of ONE checkpoint interval (checkpointing=true for Interpolating/Gauss)
    int maxit;                 // bound on the accepted steps of the forward solve (>= Smax; = Smax unless the capacity is auto-sized)
    double t0, t1, dt0, abstol, reltol;
    double loss_shift;
    int loss_kind, no_start, p_shared, cont_cost;   // loss_kind: hipadj_loss (0 cotangent, 1 lsq_shift, 2 lsq_data, 3 the model's discrete-loss bodies)
    double la, lb; int lflags;                      // as Geom: dgdu = la u + lb c for the kinds that stream a column; bit 0 of lflags drops dgdp_discrete
    int SmaxA;                 // QuadratureAdjoint: capacity (steps) of the dense ADJOINT record; its readers clamp the stored TRUE step count with it
    // ContinuousCallback (model_has_cond): per trajectory, the index of the first forward record AFTER each event (that record starts at the event time, from the affected
    // state), ascending in time — ev_s [maxev][Npad] — and the TRUE number of events — nev [Npad]; maxev = 0: no event handling
    // ... the event times ev_t [maxev][Npad] and the states just BEFORE the affect ev_ul [maxev][n][Npad] (BacksolveAdjoint keeps no forward record: at an event its
    // backsolved state is overwritten with the stored left state, as at a checkpoint — src/callback_tracking.jl:377 copy_to_integrator!)
    int maxev = 0; int* ev_s = nullptr; int* nev = nullptr; double* ev_t = nullptr; double* ev_ul = nullptr;
    // save_positions = (true, true): the states just AFTER the affect, ev_ur [maxev][n][Npad] (hipadj_event_states hands t, u-, u+ to the caller), and the caller's cotangents of
    // a loss on the saved event states, ev_dl / ev_dr [maxev][n][Npad] (hipadj_set_event_cotangents; nullptr = none)
    double* ev_ur = nullptr; const double* ev_dl = nullptr; const double* ev_dr = nullptr;
    int* ev_k = nullptr;       // which component of a VectorContinuousCallback fired, [maxev][Npad] (0 for a scalar condition)
};

// Tsit5 coefficients (Tsitouras 2011); same values as oracle/adjoint_oracle.c (order conditions checked there).
// Tables are indexed at run time (uniform indices -> scalar loads on the device).
struct TS5 {
    static HIPADJ_HD double c(int i) { const double v[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0}; return v[i]; }
    static HIPADJ_HD double a(int i, int j) {
        const double v[7][6] = {
            {0, 0, 0, 0, 0, 0},
            {0.161, 0, 0, 0, 0, 0},
            {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
            {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
            {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
            {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
            {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
        return v[i][j];
    }
    static HIPADJ_HD double bt(int i) {
        const double v[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                             0.5823571654525552, -0.45808210592918697, 0.015151515151515152};
        return v[i];
    }
    // b_j(theta) = theta (r_j0 + theta (r_j1 + theta (r_j2 + theta r_j3))); r_j0 = 0 for j >= 1
    static HIPADJ_HD double r(int j, int m) {
        const double v[7][4] = {
            {1.0, -2.763706197274826, 2.9132554618219126, -1.0530884977290216},
            {0.0, 0.13169999999999998, -0.2234, 0.1017},
            {0.0, 3.9302962368947516, -5.941033872131505, 2.490627285651253},
            {0.0, -12.411077166933676, 30.33818863028232, -16.548102889244902},
            {0.0, 37.50931341651104, -88.1789048947664, 47.37952196281928},
            {0.0, -27.896526289197286, 65.09189467479366, -34.87065786149661},
            {0.0, 1.5, -4.0, 2.5}};
        return v[j][m];
    }
    static HIPADJ_HD double b(int j, double th) {
        return j == 0 ? th * (r(0, 0) + th * (r(0, 1) + th * (r(0, 2) + th * r(0, 3)))) : th * th * (r(j, 1) + th * (r(j, 2) + th * r(j, 3)));
    }
};

HIPADJ_HD double hmax2(double a, double b) { return a > b ? a : b; }
HIPADJ_HD double hmin2(double a, double b) { return a < b ? a : b; }
HIPADJ_HD double habs(double a) { return a < 0 ? -a : a; }
HIPADJ_HD bool time_hits(double t, double target) { return habs(t - target) <= 100.0 * 2.220446049250313e-16 * hmax2(habs(t), habs(target)); }

// log and exp of the step-size controller.  The factor EEst^(7/50) / qold^(2/25) costs two logarithms, two exponentials and a division per attempt when written with the math
// library (~400 instructions: a third of a Lorenz attempt of the quad sweep, profiles/r4_tsit5_quad_counters.txt); here it is ONE logarithm (log qold is the previous accepted
// attempt's, kept) and ONE exponential of the difference, both inlined at double accuracy (a few ulp: the accept / reject decision itself never sees them, only the NEXT step's
// length does, exactly as with the round-2 exp(c log x) form of pow).  log: x = 2^e m, m in [sqrt(1/2), sqrt 2), log m = 2 atanh s, s = (m - 1) / (m + 1), |s| <= 0.172, odd
// series to s^19; exp: r = y - k ln 2 (two-part), Taylor to r^12, ldexp.  The host pass (emulator) calls the library.
HIPADJ_HD double ts5_log(double x) {      // x finite, > 0, normal
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m; e = low ? e - 1 : e;
    const double a = m - 1.0, b = m + 1.0;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    double sq = a * r;
    sq = __builtin_fma(__builtin_fma(-b, sq, a), r, sq);
    const double z = sq * sq;
    double P = 1.0 / 19.0;
    P = __builtin_fma(P, z, 1.0 / 17.0); P = __builtin_fma(P, z, 1.0 / 15.0); P = __builtin_fma(P, z, 1.0 / 13.0); P = __builtin_fma(P, z, 1.0 / 11.0);
    P = __builtin_fma(P, z, 1.0 / 9.0); P = __builtin_fma(P, z, 1.0 / 7.0); P = __builtin_fma(P, z, 1.0 / 5.0); P = __builtin_fma(P, z, 1.0 / 3.0);
    const double s2 = sq + sq;
    const double lm = __builtin_fma(s2 * z, P, s2);
    return __builtin_fma((double)e, 6.93147180559945286e-01, lm);
#else
    return log(x);
#endif
}
HIPADJ_HD double ts5_exp(double y) {      // |y| < 700
#if defined(__HIP_DEVICE_COMPILE__)
    const double kf = __builtin_rint(y * 1.44269504088896339);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, y);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double q = 1.0 / 479001600.0;
    q = __builtin_fma(q, r, 1.0 / 39916800.0); q = __builtin_fma(q, r, 1.0 / 3628800.0); q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0); q = __builtin_fma(q, r, 1.0 / 5040.0); q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0); q = __builtin_fma(q, r, 1.0 / 24.0); q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5); q = __builtin_fma(q, r, 1.0); q = __builtin_fma(q, r, 1.0);
    return __builtin_amdgcn_ldexp(q, (int)kf);
#else
    return exp(y);
#endif
}

// Stage storage of one lane: rows 0..6 = k_1..k_7 of the current step, row 7 = the step's start value.
// Device: base = LDS array + lane, stride = 64.  Host emulation: a local array, stride 1.
constexpr int KS_ROWS = 8, KS_UPREV = 7;
#ifndef HIPADJ_TS5_WIDE
#define HIPADJ_TS5_WIDE 9
#endif
#ifndef HIPADJ_TS5_PADDED
#define HIPADJ_TS5_PADDED 1
// A/B on one box, round 3 (profiles/r3_tsit5_stage_sum_ab.log): the zero-padded 6-term stage sum is 10-12 % FASTER than the sum with exactly its
// terms behind a switch (Lorenz reverse 2.21 vs 2.48 ms, LV 0.52 vs 0.58): the sweep is bound by one wave's dependent chain, not by its FMA count
#endif
constexpr int TS5_WIDE = HIPADJ_TS5_WIDE;   // widest state vector whose six stage rows are summed in one batch (tsit5_integrate)
template <int NZ> struct KStore {
    static constexpr bool IN_REGS = false, ROLLED = true;
    double* base; int stride;
    HIPADJ_HD double get(int row, int i) const { return base[(row * NZ + i) * stride]; }
    HIPADJ_HD void set(int row, int i, double v) const { base[(row * NZ + i) * stride] = v; }
};
// The same rows in registers (round 3): 8 NZ doubles per lane, every index a compile-time constant, so tsit5_integrate runs its stage loop
// unrolled — six instances of the right-hand side, each stage sum with exactly its terms and literal coefficients, no LDS round trip between a
// stage and the next.  With 157 lone waves at 10^4 trajectories the adaptive kernels are bound by one wave's instruction stream (2.96 ns per
// FP64 instruction), so the shorter stream is the whole point; 16 NZ VGPRs are affordable because a lone wave owns its SIMD's 512 registers.
// Compiled-in models with NZ <= TS5_WIDE only (ts5_in_regs): runtime models keep the LDS rows — their kernels come out of hiprtc per model, the rolled loop is
// the form that went through the compiler-defect history of DESIGN.md 6.8, and the one attempt to give them the register form (HIPADJ_TS5_REGS_USER, opt-in
// through the environment variable of the same name) produced a wrong kernel in the GPU suite (hipadj_user.hpp).  Same arithmetic, expression for expression
// (the padded sums of the LDS form add exact zeros), so step sequences are bit-identical; the host emulator runs this form for the same models.
template <int NZ> struct KRegs {
    static constexpr bool IN_REGS = true, ROLLED = false;
    double v[8][NZ];
    HIPADJ_HD double get(int row, int i) const { return v[row][i]; }
    HIPADJ_HD void set(int row, int i, double x) { v[row][i] = x; }
};
// Registers, but ONE instance of the right-hand side: the stage loop stays rolled and only the (cheap) stage sum and the row store sit behind a
// wave-uniform switch on the stage number.  For the workgroup-per-trajectory family (hipadj_wide.hpp), whose right-hand side is a whole model body
// between two barriers: six inlined copies of it would multiply the code of every runtime model.
template <int NZ> struct KRegsRolled : KRegs<NZ> { static constexpr bool ROLLED = true; };

// How the step controller's norms are summed.  Lane family: the lane holds the whole state — plain sums over NZ components.  Workgroup family: a thread
// holds its owned components, the sums run over the workgroup (hipadj_wide.hpp: WideNorm) and the divisor is the true number of components.
// `which` names the norm: 0, 1, 2 = the three sums of the Hairer-Norsett-Wanner initial step (u, f(u0), f(u1) - f(u0); h = the trial step h0), 3 = the error
// estimate of a step attempt (h = the step).  The hooks mark the points at which a policy that carries further state components OUTSIDE the integrated
// vector (the workgroup family's parameter gradient, hipadj_wide.hpp: WideAugNorm) has to act; they are empty for everything else.
struct TS5LaneNorm {
    HIPADJ_HD double sum(double x, int which, double h) const { (void)which; (void)h; return x; }
    HIPADJ_HD double count(int nz) const { return (double)nz; }
    HIPADJ_HD void begin_attempt() const {}      // top of a step attempt (k_1 is valid)
    HIPADJ_HD void after_k0() const {}           // k_1 = f(u, t) was just evaluated (first step, or after a callback changed u)
    HIPADJ_HD void after_stage(int s) const { (void)s; }   // stage s = 1..6 was just evaluated (row s holds k_{s+1})
    HIPADJ_HD void accept(double h) const { (void)h; }     // the attempt with step h is accepted: u <- u_{n+1}
    HIPADJ_HD void fsal() const {}               // k_1 <- k_7 (no callback intervened)
};
#ifndef HIPADJ_TS5_REGS
#define HIPADJ_TS5_REGS 1
#endif
template <class Mo, class = void> struct ts5_model_is_runtime { static constexpr bool value = false; };
template <class Mo> struct ts5_model_is_runtime<Mo, decltype((void)Mo::HAS_COLS)> { static constexpr bool value = true; };   // the struct hipadj_user.hpp generates
template <bool B, class A, class C> struct ts5_select { using type = A; };
template <class A, class C> struct ts5_select<false, A, C> { using type = C; };
template <class KS> HIPADJ_HD KS ts5_make_rows(double* base, int stride) {
    if constexpr (KS::IN_REGS) { (void)base; (void)stride; return KS(); } else return KS{base, stride};
}
#ifndef HIPADJ_TS5_REGS_USER
#define HIPADJ_TS5_REGS_USER 0   // runtime models: opt-in experiment (environment variable HIPADJ_TS5_REGS_USER=1, hipadj_user.hpp); off under every compiler
#endif
// (runtime models: up to 8 augmented components — at 9 the 4-state ring's kernels fill all 512 registers and start to use scratch)
template <class Mo, int NZ> struct ts5_in_regs {
    static constexpr bool value = HIPADJ_TS5_REGS && (ts5_model_is_runtime<Mo>::value ? (HIPADJ_TS5_REGS_USER && NZ <= 8) : NZ <= HIPADJ_TS5_WIDE);
};

// y = u_start + h sum_j b_j(theta) k_j : the continuous extension of the step held in K (used for the lambda values at the
// Gauss nodes; the forward solution is stored in monomial form instead, see tsit5_poly)
template <int NZ, int NOUT, class KS>
HIPADJ_HD void kstore_interp(const KS& K, double th, double h, double (&y)[NOUT]) {
#pragma unroll
    for (int i = 0; i < NOUT; ++i) y[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const double bj = TS5::b(j, th);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) y[i] += bj * K.get(j, i);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) y[i] = K.get(KS_UPREV, i) + h * y[i];
}

// monomial coefficients of the continuous extension of the accepted step held in K (see the layout note above)
template <int NZ, class KS>
HIPADJ_HD void tsit5_poly(const KS& K, double h, double (&c)[5][NZ]) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) { c[0][i] = K.get(KS_UPREV, i); c[1][i] = h * K.get(0, i); c[2][i] = 0.0; c[3][i] = 0.0; c[4][i] = 0.0; }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const double r1 = TS5::r(j, 1), r2 = TS5::r(j, 2), r3 = TS5::r(j, 3);
#pragma unroll
        for (int i = 0; i < NZ; ++i) { const double kj = K.get(j, i); c[2][i] += r1 * kj; c[3][i] += r2 * kj; c[4][i] += r3 * kj; }
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) { c[2][i] *= h; c[3][i] *= h; c[4][i] *= h; }
}
template <int NZ>
HIPADJ_HD void poly_eval(double th, const double (&c)[5][NZ], double (&y)[NZ]) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) y[i] = c[0][i] + th * (c[1][i] + th * (c[2][i] + th * (c[3][i] + th * c[4][i])));
}

// solve(prob, Tsit5(); abstol, reltol, dt, tstops, callback) for a small system.
//   rhs(du, u, t); cb(t, tprev, u, K) is called after every accepted step with the step's stages still in K (and once at
//   the start when cb_at_init) and returns true when it modified u (=> the FSAL derivative is recomputed,
//   derivative_discontinuity!).  tstops: ntstops times sorted along the integration direction.
// Returns the number of accepted steps, or -1 when max_steps was exceeded.
struct NoPre { HIPADJ_HD void operator()(double) const {} };

// w += sum_{j < S_} a(S_, j) K_j : the stage sum of stage S_ with exactly its S_ terms, in the oracle's order j = 0, 1, ...
template <int NZ, int S_, class KS>
HIPADJ_HD void tsit5_stage_sum(const KS& K, double (&w)[NZ]) {
#pragma unroll
    for (int j = 0; j < S_; ++j) {
        const double a = TS5::a(S_, j);
#pragma unroll
        for (int i = 0; i < NZ; ++i) w[i] += a * K.get(j, i);
    }
}

// pre(t) runs at the top of every step attempt, AFTER a pending k_1 = f(u, t) was evaluated: the checkpointed sweeps
// switch their interval solution there, so that every evaluation AT a checkpoint time still reads the interval above
// (as the reference's `t in interval` test does) and every stage below it reads the re-solved interval.
// one stage with the rows in registers: w = u_n + h sum_{j < S_} a(S_, j) k_j (the oracle's order), k_{S_+1} = rhs(w, t + c(S_) h)
template <int S_, int NZ, class KS, class Rhs>
HIPADJ_HD void tsit5_stage_regs(KS& K, double (&w)[NZ], double h, double t, Rhs& rhs) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) w[i] = 0.0;
    tsit5_stage_sum<NZ, S_>(K, w);
#pragma unroll
    for (int i = 0; i < NZ; ++i) w[i] = K.get(KS_UPREV, i) + h * w[i];
    double ks[NZ];
    rhs(ks, w, t + TS5::c(S_) * h);
#pragma unroll
    for (int i = 0; i < NZ; ++i) K.set(S_, i, ks[i]);
}

template <int NZ, class KS, class Rhs, class Cb, class Pre = NoPre, class Red = TS5LaneNorm>
HIPADJ_HD int tsit5_integrate(double (&u)[NZ], double tstart, double tend, double dt_hint, double abstol, double reltol,
                              const double* __restrict__ tstops, int ntstops, bool cb_at_init, int max_steps,
                              KS& K, Rhs&& rhs, Cb&& cb, Pre&& pre = NoPre(), Red red = Red(), double* dt_io = nullptr) {
    const double ncomp = red.count(NZ);
    const double EPS = 2.220446049250313e-16;
    const double tdir = tend >= tstart ? 1.0 : -1.0;
    double t = tstart, tprev = tstart;
    double w[NZ];
    if (cb_at_init) {
#pragma unroll
        for (int i = 0; i < NZ; ++i) K.set(KS_UPREV, i, u[i]);
        cb(t, tprev, u, K);
    }
    if constexpr (KS::IN_REGS) {
#pragma unroll
        for (int j = 1; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < NZ; ++i) K.set(j, i, 0.0);
    } else {
#pragma unroll 1
        for (int j = 1; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < NZ; ++i) K.set(j, i, 0.0);
    }
    bool need_k0 = true, first = true;
    double dt = 0.0, lqold = -9.21034037197618272;   // log qold, qold = 1e-4
    double dt_asked = 0.0; bool clipped = false;      // (the step the controller asked for before a stop clipped it: handed to the caller with the final proposal, dt_io)
    int its = 0, naccept = 0, guard = 0;
    double ts_cur = ntstops > 0 ? tstops[0] : tend;
#pragma unroll 1
    while (tdir * t < tdir * tend) {
        if (++guard > 16 * max_steps + 64) return -1;
        if (need_k0) {   // first step, or u was changed by a callback: k_1 = f(u, t)
            rhs(w, u, t);
#pragma unroll
            for (int i = 0; i < NZ; ++i) K.set(0, i, w[i]);
            red.after_k0();
            need_k0 = false;
        }
        if (first) {
            first = false;
            if (dt_hint > 0) dt = tdir * dt_hint;
            else {   // Hairer-Norsett-Wanner initial step; w still holds f(u0)
                double d0 = 0, d1 = 0; const double h0_unused = 0.0;
#pragma unroll
                for (int i = 0; i < NZ; ++i) { const double sc = abstol + habs(u[i]) * reltol; d0 += (u[i] / sc) * (u[i] / sc); d1 += (w[i] / sc) * (w[i] / sc); }
                d0 = sqrt(red.sum(d0, 0, h0_unused) / ncomp); d1 = sqrt(red.sum(d1, 1, h0_unused) / ncomp);
                double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                h0 = hmin2(h0, habs(tend - t));
                double u1[NZ], f1[NZ];
#pragma unroll
                for (int i = 0; i < NZ; ++i) u1[i] = u[i] + tdir * h0 * w[i];
                rhs(f1, u1, t + tdir * h0);
                double d2 = 0;
#pragma unroll
                for (int i = 0; i < NZ; ++i) { const double sc = abstol + habs(u[i]) * reltol; const double q = (f1[i] - K.get(0, i)) / sc; d2 += q * q; }
                d2 = sqrt(red.sum(d2, 2, h0) / ncomp) / h0;
                const double h1 = (hmax2(d1, d2) <= 1e-15) ? hmax2(1e-6, h0 * 1e-3) : pow(0.01 / hmax2(d1, d2), 1.0 / 5.0);
                dt = tdir * hmin2(hmin2(100.0 * h0, h1), habs(tend - t));
            }
        }
        pre(t);
        // next stop: the first tstop strictly ahead of t (beyond the 100-eps snap), else tend.  The candidate lives in a
        // register and is re-read only when the cursor moves: a dependent L2 round trip per attempt is a visible fraction
        // of a step when one wave owns the CU.
        while (its < ntstops && tdir * ts_cur <= tdir * t + 100.0 * EPS * hmax2(habs(t), habs(ts_cur))) { ++its; ts_cur = its < ntstops ? tstops[its] : tend; }
        double tstop = tend;
        if (its < ntstops && tdir * ts_cur < tdir * tend) tstop = ts_cur;
        double h = dt;
        dt_asked = dt; clipped = habs(h) > habs(tstop - t);
        if (clipped) h = tstop - t;
        if (habs((t + h) - tstop) < 100.0 * EPS * hmax2(habs(t + h), habs(tstop))) h = tstop - t;
#pragma unroll
        for (int i = 0; i < NZ; ++i) K.set(KS_UPREV, i, u[i]);
        red.begin_attempt();
        // Stage loop, rolled: ONE instance of rhs (and of the forward-solution cursor inside it).  The whole zero-padded
        // tableau row arrives in one batch of scalar loads, then a straight-line 6-term sum: rows j >= s of K hold finite
        // leftovers that the zero coefficients annihilate (non-finite leftovers are cleared when a step is rejected,
        // below).  Summation order j = 0, 1, ... is the oracle's.  (Measured alternative: forming the next stage's partial
        // sum next to rhs to hide the LDS round trip costs 9 x NZ extra FMAs per step and is 8 % slower — with one wave
        // per CU at N = 10^4 this loop is bound by instruction count, not by latency.)
        if constexpr (KS::IN_REGS && !KS::ROLLED) {
            // rows in registers: the stage loop unrolled, every sum with exactly its terms (see KRegs)
            tsit5_stage_regs<1>(K, w, h, t, rhs); tsit5_stage_regs<2>(K, w, h, t, rhs); tsit5_stage_regs<3>(K, w, h, t, rhs);
            tsit5_stage_regs<4>(K, w, h, t, rhs); tsit5_stage_regs<5>(K, w, h, t, rhs); tsit5_stage_regs<6>(K, w, h, t, rhs);
        } else if constexpr (KS::IN_REGS) {
            // rows in registers, one instance of rhs (KRegsRolled): the stage number is uniform, so the switches are scalar branches
#pragma unroll 1
            for (int s = 1; s < 7; ++s) {
#pragma unroll
                for (int i = 0; i < NZ; ++i) w[i] = 0.0;
                switch (s) {
                case 1: tsit5_stage_sum<NZ, 1>(K, w); break;
                case 2: tsit5_stage_sum<NZ, 2>(K, w); break;
                case 3: tsit5_stage_sum<NZ, 3>(K, w); break;
                case 4: tsit5_stage_sum<NZ, 4>(K, w); break;
                case 5: tsit5_stage_sum<NZ, 5>(K, w); break;
                default: tsit5_stage_sum<NZ, 6>(K, w); break;
                }
#pragma unroll
                for (int i = 0; i < NZ; ++i) w[i] = K.get(KS_UPREV, i) + h * w[i];
                double ks[NZ];
                rhs(ks, w, t + TS5::c(s) * h);
                switch (s) {
#define HIPADJ_TS5_SETROW(R) case R: { _Pragma("unroll") for (int i = 0; i < NZ; ++i) K.set(R, i, ks[i]); } break;
                HIPADJ_TS5_SETROW(1) HIPADJ_TS5_SETROW(2) HIPADJ_TS5_SETROW(3) HIPADJ_TS5_SETROW(4) HIPADJ_TS5_SETROW(5)
                default: { _Pragma("unroll") for (int i = 0; i < NZ; ++i) K.set(6, i, ks[i]); } break;
#undef HIPADJ_TS5_SETROW
                }
                red.after_stage(s);
            }
        } else {
#pragma unroll 1
        for (int s = 1; s < 7; ++s) {
#pragma unroll
            for (int i = 0; i < NZ; ++i) w[i] = 0.0;
            if constexpr (NZ <= TS5_WIDE && HIPADJ_TS5_PADDED) {
                // the zero-padded 6-term form (default): rows j >= s of K hold finite leftovers that the zero coefficients annihilate
                double as[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) as[j] = TS5::a(s, j);
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int i = 0; i < NZ; ++i) w[i] += as[j] * K.get(j, i);
            } else if constexpr (NZ <= TS5_WIDE) {
                // -DHIPADJ_TS5_PADDED=0: only the s rows the tableau really has, coefficients as literals (21 NZ instead of 36 NZ multiply-adds and LDS reads
                // per step) — measured SLOWER (see the macro), kept for A/B builds
                switch (s) {
                case 1: tsit5_stage_sum<NZ, 1>(K, w); break;
                case 2: tsit5_stage_sum<NZ, 2>(K, w); break;
                case 3: tsit5_stage_sum<NZ, 3>(K, w); break;
                case 4: tsit5_stage_sum<NZ, 4>(K, w); break;
                case 5: tsit5_stage_sum<NZ, 5>(K, w); break;
                default: tsit5_stage_sum<NZ, 6>(K, w); break;
                }
            } else {
                // wide state vectors (runtime models with n + np + n > TS5_WIDE): one stage row at a time.  Six rows in flight
                // are 12 NZ VGPRs on top of u, w and the rhs temporaries; past 256 the allocator spills inside this divergent
                // loop nest, and spilled kernels have produced wrong lanes on gfx950.
#pragma unroll 1
                for (int j = 0; j < 6; ++j) {
                    const double asj = TS5::a(s, j);
#pragma unroll
                    for (int i = 0; i < NZ; ++i) w[i] += asj * K.get(j, i);
                }
            }
#pragma unroll
            for (int i = 0; i < NZ; ++i) w[i] = K.get(KS_UPREV, i) + h * w[i];
            double ks[NZ];
            rhs(ks, w, t + TS5::c(s) * h);
#pragma unroll
            for (int i = 0; i < NZ; ++i) K.set(s, i, ks[i]);
            red.after_stage(s);
        }
        }
        // w = u_{n+1} (the seventh stage state is the 5th-order solution), K row 6 = f(u_{n+1}) (FSAL)
        double e2 = 0.0;
        {
            double err[NZ];
#pragma unroll
            for (int i = 0; i < NZ; ++i) err[i] = 0.0;
            if constexpr (NZ <= TS5_WIDE || KS::IN_REGS) {
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const double btj = TS5::bt(j);
#pragma unroll
                    for (int i = 0; i < NZ; ++i) err[i] += btj * K.get(j, i);
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < 7; ++j) {
                    const double btj = TS5::bt(j);
#pragma unroll
                    for (int i = 0; i < NZ; ++i) err[i] += btj * K.get(j, i);
                }
            }
#pragma unroll
            for (int i = 0; i < NZ; ++i) {
                const double sc = abstol + hmax2(habs(K.get(KS_UPREV, i)), habs(w[i])) * reltol;
                const double q = h * err[i] / sc;
                e2 += q * q;
            }
        }
        const double EEst = sqrt(red.sum(e2, 3, h) / ncomp);
        // q = EEst^(7/50) / qold^(2/25) as ONE exponential of (7/50) log EEst - (2/25) log qold, log qold kept from the attempt that set qold (ts5_log / ts5_exp above);
        // EEst^(7/50) alone is needed by a rejected attempt only
        const double lE = ts5_log(hmin2(hmax2(EEst, 1e-300), 1e300));
        double q = ts5_exp((7.0 / 50.0) * lE - (2.0 / 25.0) * lqold);
        q = hmax2(1.0 / 10.0, hmin2(5.0, q / 0.9));
        if (EEst <= 1.0 || habs(h) < 1e-14 * hmax2(1.0, habs(t))) {
            double tnew = t + h;
            if (habs(tnew - tstop) < 100.0 * EPS * hmax2(habs(tnew), habs(tstop))) tnew = tstop;
            lqold = hmax2(lE, -9.21034037197618272);     // qold = max(EEst, 1e-4)
#pragma unroll
            for (int i = 0; i < NZ; ++i) u[i] = w[i];
            red.accept(h);
            tprev = t; t = tnew; ++naccept;
            dt = h / q;
            if (habs(dt) < 1e-14 * hmax2(1.0, habs(tnew))) dt = tdir * 1e-14 * hmax2(1.0, habs(tnew));
            if (cb(t, tprev, u, K)) need_k0 = true;
            else {
#pragma unroll
                for (int i = 0; i < NZ; ++i) K.set(0, i, K.get(6, i));   // FSAL
                red.fsal();
            }
            if (naccept > max_steps) return -1;
        } else {
            dt = h / hmin2(5.0, ts5_exp((7.0 / 50.0) * lE) / 0.9);
            // the step's start value comes back from its LDS row: with both outcomes of the step test overwriting u, the register
            // copy of u is dead across the stage loop (2 NZ VGPRs less at the point of highest pressure; same values, bit for bit)
#pragma unroll
            for (int i = 0; i < NZ; ++i) u[i] = K.get(KS_UPREV, i);
            if constexpr (!KS::IN_REGS) {       // (the register form sums exactly its terms: no padding to protect)
            if (!(EEst < 1e300)) {            // overflowed stage derivatives must not meet the zero padding of the tableau rows
#pragma unroll 1
                for (int j = 1; j < 7; ++j)
#pragma unroll
                    for (int i = 0; i < NZ; ++i) K.set(j, i, 0.0);
            }
            }
        }
    }
    // dt_io: the controller's state for a solve that CONTINUES this one (the reverse pieces between the events of a ContinuousCallback): its last proposal — or, when the
    // last step was cut short by the end of the span, the larger of that and the step it had asked for
    if (dt_io) *dt_io = clipped ? hmax2(habs(dt), habs(dt_asked)) : habs(dt);
    return naccept;
}

// ------------------------------------------------------------------------------------------------------------
// Rosenbrock23 (round 6; VERDICT r5 missing 2 / next 8): the first stiff stepper of the lane family — Shampine & Reichelt's ode23s as OrdinaryDiffEq ships it
// [upstream-recall], the solver of /root/reference/test/Core2/stiff_adjoints.jl:53-75.  The adjoint runs with the forward solve's alg
// (src/sensitivity_interface.jl:487-491), so both directions use it: restated in oracle/adjoint_oracle.c (integrate, ORC_STEPPER_ROS23), mass matrix I:
//     W = I - d h J(u_n, t_n)                         d = 1 / (2 + sqrt 2), J = d rhs / d u, frozen over the step
//     k1 = W \ (f0 + d h dT)                          f0 = rhs(u_n, t_n) (first-same-as-last), dT = d rhs / dt
//     k2 = W \ (rhs(u_n + h/2 k1, t_n + h/2) - k1) + k1
//     u_{n+1} = u_n + h k2
//     k3 = W \ (f2 - e32 (k2 - f1) - 2 (k1 - f0) + d h dT)   f2 = rhs(u_{n+1}, t_n + h), e32 = 6 + sqrt 2
//     err = h/6 (k1 - 2 k2 + k3);  dense output u(th) = u_n + h (c1 k1 + c2 k2), c1 = th (1 - th) / (1 - 2 d), c2 = th (th - 2 d) / (1 - 2 d)
// (k3 enters the error estimate only.  Its time-derivative term is Shampine-Reichelt's `h d T`: on u' = g(t) the estimate is then h^3 g''/24, the midpoint rule's own error; with
// `h T` — the other reading of the upstream source, ORC_RECALL_ROS_K3_T in oracle/adjoint_oracle.h — it is h^2 (1 - d) g' / 6, one order low, and every reverse pass, whose
// right-hand side depends on t through the forward interpolant, takes 10-270 x the steps for the same answer: profiles/r6_rosenbrock23_steps.json.)
// dT by a forward difference (FiniteDiff's default step sqrt(eps) max(1, |t|), along the direction of integration), skipped for autonomous systems; controller:
// PI with beta1 = 7 / (10 order), beta2 = 2 / (5 order), order 2, and the steady band 1 <= q <= 6/5 -> q = 1 of the adaptive implicit algorithms.
// The forward record is the SAME monomial record as Tsit5's (degree 2 here, the cubic and quartic rows zero): every reader of a forward or adjoint record — the
// cursors, the quadrature pass — works unchanged.  Rows of K: 0 = k1, 1 = k2, 2 = f0, 3 = f2.
struct ROS23 { static constexpr double D = 0.29289321881345247560, E32 = 7.41421356237309504880; };
template <int NZ, class KS>
HIPADJ_HD void ros23_poly(const KS& K, double h, double (&c)[5][NZ]) {
    const double s = h / (1.0 - 2.0 * ROS23::D);
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const double k1 = K.get(0, i), k2 = K.get(1, i);
        c[0][i] = K.get(KS_UPREV, i); c[1][i] = s * (k1 - 2.0 * ROS23::D * k2); c[2][i] = s * (k2 - k1); c[3][i] = 0.0; c[4][i] = 0.0;
    }
}
template <int NZ, int NOUT, class KS>
HIPADJ_HD void ros23_interp(const KS& K, double th, double h, double (&y)[NOUT]) {
    const double c1 = th * (1.0 - th) / (1.0 - 2.0 * ROS23::D), c2 = th * (th - 2.0 * ROS23::D) / (1.0 - 2.0 * ROS23::D);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) y[i] = K.get(KS_UPREV, i) + h * (c1 * K.get(0, i) + c2 * K.get(1, i));
}
// M x M LU in registers, every index a compile-time constant: partial pivoting by successive exchange (row i > c moves up whenever its entry is the larger: after the
// sweep over i the pivot is the column's maximum, as in the oracle), the exchanges remembered as flags and replayed on every right-hand side.
template <int M> struct SmallLU {
    double a[M][M];
    bool sw[M][M];
    HIPADJ_HD void factor() {
#pragma unroll
        for (int c = 0; c < M; ++c) {
#pragma unroll
            for (int i = c + 1; i < M; ++i) {
                const bool x = habs(a[i][c]) > habs(a[c][c]);
                sw[c][i] = x;
#pragma unroll
                for (int j = 0; j < M; ++j) { const double u = a[c][j], v = a[i][j]; a[c][j] = x ? v : u; a[i][j] = x ? u : v; }
            }
            const double inv = 1.0 / a[c][c];
#pragma unroll
            for (int i = c + 1; i < M; ++i) {
                const double l = a[i][c] * inv;
                a[i][c] = l;
#pragma unroll
                for (int j = c + 1; j < M; ++j) a[i][j] -= l * a[c][j];
            }
        }
    }
    HIPADJ_HD void solve(double (&b)[M]) const {
#pragma unroll
        for (int c = 0; c < M; ++c)
#pragma unroll
            for (int i = c + 1; i < M; ++i) { const double u = b[c], v = b[i]; b[c] = sw[c][i] ? v : u; b[i] = sw[c][i] ? u : v; }
#pragma unroll
        for (int i = 1; i < M; ++i) {
            double s = b[i];
#pragma unroll
            for (int j = 0; j < i; ++j) s -= a[i][j] * b[j];
            b[i] = s;
        }
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            double s = b[i];
#pragma unroll
            for (int j = i + 1; j < M; ++j) s -= a[i][j] * b[j];
            b[i] = s / a[i][i];
        }
    }
};

// solve(prob, Rosenbrock23(); abstol, reltol, dt, tstops, callback) for a small system: the interface of tsit5_integrate plus `lin` — lin.factor(gh, u, t) forms and
// factors W = I - gh J(u, t), lin.solve(b) overwrites b with W \ b — and `autonomous` (dT = 0).  cb sees the step's k1, k2 in rows 0, 1 of K (ros23_poly / ros23_interp).
template <class T> struct ros_noref { using type = T; };
template <class T> struct ros_noref<T&> { using type = T; };
template <class T> struct ros_noref<T&&> { using type = T; };
template <int NZ, class KS, class Rhs, class Lin, class Cb, class Pre = NoPre>
HIPADJ_HD int ros23_integrate(double (&u)[NZ], double tstart, double tend, double dt_hint, double abstol, double reltol,
                              const double* __restrict__ tstops, int ntstops, bool cb_at_init, int max_steps,
                              KS& K, Rhs&& rhs, Lin&& lin, bool autonomous, Cb&& cb, Pre&& pre = NoPre(), double* dt_io = nullptr) {
    const double EPS = 2.220446049250313e-16;
    const double tdir = tend >= tstart ? 1.0 : -1.0;
    double t = tstart, tprev = tstart;
    double w[NZ];
    if (cb_at_init) {
#pragma unroll
        for (int i = 0; i < NZ; ++i) K.set(KS_UPREV, i, u[i]);
        cb(t, tprev, u, K);
    }
    bool need_k0 = true, first = true;
    double dt = 0.0, lqold = -9.21034037197618272;   // log qold, qold = 1e-4
    double dt_asked = 0.0; bool clipped = false;      // (the step the controller asked for before a stop clipped it: handed to the caller with the final proposal, dt_io)
    int its = 0, naccept = 0, guard = 0;
    double ts_cur = ntstops > 0 ? tstops[0] : tend;
#pragma unroll 1
    while (tdir * t < tdir * tend) {
        if (++guard > 16 * max_steps + 64) return -1;
        if (need_k0) {   // first step, or u was changed by a callback: f0 = rhs(u, t)
            rhs(w, u, t);
#pragma unroll
            for (int i = 0; i < NZ; ++i) K.set(2, i, w[i]);
            need_k0 = false;
        }
        if (first) {
            first = false;
            if (dt_hint > 0) dt = tdir * dt_hint;
            else {   // Hairer-Norsett-Wanner initial step with order 2 (exponent 1 / 3); w still holds f0
                double d0 = 0, d1 = 0;
#pragma unroll
                for (int i = 0; i < NZ; ++i) { const double sc = abstol + habs(u[i]) * reltol; d0 += (u[i] / sc) * (u[i] / sc); d1 += (w[i] / sc) * (w[i] / sc); }
                d0 = sqrt(d0 / NZ); d1 = sqrt(d1 / NZ);
                double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                h0 = hmin2(h0, habs(tend - t));
                double u1[NZ], f1[NZ];
#pragma unroll
                for (int i = 0; i < NZ; ++i) u1[i] = u[i] + tdir * h0 * w[i];
                rhs(f1, u1, t + tdir * h0);
                double d2 = 0;
#pragma unroll
                for (int i = 0; i < NZ; ++i) { const double sc = abstol + habs(u[i]) * reltol; const double q = (f1[i] - K.get(2, i)) / sc; d2 += q * q; }
                d2 = sqrt(d2 / NZ) / h0;
                const double h1 = (hmax2(d1, d2) <= 1e-15) ? hmax2(1e-6, h0 * 1e-3) : pow(0.01 / hmax2(d1, d2), 1.0 / 3.0);
                dt = tdir * hmin2(hmin2(100.0 * h0, h1), habs(tend - t));
            }
        }
        pre(t);
        while (its < ntstops && tdir * ts_cur <= tdir * t + 100.0 * EPS * hmax2(habs(t), habs(ts_cur))) { ++its; ts_cur = its < ntstops ? tstops[its] : tend; }
        double tstop = tend;
        if (its < ntstops && tdir * ts_cur < tdir * tend) tstop = ts_cur;
        double h = dt;
        dt_asked = dt; clipped = habs(h) > habs(tstop - t);
        if (clipped) h = tstop - t;
        if (habs((t + h) - tstop) < 100.0 * EPS * hmax2(habs(t + h), habs(tstop))) h = tstop - t;
#pragma unroll
        for (int i = 0; i < NZ; ++i) K.set(KS_UPREV, i, u[i]);
        const double gh = ROS23::D * h;
        lin.factor(gh, u, t);
        double dT[NZ], b[NZ], f1[NZ];
        if (autonomous) {
#pragma unroll
            for (int i = 0; i < NZ; ++i) dT[i] = 0.0;
        } else {
            const double del = tdir * 1.4901161193847656e-08 * hmax2(1.0, habs(t));
            rhs(dT, u, t + del);
#pragma unroll
            for (int i = 0; i < NZ; ++i) dT[i] = (dT[i] - K.get(2, i)) / del;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) b[i] = K.get(2, i) + gh * dT[i];
        lin.solve(b);
#pragma unroll
        for (int i = 0; i < NZ; ++i) { K.set(0, i, b[i]); w[i] = u[i] + 0.5 * h * b[i]; }
        rhs(f1, w, t + 0.5 * h);
        // mass-matrix form (Lin::MASS: a semi-explicit DAE, or its adjoint): the M^-1 f rewrite of the stages multiplied through by M — k2 = W \ (f1 - M k1) + k1,
        // k3 = W \ (f2 - e32 (M k2 - f1) - 2 (M k1 - f0) + d h dT), W = M - d h J (lin.factor) — which stays meaningful for a singular M
        constexpr bool MASS = ros_noref<Lin>::type::MASS != 0;
        double mk1[MASS ? NZ : 1];
        if constexpr (MASS) {
            double k1v[NZ];
#pragma unroll
            for (int i = 0; i < NZ; ++i) k1v[i] = K.get(0, i);
            lin.mulM(mk1, k1v);
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) b[i] = f1[i] - (MASS ? mk1[MASS ? i : 0] : K.get(0, i));
        lin.solve(b);
#pragma unroll
        for (int i = 0; i < NZ; ++i) { const double k2 = b[i] + K.get(0, i); K.set(1, i, k2); w[i] = u[i] + h * k2; }
        rhs(b, w, t + h);                                  // f2 (first-same-as-last of the next step)
        if constexpr (MASS) {
            double k2v[NZ], mk2[NZ];
#pragma unroll
            for (int i = 0; i < NZ; ++i) k2v[i] = K.get(1, i);
            lin.mulM(mk2, k2v);
#pragma unroll
            for (int i = 0; i < NZ; ++i) { K.set(3, i, b[i]); b[i] = b[i] - ROS23::E32 * (mk2[i] - f1[i]) - 2.0 * (mk1[i] - K.get(2, i)) + gh * dT[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < NZ; ++i) { K.set(3, i, b[i]); b[i] = b[i] - ROS23::E32 * (K.get(1, i) - f1[i]) - 2.0 * (K.get(0, i) - K.get(2, i)) + gh * dT[i]; }
        }
        lin.solve(b);
        double e2 = 0.0;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const double err = h / 6.0 * (K.get(0, i) - 2.0 * K.get(1, i) + b[i]);
            const double sc = abstol + hmax2(habs(u[i]), habs(w[i])) * reltol;
            const double q = err / sc;
            e2 += q * q;
        }
        const double EEst = sqrt(e2 / NZ);
        const double lE = ts5_log(hmin2(hmax2(EEst, 1e-300), 1e300));
        double q = ts5_exp((7.0 / 20.0) * lE - (1.0 / 5.0) * lqold);
        q = hmax2(1.0 / 10.0, hmin2(5.0, q / 0.9));
        if (EEst <= 1.0 || habs(h) < 1e-14 * hmax2(1.0, habs(t))) {
            double tnew = t + h;
            if (habs(tnew - tstop) < 100.0 * EPS * hmax2(habs(tnew), habs(tstop))) tnew = tstop;
            if (q >= 1.0 && q <= 1.2) q = 1.0;            // the steady band of the adaptive implicit algorithms
            lqold = hmax2(lE, -9.21034037197618272);
#pragma unroll
            for (int i = 0; i < NZ; ++i) u[i] = w[i];
            tprev = t; t = tnew; ++naccept;
            dt = h / q;
            if (habs(dt) < 1e-14 * hmax2(1.0, habs(tnew))) dt = tdir * 1e-14 * hmax2(1.0, habs(tnew));
            if (cb(t, tprev, u, K)) need_k0 = true;
            else {
#pragma unroll
                for (int i = 0; i < NZ; ++i) K.set(2, i, K.get(3, i));   // first-same-as-last
            }
            if (naccept > max_steps) return -1;
        } else {
            dt = h / hmin2(5.0, ts5_exp((7.0 / 20.0) * lE) / 0.9);
        }
    }
    // dt_io: the controller's state for a solve that CONTINUES this one (the reverse pieces between the events of a ContinuousCallback): its last proposal — or, when the
    // last step was cut short by the end of the span, the larger of that and the step it had asked for
    if (dt_io) *dt_io = clipped ? hmax2(habs(dt), habs(dt_asked)) : habs(dt);
    return naccept;
}

// ---- semi-explicit DAEs (model_dae<Mo>) -------------------------------------------------------------------------------------------------------------------------------
// the algebraic block of J' (TR) or of J, identity on the differential rows and columns — one factorisation serves every solve on the algebraic variables
template <class Mo, bool TR> HIPADJ_HD void dae_alg_block(SmallLU<Mo::N>& B, const double (&y)[Mo::N], const double (&pv)[Mo::NP], double t) {
    double J[Mo::N][Mo::N];
    model_jacobian<Mo>(J, y, pv, t);
#pragma unroll
    for (int r = 0; r < Mo::N; ++r)
#pragma unroll
        for (int c = 0; c < Mo::N; ++c) {
            const bool aa = Mo::isalg(r) && Mo::isalg(c);
            if (TR) B.a[c][r] = aa ? J[r][c] : (r == c ? 1.0 : 0.0); else B.a[r][c] = aa ? J[r][c] : (r == c ? 1.0 : 0.0);
        }
    B.factor();
}
// BrownFullBasicInit [upstream-recall]: the differential variables keep their values, the algebraic ones are solved from 0 = f_alg(u) by Newton (the oracle's dae_consistent_init)
template <class Mo> HIPADJ_HD bool dae_consistent_init(double (&u)[Mo::N], const double (&pv)[Mo::NP], double t) {
#pragma unroll 1
    for (int it = 0; it < 50; ++it) {
        double f[Mo::N], r[Mo::N];
        Mo::f(f, u, pv, t);
        double nr = 0.0;
#pragma unroll
        for (int j = 0; j < Mo::N; ++j) { r[j] = Mo::isalg(j) ? -f[j] : 0.0; nr = hmax2(nr, habs(r[j])); }
        if (nr <= 1e-13) return true;
        SmallLU<Mo::N> B; dae_alg_block<Mo, false>(B, u, pv, t);
        B.solve(r);
#pragma unroll
        for (int j = 0; j < Mo::N; ++j) if (Mo::isalg(j)) u[j] += r[j];
    }
    return false;
}

// W = I - gh (df/du)(u, t) of the forward problem: (df/du)' e_r = row r of the Jacobian, from the model's VJP
template <class Mo> struct RosLinFwd {
    static constexpr bool MASS = model_dae<Mo>::value;
    const double (&pv)[Mo::NP];
    SmallLU<Mo::N> lu;
    HIPADJ_HD void mulM(double (&out)[Mo::N], const double (&x)[Mo::N]) const {
        if constexpr (MASS) {
#pragma unroll
            for (int i = 0; i < Mo::N; ++i) { double s = 0.0;
#pragma unroll
                for (int j = 0; j < Mo::N; ++j) s += Mo::mass(i, j) * x[j];
                out[i] = s; }
        } else {
#pragma unroll
            for (int i = 0; i < Mo::N; ++i) out[i] = x[i];
        }
    }
    HIPADJ_HD explicit RosLinFwd(const double (&p_)[Mo::NP]) : pv(p_) {}
    HIPADJ_HD void factor(double gh, const double (&u)[Mo::N], double t) {
        double J[Mo::N][Mo::N];
        model_jacobian<Mo>(J, u, pv, t);
#pragma unroll
        for (int r = 0; r < Mo::N; ++r)
#pragma unroll
            for (int c = 0; c < Mo::N; ++c) { double mrc = (r == c ? 1.0 : 0.0); if constexpr (MASS) mrc = Mo::mass(r, c); lu.a[r][c] = mrc - gh * J[r][c]; }
        lu.factor();
    }
    HIPADJ_HD void solve(double (&b)[Mo::N]) const { lu.solve(b); }
};

// ------------------------------------------------------------------------------------------------------------
// forward dense solve; also out = sol(ts) (outT [M][n][Npad]) and the checkpoint states sol(c_j) (ckpt [nck][n][Npad])
// STEP: 0 = Tsit5, 1 = Rosenbrock23 (ros23_integrate; the records both write are the same monomial records)
template <class Mo, int STEP = 0>
HIPADJ_HD void forward_tsit5_lane(const AdaptGeom& g, long i, const double* __restrict__ u0, const double* __restrict__ p,
                                  double* __restrict__ rec, int* __restrict__ nsteps, const double* __restrict__ save_t,
                                  double* __restrict__ outT, const double* __restrict__ ck_t, double* __restrict__ ckpt,
                                  double* __restrict__ yT, int* __restrict__ flag, double* kbase, int kstride) {
    constexpr int N = Mo::N, RW = 2 + 5 * N;
    using KS = typename ts5_select<ts5_in_regs<Mo, N>::value, KRegs<N>, const KStore<N>>::type;
    KS K = ts5_make_rows<KS>(kbase, kstride);
    double pv[Mo::NP];
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) pv[j] = g.p_shared ? p[j] : p[i * Mo::NP + j];
    double u[N];
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = u0[i * N + j];
    int s = 0, ms = 0, mc = 0;
    bool overflow = false;
    if constexpr (model_dae<Mo>::value) { if (!dae_consistent_init<Mo>(u, pv, g.t0)) overflow = true; }      // (no consistent state found: reported like a step overflow)
    // points that coincide with t0
    while (outT && ms < g.M && save_t[ms] <= g.t0) {
#pragma unroll
        for (int j = 0; j < N; ++j) outT[((long)ms * N + j) * g.Npad + i] = u[j];
        ++ms; }
    while (ckpt && mc < g.nck && ck_t[mc] <= g.t0) {
#pragma unroll
        for (int j = 0; j < N; ++j) ckpt[((long)mc * N + j) * g.Npad + i] = u[j];
        ++mc; }
    // next output / checkpoint time, cached in registers (one dependent L2 load per accepted step otherwise)
    const double TINF = 1.7976931348623157e308;
    double ts_next = (outT && ms < g.M) ? save_t[ms] : TINF, tc_next = (ckpt && mc < g.nck) ? ck_t[mc] : TINF;
    auto frhs = [&](double (&du)[N], const double (&uu)[N], double t) { Mo::f(du, uu, pv, t); };
    // ContinuousCallback (model_has_cond; the oracle's section 3b, src/callback_tracking.jl:1-223): the sign of the condition at ten points of the accepted step's dense output
    // against its sign at the step's start (right after an event: at 1/100 of the step); the first bracket halved 52 times; the step is cut at the bracket's upper end — the
    // record rescaled to [tprev, t_event] —, u <- affect(u), t <- t_event, and the integrator recomputes the derivative; its proposal for the next step stands
    // VectorContinuousCallback (Mo::NCOND > 1: `out[k] = ...` in the condition body, `idx` in the affect body): the scan watches every component; in the first tenth of the step where any of them
    // crosses, each crossing component is bisected on its own and the EARLIEST root is the event (simultaneous fires of several components are not merged: DESIGN.md section 4.12)
    constexpr int NC = model_ncond<Mo>::value;
    constexpr int CDIR = model_cdir<Mo>::value;      // 0: both directions fire; +1 / -1: only crossings upward / downward (hipadj_model_set_callback_direction)
    auto crosses = [](double a, double b) -> bool { return (a * b < 0.0 || (b == 0.0 && a != 0.0)) && (CDIR == 0 || (CDIR > 0 ? a < 0.0 : a > 0.0)); };
    double cprev[NC]; bool nudge = false, terminated = false; int nevl = 0, evk = 0;
    for (int k = 0; k < NC; ++k) cprev[k] = 0.0;
    if constexpr (model_has_cond<Mo>::value) { if (g.maxev > 0) { Mo::cond(cprev, u, pv, g.t0); for (int k = 0; k < NC; ++k) nudge = nudge || (cprev[k] == 0.0); } }
    auto fcb = [&](double& t, double tprev, double (&un)[N], const auto& KK) -> bool {
            double h = t - tprev;
            double c[5][N];
            bool event = false;
            double uleft[model_has_cond<Mo>::value ? N : 1];
            if constexpr (STEP == 1) ros23_poly<N>(KK, h, c); else tsit5_poly<N>(KK, h, c);
            if constexpr (model_has_cond<Mo>::value) {
                if (g.maxev > 0 && h != 0.0) {
                    double y[N], cv[NC], ca[NC];
                    if (nudge) { poly_eval<N>(0.01, c, y); Mo::cond(cprev, y, pv, tprev + 0.01 * h); nudge = false; }
                    double tha = 0.0, thb = 0.0; int kx = -1; bool any = false;
#pragma unroll
                    for (int k = 0; k < NC; ++k) ca[k] = cprev[k];
#pragma unroll 1
                    for (int j = 1; j <= 10 && !any; ++j) {
                        thb = j < 10 ? 0.1 * j : 1.0;
                        if (j < 10) poly_eval<N>(thb, c, y);
                        else {
#pragma unroll
                            for (int q = 0; q < N; ++q) y[q] = un[q]; }
                        Mo::cond(cv, y, pv, tprev + thb * h);
#pragma unroll
                        for (int k = 0; k < NC; ++k) any = any || crosses(ca[k], cv[k]);
                        if (!any) { tha = thb;
#pragma unroll
                            for (int k = 0; k < NC; ++k) ca[k] = cv[k]; }
                    }
                    if (!any) {
#pragma unroll
                        for (int k = 0; k < NC; ++k) cprev[k] = cv[k];
                    } else {
                        // every component that crosses in this tenth is bisected on its own; the EARLIEST root is the event (ties: the lowest component)
                        double best = 2.0, cend[NC];
#pragma unroll
                        for (int k = 0; k < NC; ++k) cend[k] = cv[k];
#pragma unroll 1
                        for (int k = 0; k < NC; ++k) {
                            double cak = ca[0], cek = cend[0];
#pragma unroll
                            for (int q = 1; q < NC; ++q) { cak = (q == k) ? ca[q] : cak; cek = (q == k) ? cend[q] : cek; }
                            if (!crosses(cak, cek)) continue;
                            double lo = tha, hi = thb;
#pragma unroll 1
                            for (int it = 0; it < 52; ++it) {
                                const double thm = 0.5 * (lo + hi);
                                poly_eval<N>(thm, c, y);
                                Mo::cond(cv, y, pv, tprev + thm * h);
                                double cm = cv[0];
#pragma unroll
                                for (int q = 1; q < NC; ++q) cm = (q == k) ? cv[q] : cm;
                                if (cak * cm < 0.0 || (cm == 0.0 && cak != 0.0)) hi = thm; else { lo = thm; cak = cm; }
                            }
                            if (hi < best) { best = hi; kx = k; }
                        }
                        thb = best;
                        const double tev = tprev + thb * h;
                        if (!(tev < g.t1) || time_hits(tev, g.t1)) Mo::cond(cprev, un, pv, t);      // (an event at the end of the span changes nothing that is observed)
                        else {
                            poly_eval<N>(thb, c, y);
                            double r = thb;
#pragma unroll
                            for (int m = 1; m < 5; ++m) {
#pragma unroll
                                for (int q = 0; q < N; ++q) c[m][q] *= r;
                                r *= thb; }
                            // terminate!(integrator) (the affect body set `terminate`): the event is recorded with bit 8 of its component index, the lane's solve ends here, the save and
                            // checkpoint times after it hold the state after the affect
                            terminated = Mo::cc_affect(un, y, pv, tev, kx);
#pragma unroll
                            for (int q = 0; q < N; ++q) uleft[q] = y[q];
                            t = tev; h = tev - tprev; nudge = true; event = true; evk = kx | (terminated ? 256 : 0);
                        }
                    }
                }
            }
            if (s < g.Smax) {
                if (rec) {
                    rec[((long)s * RW + 0) * g.Npad + i] = tprev; rec[((long)s * RW + 1) * g.Npad + i] = t;
#pragma unroll
                    for (int m = 0; m < 5; ++m)
#pragma unroll
                        for (int j = 0; j < N; ++j) rec[((long)s * RW + 2 + m * N + j) * g.Npad + i] = c[m][j];
                }
            } else if (rec) overflow = true;      // (BacksolveAdjoint keeps no dense record: only the step bound g.maxit of the integrator limits it — with max_steps = 0 the reference's maxiters)
            ++s;
            if constexpr (model_has_cond<Mo>::value) {
                if (event) {
                    if (nevl < g.maxev) {
                        g.ev_s[(long)nevl * g.Npad + i] = s; g.ev_t[(long)nevl * g.Npad + i] = t; g.ev_k[(long)nevl * g.Npad + i] = evk;
#pragma unroll
                        for (int q = 0; q < N; ++q) { g.ev_ul[((long)nevl * N + q) * g.Npad + i] = uleft[q]; g.ev_ur[((long)nevl * N + q) * g.Npad + i] = un[q]; }
                    }
                    else { overflow = true; t = g.t1; }      // more events than the list holds (an accumulation point of events, or max_events too small): reported, and the solve ends here
                    ++nevl;
                }
            }
            while (ts_next <= t || time_hits(ts_next, t)) {
                double y[N]; poly_eval<N>((ts_next - tprev) / h, c, y);
#pragma unroll
                for (int j = 0; j < N; ++j) outT[((long)ms * N + j) * g.Npad + i] = y[j];
                ++ms; ts_next = ms < g.M ? save_t[ms] : TINF; }
            while (tc_next <= t || time_hits(tc_next, t)) {
                double y[N]; poly_eval<N>((tc_next - tprev) / h, c, y);
#pragma unroll
                for (int j = 0; j < N; ++j) ckpt[((long)mc * N + j) * g.Npad + i] = y[j];
                ++mc; tc_next = mc < g.nck ? ck_t[mc] : TINF; }
            (void)un;
            if constexpr (model_has_cond<Mo>::value) { if (terminated) t = g.t1; }      // (after the step's save / checkpoint times were served up to the event: the integrator's loop ends)
            return event;
        };
    int na;
    if constexpr (STEP == 1) {
        RosLinFwd<Mo> lin(pv);
        na = ros23_integrate<N>(u, g.t0, g.t1, g.dt0, g.abstol, g.reltol, nullptr, 0, false, g.maxit, K, frhs, lin, !Mo::TIME_DEP, fcb);
    } else na = tsit5_integrate<N>(u, g.t0, g.t1, g.dt0, g.abstol, g.reltol, nullptr, 0, false, g.maxit, K, frhs, fcb);
    nsteps[i] = s;   // the TRUE count, also beyond the capacity: the host sizes the buffers from it (flag bit 4 marks the overflow)
    if constexpr (model_has_cond<Mo>::value) {
        if (g.maxev > 0) g.nev[i] = nevl;
        if (terminated) {      // the reference's solution ends at the event; here every later save / checkpoint time holds the final state (their cotangents are ignored by the reverse pass)
            while (outT && ms < g.M) {
#pragma unroll
                for (int j = 0; j < N; ++j) outT[((long)ms * N + j) * g.Npad + i] = u[j];
                ++ms; }
            while (ckpt && mc < g.nck) {
#pragma unroll
                for (int j = 0; j < N; ++j) ckpt[((long)mc * N + j) * g.Npad + i] = u[j];
                ++mc; }
        }
    }
    if (yT) {
#pragma unroll
        for (int j = 0; j < N; ++j) yT[(long)j * g.Npad + i] = u[j]; }
    if (na < 0 || overflow) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(flag, 4);
#else
        *flag |= 4;
#endif
    }
}

// cursor into one trajectory's forward dense solution: the step containing t, its coefficients cached in registers.
// The walk touches only the step end points (one load per visited step: consecutive steps share an end point);
// the 5 n coefficients are loaded once per step actually used.
#ifndef HIPADJ_TS5_PREFETCH
#define HIPADJ_TS5_PREFETCH 0   // A/B hook: 1 = fetch the record below one step ahead into a second register set.  Measured SLOWER twice: round 1 (-2 % .. +7 %)
                                // and round 3 on the register-resident sweep (Lorenz 10^4: Interpolating 1.83 -> 1.94 ms, Gauss 2.43 -> 2.79; profiles/r3_tsit5_prefetch_ab.log)
#endif
template <class Mo, bool PF = (HIPADJ_TS5_PREFETCH != 0)> struct FwdCursor {
    static constexpr int N = Mo::N, RW = 2 + 5 * Mo::N;
    const double* rec; long Npad, i; int ns, sc, lc;
    int smin = 0;            // lowest record the walk may reach (events: the first record of the piece the reverse solve stands in; a stage time that rounds one ulp below the
                             // event time must not read the state from before the affect)
    double ta, tb, c[5][Mo::N];
    // the reverse solve crosses an event downward: onto record s (the step the event cut short, which ends at the event time), with the piece below as the new range
    HIPADJ_HD void below(int s, int smin_) { sc = s; smin = smin_; ta = rec[((long)sc * RW + 0) * Npad + i]; tb = rec[((long)sc * RW + 1) * Npad + i]; }
    // PF (off by default, see the macro): the reverse sweeps walk downward and the 64 lanes of a wave change records at different step attempts, so most
    // attempts of the wave wait for SOME lane's 5 n coefficient loads (SQ_WAIT_ANY is a third of the wave cycles, profiles/r3_tsit5_counters.txt).  With PF
    // the cursor also issues the loads of record s - 1 into a second register set when it arrives on record s.  It did not pay: the extra registers and
    // the copies cost more than the round trips they hide.
    // (PF = false: the quadrature lanes, which jump between nodes.)
    double pc[PF ? 5 : 1][PF ? Mo::N : 1]; int pn;
    HIPADJ_HD void init(const double* r, long np, long ii, int nsteps) {
        rec = r; Npad = np; i = ii; ns = nsteps; sc = nsteps - 1; lc = -1; pn = -2; smin = 0;
        ta = rec[((long)sc * RW + 0) * Npad + i]; tb = rec[((long)sc * RW + 1) * Npad + i];
    }
    // position the cursor on the step containing t by bisection (quadrature lanes start anywhere in [t0, T])
    HIPADJ_HD void seek(double t) {
        int lo = 0, hi = ns - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[((long)mid * RW + 1) * Npad + i] < t) lo = mid + 1; else hi = mid; }
        sc = lo; ta = rec[((long)sc * RW + 0) * Npad + i]; tb = rec[((long)sc * RW + 1) * Npad + i];
    }
    // y = sol(t): the reverse sweep moves mostly downward, so a linear cursor walk replaces the binary search.
    // (Measured: fetching the record below one step ahead into a second register set changes the sweep by -2 % .. +7 %:
    // the sweep is bound by the instruction count of a step, not by this round trip.)
    HIPADJ_HD void eval(double t, double (&y)[Mo::N]) {
        while (t < ta && sc > smin) { --sc; tb = ta; ta = rec[((long)sc * RW + 0) * Npad + i]; }
        while (t > tb && sc < ns - 1) { ++sc; ta = tb; tb = rec[((long)sc * RW + 1) * Npad + i]; }
        if (sc != lc) {
            lc = sc;
            bool have = false;
            if constexpr (PF) {
                if (sc == pn) {
                    have = true;
#pragma unroll
                    for (int m = 0; m < 5; ++m)
#pragma unroll
                        for (int j = 0; j < N; ++j) c[m][j] = pc[m][j];
                }
            }
            if (!have) {
                const long base = ((long)sc * RW + 2) * Npad + i;
#pragma unroll
                for (int m = 0; m < 5; ++m)
#pragma unroll
                    for (int j = 0; j < N; ++j) c[m][j] = rec[base + (long)(m * N + j) * Npad];
            }
            if constexpr (PF) {
                if (sc > 0) {          // the record below: issued now, consumed when the cursor gets there
                    pn = sc - 1;
                    const long base = ((long)pn * RW + 2) * Npad + i;
#pragma unroll
                    for (int m = 0; m < 5; ++m)
#pragma unroll
                        for (int j = 0; j < N; ++j) pc[m][j] = rec[base + (long)(m * N + j) * Npad];
                }
            }
        }
        poly_eval<N>((t - ta) / (tb - ta), c, y);
    }
};

// reverse sweeps.  ALG: 0 Interpolating (z = [lam; mu]), 1 Backsolve (z = [lam; mu; y]), 2 Gauss (z = lam, mu by quadrature),
// 3 Quadrature pass 1 (z = lam, recorded densely), 4 GaussKronrod (as Gauss with a per-step adaptive (7,15) rule)
// ALG 3 = Quadrature pass 1: z = lam, every accepted step is recorded (dense adjoint solution, src/quadrature_adjoint.jl:527-530)
template <class Mo, int ALG> struct AdjNZ { static constexpr int value = ALG == 0 ? Mo::N + Mo::NP : (ALG == 1 ? 2 * Mo::N + Mo::NP : Mo::N); };

// cursor into the dense ADJOINT solution of one trajectory: records in order of decreasing time, record s covers
// [t_end, t_start] with t_end < t_start, monomial coefficients in theta = (t - t_start) / (t_end - t_start)
template <class Mo> struct AdjCursor {
    static constexpr int N = Mo::N, RW = 2 + 5 * Mo::N;
    const double* rec; long Npad, i; int ns, sc, lc;
    double ts, te, c[5][Mo::N];   // ts = start (upper), te = end (lower)
    HIPADJ_HD void init(const double* r, long np, long ii, int nsteps, double t) {
        rec = r; Npad = np; i = ii; ns = nsteps; lc = -1;
        int lo = 0, hi = ns - 1;   // first record whose lower end is <= t
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rec[((long)mid * RW + 1) * Npad + i] > t) lo = mid + 1; else hi = mid; }
        sc = lo; ts = rec[((long)sc * RW + 0) * Npad + i]; te = rec[((long)sc * RW + 1) * Npad + i];
    }
    HIPADJ_HD void eval(double t, double (&lam)[Mo::N]) {
        while (t < te && sc < ns - 1) { ++sc; ts = te; te = rec[((long)sc * RW + 1) * Npad + i]; }
        while (t > ts && sc > 0) { --sc; te = ts; ts = rec[((long)sc * RW + 0) * Npad + i]; }
        if (sc != lc) {
            lc = sc;
            const long base = ((long)sc * RW + 2) * Npad + i;
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int j = 0; j < N; ++j) c[m][j] = rec[base + (long)(m * N + j) * Npad];
        }
        poly_eval<N>((t - ts) / (te - ts), c, lam);
    }
};

// CK = true (Interpolating / Gauss with checkpointing=true, src/interpolating_adjoint.jl:54-109, 207-277): no dense forward
// solution exists; `lrec` is this lane's buffer for ONE checkpoint interval [c_j, c_{j+1}] (capacity g.SmaxI steps), re-solved
// from the stored sol(c_j) with the forward tolerances and dt = |last step of the previous interval solution| (:245-251)
// whenever the sweep steps below the current interval; the last interval is solved eagerly (:88-92).
template <class Mo, int ALG, int CC, bool CK = false, int STEP = 0>
HIPADJ_HD void adjoint_tsit5_lane(const AdaptGeom& g, long i, const double* __restrict__ p, const double* __restrict__ rec,
                                  const int* __restrict__ nsteps, const double* __restrict__ yT, const double* __restrict__ ckpt,
                                  const double* __restrict__ ck_t, const double* __restrict__ save_t, const double* __restrict__ tstops_desc,
                                  int ntstops, const double* __restrict__ cotT, double (&lam_out)[Mo::N], double (&mu_out)[Mo::NP], int* __restrict__ flag,
                                  double* kbase, int kstride, double* __restrict__ arec = nullptr, int* __restrict__ nsteps_adj = nullptr, int SmaxA = 0,
                                  double* kfbase = nullptr, double* lrec = nullptr) {
    constexpr int N = Mo::N, NP = Mo::NP, NZ = AdjNZ<Mo, ALG>::value;
    static_assert(STEP == 0 || ALG != 1 || !model_dae<Mo>::value, "Rosenbrock23: BacksolveAdjoint is not offered on a semi-explicit DAE (the reference documents it to fail there)");
#ifndef HIPADJ_TS5_REGS_CK
#define HIPADJ_TS5_REGS_CK 1   // checkpointing = true: the rows of the sweep AND of the interval re-solve in registers (A/B hook)
#endif
    using KS = typename ts5_select<ts5_in_regs<Mo, NZ>::value && (!CK || HIPADJ_TS5_REGS_CK), KRegs<NZ>, const KStore<NZ>>::type;
    using KSF = typename ts5_select<ts5_in_regs<Mo, NZ>::value && CK && HIPADJ_TS5_REGS_CK, KRegs<N>, const KStore<N>>::type;   // ... of the interval re-solve
    KS K = ts5_make_rows<KS>(kbase, kstride);
    double pv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) pv[j] = g.p_shared ? p[j] : p[i * NP + j];
    FwdCursor<Mo, (HIPADJ_TS5_PREFETCH != 0) && !CK> cur;   // (CK: the sweep's and the re-solve's rows already fill the registers; with the second coefficient set on top one
                                                          //  kernel — LinDiag, GaussKronrod — came back with the spill placement tests/tools/isa_lint.py flags)
    int icur = g.nck - 2;            // CK: checkpoint interval the cursor's records belong to
    bool ck_overflow = false;
    // ContinuousCallback with checkpointing = true: a checkpoint interval is re-solved only as far as the current PIECE (between two of this trajectory's events) reaches — from
    // the state just after the piece's lower event (stored by the forward lane, ev_ur) or the checkpoint, to the piece's upper event or the next checkpoint: no event lies inside
    // a re-solve and none is searched for
    int pc_ev = -1; double pc_lo = g.t0, pc_hi = g.t1;
    if constexpr (model_has_cond<Mo>::value && CK) {
        if (g.maxev > 0) { const int ne0 = g.nev[i] < g.maxev ? g.nev[i] : g.maxev; if (ne0 > 0) { pc_ev = ne0 - 1; pc_lo = g.ev_t[(long)pc_ev * g.Npad + i]; } }
    }
    auto resolve = [&](int j, double dt_hint) {
        constexpr int RW = 2 + 5 * N;
        KSF KF = ts5_make_rows<KSF>(kfbase, kstride);
        double uu[N];
        double rta = ck_t[j], rtb = ck_t[j + 1];
#pragma unroll
        for (int jj = 0; jj < N; ++jj) uu[jj] = ckpt[((long)j * N + jj) * g.Npad + i];
        if constexpr (model_has_cond<Mo>::value && CK) {
            if (pc_ev >= 0 && pc_lo > rta) { rta = pc_lo;
#pragma unroll
                for (int jj = 0; jj < N; ++jj) uu[jj] = g.ev_ur[((long)pc_ev * N + jj) * g.Npad + i]; }
            if (pc_hi < rtb) rtb = pc_hi;
        }
        int sl = 0;
        auto frhs = [&](double (&du)[N], const double (&u_)[N], double t) { Mo::f(du, u_, pv, t); };
        auto fcb = [&](double t, double tprev, double (&un)[N], const auto& KK) -> bool {
                (void)un;
                if (sl < g.SmaxI) {
                    double c[5][N];
                    if constexpr (STEP == 1) ros23_poly<N>(KK, t - tprev, c); else tsit5_poly<N>(KK, t - tprev, c);
                    lrec[((long)sl * RW + 0) * g.Npad + i] = tprev; lrec[((long)sl * RW + 1) * g.Npad + i] = t;
#pragma unroll
                    for (int m = 0; m < 5; ++m)
#pragma unroll
                        for (int jj = 0; jj < N; ++jj) lrec[((long)sl * RW + 2 + m * N + jj) * g.Npad + i] = c[m][jj];
                } else ck_overflow = true;
                ++sl;
                return false;
            };
        int nr;
        if constexpr (STEP == 1) {      // the interval re-solved with the forward problem's own stepper (src/interpolating_adjoint.jl:245-251); a DAE's checkpoint is a consistent state already
            RosLinFwd<Mo> flin(pv);
            nr = ros23_integrate<N>(uu, rta, rtb, dt_hint > 0 ? dt_hint : g.dt0, g.abstol, g.reltol, nullptr, 0, false, g.SmaxI, KF, frhs, flin, !Mo::TIME_DEP, fcb);
        } else nr = tsit5_integrate<N>(uu, rta, rtb, dt_hint > 0 ? dt_hint : g.dt0, g.abstol, g.reltol, nullptr, 0, false, g.SmaxI, KF, frhs, fcb);
        if (nr < 0) ck_overflow = true;
        cur.init(lrec, g.Npad, i, sl < g.SmaxI ? sl : g.SmaxI);
        icur = j;
    };
    if (CK) resolve(g.nck - 2, 0.0);
    else if (ALG != 1) cur.init(rec, g.Npad, i, nsteps[i] < g.Smax ? nsteps[i] : g.Smax);   // clamped: an overflowed forward pass is an error, not a fault
    double z[NZ];
#pragma unroll
    for (int j = 0; j < NZ; ++j) z[j] = 0.0;
    if (ALG == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) z[N + NP + j] = yT[(long)j * g.Npad + i]; }
    double gacc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) gacc[j] = 0.0;
    int cur_time = g.M, bs_cur = g.nck;
    double t_loss = g.M > 0 ? save_t[g.M - 1] : 0.0;
    if (ALG == 1 && bs_cur >= 1 && time_hits(g.t1, ck_t[bs_cur - 1])) --bs_cur;
    double t_ck = (ALG == 1 && ckpt && bs_cur >= 1) ? ck_t[bs_cur - 1] : 0.0;

    auto rhs = [&](double (&dz)[NZ], const double (&zz)[NZ], double t) {
        double y[N], lam[N], dl[N], dg[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) lam[j] = zz[j];
        if (ALG == 1) {
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = zz[N + NP + j];
        } else cur.eval(t, y);
        Mo::vjp_u(dl, lam, y, pv, t);
        double gu[N]; cost_grad_u<Mo, CC>(y, pv, t, gu);
#pragma unroll
        for (int j = 0; j < N; ++j) dz[j] = -dl[j] - gu[j];
        if (ALG == 0 || ALG == 1) {
            Mo::vjp_p(dg, lam, y, pv, t);
            if (cost_has_gp<CC>::value) {   // dgrad -= g_p
                double gp[NP]; cost_grad_p<Mo, CC>(y, pv, t, gp);
#pragma unroll
                for (int j = 0; j < NP; ++j) dg[j] += gp[j];
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) dz[N + j] = -dg[j];
        }
        if (ALG == 1) {
            double f[N]; Mo::f(f, y, pv, t);
#pragma unroll
            for (int j = 0; j < N; ++j) dz[N + NP + j] = f[j];
        }
    };
    int sa = 0;
    bool aoverflow = false;
    double dcorr[model_dae<Mo>::value ? NP : 1];      // semi-explicit DAE: sum over the loss jumps of f_p' [0; dlam_a]
    for (int j = 0; j < (model_dae<Mo>::value ? NP : 1); ++j) dcorr[j] = 0.0;
    auto cb = [&](double t, double tprev, double (&zz)[NZ], const auto& KK) -> bool {
        bool mod = false;
        if (ALG == 3 && t != tprev) {   // dense adjoint solution for the quadrature pass
            if (sa < SmaxA) {
                constexpr int RW = 2 + 5 * N;
                double c[5][NZ];
                if constexpr (STEP == 1) ros23_poly<NZ>(KK, t - tprev, c); else tsit5_poly<NZ>(KK, t - tprev, c);
                arec[((long)sa * RW + 0) * g.Npad + i] = tprev; arec[((long)sa * RW + 1) * g.Npad + i] = t;
#pragma unroll
                for (int m = 0; m < 5; ++m)
#pragma unroll
                    for (int j = 0; j < N; ++j) arec[((long)sa * RW + 2 + m * N + j) * g.Npad + i] = c[m][j];
            } else aoverflow = true;
            ++sa;
        }
        if (ALG == 2 && t != tprev) {   // IntegratingSumCallback: 3-point Gauss-Legendre of -(df/dp)^T lam on [tprev, t]
            const double half = 0.5 * (t - tprev), mid = 0.5 * (t + tprev), h = t - tprev;
#pragma unroll 1
            for (int q = (STEP == 1 ? 1 : 0); q < (STEP == 1 ? 2 : 3); ++q) {      // Rosenbrock23 (order 2): div(order + 1, 2) = ONE node, the midpoint rule [upstream-recall]
                const double xq = q == 0 ? -0.7745966692414833770 : (q == 1 ? 0.0 : 0.7745966692414833770);
                const double wq = STEP == 1 ? 2.0 : (q == 1 ? 8.0 / 9.0 : 5.0 / 9.0);
                const double tt = half * xq + mid;
                double y[N], W[NP], lamq[N];
                if constexpr (STEP == 1) ros23_interp<NZ, N>(KK, (tt - tprev) / h, h, lamq); else kstore_interp<NZ, N>(KK, (tt - tprev) / h, h, lamq);
                cur.eval(tt, y);
                Mo::vjp_p(W, lamq, y, pv, tt);
                if (cost_has_gp<CC>::value) {   // + g_p at the node; sign: DESIGN.md 6.5 (Gauss == Interpolating == Quadrature)
                    double gp[NP]; cost_grad_p<Mo, CC>(y, pv, tt, gp);
#pragma unroll
                    for (int j = 0; j < NP; ++j) W[j] += ((g.lflags & 2) ? -1.0 : 1.0) * gp[j];     // lflags bit 1: the reference's line as written (hipadj_config.reference_literal)
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) gacc[j] += half * wq * (-W[j]);
            }
        }
        if (ALG == 4 && t != tprev) {
            // IntegratingGKSumCallback [upstream-recall, DiffEqCallbacks is not vendored; src/gauss_adjoint.jl:820-825 only
            // constructs it]: (7,15) Gauss-Kronrod rule of the same integrand on the step, halved recursively (left half first)
            // while the Euclidean norm of Kronrod - Gauss exceeds 1e-7; same restatement as oracle `gk_panel`.
            constexpr int GKD = 12;
            const double hstep = t - tprev;
            double pa[GKD + 2], pb[GKD + 2]; int pd[GKD + 2]; int sp = 1;
            pa[0] = tprev; pb[0] = t; pd[0] = 0;
#pragma unroll 1
            while (sp > 0) {
                --sp;
                const double a = pa[sp], b = pb[sp]; const int d = pd[sp];
                const double c = 0.5 * (a + b), h = 0.5 * (b - a);
                double IK[NP], IG[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) { IK[j] = 0.0; IG[j] = 0.0; }
#pragma unroll 1
                for (int jn = 0; jn < 15; ++jn) {
                    const int q = jn < 7 ? jn : (jn == 7 ? 7 : 14 - jn);
                    const double x = jn < 7 ? -GK15::X[q] : (jn == 7 ? 0.0 : GK15::X[q]);
                    const double tt = c + h * x;
                    double y[N], W[NP], lamq[N];
                    if constexpr (STEP == 1) ros23_interp<NZ, N>(KK, (tt - tprev) / hstep, hstep, lamq); else kstore_interp<NZ, N>(KK, (tt - tprev) / hstep, hstep, lamq);
                    cur.eval(tt, y);
                    Mo::vjp_p(W, lamq, y, pv, tt);
                    if (cost_has_gp<CC>::value) {
                        double gp[NP]; cost_grad_p<Mo, CC>(y, pv, tt, gp);
#pragma unroll
                        for (int j = 0; j < NP; ++j) W[j] += ((g.lflags & 2) ? -1.0 : 1.0) * gp[j];
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) { IK[j] += GK15::WK[q] * (-W[j]); if (q & 1) IG[j] += GK15::WG[q / 2] * (-W[j]); }
                }
                double e = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) { IK[j] *= h; IG[j] *= h; const double dd = IK[j] - IG[j]; e += dd * dd; }
                if (sqrt(e) <= 1e-7 || d >= GKD) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) gacc[j] += IK[j];
                } else {
                    pa[sp] = c; pb[sp] = b; pd[sp] = d + 1; ++sp;      // right half: after the left one
                    pa[sp] = a; pb[sp] = c; pd[sp] = d + 1; ++sp;
                }
            }
        }
        if (ALG == 1 && ckpt && bs_cur >= 1 && time_hits(t, t_ck)) {               // backsolve_checkpoint_callbacks (t_ck = ck_t[bs_cur - 1], cached)
#pragma unroll
            for (int j = 0; j < N; ++j) zz[N + NP + j] = ckpt[((long)(bs_cur - 1) * N + j) * g.Npad + i];
            --bs_cur; mod = true;
            t_ck = bs_cur >= 1 ? ck_t[bs_cur - 1] : 0.0;
        }
        if (cur_time >= 1 && time_hits(t, t_loss)) {                                  // ReverseLossCallback (t_loss = save_t[cur_time - 1], cached)
            if (!(g.no_start && ALG != 1 && cur_time == 1)) {
                double y[N];
                if (ALG == 1) {
#pragma unroll
                    for (int j = 0; j < N; ++j) y[j] = zz[N + NP + j];
                } else cur.eval(t, y);
                if constexpr (model_has_dloss<Mo>::value && !model_dae<Mo>::value) {   // a model with discrete-loss bodies (hipadj_model_set_discrete_loss): dgdu_discrete / dgdp_discrete evaluated here when the handle selects them (a DAE model takes the branch below whatever bodies it carries: HIPADJ_LOSS_MODEL is refused for its stepper)
                    double gl[N];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        gl[j] = (g.loss_kind == 1) ? (y[j] - g.loss_shift) : __builtin_fma(g.la, y[j], g.lb * cotT[((long)(cur_time - 1) * N + j) * g.Npad + i]);
                    if (g.loss_kind == 3) {      // gl holds the data column
                        double d[N], o[N], gpd[NP];
#pragma unroll
                        for (int j = 0; j < N; ++j) d[j] = gl[j];
                        Mo::dgdu_disc(o, y, pv, t, cur_time - 1, d);
#pragma unroll
                        for (int j = 0; j < N; ++j) gl[j] = o[j];
                        if (!(g.lflags & 1)) {
                            Mo::dgdp_disc(gpd, y, pv, t, cur_time - 1, d);
#pragma unroll
                            for (int j = 0; j < NP; ++j) { if constexpr (ALG == 0 || ALG == 1) zz[N + j] += gpd[j]; else gacc[j] += gpd[j]; }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < N; ++j) zz[j] += gl[j];
                } else if constexpr (model_dae<Mo>::value) {
                    // src/adjoint_common.jl:790-813 for a semi-explicit DAE (the oracle's loss_jump, g_mm_dae):  dlam_a = -(J_aa' \ g_a);  dlam_d = g_d + (J' [0; dlam_a])_d;
                    // dp += f_p' [0; dlam_a] (:803 with src/sensitivity_interface.jl:510-521 — kept apart from the integrated parameter block so that the controller's
                    // error norm sees what the oracle's does);  M'[diff, diff] \ dlam_d;  lam_d += dlam_d;  lam_a re-initialised from 0 = (J' lam)_a
                    double gl[N], x[N], v[N], dla[N];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        gl[j] = (g.loss_kind == 1) ? (y[j] - g.loss_shift) : __builtin_fma(g.la, y[j], g.lb * cotT[((long)(cur_time - 1) * N + j) * g.Npad + i]);
                    SmallLU<N> B; dae_alg_block<Mo, true>(B, y, pv, t);
#pragma unroll
                    for (int j = 0; j < N; ++j) x[j] = Mo::isalg(j) ? gl[j] : 0.0;
                    B.solve(x);
#pragma unroll
                    for (int j = 0; j < N; ++j) dla[j] = Mo::isalg(j) ? -x[j] : 0.0;
                    { double W[NP]; Mo::vjp_p(W, dla, y, pv, t);
#pragma unroll
                      for (int j = 0; j < NP; ++j) dcorr[j] += W[j]; }
                    Mo::vjp_u(v, dla, y, pv, t);
#pragma unroll
                    for (int j = 0; j < N; ++j) x[j] = Mo::isalg(j) ? 0.0 : gl[j] + v[j];
                    { SmallLU<N> Mt;
#pragma unroll
                      for (int r = 0; r < N; ++r)
#pragma unroll
                          for (int c = 0; c < N; ++c) Mt.a[r][c] = (!Mo::isalg(r) && !Mo::isalg(c)) ? Mo::mass(c, r) : (r == c ? 1.0 : 0.0);
                      Mt.factor(); Mt.solve(x); }
#pragma unroll
                    for (int j = 0; j < N; ++j) if (!Mo::isalg(j)) zz[j] += x[j];
#pragma unroll
                    for (int j = 0; j < N; ++j) dla[j] = Mo::isalg(j) ? 0.0 : zz[j];
                    Mo::vjp_u(v, dla, y, pv, t);
#pragma unroll
                    for (int j = 0; j < N; ++j) x[j] = Mo::isalg(j) ? -v[j] : 0.0;
                    B.solve(x);
#pragma unroll
                    for (int j = 0; j < N; ++j) if (Mo::isalg(j)) zz[j] = x[j];
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j)      // u - shift, or la u + lb c with the streamed column c: the cotangent (0, 1), or the data of HIPADJ_LOSS_LSQ_DATA (w, -w)
                        zz[j] += (g.loss_kind == 1) ? (y[j] - g.loss_shift) : __builtin_fma(g.la, y[j], g.lb * cotT[((long)(cur_time - 1) * N + j) * g.Npad + i]);
                }
                mod = true;
            }
            --cur_time;
            t_loss = cur_time >= 1 ? save_t[cur_time - 1] : 0.0;
        }
        return mod;
    };
    const bool cb_at_init = g.M > 0 && time_hits(g.t1, save_t[g.M - 1]);
    auto pre = [&](double t) {
        if (CK) {
            if (icur > 0 && !(t > ck_t[icur])) {   // the sweep stands on (or below) the lower end of its interval: the next stages need the one below
                const double dtl = cur.ns > 0 ? habs(lrec[((long)(cur.ns - 1) * (2 + 5 * N) + 1) * g.Npad + i] - lrec[((long)(cur.ns - 1) * (2 + 5 * N) + 0) * g.Npad + i]) : 0.0;
                resolve(icur - 1, dtl);
            }
        }
    };
    // one reverse solve from ta down to tb (the whole span, or — ContinuousCallback — the piece between two events)
    double dt_carry = 0.0;      // the controller's proposal at the end of the piece above (events: the reverse solve does not restart its step size at an event, like the reference's)
    auto run_piece = [&](double ta_, double tb_, bool at_init) -> int {
    const double dt_first = dt_carry > 0.0 ? dt_carry : g.dt0;
    int na;
    if constexpr (STEP == 1) {
        // W = I - gh A(t_n) for the adjoint system z' = A(t) z + b(t):  A_ll = -J(y(t))', A_ml = -f_p(y(t))' (Interpolating), nothing else — so W is block triangular:
        // the lambda block by an n x n LU of I + gh J', the parameter block by substitution, x_mu = b_mu - gh f_p' x_lam
        struct Lin {
            enum { MASS = model_dae<Mo>::value ? 1 : 0 };      // (a local class: no static data member) semi-explicit DAE: mass matrix [M' 0; 0 I] of the adjoint system (M' alone for the lambda-only sensealgs)
            const double (&pv)[NP]; decltype(cur)& cu; SmallLU<N> lu; double y[N], gh, t;
            // BacksolveAdjoint (ALG == 1), z = [lam; mu; y]: the system is NOT affine in y.  Rosenbrock23 is a W-method (Shampine-Reichelt: its order does not depend on the
            // Jacobian being exact), and W is formed from the first-derivative blocks only — d(lam')/d lam = -J', d(mu')/d lam = -f_p', d(y')/dy = J; the second-derivative
            // blocks d(-J(y)' lam)/dy, d(-f_p(y)' lam)/dy are dropped (the reference's W carries them, by AD of the whole right-hand side: DESIGN 6, deliberate deviation) —
            // so W stays block triangular: two n x n factorisations and one substitution per step.
            SmallLU<ALG == 1 ? N : 1> luy;
            HIPADJ_HD void mulM(double (&out)[NZ], const double (&x)[NZ]) const {
#pragma unroll
                for (int r = 0; r < NZ; ++r) {
                    double s_ = x[r];
                    if constexpr (MASS) { if (r < N) { s_ = 0.0;
#pragma unroll
                        for (int c = 0; c < N; ++c) s_ += Mo::mass(c, r < N ? r : 0) * x[c]; } }
                    out[r] = s_;
                }
            }
            HIPADJ_HD void factor(double gh_, const double (&zc)[NZ], double t_) {
                gh = gh_; t = t_;
                if constexpr (ALG == 1) {
#pragma unroll
                    for (int j = 0; j < N; ++j) y[j] = zc[N + NP + j];      // the backsolved state
                } else cu.eval(t, y);
                double J[N][N];
                model_jacobian<Mo>(J, y, pv, t);                 // row c of J = column c of J'
#pragma unroll
                for (int c = 0; c < N; ++c) {
#pragma unroll
                    for (int r = 0; r < N; ++r) { double mrc = (r == c ? 1.0 : 0.0); if constexpr (MASS) mrc = Mo::mass(c, r); lu.a[r][c] = mrc + gh * J[c][r]; }
                    if constexpr (ALG == 1) {
#pragma unroll
                        for (int r = 0; r < N; ++r) luy.a[c][r] = (r == c ? 1.0 : 0.0) - gh * J[c][r];      // I - gh J, row c
                    }
                }
                lu.factor();
                if constexpr (ALG == 1) luy.factor();
            }
            HIPADJ_HD void solve(double (&b)[NZ]) const {
                double x[N];
#pragma unroll
                for (int j = 0; j < N; ++j) x[j] = b[j];
                lu.solve(x);
#pragma unroll
                for (int j = 0; j < N; ++j) b[j] = x[j];
                if constexpr (ALG == 0 || ALG == 1) {
                    double W[NP]; Mo::vjp_p(W, x, y, pv, t);
#pragma unroll
                    for (int j = 0; j < NP; ++j) b[N + j] -= gh * W[j];
                }
                if constexpr (ALG == 1) {
                    double xy[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) xy[j] = b[N + NP + j];
                    luy.solve(xy);
#pragma unroll
                    for (int j = 0; j < N; ++j) b[N + NP + j] = xy[j];
                }
            }
        } lin{pv, cur, {}, {}, 0.0, 0.0, {}};
        na = ros23_integrate<NZ>(z, ta_, tb_, dt_first, g.abstol, g.reltol, tstops_desc, ntstops, at_init, 8 * g.maxit, K, rhs, lin, false, cb, pre, model_has_cond<Mo>::value ? &dt_carry : nullptr);
    } else na = tsit5_integrate<NZ>(z, ta_, tb_, dt_first, g.abstol, g.reltol, tstops_desc, ntstops, at_init, 8 * g.maxit, K, rhs, cb, pre, TS5LaneNorm(), model_has_cond<Mo>::value ? &dt_carry : nullptr);
    return na;
    };
    int na = 0;
    if constexpr (model_has_cond<Mo>::value && CC == 0) {
        {
            // ContinuousCallback (the oracle's section 3b; src/callback_tracking.jl:232-479 with save_positions = (false, false)): the reverse solve runs piece by piece between
            // this trajectory's events (a piece starts from the step size the piece above ended with, dt_carry) and at each event, - / + the limits from below / above,
            //     kappa = lam+ . (a_u f- + a_t - f+) / (c_u . f- + c_t)      lam- = a_u' lam+ - kappa c_u      dp += a_p' lam+ - kappa c_p
            // A loss time that coincides with an event is taken at the end of the piece above it (it sees the affected state).
            // (ONE call site of the integrator: without events the loop runs once over the whole span)
            // BacksolveAdjoint (z = [lam; mu; y], no forward record): y+ is the backsolved state, y- the left state the forward solve stored; the y block goes on from y-
            const int nevl = g.maxev > 0 ? (g.nev[i] < g.maxev ? g.nev[i] : g.maxev) : 0;
            if constexpr (ALG != 1 && !CK) cur.smin = nevl > 0 ? g.ev_s[(long)(nevl - 1) * g.Npad + i] : 0;
            // terminate!: the last event ended the lane's forward solve — nothing lies above it: the piece (t*, T) is skipped (lam = 0 there), the loss and checkpoint times above t*
            // are passed over, and the jump at t* sees lam+ = 0 (the loss on the final state arrives as the event's dr, hipadj_set_event_cotangents)
            const bool term = nevl > 0 && (g.ev_k[(long)(nevl - 1) * g.Npad + i] & 256) != 0;
            if (term) {
                const double tte = g.ev_t[(long)(nevl - 1) * g.Npad + i];
                while (cur_time >= 1 && save_t[cur_time - 1] > tte && !time_hits(save_t[cur_time - 1], tte)) --cur_time;
                t_loss = cur_time >= 1 ? save_t[cur_time - 1] : 0.0;
                if (ALG == 1 && ckpt) { while (bs_cur >= 1 && ck_t[bs_cur - 1] > tte) --bs_cur; t_ck = bs_cur >= 1 ? ck_t[bs_cur - 1] : 0.0; }
            }
#pragma unroll 1
            for (int e = nevl; e >= 0; --e) {
                const double t_hi = (e == nevl) ? g.t1 : g.ev_t[(long)e * g.Npad + i];
                const double t_lo = e > 0 ? g.ev_t[(long)(e - 1) * g.Npad + i] : g.t0;
                if (!(term && e == nevl)) {
                const int r = run_piece(t_hi, t_lo, e == nevl && cb_at_init);
                if (r < 0) { na = -1; break; }
                na += r;
                }
                if (e == 0) break;
                double yp[N], ym[N], fm[N], fp[N], gu[N], gp[NP], jf[N], lo[N], go[NP], lamv[N], gt = 0.0;
                if constexpr (ALG == 1) {
#pragma unroll
                    for (int j = 0; j < N; ++j) { yp[j] = z[N + NP + j]; ym[j] = g.ev_ul[((long)(e - 1) * N + j) * g.Npad + i]; z[N + NP + j] = ym[j]; }
                } else if constexpr (CK) {
                    // checkpointing = true: the two limits as the forward lane stored them; the interval around the event is re-solved for the piece below it
#pragma unroll
                    for (int j = 0; j < N; ++j) { yp[j] = g.ev_ur[((long)(e - 1) * N + j) * g.Npad + i]; ym[j] = g.ev_ul[((long)(e - 1) * N + j) * g.Npad + i]; }
                    pc_hi = t_lo; pc_ev = e - 2; pc_lo = e >= 2 ? g.ev_t[(long)(e - 2) * g.Npad + i] : g.t0;
                    int jv = icur;
                    while (jv > 0 && !(t_lo > ck_t[jv])) --jv;
                    while (jv < g.nck - 2 && t_lo > ck_t[jv + 1]) ++jv;
                    resolve(jv, 0.0);
                } else {
                    const int sp = g.ev_s[(long)(e - 1) * g.Npad + i];
                    cur.eval(t_lo, yp);
                    cur.below(sp - 1, e >= 2 ? g.ev_s[(long)(e - 2) * g.Npad + i] : 0);
                    cur.eval(t_lo, ym);
                }
                const int kx = g.ev_k[(long)(e - 1) * g.Npad + i] & 255;      // the component that fired (0 for a scalar condition)
                Mo::f(fm, ym, pv, t_lo); Mo::f(fp, yp, pv, t_lo);
                Mo::cond_grad(gu, gp, gt, kx, ym, pv, t_lo);
                Mo::cc_affect_jvp(jf, ym, fm, pv, t_lo, kx);
                // a loss on the SAVED event states (save_positions = (true, true), src/callback_tracking.jl:385-401, 439-452): with dl / dr its cotangents at u- / u+,
                //     kappa = [lam+ . (a_u f- + a_t - f+) + dr . (a_u f- + a_t) + dl . f-] / (c_u . f- + c_t)      lam- = a_u' (lam+ + dr) + dl - kappa c_u      dp += a_p' (lam+ + dr) - kappa c_p
                // (the saved states sit AT the event time: they move with it along f- resp. a_u f- + a_t, not along the later flow)
                double num = 0.0, den = 0.0, dlv[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double drj = g.ev_dr ? g.ev_dr[((long)(e - 1) * N + j) * g.Npad + i] : 0.0;
                    dlv[j] = g.ev_dl ? g.ev_dl[((long)(e - 1) * N + j) * g.Npad + i] : 0.0;
                    lamv[j] = z[j] + drj; num += z[j] * (jf[j] - fp[j]) + drj * jf[j] + dlv[j] * fm[j]; den += gu[j] * fm[j]; }
                const double kappa = num / (den + gt);
                Mo::cc_affect_vjp(lo, go, lamv, ym, pv, t_lo, kx);
#pragma unroll
                for (int j = 0; j < N; ++j) z[j] = lo[j] + dlv[j] - kappa * gu[j];
#pragma unroll
                for (int j = 0; j < NP; ++j) { if constexpr (ALG == 0 || ALG == 1) z[N + j] += go[j] - kappa * gp[j]; else gacc[j] += go[j] - kappa * gp[j]; }
            }
        }
    } else na = run_piece(g.t1, g.t0, cb_at_init);
#pragma unroll
    for (int j = 0; j < N; ++j) lam_out[j] = z[j];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if constexpr (ALG == 2 || ALG == 4) mu_out[j] = gacc[j];
        else if constexpr (ALG == 3) mu_out[j] = gacc[j];  // dp comes from the quadrature pass; gacc: the sum of dgdp_discrete of a model's discrete loss (else zero)
        else mu_out[j] = z[N + j];
        if constexpr (model_dae<Mo>::value) mu_out[j] += dcorr[j];
    }
    if (ALG == 3) nsteps_adj[i] = sa;   // the TRUE count, also beyond the capacity: the host sizes the buffer from it (readers clamp with g.SmaxA)
    if (na < 0 || aoverflow || ck_overflow) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(flag, 4);
#else
        *flag |= 4;
#endif
    }
}

// ---- QuadratureAdjoint pass 2 on the adaptive solutions: quadgk(t -> f_p(y(t))^T lam(t) + g_p, a, b; atol, rtol) -----------
// (src/quadrature_adjoint.jl:486-502, 537-616).  Same Gauss-Kronrod (7,15) rule and worst-segment bisection as
// quad_gk_lane (hipadj_lane.hpp); the integrand reads both dense solutions through cursors.
template <int NP, class F>
HIPADJ_HD double gk15_eval_f(F&& integrand, double a, double b, double (&I)[NP]) {
    const double c = 0.5 * (a + b), h = 0.5 * (b - a);
    double Ig[NP], f1[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { I[j] = 0.0; Ig[j] = 0.0; }
    // the 15 nodes in ascending time: the two cursors behind `integrand` then only walk forward through their records
    // (the symmetric-pair order of QUADPACK would make them jump across the panel 14 times)
#pragma unroll 1
    for (int jn = 0; jn < 15; ++jn) {
        const int q = jn < 7 ? jn : (jn == 7 ? 7 : 14 - jn);
        integrand(c + h * (jn < 7 ? -GK15::X[q] : (jn == 7 ? 0.0 : GK15::X[q])), f1);
#pragma unroll
        for (int j = 0; j < NP; ++j) { I[j] += GK15::WK[q] * f1[j]; if (q & 1) Ig[j] += GK15::WG[q / 2] * f1[j]; }
    }
    double e = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        I[j] *= h; Ig[j] *= h;
        const double d = I[j] - Ig[j]; e += d * d;
    }
    return sqrt(e);
}

template <class Mo, int MAXSEG, int CC>
HIPADJ_HD void quad_gk_tsit5_lane(const AdaptGeom& g, long i, const double* __restrict__ p, const double* __restrict__ rec,
                                  const int* __restrict__ nsteps, const double* __restrict__ arec, const int* __restrict__ nsteps_adj,
                                  double a, double b, double atol, double rtol, double (&res)[Mo::NP]) {
    constexpr int N = Mo::N, NP = Mo::NP;
    double pv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) pv[j] = g.p_shared ? p[j] : p[i * NP + j];
    FwdCursor<Mo, false> cf; cf.init(rec, g.Npad, i, nsteps[i] < g.Smax ? nsteps[i] : g.Smax); cf.seek(0.5 * (a + b));
    AdjCursor<Mo> ca; ca.init(arec, g.Npad, i, nsteps_adj[i] < g.SmaxA ? nsteps_adj[i] : g.SmaxA, 0.5 * (a + b));
    auto integrand = [&](double t, double (&out)[NP]) {
        double y[N], lam[N];
        cf.eval(t, y); ca.eval(t, lam);
        Mo::vjp_p(out, lam, y, pv, t);
        if (cost_has_gp<CC>::value) {
            double gp[NP]; cost_grad_p<Mo, CC>(y, pv, t, gp);
#pragma unroll
            for (int j = 0; j < NP; ++j) out[j] += gp[j];
        }
    };
    // ContinuousCallback: lam (and y) jump at this trajectory's events — the interval is split there and every part integrated on its own (the nodes of the rule are interior
    // points: no evaluation lands on a jump); the parts share the interval's tolerances
    double acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = 0.0;
    // (the intervals are handed over ascending: a < b, hipadj_plan.hpp)
    int ke = 0, nevl = 0;
    if constexpr (model_has_cond<Mo>::value) { if (g.maxev > 0) nevl = g.nev[i] < g.maxev ? g.nev[i] : g.maxev; }
    double b_cap = b;                        // terminate!: lam = 0 above the event that ended the lane's solve (and no adjoint record exists there)
    if constexpr (model_has_cond<Mo>::value) {
        if (nevl > 0 && (g.ev_k[(long)(nevl - 1) * g.Npad + i] & 256)) { const double tte = g.ev_t[(long)(nevl - 1) * g.Npad + i]; if (tte < b_cap) b_cap = tte; }
        if (!(a < b_cap)) {
#pragma unroll
            for (int j = 0; j < NP; ++j) res[j] = 0.0;
            return; }
    }
#pragma unroll 1
    for (;;) {
    if constexpr (model_has_cond<Mo>::value) {
        while (ke < nevl && !(g.ev_t[(long)ke * g.Npad + i] > a)) ++ke;      // events at or below the part's start
        b = (ke < nevl && g.ev_t[(long)ke * g.Npad + i] < b_cap) ? g.ev_t[(long)ke * g.Npad + i] : b_cap;
    }
    double sa[MAXSEG], sb[MAXSEG], sE[MAXSEG], sI[MAXSEG][NP];
    double I[NP];
    int ns = 1;
    sa[0] = a; sb[0] = b;
    { double I0[NP]; sE[0] = gk15_eval_f<NP>(integrand, a, b, I0);
      for (int j = 0; j < NP; ++j) { sI[0][j] = I0[j]; I[j] = I0[j]; } }
    double E = sE[0];
    for (;;) {
        double nrm = 0.0;
        for (int j = 0; j < NP; ++j) nrm += I[j] * I[j];
        nrm = sqrt(nrm);
        const double tol = atol > rtol * nrm ? atol : rtol * nrm;
        if (E <= tol || ns + 1 > MAXSEG) break;
        int w = 0;
        for (int s = 1; s < ns; ++s) if (sE[s] > sE[w]) w = s;
        const double wa = sa[w], wb = sb[w], mid = 0.5 * (wa + wb);
        if (!(mid > (wa < wb ? wa : wb) && mid < (wa < wb ? wb : wa))) break;
        double I1[NP], I2[NP];
        const double E1 = gk15_eval_f<NP>(integrand, wa, mid, I1);
        const double E2 = gk15_eval_f<NP>(integrand, mid, wb, I2);
        for (int j = 0; j < NP; ++j) { I[j] += I1[j] + I2[j] - sI[w][j]; sI[w][j] = I1[j]; sI[ns][j] = I2[j]; }
        E += E1 + E2 - sE[w];
        sa[w] = wa; sb[w] = mid; sE[w] = E1;
        sa[ns] = mid; sb[ns] = wb; sE[ns] = E2;
        ++ns;
    }
    for (int j = 0; j < NP; ++j) { double s = 0.0; for (int q = 0; q < ns; ++q) s += sI[q][j]; acc[j] += s; }
    if constexpr (model_has_cond<Mo>::value) {
        if (b < b_cap) { a = b; continue; }
    }
    break;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) res[j] = acc[j];
}

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
// one lane = one trajectory; lanes of a wave take their own step sequences (accept/reject and the cursor walks
// diverge under the exec mask), a wave retires when its slowest trajectory does.  Stage storage: 8 x NZ x 64 doubles
// of LDS per wave (NZ = 6: 24.5 KB => 6 waves per CU).
template <class Mo, int STEP = 0>
__global__ void __launch_bounds__(64) k_forward_tsit5(AdaptGeom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                      double* __restrict__ rec, int* __restrict__ nsteps, const double* __restrict__ save_t,
                                                      double* __restrict__ outT, const double* __restrict__ ck_t, double* __restrict__ ckpt,
                                                      double* __restrict__ yT, int* __restrict__ flag) {
    __shared__ double ks[KS_ROWS * Mo::N * 64];
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= g.N) return;
    forward_tsit5_lane<Mo, STEP>(g, i, u0, p, rec, nsteps, save_t, outT, ck_t, ckpt, yT, flag, ks + threadIdx.x, 64);
}

template <class Mo, int ALG, int CC, bool CK = false, int STEP = 0>
__global__ void __launch_bounds__(64) k_adjoint_tsit5(AdaptGeom g, const double* __restrict__ p, const double* __restrict__ rec,
                                                      const int* __restrict__ nsteps, const double* __restrict__ yT, const double* __restrict__ ckpt,
                                                      const double* __restrict__ ck_t, const double* __restrict__ save_t,
                                                      const double* __restrict__ tstops_desc, int ntstops, const double* __restrict__ cotT,
                                                      double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag,
                                                      double* __restrict__ arec, int* __restrict__ nsteps_adj, int SmaxA) {
    __shared__ double ks[KS_ROWS * (AdjNZ<Mo, ALG>::value + (CK ? Mo::N : 0)) * 64];   // CK: + the stage rows of the interval re-solve
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= g.N) return;
    double lam[Mo::N], mu[Mo::NP];
    adjoint_tsit5_lane<Mo, ALG, CC, CK, STEP>(g, i, p, rec, nsteps, yT, ckpt, ck_t, save_t, tstops_desc, ntstops, cotT, lam, mu, flag, ks + threadIdx.x, 64, arec, nsteps_adj, SmaxA,
                                        ks + KS_ROWS * AdjNZ<Mo, ALG>::value * 64 + threadIdx.x, CK ? const_cast<double*>(rec) : nullptr);
#pragma unroll
    for (int j = 0; j < Mo::N; ++j) du0[i * Mo::N + j] = lam[j];
    if (ALG != 3 || model_has_dloss<Mo>::value || model_dae<Mo>::value || model_has_cond<Mo>::value) {   // QuadratureAdjoint: k_quad_sum writes dp_traj from the quadrature (and ADDS to it for a model with discrete-loss bodies or a semi-explicit DAE, whose loss jumps leave their parameter term here)
#pragma unroll
        for (int j = 0; j < Mo::NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[j]; }
}

// one lane per (trajectory, quadrature interval); qres [interval][NP][Npad]
template <class Mo, int CC>
__global__ void __launch_bounds__(64) k_quad_gk_tsit5(AdaptGeom g, const double* __restrict__ p, const double* __restrict__ rec,
                                                      const int* __restrict__ nsteps, const double* __restrict__ arec,
                                                      const int* __restrict__ nsteps_adj, const double* __restrict__ qa,
                                                      const double* __restrict__ qb, double atol, double rtol, double* __restrict__ qres) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    const int q = blockIdx.y;
    if (i >= g.N) return;
    double res[Mo::NP];
    quad_gk_tsit5_lane<Mo, 128, CC>(g, i, p, rec, nsteps, arec, nsteps_adj, qa[q], qb[q], atol, rtol, res);
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) qres[((long)q * Mo::NP + j) * g.Npad + i] = res[j];
}
#endif

}  // namespace hipadj
