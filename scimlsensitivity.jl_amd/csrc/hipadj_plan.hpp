// hipadj_plan.hpp — host-side planning shared by the C-ABI library and the test-only lane emulator:
// validates a hipadj_config and derives the step grid, the knot -> loss-time / checkpoint maps, the time
// segmentation and the quadrature interval list.  Pure C++ (no HIP).
//
// Reference behaviour restated here:
//   loss times become tstops of the reverse solve via PresetTimeCallback      src/adjoint_common.jl:848-855
//   default Backsolve checkpoints = sol.t of the saveat solve                 src/backsolve_adjoint.jl:132
//   QuadratureAdjoint interval order (end correction, t[i]..t[i+1] descending, start correction)
//                                                                             src/quadrature_adjoint.jl:563-616
#pragma once
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/hipadj.h"
#include "hipadj_models.hpp"

namespace hipadj {

struct Plan {
    int n = 0, np = 0;
    long N = 0, Npad = 0;
    int S = 0, M = 0, nck = 0, nseg = 1, nq = 0;
    int fgroup = 0, tree_radix = 4;   // grouped one-launch pass (k_interp_fused_g): waves per workgroup (0 = the plain form) and the radix of the HBM tree above the groups
    std::vector<double> save_times;
    std::vector<int> save_of_knot, ckpt_of_knot, seg_bounds;
    std::vector<int> save_of_knot_rev;   // the map the REVERSE kernels read: save_of_knot, minus the jump no_start suppresses when it sits at T (see make_plan)
    std::vector<double> qa, qb;
    bool bs_ckpt = false;
    bool ip_ckpt = false;    // Interpolating/Gauss with checkpointing=true: checkpoint tiles + in-kernel interval re-solve
    std::vector<int> prev_ck; // largest checkpoint knot < k
    int ck_longest = 0;      // longest checkpoint interval in steps (fixed-step checkpointed Interpolating/Gauss)
    bool field = false;      // workgroup-per-trajectory family (hipadj_field.hpp)
    bool mlp = false;        // FP64-MFMA family (hipadj_mlp.hpp)
    bool adaptive = false;   // adaptive Tsit5 (hipadj_adaptive.hpp)
    bool user = false;       // runtime-compiled right-hand side (hipadj_user.hpp)
    bool wide = false;       // ... of the workgroup-per-trajectory family (hipadj_wide.hpp): planned like the PDE family, BacksolveAdjoint included
    int Smax = 0;            // capacity of the per-trajectory dense solution (adaptive)
    int SmaxI = 0;           // adaptive + checkpointing=true (Interpolating/Gauss): record capacity of ONE checkpoint interval
    std::vector<double> ck_times, tstops_desc;   // adaptive: checkpoint times (ascending), reverse tstops (descending)
    int NQ = 0;              // activation records per step (MLP)
    // fixed-step RK4 with loss times off the step grid (InterpolatingAdjoint): the reverse step sequence, the same for every
    // trajectory (hipadj_lane.hpp, interp_offgrid_lane)
    bool offgrid = false;
    double h_last = 0.0;     // length of the last forward step (= dt unless the span is not a multiple of dt)
    std::vector<double> rs_t, rs_h, rs_te;
    std::vector<int> rs_save, rs_ck;
    int rs_save_at_start = -1;
    // ... and Interpolating / Gauss / GaussKronrod with checkpointing = true on top (offgrid_ckpt_lane, round 5): per checkpoint interval [c_j, c_{j+1}] the steps of its
    // re-solve, the length of the last one (it lands on c_{j+1}) and its share [q_lo, q_hi) of the reverse step list
    bool og_ck = false;
    std::vector<int> og_S, og_qlo, og_qhi;
    std::vector<double> og_hlast;
    int og_tile_knots = 0;
};

// Event knots of the fixed-step forward solve (forward_lane_ev, hipadj_lane.hpp): every knot where something besides the plain step
// happens — a save time (out = sol(ts) on the grid), a checkpoint, and knot S - 1 when the last step is shortened (span not a multiple of dt).
inline void forward_events(const Plan& P, double dt, std::vector<int>& knot, std::vector<int>& save, std::vector<int>& ckpt) {
    knot.clear(); save.clear(); ckpt.clear();
    const bool ragged = P.h_last != dt;
    for (int k = 0; k <= P.S; ++k) {
        const int sv = k < (int)P.save_of_knot.size() ? P.save_of_knot[k] : -1, ck = k < (int)P.ckpt_of_knot.size() ? P.ckpt_of_knot[k] : -1;
        if (sv >= 0 || ck >= 0 || (ragged && k == P.S - 1)) { knot.push_back(k); save.push_back(sv); ckpt.push_back(ck); }
    }
}

// The reverse solve of the reference on a fixed step with tstops at the loss times [upstream-recall, restated from the oracle's
// `integrate`]: dt = min(|dt|, |tstop - t|), a step whose remainder would be a roundoff sliver lands on the tstop, t snaps onto
// the tstop within 100 eps, and after a stop the solver continues with the full dt.  save[q] = the loss time that fires at the
// end of step q (PresetTimeCallback), honouring no_start (src/adjoint_common.jl:761).
inline void plan_reverse_steps(const hipadj_config* cfg, Plan& P) {
    const double EPS = 2.220446049250313e-16;
    const std::vector<double>& st = P.save_times;
    const bool bs = cfg->alg == HIPADJ_ALG_BACKSOLVE;      // Backsolve: no_start is not consulted (src/adjoint_common.jl:761 reads it for the others), checkpoint stops
    auto hits = [&](double a, double b) { return std::fabs(a - b) <= 100 * EPS * std::fmax(std::fabs(a), std::fabs(b)); };
    // tstops along the integration direction (descending): loss times and, with checkpoints (Backsolve; the checkpointed sweeps, src/sensitivity_interface.jl:484-486), their times
    std::vector<double> ts(st.rbegin(), st.rend());
    if (bs || P.og_ck) { ts.insert(ts.end(), P.ck_times.begin(), P.ck_times.end());
              for (size_t a = 1; a < ts.size(); ++a) { const double v = ts[a]; size_t b = a; while (b > 0 && ts[b - 1] < v) { ts[b] = ts[b - 1]; --b; } ts[b] = v; } }
    ts.push_back(cfg->t0);
    auto loss_at = [&](double t) {   // the callback's time test (within 100 eps), honouring no_start for the first loss time
        for (int i = 0; i < (int)st.size(); ++i) if (hits(st[i], t)) return (cfg->no_start && !bs && i == 0) ? -1 : i;
        return -1; };
    auto ck_at = [&](double t) { for (int i = 0; i < (int)P.ck_times.size(); ++i) if (hits(P.ck_times[i], t)) return i; return -1; };
    P.rs_t.clear(); P.rs_h.clear(); P.rs_te.clear(); P.rs_save.clear(); P.rs_ck.clear();
    P.rs_save_at_start = loss_at(cfg->t1);
    double t = cfg->t1;
    size_t its = 0;
    while (t > cfg->t0) {
        while (its < ts.size() && -ts[its] <= -t + 100 * EPS * std::fmax(std::fabs(t), std::fabs(ts[its]))) ++its;
        if (its >= ts.size()) break;
        const double tstop = ts[its];
        double d = -cfg->dt;
        if (std::fabs(d) > std::fabs(tstop - t)) d = tstop - t;
        if (std::fabs((t + d) - tstop) < 100 * EPS * std::fmax(std::fabs(t + d), std::fabs(tstop))) d = tstop - t;
        double tnew = t + d;
        if (std::fabs(tnew - tstop) < 100 * EPS * std::fmax(std::fabs(tnew), std::fabs(tstop))) tnew = tstop;
        P.rs_t.push_back(t); P.rs_h.push_back(-d); P.rs_te.push_back(tnew); P.rs_save.push_back(loss_at(tnew)); P.rs_ck.push_back(bs ? ck_at(tnew) : -1);
        t = tnew;
    }
}

// runtime-registered models (ids >= HIPADJ_MODEL_USER_BASE, hipadj_user.hpp): sizes come from the registry
typedef int (*plan_user_sizes_fn)(int32_t model, int32_t* n, int32_t* np);
inline plan_user_sizes_fn& plan_user_sizes_hook() { static plan_user_sizes_fn f = nullptr; return f; }
typedef bool (*plan_user_wide_fn)(int32_t);   // is this runtime model one of the workgroup-per-trajectory family (hipadj_wmodel_register)?
inline plan_user_wide_fn& plan_user_wide_hook() { static plan_user_wide_fn f = nullptr; return f; }
typedef bool (*plan_user_dae_fn)(int32_t);    // does this runtime model carry a SINGULAR mass matrix (semi-explicit DAE, hipadj_model_set_mass_matrix)?
inline plan_user_dae_fn& plan_user_dae_hook() { static plan_user_dae_fn f = nullptr; return f; }
typedef int (*plan_user_events_fn)(int32_t);  // capacity of the per-trajectory event list when this runtime model carries a ContinuousCallback (hipadj_model_set_continuous_callback), else 0
inline plan_user_events_fn& plan_user_events_hook() { static plan_user_events_fn f = nullptr; return f; }
inline bool plan_user_model(int m) { return m >= HIPADJ_MODEL_USER_BASE; }
inline int plan_user_events(int32_t model) { return (plan_user_model(model) && plan_user_events_hook()) ? plan_user_events_hook()(model) : 0; }
inline bool plan_small_model(int m) { return (m >= HIPADJ_MODEL_LV && m <= HIPADJ_MODEL_FALLMASS) || plan_user_model(m); }

inline int plan_model_sizes(int32_t model, const int32_t dims[4], int32_t* n, int32_t* np) {
    switch (model) {
    case HIPADJ_MODEL_LV: case HIPADJ_MODEL_LVT: *n = 2; *np = 4; return HIPADJ_OK;
    case HIPADJ_MODEL_LORENZ: *n = 3; *np = 3; return HIPADJ_OK;
    case HIPADJ_MODEL_LINDIAG: case HIPADJ_MODEL_FALLMASS: *n = 2; *np = 2; return HIPADJ_OK;
    case HIPADJ_MODEL_MLP:
        if (!dims || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) return HIPADJ_ERR_INVALID_ARG;
        *n = dims[0] * dims[2]; *np = dims[1] * dims[0] + dims[1] + dims[1] * dims[1] + dims[1] + dims[0] * dims[1] + dims[0];
        return HIPADJ_OK;
    case HIPADJ_MODEL_BRUSS:
        if (!dims || dims[0] <= 1) return HIPADJ_ERR_INVALID_ARG;
        *n = 2 * dims[0] * dims[0]; *np = 3; return HIPADJ_OK;
    default:
        if (plan_user_model(model) && plan_user_sizes_hook()) return plan_user_sizes_hook()(model, n, np);
        return HIPADJ_ERR_INVALID_ARG;
    }
}

// Widest segment state a lane carries in registers: (1 + n) columns of n + np doubles.  160 admits every runtime model up to n = 8 states with
// np = n + 1 parameters (9 x 17 = 153 doubles of 512 registers per lane); measured with the ring models, 10^4 trajectories x 1000 steps, 13
// segments against one: InterpolatingAdjoint 3.2x / 2.8x / 1.6x faster at n = 5 / 6 / 8 (profiles/r2_user_segments_ab.log).  The cap used to be 64
// because wider kernels came back wrong from the compiler then bound for runtime models.  HIPADJ_SEG_CAP is a tuning hook.
typedef int (*plan_segcap_fn)();   // runtime models: the cap the bound compiler is trusted with (hipadj_user.hpp: 64 when the toolkit's hiprtc could not be bound)
inline plan_segcap_fn& plan_user_segcap_hook() { static plan_segcap_fn f = nullptr; return f; }
inline int plan_seg_cap() {
    static const int cap = [] { const char* e = std::getenv("HIPADJ_SEG_CAP"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 160; }();
    if (plan_user_segcap_hook()) { const int c = plan_user_segcap_hook()(); if (c > 0 && c < cap && !std::getenv("HIPADJ_SEG_CAP")) return c; }
    return cap;
}
inline bool plan_seg_fits(int n, int np) { return (1 + n) * (n + np) <= plan_seg_cap(); }

// Number of time segments per trajectory for the linear (Interpolating) reverse pass: enough
// (trajectory-wave x segment) workgroups to put ~2 waves on each of the 1024 SIMDs of an MI355X, each segment
// keeping >= 16 steps.  A non-top segment carries 1 + n columns, so segmentation only pays once the added
// parallelism exceeds that factor.
inline int plan_auto_segments(long N, int S, int n, int np = 0) {
    const long waves = (N + 63) / 64;
    if (!plan_seg_fits(n, np)) return 1;   // a segment lane holds (1 + n) columns of n + np doubles in VGPRs
    // the segmented kernel needs ~250 VGPRs => 2 resident waves per SIMD => 2048 wave slots on 256 CUs x 4 SIMDs;
    // floor() keeps the grid within ONE residency round (a partial second round would double the makespan)
    long target = 2048 / waves;
    long maxseg = S / 16; if (maxseg < 1) maxseg = 1;
    if (target > maxseg) {
        // clipped by the 16-step minimum: between one and two residency rounds the SIMDs that hold two waves set the makespan, so prefer exactly
        // ONE wave per SIMD (1250 Lorenz trajectories x 1000 steps: 51 segments 40.0 us per reverse pass, 62 segments 44.0 us; profiles/r2_compose_w_segments.log)
        target = maxseg;
        if (waves * maxseg > 1024 && 1024 / waves >= 1) target = 1024 / waves < maxseg ? 1024 / waves : maxseg;
    }
    if ((double)target < 1.0 + n) return 1;
    return (int)target;
}

// The GROUPED one-launch pass (hipadj_fused.hpp; the stage-operator sweep of the compiled-in Lorenz model with shared parameters and the fused LSQ_SHIFT loss — BASELINE
// configs[1] and its shards): G consecutive segments per workgroup, first composition level in LDS.  Measured on MI355X, 1000 steps (profiles/r6_shard_group_ab.jsonl), best of
// the forms tried — 1250 trajectories (20 blocks): G = 4 x 12 groups, radix 4: 23.8 us (plain, 51 segments: 27.2); 2500 (40): G = 8 x 6, radix 8: 33.5 (40.5); 5000 (79):
// G = 8 x 3: 52.8 (61.4); 10^4 (157): G = 4 x 3: 107.4 (110.0).  The rule behind those: an 8-wave workgroup (84 KB of LDS, 2 waves per SIMD) owns a CU, so at most 256 of them;
// 4-wave workgroups pair up on a CU (<= 512), and a shard too small to fill the chip twice keeps one wave per SIMD (G = 4, <= 256 workgroups).  Returns false where nothing
// was measured to gain (fewer than 3 groups per block, segments shorter than 8 steps).
inline bool plan_group_choice(long N, int S, int& G, int& segs, int& radix) {
    const long blocks = (N + 63) / 64;
    long groups;
    if (blocks <= 25) { G = 4; groups = 256 / blocks; }
    else if (blocks <= 128) { G = 8; groups = 256 / blocks; }
    else { G = 4; groups = 512 / blocks; }
    while (groups > 1 && groups * G > S / (G == 8 ? 10 : 16)) --groups;      // segments of at least 16 steps (10 in the two-waves-per-SIMD form)
    if (groups > 16) groups = 16;                                            // one level of a radix-16 tree
    if (groups < 3) return false;
    segs = (int)(groups * G);
    radix = groups <= 4 ? 4 : (groups <= 8 ? 8 : 4);                          // 12 groups: 4 + 4 + 4 under a root of 3 (measured ahead of one radix-16 level, 23.8 vs 24.1 us)
    return true;
}

inline int plan_check_cost(const hipadj_config* cfg, std::string& err) {
    if (cfg->cont_cost < HIPADJ_CCOST_NONE || cfg->cont_cost > HIPADJ_CCOST_MODEL) { err = "unknown cont_cost"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->cont_cost == HIPADJ_CCOST_MODEL && !plan_user_model(cfg->model)) { err = "HIPADJ_CCOST_MODEL needs a runtime-registered model with hipadj_model_set_cost"; return HIPADJ_ERR_INVALID_ARG; }
    return HIPADJ_OK;
}

// the `checkpoints` list of the configuration (ncheckpoints > 0): strictly ascending inside [t0, t1], not combined with ckpt_stride
inline int plan_check_checkpoint_list(const hipadj_config* cfg, std::string& err) {
    if (cfg->ncheckpoints < 0 || (cfg->ncheckpoints > 0 && !cfg->checkpoints)) { err = "checkpoints missing"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ncheckpoints > 0 && cfg->ckpt_stride > 0) { err = "give either ckpt_stride or an explicit checkpoint list, not both"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ncheckpoints > 0 && cfg->alg == HIPADJ_ALG_QUADRATURE) { err = "QuadratureAdjoint has no checkpointing (src/sensitivity_algorithms.jl:1665-1677)"; return HIPADJ_ERR_INVALID_ARG; }
    const double slack = 1e-9 * std::fmax(1.0, std::fabs(cfg->t1 - cfg->t0));
    for (int i = 0; i < cfg->ncheckpoints; ++i) {
        if (!(cfg->checkpoints[i] >= cfg->t0 - slack && cfg->checkpoints[i] <= cfg->t1 + slack)) { err = "checkpoints must lie inside [t0, t1]"; return HIPADJ_ERR_INVALID_ARG; }
        if (i > 0 && !(cfg->checkpoints[i] > cfg->checkpoints[i - 1])) { err = "checkpoints must be strictly ascending"; return HIPADJ_ERR_INVALID_ARG; }
    }
    return HIPADJ_OK;
}

inline int make_plan(const hipadj_config* cfg, Plan& P, std::string& err) {
    if (!cfg) { err = "cfg == NULL"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->struct_size != sizeof(hipadj_config)) { err = "hipadj_config.struct_size mismatch (ABI)"; return HIPADJ_ERR_INVALID_ARG; }
    int32_t n, np;
    if (plan_model_sizes(cfg->model, cfg->dims, &n, &np) != HIPADJ_OK) { err = "unknown model id or bad dims"; return HIPADJ_ERR_INVALID_ARG; }
    P.field = cfg->model == HIPADJ_MODEL_BRUSS;
    P.mlp = cfg->model == HIPADJ_MODEL_MLP;
    P.user = plan_user_model(cfg->model);
    P.wide = P.user && plan_user_wide_hook() && plan_user_wide_hook()(cfg->model);
    if (!plan_small_model(cfg->model) && !P.field && !P.mlp) { err = "unknown model"; return HIPADJ_ERR_UNSUPPORTED; }
    if (P.wide) {   // what the wide family offers so far: fixed-step RK4, loss times on the step grid, the four sensealgs, discrete losses
        const bool ts5 = cfg->stepper == HIPADJ_STEPPER_TSIT5_ADAPTIVE;
        if (ts5 && (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_BACKSOLVE) && np > 8192) {
            err = "wide models: Interpolating- / BacksolveAdjoint on the adaptive solution keep five parameter-sized rows in LDS (np <= 8192 at most; the exact budget is checked when the handle is created) — GaussAdjoint has no such limit"; return HIPADJ_ERR_UNSUPPORTED; }
        if (!ts5 && cfg->checkpointing && cfg->alg == HIPADJ_ALG_QUADRATURE) { err = "wide models: QuadratureAdjoint keeps the dense forward solution (its second pass integrates over it); checkpointing = true is offered for Interpolating / Gauss / GaussKronrod / BacksolveAdjoint"; return HIPADJ_ERR_UNSUPPORTED; }
    }
    if (P.mlp) {
        if (cfg->dims[0] != 2) { err = "MLP family: state width d must be 2 (docs/src/Benchmark.md:62 shape)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->dims[1] != 32 && cfg->dims[1] != 64 && cfg->dims[1] != 128) { err = "MLP family: hidden width must be 32, 64 or 128 (the weight matrix lives in LDS)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->dims[2] % 16 != 0) { err = "MLP family: batch must be a multiple of 16 (one workgroup per 16 columns)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->alg != HIPADJ_ALG_GAUSS && cfg->alg != HIPADJ_ALG_INTERPOLATING && cfg->alg != HIPADJ_ALG_BACKSOLVE && cfg->alg != HIPADJ_ALG_QUADRATURE) { err = "MLP family offers Gauss-, Interpolating-, Backsolve- and QuadratureAdjoint"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->checkpointing && cfg->alg != HIPADJ_ALG_BACKSOLVE) { err = "MLP family: checkpointing = true is available for BacksolveAdjoint (the sweeps read the forward knots, 64 B per column and step)"; return HIPADJ_ERR_UNSUPPORTED; }
        P.NQ = cfg->alg == HIPADJ_ALG_GAUSS ? 2 : 4;
    }
    if (P.field) {
        if (cfg->dims[0] != 8 && cfg->dims[0] != 16 && cfg->dims[0] != 32) { err = "Brusselator grid must be 8, 16 or 32 (one workgroup per trajectory)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->alg == HIPADJ_ALG_BACKSOLVE) { err = "BacksolveAdjoint is not offered for the PDE family: backward diffusion is ill-posed (src/sensitivity_algorithms.jl:168-198)"; return HIPADJ_ERR_UNSUPPORTED; }
    }
    if (cfg->alg < HIPADJ_ALG_INTERPOLATING || cfg->alg > HIPADJ_ALG_GAUSS_KRONROD) { err = "unknown sensealg"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->alg == HIPADJ_ALG_GAUSS_KRONROD && (P.mlp || (P.field && cfg->stepper != HIPADJ_STEPPER_RK4_FIXED))) { err = "GaussKronrodAdjoint is offered for the lane-per-trajectory, wide and (RK4) PDE families"; return HIPADJ_ERR_UNSUPPORTED; }
    if (cfg->stepper != HIPADJ_STEPPER_RK4_FIXED && cfg->stepper != HIPADJ_STEPPER_TSIT5_ADAPTIVE && cfg->stepper != HIPADJ_STEPPER_ETDRK4_FIXED && cfg->stepper != HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) { err = "unknown stepper"; return HIPADJ_ERR_INVALID_ARG; }
    if (plan_user_model(cfg->model) && plan_user_dae_hook() && plan_user_dae_hook()(cfg->model)) {      // M u' = f with a singular M (src/adjoint_common.jl:117-135, 790-803)
        if (cfg->stepper != HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) { err = "the model's mass matrix is singular (a semi-explicit DAE): HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE integrates it in mass-matrix form; the explicit steppers cannot"; return HIPADJ_ERR_UNSUPPORTED; }
    }
    if (plan_user_events(cfg->model) > 0) {      // a ContinuousCallback (src/callback_tracking.jl:232-479): detected on the dense output of the adaptive steppers, per trajectory
        if (cfg->stepper != HIPADJ_STEPPER_TSIT5_ADAPTIVE && cfg->stepper != HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) { err = "the model carries a ContinuousCallback: events are located on the dense output of the adaptive steppers (HIPADJ_STEPPER_TSIT5_ADAPTIVE, HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->cont_cost != HIPADJ_CCOST_NONE) { err = "ContinuousCallback: no continuous cost"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->loss_kind == HIPADJ_LOSS_MODEL) { err = "ContinuousCallback: discrete losses by cotangents, HIPADJ_LOSS_LSQ_SHIFT or HIPADJ_LOSS_LSQ_DATA (the reference refuses dgdp with callbacks, src/callback_tracking.jl:289-290)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (plan_user_dae_hook() && plan_user_dae_hook()(cfg->model)) { err = "ContinuousCallback on a semi-explicit DAE is not offered"; return HIPADJ_ERR_UNSUPPORTED; }
    }
    if (cfg->stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) {   // the stiff stepper of the lane family (hipadj_adaptive.hpp ros23_integrate): planned like adaptive Tsit5 below
        if (!plan_small_model(cfg->model) || P.wide) { err = "Rosenbrock23 is available for the lane-per-trajectory models (n <= 8)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->alg == HIPADJ_ALG_BACKSOLVE && plan_user_model(cfg->model) && plan_user_dae_hook() && plan_user_dae_hook()(cfg->model)) { err = "Rosenbrock23: BacksolveAdjoint is not offered on a semi-explicit DAE (the reference documents it to fail there, test/Core3/adjoint.jl:1516-1530)"; return HIPADJ_ERR_UNSUPPORTED; }
        if (plan_user_model(cfg->model) && plan_user_dae_hook() && plan_user_dae_hook()(cfg->model)) {      // a semi-explicit DAE: its loss jump (src/adjoint_common.jl:790-813) is built for the plain loss routes
            if (cfg->cont_cost != HIPADJ_CCOST_NONE) { err = "Rosenbrock23 on a semi-explicit DAE: no continuous cost"; return HIPADJ_ERR_UNSUPPORTED; }
            if (cfg->loss_kind == HIPADJ_LOSS_MODEL) { err = "Rosenbrock23 on a semi-explicit DAE: discrete losses by cotangents, HIPADJ_LOSS_LSQ_SHIFT or HIPADJ_LOSS_LSQ_DATA"; return HIPADJ_ERR_UNSUPPORTED; }
        }
    }
    if (cfg->stepper == HIPADJ_STEPPER_ETDRK4_FIXED) {   // the exponential stepper: everything below treats it as a fixed-step scheme with Hermite dense output
        if (!P.field) { err = "HIPADJ_STEPPER_ETDRK4_FIXED integrates the semilinear PDE family (HIPADJ_MODEL_BRUSS): its linear part is diagonal in the DFT basis"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->alg != HIPADJ_ALG_INTERPOLATING && cfg->alg != HIPADJ_ALG_QUADRATURE && cfg->alg != HIPADJ_ALG_GAUSS) { err = "HIPADJ_STEPPER_ETDRK4_FIXED: Interpolating-, Gauss- and QuadratureAdjoint"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->cont_cost != 0) { err = "HIPADJ_STEPPER_ETDRK4_FIXED: no continuous cost"; return HIPADJ_ERR_UNSUPPORTED; }
    }
    if (cfg->stepper == HIPADJ_STEPPER_TSIT5_ADAPTIVE || cfg->stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) {
        // adaptive path: no step grid; arbitrary ascending loss times inside [t0, t1]
        if (!plan_small_model(cfg->model)) { err = "adaptive Tsit5 is available for the lane-per-trajectory models"; return HIPADJ_ERR_UNSUPPORTED; }
        if (cfg->alg == HIPADJ_ALG_QUADRATURE && cfg->checkpointing) { err = "QuadratureAdjoint has no checkpointing (src/sensitivity_algorithms.jl:1665-1677)"; return HIPADJ_ERR_INVALID_ARG; }
        if (cfg->ntraj <= 0) { err = "ntraj must be positive"; return HIPADJ_ERR_INVALID_ARG; }
        if (!(cfg->t1 > cfg->t0)) { err = "need t1 > t0"; return HIPADJ_ERR_INVALID_ARG; }
        if (!(cfg->abstol > 0) || !(cfg->reltol > 0)) { err = "adaptive Tsit5 needs abstol > 0 and reltol > 0"; return HIPADJ_ERR_INVALID_ARG; }
        if (cfg->nsave < 0 || (cfg->nsave > 0 && !cfg->save_times)) { err = "save_times missing"; return HIPADJ_ERR_INVALID_ARG; }
        if (cfg->loss_kind < HIPADJ_LOSS_COTANGENT || cfg->loss_kind > HIPADJ_LOSS_MODEL) { err = "unknown loss_kind"; return HIPADJ_ERR_INVALID_ARG; }
        if (cfg->loss_kind == HIPADJ_LOSS_MODEL && cfg->model < HIPADJ_MODEL_USER_BASE) { err = "loss_kind = HIPADJ_LOSS_MODEL needs a runtime-registered model with discrete-loss bodies (hipadj_model_set_discrete_loss); compiled-in models take HIPADJ_LOSS_LSQ_DATA"; return HIPADJ_ERR_INVALID_ARG; }
        { const int crc = plan_check_cost(cfg, err); if (crc != HIPADJ_OK) return crc; }
        if (cfg->max_steps < 0) { err = "max_steps must be >= 0"; return HIPADJ_ERR_INVALID_ARG; }
        if (!P.wide) {   // the 8 x NZ stage rows of a wave live in LDS (hipadj_adaptive.hpp): 8 * NZ * 64 lanes * 8 B <= 160 KB
            const int NZ = cfg->alg == HIPADJ_ALG_INTERPOLATING ? n + np : (cfg->alg == HIPADJ_ALG_BACKSOLVE ? 2 * n + np : n);   // Gauss, Quadrature: lam only
            const bool ipck = (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD) && cfg->checkpointing;   // + the rows of the interval re-solve
            if (8L * (NZ + (ipck ? n : 0)) * 64 * 8 > 160L * 1024) { err = "adaptive Tsit5: augmented state too large for the LDS stage storage (need 8 * NZ * 512 B <= 160 KB)"; return HIPADJ_ERR_UNSUPPORTED; }
        }
        P.adaptive = true; P.n = n; P.np = np; P.N = cfg->ntraj; P.Npad = ((cfg->ntraj + 63) / 64) * 64; P.S = 0; P.M = cfg->nsave;
        P.Smax = cfg->max_steps > 0 ? cfg->max_steps : 2048;
        P.save_times.assign(cfg->save_times, cfg->save_times + cfg->nsave);
        for (int i = 0; i < cfg->nsave; ++i) {
            if (!(cfg->save_times[i] >= cfg->t0 && cfg->save_times[i] <= cfg->t1)) { err = "save_times must lie inside [t0, t1]"; return HIPADJ_ERR_INVALID_ARG; }
            if (i > 0 && !(cfg->save_times[i] > cfg->save_times[i - 1])) { err = "save_times must be strictly ascending (duplicate event times are out of scope)"; return HIPADJ_ERR_INVALID_ARG; }
        }
        P.bs_ckpt = cfg->alg == HIPADJ_ALG_BACKSOLVE && cfg->checkpointing;
        P.ip_ckpt = (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD) && cfg->checkpointing;
        P.ck_times.clear();
        { const int crc = plan_check_checkpoint_list(cfg, err); if (crc != HIPADJ_OK) return crc; }
        if (P.bs_ckpt || P.ip_ckpt) {   // default checkpoints = sol.t of the saveat solve: t0, save times, t1 (src/backsolve_adjoint.jl:132); or the caller's list
            std::vector<double> src = cfg->ncheckpoints > 0 ? std::vector<double>(cfg->checkpoints, cfg->checkpoints + cfg->ncheckpoints) : P.save_times;
            if (src.empty() || src.front() > cfg->t0) P.ck_times.push_back(cfg->t0);
            for (double t : src) P.ck_times.push_back(t);
            if (P.ck_times.back() < cfg->t1) P.ck_times.push_back(cfg->t1);
        }
        P.nck = (int)P.ck_times.size();
        if (P.ip_ckpt) { const int per = 2 * P.Smax / (P.nck > 1 ? P.nck - 1 : 1); P.SmaxI = per < 64 ? 64 : per; }   // twice the average share of an interval
        // reverse tstops: loss times (PresetTimeCallback) + checkpoint times, descending
        std::vector<double> ts(P.save_times); ts.insert(ts.end(), P.ck_times.begin(), P.ck_times.end());
        for (size_t a = 1; a < ts.size(); ++a) { const double v = ts[a]; size_t b = a; while (b > 0 && ts[b - 1] < v) { ts[b] = ts[b - 1]; --b; } ts[b] = v; }
        P.tstops_desc = ts;
        P.nseg = 1; P.seg_bounds.assign(2, 0);
        P.save_of_knot.assign(1, -1); P.save_of_knot_rev = P.save_of_knot; P.ckpt_of_knot.assign(1, -1); P.prev_ck.assign(1, 0);
        P.qa.clear(); P.qb.clear();
        if (cfg->alg == HIPADJ_ALG_QUADRATURE) {   // interval order of src/quadrature_adjoint.jl:563-616 (end correction, pairs descending, start correction)
            const auto& t = P.save_times;
            if (t.empty()) { P.qa.push_back(cfg->t0); P.qb.push_back(cfg->t1); }
            else {
                if (t.back() != cfg->t1) { P.qa.push_back(t.back()); P.qb.push_back(cfg->t1); }
                for (int i = (int)t.size() - 2; i >= 0; --i) { P.qa.push_back(t[i]); P.qb.push_back(t[i + 1]); }
                if (t.front() != cfg->t0) { P.qa.push_back(cfg->t0); P.qb.push_back(t.front()); }
            }
        }
        P.nq = (int)P.qa.size();
        return HIPADJ_OK;
    }
    if (cfg->ntraj <= 0) { err = "ntraj must be positive"; return HIPADJ_ERR_INVALID_ARG; }
    if (!(cfg->dt > 0) || !(cfg->t1 > cfg->t0)) { err = "need dt > 0 and t1 > t0"; return HIPADJ_ERR_INVALID_ARG; }
    // Number of forward steps.  A span that is not a multiple of dt ends with a shortened step, as the reference's fixed-step solve does
    // (dt = min(dt, tend - t)): S = ceil, h_last = the remainder; the reverse solve then starts from T with the full dt, so its steps never
    // coincide with the forward knots and the configuration runs the off-grid sweeps (lane-per-trajectory models; forced below).
    const double sreal = (cfg->t1 - cfg->t0) / cfg->dt; long S = std::lround(sreal);
    bool ragged = false;
    if (S < 1 || std::fabs(sreal - (double)S) > 1e-6 * (double)(S < 1 ? 1 : S)) {
        S = (long)std::floor(sreal * (1.0 + 1e-12)) + 1;
        ragged = sreal > 0.0;
    }
    if (S < 1 || S > 100000000L) { err = "(t1 - t0)/dt must give between 1 and 1e8 steps"; return HIPADJ_ERR_INVALID_ARG; }
    P.h_last = ragged ? (cfg->t1 - cfg->t0) - (double)(S - 1) * cfg->dt : cfg->dt;
    if (ragged && (P.field || P.mlp || P.wide)) { err = "a span that is not a multiple of dt (shortened last step) is offered for the lane-per-trajectory models"; return HIPADJ_ERR_UNSUPPORTED; }
    if (cfg->nsave < 0 || (cfg->nsave > 0 && !cfg->save_times)) { err = "save_times missing"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->loss_kind < HIPADJ_LOSS_COTANGENT || cfg->loss_kind > HIPADJ_LOSS_MODEL) { err = "unknown loss_kind"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->loss_kind == HIPADJ_LOSS_MODEL && cfg->model < HIPADJ_MODEL_USER_BASE) { err = "loss_kind = HIPADJ_LOSS_MODEL needs a runtime-registered model with discrete-loss bodies (hipadj_model_set_discrete_loss); compiled-in models take HIPADJ_LOSS_LSQ_DATA"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->time_segments < 0) { err = "time_segments must be >= 0"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ckpt_stride < 0) { err = "ckpt_stride must be >= 0"; return HIPADJ_ERR_INVALID_ARG; }
    { const int crc = plan_check_checkpoint_list(cfg, err); if (crc != HIPADJ_OK) return crc; }
    { const int crc = plan_check_cost(cfg, err); if (crc != HIPADJ_OK) return crc; }
    if (cfg->cont_cost != HIPADJ_CCOST_NONE && P.mlp) { err = "continuous costs are available for the lane-per-trajectory, wide and PDE families"; return HIPADJ_ERR_UNSUPPORTED; }
    if (cfg->cont_cost != HIPADJ_CCOST_NONE && P.field && cfg->cont_cost != HIPADJ_CCOST_HALF_SQ_SUM && cfg->cont_cost != HIPADJ_CCOST_U1SQ_PLUS_P1) {
        err = "the PDE family takes the built-in continuous costs (HIPADJ_CCOST_HALF_SQ_SUM, HIPADJ_CCOST_U1SQ_PLUS_P1)"; return HIPADJ_ERR_UNSUPPORTED; }
    P.n = n; P.np = np; P.N = cfg->ntraj; P.Npad = ((cfg->ntraj + 63) / 64) * 64; P.S = (int)S; P.M = cfg->nsave;
    P.save_of_knot.assign(S + 1, -1); P.ckpt_of_knot.assign(S + 1, -1);
    P.save_times.assign(cfg->save_times, cfg->save_times + cfg->nsave);
    for (int i = 0; i < cfg->nsave; ++i) {
        const double kr = (cfg->save_times[i] - cfg->t0) / cfg->dt; const long k = std::lround(kr);
        if (!(cfg->save_times[i] >= cfg->t0 - 1e-6 * cfg->dt && cfg->save_times[i] <= cfg->t1 + 1e-6 * cfg->dt)) { err = "save_times must lie inside [t0, t1]"; return HIPADJ_ERR_INVALID_ARG; }
        if (i > 0 && !(cfg->save_times[i] > cfg->save_times[i - 1])) { err = "save_times must be strictly ascending (duplicate event times are out of scope)"; return HIPADJ_ERR_INVALID_ARG; }
        if (k < 0 || k > S || std::fabs(kr - (double)k) > 1e-6) P.offgrid = true;
        else P.save_of_knot[k] = i;
    }
    if (ragged) P.offgrid = true;   // the reverse steps leave the knots right from T
    if (!P.offgrid && cfg->alg == HIPADJ_ALG_GAUSS_KRONROD && cfg->checkpointing && !P.wide) {
        err = "GaussKronrodAdjoint(checkpointing=true) with the loss times on the step grid is offered for wide models; the lane family has it with adaptive Tsit5 and over the reverse step list (loss times off the grid)"; return HIPADJ_ERR_UNSUPPORTED; }
    if (P.offgrid) {
        // loss times off the step grid t0 + k*dt: the reverse steps leave the forward knots (hipadj_lane.hpp, interp_offgrid_lane)
        const bool og_gk = cfg->alg == HIPADJ_ALG_GAUSS_KRONROD && !(P.wide && cfg->checkpointing);   // round 5: gauss_offgrid_lane's GKR branch (sequential in time) / k_wide_adjoint_og<., 4>
        const bool ig_alg = cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS;
        P.og_ck = (ig_alg || og_gk) && cfg->checkpointing && !P.wide;         // lane models (round 5): offgrid_ckpt_lane, sequential in time
        const bool og_ig = (ig_alg && !cfg->checkpointing) || og_gk || P.og_ck;
        const bool og_bs = cfg->alg == HIPADJ_ALG_BACKSOLVE;    // any checkpoint choice (round 5: ckpt_stride and explicit lists too — the reverse step list stops at whatever times they name)
        const bool og_q = cfg->alg == HIPADJ_ALG_QUADRATURE && !P.wide;       // lane models: compiled-in (round 2) and runtime-registered (round 5)
        // wide models: Interpolating / Gauss (k_wide_adjoint_og), Backsolve (k_wide_backsolve_og) and Quadrature (k_wide_quad_adj_og + the GK pass over the reverse step list, round 5)
        const bool og_wide = P.wide && (og_ig || og_bs || (cfg->alg == HIPADJ_ALG_QUADRATURE && !cfg->checkpointing));
        if (!(og_ig || og_bs || og_q || og_wide) || P.field || P.mlp || (P.wide && !og_wide)) {
            err = "save_times off the step grid t0 + k*dt are offered for InterpolatingAdjoint / GaussAdjoint / GaussKronrodAdjoint (checkpointing = true over the reverse step list: lane models), QuadratureAdjoint and "
                  "BacksolveAdjoint on the lane-per-trajectory and wide models; other configurations need times on the step grid, or the adaptive stepper (arbitrary times)";
            return HIPADJ_ERR_UNSUPPORTED; }
        for (int i = 0; i < cfg->nsave; ++i) {   // the sweep takes the times literally; they must not leave the span
            if (P.save_times[i] < cfg->t0) P.save_times[i] = cfg->t0;
            if (P.save_times[i] > cfg->t1) P.save_times[i] = cfg->t1;
        }
        P.save_of_knot.assign(S + 1, -1);   // no fused select on the knots: jumps are driven by the reverse step list
        P.ck_times.clear();
        if ((og_bs || P.og_ck) && cfg->checkpointing) {  // default checkpoints = sol.t of the saveat solve: t0, the save times, t1 (src/backsolve_adjoint.jl:132); or the caller's list
                                            // (src/sensitivity_interface.jl:484-486: any times inside the span), or every ckpt_stride-th knot of the forward grid
            std::vector<double> src;
            if (cfg->ncheckpoints > 0) src.assign(cfg->checkpoints, cfg->checkpoints + cfg->ncheckpoints);
            else if (cfg->ckpt_stride > 0) { for (long k = 0; k < S; k += cfg->ckpt_stride) src.push_back(cfg->t0 + (double)k * cfg->dt); }
            else src = P.save_times;
            for (double& t : src) { if (t < cfg->t0) t = cfg->t0; if (t > cfg->t1) t = cfg->t1; }
            if (src.empty() || src.front() > cfg->t0) P.ck_times.push_back(cfg->t0);
            for (double t : src) if (P.ck_times.empty() || t > P.ck_times.back()) P.ck_times.push_back(t);
            if (P.ck_times.back() < cfg->t1) P.ck_times.push_back(cfg->t1);
        }
        plan_reverse_steps(cfg, P);
        if (P.og_ck) {
            const double EPS = 2.220446049250313e-16;
            const int nint = (int)P.ck_times.size() - 1;
            P.og_S.assign(nint, 0); P.og_hlast.assign(nint, cfg->dt); P.og_qlo.assign(nint, 0); P.og_qhi.assign(nint, 0); P.og_tile_knots = 0;
            for (int j = 0; j < nint; ++j) {   // the re-solve of [c_j, c_{j+1}]: full dt-steps, the last one shortened onto c_{j+1}; a remainder that is a roundoff sliver joins the step before it
                const double a = P.ck_times[j], b = P.ck_times[j + 1], L = b - a;
                long Sj = (long)std::floor(L / cfg->dt);
                if (Sj < 1) Sj = 0;
                const double rem = L - (double)Sj * cfg->dt;
                if (Sj == 0 || rem > 100 * EPS * std::fmax(std::fabs(a), std::fabs(b))) Sj += 1;
                P.og_S[j] = (int)Sj; P.og_hlast[j] = L - (double)(Sj - 1) * cfg->dt;
                if ((int)Sj + 1 > P.og_tile_knots) P.og_tile_knots = (int)Sj + 1;
            }
            int q = 0;
            const int nrs = (int)P.rs_t.size();
            for (int j = nint - 1; j >= 0; --j) {   // every checkpoint is a stop: a reverse step lies inside one interval
                P.og_qlo[j] = q;
                while (q < nrs && P.rs_t[q] - 0.5 * P.rs_h[q] > P.ck_times[j]) ++q;
                P.og_qhi[j] = q;
                if (P.og_qhi[j] <= P.og_qlo[j]) { err = "two checkpoint times are closer than the time resolution of the reverse solve (100 eps): no step falls between them"; return HIPADJ_ERR_INVALID_ARG; }
            }
        }
    }
    // checkpoints: BacksolveAdjoint only.  Interpolating/Gauss checkpointing re-solves, on this fixed grid,
    // bit-identical knots from the stored values; the dense tiles are kept instead.
    P.bs_ckpt = cfg->alg == HIPADJ_ALG_BACKSOLVE && cfg->checkpointing;
    P.ip_ckpt = (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS || (P.wide && cfg->alg == HIPADJ_ALG_GAUSS_KRONROD)) && cfg->checkpointing && !P.field && !P.mlp;   // wide models: k_wide_adjoint_ck
    P.nck = P.offgrid ? (int)P.ck_times.size() : 0;   // off-grid Backsolve: checkpoint TIMES (interpolated states), not knots
    if ((P.bs_ckpt || P.ip_ckpt) && !P.offgrid) {
        int c = 0;
        if (cfg->ncheckpoints > 0) {
            // explicit list (adjoint_sensitivities(...; checkpoints), src/sensitivity_interface.jl:484-486): any spacing, every time a knot;
            // t0 and T are checkpoints as well (the interval construction of src/interpolating_adjoint.jl:54-58 adds the tail up to T,
            // the forward solution starts at t0)
            std::vector<char> is_ck((size_t)S + 1, 0);
            is_ck[0] = is_ck[S] = 1;
            for (int i = 0; i < cfg->ncheckpoints; ++i) {
                const double kr = (cfg->checkpoints[i] - cfg->t0) / cfg->dt; const long k = std::lround(kr);
                if (k < 0 || k > S || std::fabs(kr - (double)k) > 1e-6) { err = "fixed-step RK4: checkpoints must lie on the step grid t0 + k*dt (arbitrary times: the adaptive stepper)"; return HIPADJ_ERR_UNSUPPORTED; }
                is_ck[k] = 1;
            }
            for (long k = 0; k <= S; ++k) if (is_ck[k]) P.ckpt_of_knot[k] = c++;
        }
        else if (cfg->ckpt_stride > 0) { for (long k = 0; k <= S; k += cfg->ckpt_stride) P.ckpt_of_knot[k] = c++; if (P.ckpt_of_knot[S] < 0) P.ckpt_of_knot[S] = c++; }
        else { for (long k = 0; k <= S; ++k) if (k == 0 || k == S || P.save_of_knot[k] >= 0) P.ckpt_of_knot[k] = c++; }
        P.nck = c;
    }
    if (P.ip_ckpt && !P.og_ck && P.user && !P.wide && (1 + n) * (n + np) > 64) {   // the checkpointed sweeps carry the 1 + n segment columns in VGPRs next to the re-solve state (not re-measured beyond 64)
        err = "checkpointing=true for Interpolating/Gauss on the fixed step needs (1 + n)(n + np) <= 64 for a runtime-compiled model (wider models: the adaptive stepper)"; return HIPADJ_ERR_UNSUPPORTED; }
    P.prev_ck.assign(S + 1, 0);
    if (P.ip_ckpt) {
        int last = 0, longest = 0;
        for (long k = 1; k <= S; ++k) { P.prev_ck[k] = last; if (P.ckpt_of_knot[k] >= 0) { if ((int)k - last > longest) longest = (int)k - last; last = (int)k; } }
        P.ck_longest = longest;   // <= HIPADJ_CKPT_KMAX: the re-solve tile lives in LDS; longer intervals: a per-wave slice of an HBM scratch buffer
    }
    P.nseg = 1;
    const bool seg_alg = !P.field && !P.mlp && !P.wide && (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS || cfg->alg == HIPADJ_ALG_GAUSS_KRONROD || (cfg->alg == HIPADJ_ALG_BACKSOLVE && P.bs_ckpt));
    // off-grid Interpolating / Gauss: the reverse STEP LIST is what gets segmented (it does not depend on the
    // trajectory); bounds are then positions r = nrs - q in that list instead of knot indices (k_offgrid_seg)
    const bool seg_offgrid = P.offgrid && !P.og_ck && (!P.user || plan_seg_fits(n, np)) && (cfg->alg == HIPADJ_ALG_INTERPOLATING || cfg->alg == HIPADJ_ALG_GAUSS);
    const long L = seg_offgrid ? (long)P.rs_t.size() : S;      // length of the axis the segments cut
    if (seg_alg && (!P.offgrid || seg_offgrid)) {
        P.nseg = cfg->time_segments == 0 ? plan_auto_segments(P.N, (int)L, n, np) : cfg->time_segments;
        {   // the grouped one-launch pass where it was measured to win (plan_group_choice); HIPADJ_FUSED_GROUP = 0 keeps the plain form, 4 / 8 force a group size on the
            // segment count given (A/B runs, tests)
            const bool eligible = cfg->model == HIPADJ_MODEL_LORENZ && cfg->alg == HIPADJ_ALG_INTERPOLATING && cfg->p_shared && cfg->loss_kind != HIPADJ_LOSS_MODEL &&
                                  cfg->cont_cost == HIPADJ_CCOST_NONE && !cfg->checkpointing && !P.offgrid && !std::getenv("HIPADJ_NO_OPS") && !std::getenv("HIPADJ_WPB");
            const char* e = std::getenv("HIPADJ_FUSED_GROUP"); const int forced = e ? std::atoi(e) : -1;
            if (eligible && forced != 0) {
                int G = 0, segs = 0, radix = 4;
                if (forced == 4 || forced == 8) { P.fgroup = forced; if (const char* r = std::getenv("HIPADJ_TREE_RADIX")) { const int v = std::atoi(r); P.tree_radix = (v == 8 || v == 16) ? v : 4; } }
                else if (cfg->time_segments == 0 && plan_group_choice(P.N, (int)L, G, segs, radix)) { P.fgroup = G; P.nseg = segs; P.tree_radix = radix; }
            }
        }
        if (!plan_seg_fits(n, np)) P.nseg = 1;   // segment lanes would not fit the register file
        if (P.nseg > L) P.nseg = (int)L;
        if (P.nseg < 1) P.nseg = 1;
    }
    // the top segment carries 1 column, the others 1 + n: give the top segment a proportionally longer span
    {
        const int C = P.nseg; P.seg_bounds.assign(C + 1, 0);
        if (C == 1) { P.seg_bounds[1] = (int)L; }
        else {
            // a 1-column lane advances ~w_top steps per (1+n)-column step: the ratio of the two step bodies' VALU instruction counts
            // (Lorenz, stage-operator form of the multi-column step with shared parameters: 263 : 101, profiles/README.md round 2)
            double w_top = 1.0 + 0.8 * n;
            if (cfg->model == HIPADJ_MODEL_LORENZ && cfg->p_shared && cfg->alg == HIPADJ_ALG_INTERPOLATING && cfg->cont_cost == HIPADJ_CCOST_NONE && !cfg->checkpointing) w_top = 2.6;
            // grouped form with ONE wave per SIMD (G = 4 on a small shard): a lone wave issues one instruction per ~5.5 cycles whatever its kind, so the one-column step is
            // relatively dearer than its instruction count says — and the top segment's wave is the one that folds its group (1250 trajectories, 48 segments: w_top 1.8 / 2.1 /
            // 2.4 / 2.6 / 3.0 -> 23.3 / 23.1 / 23.5 / 24.0 / 25.4 us; two waves per SIMD are flat between 1.8 and 2.6: profiles/r6_wtop_grouped.jsonl)
            if (P.fgroup == 4 && (P.N + 63) / 64 <= 25 && cfg->time_segments == 0) w_top = 2.1;
            if (seg_offgrid) w_top = 1.0 + 0.27 * n;   // the general-theta Hermite evaluations and the cursor walk of a step are shared by its columns, so a 1-column step is relatively dearer (Lorenz, measured: 1.8 best of 1.8 / 2.1 / 2.35 / 2.7 / 3.4)
            if (const char* e = std::getenv("HIPADJ_WTOP")) { const double v = std::atof(e); if (v > 0) w_top = v; }   // tuning hook
            const double unit = (double)L / ((C - 1) + w_top);
            double acc = 0.0;
            for (int s = 1; s < C; ++s) { acc += unit; int b = (int)std::lround(acc); if (b <= P.seg_bounds[s - 1]) b = P.seg_bounds[s - 1] + 1; P.seg_bounds[s] = b; }
            P.seg_bounds[C] = (int)L;
            for (int s = C - 1; s >= 1; --s) if (P.seg_bounds[s] >= P.seg_bounds[s + 1]) P.seg_bounds[s] = P.seg_bounds[s + 1] - 1;
        }
    }
    if ((cfg->alg == HIPADJ_ALG_BACKSOLVE || P.ip_ckpt) && P.nseg > 1) {
        // Backsolve segments may only be cut where y is known independently of the segments above: at checkpoint
        // knots.  Snap every interior bound to the nearest checkpoint knot and drop duplicates.
        std::vector<int> ck;
        for (int k = 1; k < (int)S; ++k) if (P.ckpt_of_knot[k] >= 0) ck.push_back(k);
        std::vector<int> b; b.push_back(0);
        for (int s = 1; s < P.nseg; ++s) {
            int best = -1; long bd = 0;
            for (int k : ck) { const long d = std::labs((long)k - P.seg_bounds[s]); if (best < 0 || d < bd) { best = k; bd = d; } }
            if (best > b.back()) b.push_back(best);
        }
        b.push_back((int)S);
        P.seg_bounds = b; P.nseg = (int)b.size() - 1;
    }
    // no_start drops the jump of the FIRST loss time wherever it lies (`cur_time == 1 && no_start`, src/adjoint_common.jl:761; not
    // for Backsolve).  The sweeps test it on every step (`s == 0`), but the jump that fires at initialisation (first loss time
    // == T, i.e. the only one) is applied unconditionally by the kernels: the reverse pass therefore reads a copy of the map with
    // that entry cleared.  The forward pass keeps the full map (out = sol(ts) still has the column).
    P.save_of_knot_rev = P.save_of_knot;
    if (cfg->no_start && cfg->alg != HIPADJ_ALG_BACKSOLVE && P.save_of_knot[S] == 0) P.save_of_knot_rev[S] = -1;
    P.qa.clear(); P.qb.clear();
    if (cfg->alg == HIPADJ_ALG_QUADRATURE) {
        const auto& t = P.save_times;
        if (t.empty()) { P.qa.push_back(cfg->t0); P.qb.push_back(cfg->t1); }
        else {
            // end / start corrections when T / t0 is not a loss time (:563-616); off the grid the knot map is empty, so the times decide
            const bool last_is_T = P.offgrid ? !(t.back() < cfg->t1) : P.save_of_knot[S] >= 0;
            const bool first_is_t0 = P.offgrid ? !(t.front() > cfg->t0) : P.save_of_knot[0] >= 0;
            if (!last_is_T) { P.qa.push_back(t.back()); P.qb.push_back(cfg->t1); }
            for (int i = (int)t.size() - 2; i >= 0; --i) { P.qa.push_back(t[i]); P.qb.push_back(t[i + 1]); }
            if (!first_is_t0) { P.qa.push_back(cfg->t0); P.qb.push_back(t.front()); }
        }
    }
    P.nq = (int)P.qa.size();
    return HIPADJ_OK;
}

}  // namespace hipadj
