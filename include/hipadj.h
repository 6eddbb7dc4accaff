/*
 * hipadj.h — C ABI of libhipadj: MI355X (gfx950) batched continuous-adjoint sensitivity engine.
 *
 * Drop-in boundary: the WHOLE reverse pass of a WHOLE ensemble sits behind
 * these entry points.  Each entry point names the reference interface it replaces (paths relative to the
 * SciMLSensitivity.jl tree).  The reference has no FFI of its own (pure Julia); the Julia-side binding a
 * maintainer would add is the `ccall` stub in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, Float64, caller-owned buffers, row-major as stated per argument.
 * Every function returns HIPADJ_OK (0) or a negative hipadj_status; nothing throws across the ABI;
 * hipadj_last_error() gives the message.  A handle is not thread-safe; distinct handles are independent
 * (the reference gets concurrency from many solves on Julia threads, test/Core4/ensembles.jl:16-20).
 * There is NO CPU fallback: hipadj_create fails with HIPADJ_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef HIPADJ_H
#define HIPADJ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPADJ_VERSION 110 /* 0.1.8: + hipadj_model_set_continuous_callback, hipadj_model_set_vector_continuous_callback, hipadj_model_set_callback_direction, hipadj_event_counts, hipadj_event_states, hipadj_event_components, hipadj_set_event_cotangents (ContinuousCallback on the adaptive lane steppers); additive, no struct changed. History of 109 and earlier: docs/ABI_HISTORY.md */

typedef enum {
    HIPADJ_OK = 0,
    HIPADJ_ERR_INVALID_ARG = -1,  /* mirrors the reference's `error(...)` on misuse, src/interpolating_adjoint.jl:321-326 */
    HIPADJ_ERR_NO_DEVICE = -2,
    HIPADJ_ERR_HIP = -3,
    HIPADJ_ERR_NONFINITE = -4,    /* a trajectory produced NaN/Inf — the reference's retcode checks */
    HIPADJ_ERR_STATE = -5,        /* adjoint requested before forward */
    HIPADJ_ERR_UNSUPPORTED = -6,
    HIPADJ_ERR_MAXITERS = -7,     /* adaptive solve ran out of steps (max_steps) — the reference's ReturnCode.MaxIters */
    HIPADJ_ERR_RCCL = -8          /* RCCL missing or a collective failed (hipadj_comm_*) */
} hipadj_status;

/* compile-time model registry: device-inlined f, (df/du)^T lam, (df/dp)^T lam — the reference's user-VJP seam
 * ODEFunction(f; vjp, vjp_p)  (src/derivative_wrappers.jl:284-359, test/Core3/user_vjp.jl:14-38) */
typedef enum {
    HIPADJ_MODEL_LV = 0,       /* Lotka-Volterra n=2 np=4 (test/Core3/user_vjp.jl:6-10) */
    HIPADJ_MODEL_LVT = 1,      /* time-dependent LV `fb` (test/Core3/adjoint.jl:8-12) */
    HIPADJ_MODEL_LORENZ = 2,   /* Lorenz-63 n=3 np=3 (test/Core3/adjoint.jl:1160-1166) */
    HIPADJ_MODEL_LINDIAG = 3,  /* u' = p .* u, n=np=2 (test/Core1/sparse_adjoint.jl:6-8) */
    HIPADJ_MODEL_FALLMASS = 4, /* u' = [u2, -g] (test/Core7/physical_ode_regression.jl:20-23) */
    HIPADJ_MODEL_MLP = 5,      /* tanh MLP d->H->H->d on a d x B state; dims = {d, H, B} */
    HIPADJ_MODEL_BRUSS = 6,    /* 2-D Brusselator, dims = {Ngrid} (docs/src/examples/pde/brusselator.md:98-112) */
    HIPADJ_MODEL_USER_BASE = 1000 /* ids >= this come from hipadj_model_register (runtime-compiled right-hand sides) */
} hipadj_model;

/* sensealg — src/sensitivity_algorithms.jl:254-272 (Backsolve), 378-396 (Interpolating), 486-503 (Quadrature),
 * 591-607 (Gauss) */
typedef enum {
    HIPADJ_ALG_INTERPOLATING = 0,
    HIPADJ_ALG_BACKSOLVE = 1,
    HIPADJ_ALG_GAUSS = 2,
    HIPADJ_ALG_QUADRATURE = 3,
    HIPADJ_ALG_GAUSS_KRONROD = 4  /* GaussKronrodAdjoint (src/sensitivity_algorithms.jl:612-711): GaussAdjoint with a per-step adaptive
                                     (7,15) Gauss-Kronrod rule.  Its callback lives in DiffEqCallbacks (not vendored): restated from
                                     recall, pinned only by GaussKronrod == Gauss == Interpolating.  Lane-per-trajectory models; RK4
                                     (time-segmented like Gauss) and Tsit5; checkpointing=true with Tsit5 only. */
} hipadj_alg;

typedef enum {
    HIPADJ_STEPPER_RK4_FIXED = 0,      /* fixed-step classic RK4, cubic-Hermite dense output; loss times on the step grid */
    HIPADJ_STEPPER_ETDRK4_FIXED = 2,   /* fixed-step exponential RK4 (Cox & Matthews 2002; OrdinaryDiffEq's ETDRK4) for the semilinear PDE family (HIPADJ_MODEL_BRUSS): the
                                        * diffusion term is integrated exactly in the 2-D DFT basis, dt is bound by the reaction terms only — the stiff stepper for the
                                        * horizon the reference documents (docs/src/examples/pde/brusselator.md:115 uses FBDF).  Interpolating-, Gauss- and QuadratureAdjoint,
                                        * loss times on the step grid, cubic-Hermite dense output like RK4_FIXED.  */
    HIPADJ_STEPPER_TSIT5_ADAPTIVE = 1, /* adaptive Tsit5 with per-trajectory step control and its own interpolant (the stepper of
                                          the reference's tests); arbitrary loss times; lane-per-trajectory models; all four
                                          sensealgs (checkpointing=true: Backsolve only) */
    HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE = 3 /* adaptive Rosenbrock23 (ode23s as OrdinaryDiffEq ships it; test/Core2/stiff_adjoints.jl:53-75): the STIFF stepper of the
                                          lane-per-trajectory models (compiled-in and runtime-registered, n <= 8) — W = I - d h J factored per lane in registers (J from the
                                          model's VJP), forward solve AND reverse solve (the adjoint runs with the forward solve's alg, src/sensitivity_interface.jl:487-491;
                                          the adjoint system's W is block triangular: an n x n solve plus a substitution).  Interpolating-, Gauss-, GaussKronrod- and
                                          QuadratureAdjoint, arbitrary loss times, checkpointing = true (intervals re-solved with this stepper), and BacksolveAdjoint (its W formed from the first-derivative
                                          blocks: the stepper is a W-method; not on a DAE); continuous costs and discrete-loss bodies like Tsit5 (not on a DAE); a non-singular mass
                                          matrix enters through the M^{-1} form of the runtime models like everywhere in the lane family, a singular semi-explicit one makes the
                                          model a DAE that ONLY this stepper integrates (hipadj_model_set_mass_matrix).  ABI 109 */
} hipadj_stepper;

/* how dgdu_discrete(out, u, p, t, i) — and dgdp_discrete — is evaluated at loss time t_i (src/adjoint_common.jl:771-779).  Kinds 1-3 keep the loss ON THE DEVICE:
 * nothing crosses the host link between the forward solve and the reverse pass (a cotangent block of BASELINE configs[1] is 24 MB each way). */
typedef enum {
    HIPADJ_LOSS_COTANGENT = 0, /* out = dLdu[:, i]   — the AD path, src/concrete_solve.jl:842-851 */
    HIPADJ_LOSS_LSQ_SHIFT = 1, /* out = u - loss_shift — test/Core3/adjoint.jl:49-51, 1169-1172; fused in-kernel */
    HIPADJ_LOSS_LSQ_DATA = 2,  /* out = loss_scale * (u - data[:, i]) with a per-(trajectory, time) data block handed over ONCE (hipadj_set_loss_data[_dev]);
                                  loss_scale = 2 is the gradient of sum(abs2, sol .- data), the loss of the reference's tutorials and benchmark (docs/src/Benchmark.md:80,
                                  docs/src/tutorials/parameter_estimation_ode.md:43, docs/src/tutorials/data_parallel.md:115-116, docs/src/examples/pde/brusselator.md:221).  Every kernel family, both steppers, all sensealgs. */
    HIPADJ_LOSS_MODEL = 3      /* out = the dgdu_discrete body attached to a runtime-registered model (hipadj_model_set_discrete_loss / hipadj_wmodel_set_discrete_loss),
                                  evaluated in the sweep at every loss time, together with its dgdp_discrete body (src/adjoint_common.jl:775-779,
                                  src/quadrature_adjoint.jl:545-552, 601-605; test/Core7/mixed_costs.jl:199-390).  The bodies see the data block of
                                  hipadj_set_loss_data when one was set. */
} hipadj_loss;

/* continuous costs g(u, p, t) with device-inlined dgdu_continuous / dgdp_continuous
 * (accumulate_cost!, src/derivative_wrappers.jl:1411-1442; AdjointSensitivityIntegrand `out .+= dgdp`, src/quadrature_adjoint.jl:497-500).
 * GaussAdjoint / GaussKronrodAdjoint take the parameter term with the sign of the other algorithms (Gauss == Interpolating == Quadrature); the
 * reference's own line (src/gauss_adjoint.jl:755-758, `out .+= dgdp` after the negation, no test) reads the other way — DESIGN.md 6.5. */
typedef enum {
    HIPADJ_CCOST_NONE = 0,
    HIPADJ_CCOST_HALF_SQ_SUM = 1, /* g = (sum(u))^2 / 2, dgdu = sum(u) in every component, dgdp = 0 (test/Core3/adjoint.jl:913-919) */
    HIPADJ_CCOST_U1SQ_PLUS_P1 = 2,/* g = u[1]^2 + p[1], dgdu = [2 u1, 0, ...], dgdp = [1, 0, ...] (test/Core7/mixed_costs.jl:46-57) */
    HIPADJ_CCOST_MODEL = 3        /* the cost attached to a runtime-registered model with hipadj_model_set_cost */
} hipadj_cont_cost;

/* kernel family selection (hipadj_config.family, hipadj_stats.routed_family; ABI 109) */
typedef enum {
    HIPADJ_FAMILY_AUTO = 0,
    HIPADJ_FAMILY_AS_REGISTERED = 1,
    HIPADJ_FAMILY_MFMA = 3          /* the FP64-MFMA family of HIPADJ_MODEL_MLP (csrc/hipadj_mlp*.hpp) */
} hipadj_family;
typedef enum { HIPADJ_ACT_TANH = 1 } hipadj_activation;

typedef struct {
    uint32_t struct_size;      /* = sizeof(hipadj_config); ABI guard */
    int32_t model;             /* hipadj_model */
    int32_t alg;               /* hipadj_alg */
    int32_t stepper;           /* hipadj_stepper */
    int32_t dims[4];           /* model shape parameters (MLP, BRUSS), else 0 */
    int64_t ntraj;             /* N trajectories in this handle's shard */
    double t0, t1, dt;         /* tspan; RK4: fixed step; a span that is not a multiple of dt ends with a shortened step (lane-per-trajectory
                                  models: such spans run the off-grid sweeps; PDE / MLP families: (t1 - t0)/dt must be an integer);
                                  Tsit5: initial step (<= 0: automatic) */
    int32_t nsave;             /* M loss/save times */
    const double *save_times;  /* [M] strictly ascending inside [t0, t1] (copied at create).  RK4: on the step grid t0 + k*dt, or anywhere: the reverse
                                  solve then stops at them like the reference's PresetTimeCallback tstops (src/adjoint_common.jl:848-855) and out = sol(ts)
                                  is interpolated (src/concrete_solve.jl:718-727).  Off-grid times are taken by every sensealg on the lane-per-trajectory and
                                  the wide models — lane models also with checkpointing = true / ckpt_stride / a checkpoint list (the checkpoints are stops
                                  too and may lie anywhere; BacksolveAdjoint's likewise on wide models); the PDE / MLP families and the checkpointed
                                  Interpolating / Gauss / GaussKronrod sweeps of wide models need times on the grid (HIPADJ_ERR_UNSUPPORTED otherwise).
                                  Tsit5: arbitrary times */
    int32_t loss_kind;         /* hipadj_loss */
    double loss_shift;
    int32_t checkpointing;     /* sensealg.checkpointing (Backsolve default true: sensitivity_algorithms.jl:260-265) */
    int32_t ckpt_stride;       /* checkpoint every ckpt_stride steps; 0 => checkpoints = save_times (backsolve_adjoint.jl:132) */
    double quad_abstol, quad_reltol; /* QuadratureAdjoint(abstol = 1e-6, reltol = 1e-3) */
    int32_t no_start;          /* suppress the loss jump at t0 (src/adjoint_common.jl:761) */
    int32_t p_shared;          /* 1: p[np] shared by all trajectories and dp[np] = sum_i dp_i ; 0: p[N][np], dp[N][np] */
    int32_t device;            /* HIP device ordinal */
    int32_t time_segments;     /* 0 = automatic; 1 = strictly sequential in time; C > 1 = C time segments per trajectory */
    int32_t cont_cost;         /* hipadj_cont_cost: continuous cost g(u,p,t) added to the loss as int g dt (accumulate_cost!,
                                  src/derivative_wrappers.jl:1411-1442; adjoint_sensitivities(...; g, dgdu_continuous)) */
    int32_t max_steps;         /* Tsit5: capacity of the per-trajectory dense solution in accepted steps.  0 => automatic: the buffers
                                  start at 128 steps; hipadj_forward reads the true step counts back after the pass and, when they
                                  do not fit, regrows the buffers and repeats the pass (one host sync per forward; bound 100 000
                                  steps = the reference's maxiters) */
    double abstol, reltol;     /* Tsit5: tolerances of the forward AND reverse solves (src/sensitivity_interface.jl:432 defaults 1e-6 / 1e-3) */
    int32_t ncheckpoints;      /* > 0: the `checkpoints` keyword of adjoint_sensitivities (src/sensitivity_interface.jl:484-486,
                                  src/backsolve_adjoint.jl:132, src/interpolating_adjoint.jl:54-58): an explicit, strictly ascending list of
                                  checkpoint times inside [t0, t1] (copied at create); t0 and t1 are added when missing, as the reference's
                                  interval construction does.  RK4: every time on the step grid t0 + k*dt, any spacing (ckpt_stride must be
                                  0).  Tsit5: arbitrary times.  0: ckpt_stride, or the save times (the reference default) */
    const double *checkpoints; /* [ncheckpoints] */
    double loss_scale;         /* HIPADJ_LOSS_LSQ_DATA: the factor w of dgdu = w (u - data); 0 means 1 */
    int32_t ndevices;          /* > 1: ONE handle over several devices — the ensemble is cut into ndevices contiguous trajectory ranges (the partitioning of the reference's
                                  EnsembleDistributed pattern, docs/src/tutorials/data_parallel.md:77-136, inside one process and one `solve` call): every range gets its own
                                  stream, workspaces and kernels on its device; host-pointer calls scatter u0 / gather out, du0 and sum dp over the ranges in range order;
                                  device-pointer calls take buffers of device_ids[0] and move the slices by peer copies.  0 / 1: the single device `device`. */
    const int32_t *device_ids; /* [ndevices] HIP ordinals (copied at create); the same ordinal may repeat ("virtual shards": how a 1-GPU box tests the path); NULL: 0 .. ndevices-1 */
    int32_t reference_literal; /* 1: reproduce the reference's lines where this library deliberately deviates from them: GaussAdjoint / GaussKronrodAdjoint take
                                  dgdp_continuous with the sign src/gauss_adjoint.jl:753-758 has as written (-f_p' lam + g_p under the reversed-time sum), and drop dgdp_discrete
                                  (ReverseLossCallback skips it for `isq` algorithms, src/adjoint_common.jl:776, and src/gauss_adjoint.jl adds it nowhere).  0 (default): the
                                  mathematically consistent forms (Gauss == Interpolating == Quadrature).  Exists so that a reference-generated fixture can decide each with one number. */
    int32_t family;            /* a hipadj_family value; this field was reserved1 before ABI 109.  HIPADJ_FAMILY_AUTO = 0: hipadj_create picks the kernel family — a wide model declared as a dense chain
                                  (hipadj_wmodel_declare_dense_chain) of shape 2 -> H -> H -> 2, H in {32, 64, 128}, tanh, on fixed-step RK4 with shared parameters and a multiple
                                  of 16 trajectories runs on the FP64-MFMA family (the trajectories become the batch columns of HIPADJ_MODEL_MLP; same handle API and array shapes;
                                  hipadj_stats.routed_family says so).  HIPADJ_FAMILY_AS_REGISTERED = 1: always the family the model was registered for. */
} hipadj_config;

typedef struct {
    uint32_t struct_size;
    int32_t n, np;
    int64_t ntraj, nsteps;
    int32_t time_segments;         /* segments actually used by the last adjoint */
    double forward_ms_last, adjoint_ms_last;      /* device time of the last call (HIP events on the handle stream) */
    double forward_ms_total, adjoint_ms_total;
    int64_t forward_calls, adjoint_calls;
    double adjoint_main_kernel_ms_last, adjoint_main_kernel_ms_total; /* dominant reverse kernel only */
    double adjoint_algorithmic_bytes; /* per adjoint call, SURVEY.md §8d definition */
    double vjp_steps;                 /* N * S * 4 per adjoint call (src/derivative_wrappers.jl:256 equivalents) */
    double workspace_bytes;
    int32_t launches_per_pass;        /* kernel launches of one reverse pass as configured: 1 = the sweep kernel finishes the pass itself (composition tree
                                         and dp reduction in-launch, csrc/hipadj_fused.hpp), 3 = sweep + composition + reduction (lane family), wide models: the real count (sweep
                                         [+ GK15 pass] [+ k_wide_reduce_dp for shared parameters]), 0 = other sequences */
    int32_t routed_family;            /* hipadj_family the library chose for this handle when it differs from the registered one (HIPADJ_FAMILY_MFMA), else 0 (was reserved0; ABI 109) */
} hipadj_stats;

typedef struct hipadj_handle hipadj_handle;

int hipadj_version(void);
const char *hipadj_status_string(int status);
/* message of the last failing call on this handle; handle == NULL: last hipadj_create failure of this thread */
const char *hipadj_last_error(const hipadj_handle *h);
/* HIP devices the library sees (0 without a usable device): what a host expands `devices = :all` with for hipadj_config.device_ids */
int hipadj_device_count(void);
/* n and np of a registered model (the reference reads them off u0 / p) */
int hipadj_model_sizes(int32_t model, const int32_t dims[4], int32_t *n, int32_t *np);

/* Runtime model ingestion — the reference's ODEFunction(f!; vjp, vjp_p) seam (src/derivative_wrappers.jl:284-359,
 * test/Core3/user_vjp.jl:14-38, 77): the caller supplies the BODIES of the three device functions as HIP C++ text;
 * hipadj_create compiles the lane-per-trajectory kernels for them with hiprtc (gfx950) on first use.
 *   f_body      writes du[0..n)   from  u[0..n), p[0..np), t            (in-place f!(du, u, p, t))
 *   vjp_u_body  writes out[0..n)  = (df/du)^T lam  from lam, u, p, t    (vjp(dlam, lam, u, p, t), UN-negated)
 *   vjp_p_body  writes out[0..np) = (df/dp)^T lam  from lam, u, p, t    (vjp_p(dgrad, lam, u, p, t), UN-negated)
 * All are `double`; local variables and device math functions are allowed; no global memory access.
 * vjp_u_body == vjp_p_body == NULL selects AUTOMATIC VJPs — the reference's `autojacvec = true` (ForwardDiff Jacobian,
 * src/sensitivity_algorithms.jl:629-660): f_body is then compiled a second time with forward-mode dual numbers, so it must
 * declare its locals as `real` (or `auto`) instead of `double`; + - * / comparisons and sin cos tan exp log sqrt tanh sinh
 * cosh atan fabs pow are available on `real`.
 * Limits: 1 <= n <= 8, 1 <= np <= 32.  On success *model_id (>= HIPADJ_MODEL_USER_BASE) is valid for hipadj_config.model
 * for the lifetime of the process; registering the same name again replaces the sources (new id). */
int hipadj_model_register(const char *name, int32_t n, int32_t np, const char *f_body, const char *vjp_u_body,
                          const char *vjp_p_body, int32_t *model_id);
/* Attaches a continuous cost to a runtime-registered model — dgdu_continuous / dgdp_continuous of adjoint_sensitivities
 * (src/sensitivity_interface.jl:373-526): dgdu_body writes out[0..n) = dg/du, dgdp_body writes out[0..np) = dg/dp from u, p, t.
 * Selected per handle with cont_cost = HIPADJ_CCOST_MODEL. */
int hipadj_model_set_cost(int32_t model_id, const char *dgdu_body, const char *dgdp_body);
/* Same, from the cost itself: g_body assigns `g` (declared `real g`) from u, p, t with `real` locals; dg/du and dg/dp are
 * generated by forward-mode dual numbers — the reference's gradient!(g) fallback (src/derivative_wrappers.jl:1428-1441). */
int hipadj_model_set_cost_function(int32_t model_id, const char *g_body);
/* The continuous cost of a WIDE model (hipadj_wmodel_register) — dgdu_continuous / dgdp_continuous of adjoint_sensitivities for a model beyond the lane family — as ONE
 * SPMD body with the conventions of its vjp body:  cost<WP>(dlam, gp, acc, w, u, p, t, ws, tid)  ADDS dg/du into dlam[0..n) (entry i from the thread that owns it:
 * HIPADJ_W_FOR loops do) and, under `if (WP)`, w * dg/dp into gp[...] (entries owned by one thread) or acc[...] (the model's reduced parameters).  It is evaluated
 * with every joint VJP (accumulate_cost!, src/derivative_wrappers.jl:1411-1442).  NULL / "" removes it.  Selected per handle with cont_cost = HIPADJ_CCOST_MODEL. */
int hipadj_wmodel_set_cost(int32_t model_id, const char *cost_body);
/* ODEFunction(f; mass_matrix = M) for a runtime-registered model: M u' = f(u, p, t) with a CONSTANT n x n matrix M, non-singular or singular of the
 * semi-explicit form described below (row-major; NULL removes it) — test/Core3/adjoint.jl:1315-1376.  The reference hands M to the forward solver and M' (resp.
 * [M' 0; 0 I], [M' 0 0; 0 I 0; 0 0 M]) to the adjoint problems (src/interpolating_adjoint.jl:413-426, src/backsolve_adjoint.jl:232-247,
 * src/quadrature_adjoint.jl:194-206, src/gauss_adjoint.jl:403-415) and divides the loss jumps by lu(M') (src/adjoint_common.jl:110-135,
 * 805-807).  For a non-singular M the generated model is F = M^{-1} f (every stepper, Rosenbrock23 included) with F_u' nu = f_u' (M^{-T} nu): the sweep
 * integrates nu = M' lam, the parameter integrand f_p' lam is unchanged, and du0 is mapped back to lam(t0) = M^{-T} nu(t0) — what the
 * reference returns (src/sensitivity_interface.jl:500; note that dG/du0 itself is M' du0).  All sensealgs, every stepper of the lane family.
 * A SINGULAR M of the semi-explicit form [Md 0; 0 0] — zero rows that are also zero columns (the algebraic variables, src/adjoint_common.jl:116-122), Md non-singular (:131-133) —
 * makes the model a DAE (round 6; test/Core3/adjoint.jl:1434-1530): HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE integrates M u' = f and M' lam' = -J' lam in mass-matrix form
 * (W = M - d h J), from a consistent state (the algebraic entries of u0 are solved for, BrownFullBasicInit), with the loss jumps of src/adjoint_common.jl:790-813 (algebraic
 * part of the cotangent eliminated through J_aa, its parameter term added to dp, the algebraic adjoints re-initialised); du0 = lam(t0).  Interpolating-, Gauss-, GaussKronrod-,
 * QuadratureAdjoint; every other stepper refuses such a model at hipadj_create (HIPADJ_ERR_UNSUPPORTED).  Any other singular M: HIPADJ_ERR_UNSUPPORTED here.
 * Handles created earlier keep the matrix they were created with. */
int hipadj_model_set_mass_matrix(int32_t model_id, const double *M);
/* DiscreteCallback at preset times with an affect  (u, p) <- a(u, p, t)  (test/Callbacks1/discrete_callbacks.jl:260-330, incl. the parameter-changing
 * case :303-312; the reverse pass of src/callback_tracking.jl:232-470 for a DiscreteCallback: lam <- (da/du)' lam at the LEFT state, grad += (da/dp)' lam).
 * affect_body edits un[0..n) and / or pn[0..np) — which start as copies of u and p — from u, p, t (declare locals `real`: the Jacobian products are
 * generated by forward-mode dual numbers).  An event problem is composed on the host from per-piece solves (the spans between consecutive event times
 * are ordinary handles): hipadj_affect_apply maps the end state (and parameters) of a piece to the start state (and parameters) of the next;
 * hipadj_affect_vjp is the reverse callback for the map (u, p) -> (un, pn):
 *     lam_out = (dun/du)' lam + (dpn/du)' gp,      gp_out = (dun/dp)' lam + (dpn/dp)' gp
 * with lam = du0 of the upper piece and gp [N][np] = the gradient with respect to the parameters after the event of everything later in time
 * (an affect that leaves p alone: gp_out = (dun/dp)' lam + gp).  Host pointers, synchronous; u, lam, out: [N][n]; p: [np] (p_shared) or [N][np];
 * p_out (may be NULL), gp, gp_out: [N][np].  Not covered: ContinuousCallback (root finding + the implicit event-time corrections), save_positions
 * other than (false, false) — the host mirror defines the value saved AT an event time as the right limit. */
int hipadj_model_set_affect(int32_t model_id, const char *affect_body);

/* The same for a wide model (hipadj_wmodel_register; ABI 108, round 5): dual numbers do not scale to 4096 states, so the reverse callback comes as text too.
 * Both bodies are SERIAL code run by one thread per trajectory (an event happens a handful of times per solve), over plain arrays:
 *   affect_body      edits un[0..N) / pn[0..NP) — copies of u / p on entry — from u, p, t
 *   affect_vjp_body  edits lo[0..N) / go[0..NP) — copies of lam / gp on entry, i.e. the reverse callback of the identity — from lam, gp, u, p, t, so that
 *                    lo = (dun/du)' lam + (dpn/du)' gp,  go = (dun/dp)' lam + (dpn/dp)' gp.  "" = the affect's Jacobian is the identity (a constant dose).
 * e.g. affect "for (int i = 0; i < N; ++i) un[i] += p[1] / 8.0 * sin(u[i]);"  with  vjp "for (int i = 0; i < N; ++i) { lo[i] = lam[i] * (1.0 + p[1] / 8.0 * cos(u[i])); go[1] += lam[i] * sin(u[i]) / 8.0; }".
 * Both NULL removes the affect.  hipadj_affect_apply / hipadj_affect_vjp below serve both families. */
int hipadj_wmodel_set_affect(int32_t model_id, const char *affect_body, const char *affect_vjp_body);
int hipadj_affect_apply(int32_t model_id, int32_t device, int64_t N, const double *u, const double *p, int32_t p_shared, double t, double *out, double *p_out);
int hipadj_affect_vjp(int32_t model_id, int32_t device, int64_t N, const double *u, const double *p, int32_t p_shared, double t, const double *lam,
                      const double *gp, double *lam_out, double *gp_out);

/* ContinuousCallback(condition, affect!) on a runtime lane model  (src/callback_tracking.jl:1-223 forward tracking, :232-479 reverse callbacks;
 * test/Callbacks2/continuous_callbacks.jl — the bouncing ball).  Every handle created on the model afterwards, on HIPADJ_STEPPER_TSIT5_ADAPTIVE or
 * HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE, locates the zero crossings of the condition on the dense output of each accepted step, PER TRAJECTORY (either direction: affect_neg! =
 * affect!), cuts the step there, applies the affect and goes on; the reverse solve of every sensealg (Interpolating-,
 * Backsolve-, Gauss-, GaussKronrod-, QuadratureAdjoint) runs piece by piece between the events of each trajectory and applies, at each of them, the jump with the event-time term:
 *     kappa = lam+ . (a_u f- + a_t - f+) / (c_u . f- + c_t)      lam- = a_u' lam+ - kappa c_u      dp += a_p' lam+ - kappa c_p
 *   condition_body  assigns `c` from u[0..N), p[0..NP), t       e.g. "c = u[0];"  or  "c = u[0] - 0.75 * p[0];"
 *   affect_body     edits un[0..N) — a copy of u on entry — from u, p, t (NULL or "": the identity)       e.g. "un[1] = -p[1] * u[1];"
 *                   `terminate = true;` in it is terminate!(integrator): THAT trajectory's solve ends at the event; its later save times hold the final state and carry no
 *                   loss (the loss on the solution's last point: the event's dr, hipadj_set_event_cotangents)
 *   max_events      capacity of the event list of one trajectory (0 = 64); a trajectory with more events fails the forward call with HIPADJ_ERR_MAXITERS
 * Both bodies are compiled for double and for dual numbers (declare locals `real`); every derivative of the jump comes from them.  Both NULL removes the callback.
 * BacksolveAdjoint (the algorithm the reference's callback tests lean on, also with checkpointing = true, its default): the forward solve stores every event's time and left
 * state; the backsolved state is overwritten with it at the event, as at a checkpoint.
 * QuadratureAdjoint: the dense adjoint record runs through the jumps; its quadrature intervals are split at each trajectory's events.
 * checkpointing = true (Interpolating / Gauss / GaussKronrod): a checkpoint interval is re-solved only as far as the piece between two events reaches, from the stored state
 * after the lower event.
 * Refused with HIPADJ_ERR_UNSUPPORTED at hipadj_create: the fixed-step steppers, continuous costs, HIPADJ_LOSS_MODEL;
 * here: wide models, models with a mass matrix, affects that edit the parameters (pn).  A save time that coincides with an event sees the affected
 * state; save_positions = (true, true): hipadj_event_states / hipadj_set_event_cotangents below. */
int hipadj_model_set_continuous_callback(int32_t model_id, const char *condition_body, const char *affect_body, int32_t max_events);
/* VectorContinuousCallback(condition, affect!, len) (test/Callbacks2/vector_continuous_callbacks.jl): ncond conditions (1 .. 8) watched together; the event is the first zero
 * crossing of any of them, and the affect sees which one fired.
 *   condition_body  assigns out[0 .. ncond)                 e.g. "out[0] = u[0]; out[1] = (u[2] - 10.0) * u[2];"
 *   affect_body     edits un from u, p, t and `idx` (int)   e.g. "if (idx == 0) un[1] = -p[1] * u[1]; else un[3] = -p[1] * u[3];"
 * Everything else as hipadj_model_set_continuous_callback (which is the case ncond = 1, `c` an alias of out[0]).  Components that cross in the same tenth of
 * a step are each located on their own and the earliest root is the event; the SIMULTANEOUS fire of several components (the reference's event_idx mask, :118-230) is not merged
 * into one event: the lowest index fires. */
int hipadj_model_set_vector_continuous_callback(int32_t model_id, int32_t ncond, const char *condition_body, const char *affect_body, int32_t max_events);
/* ContinuousCallback(condition, affect!, affect_neg!) with one of the two affects `nothing`: direction +1 = only upcrossings of the condition fire (negative to positive: affect!
 * with affect_neg! = nothing), -1 = only downcrossings, 0 = both (the default: affect_neg! = affect!).  For a callback set before; applies to every handle created afterwards. */
int hipadj_model_set_callback_direction(int32_t model_id, int32_t direction);
/* events per trajectory of the handle's last forward solve: counts[ntraj], host pointer, synchronous.  HIPADJ_ERR_UNSUPPORTED when the model carries no ContinuousCallback. */
int hipadj_event_counts(hipadj_handle *h, int32_t *counts);
/* save_positions = (true, true) — the constructor's default, and the setting of most of the reference's callback tests (test/Callbacks2/continuous_callbacks.jl:200-250): the
 * solution also holds the state just before and just after every affect, and a loss may take them.  Outputs and losses of a handle live at the save times; the saved event
 * states are a second, ragged set handed over by these two calls (max_events as the model was registered with, 0 = 64):
 *   hipadj_event_states        t [ntraj][max_events], ul, ur [ntraj][max_events][n] of the last forward solve: event times, states before / after the affect (host pointers, any
 *                              may be NULL; entries beyond a trajectory's event count are zero; synchronous)
 *   hipadj_set_event_cotangents  dl, dr [ntraj][max_events][n]: the cotangents of the caller's loss at those states, used by every following hipadj_adjoint* call (host pointers;
 *                              either may be NULL = zero, both NULL removes them; entries beyond a trajectory's event count are ignored).  The reverse jump then reads
 *                              kappa = [lam+ . (a_u f- + a_t - f+) + dr . (a_u f- + a_t) + dl . f-] / (c_u . f- + c_t),  lam- = a_u' (lam+ + dr) + dl - kappa c_u,
 *                              dp += a_p' (lam+ + dr) - kappa c_p  (src/callback_tracking.jl:385-401, 439-452: the saved states move with the event time).
 * HIPADJ_ERR_UNSUPPORTED when the model carries no ContinuousCallback; hipadj_event_states before the first forward solve: HIPADJ_ERR_STATE. */
int hipadj_event_states(hipadj_handle *h, double *t, double *ul, double *ur);
/* which component of a VectorContinuousCallback fired at each event of the last forward solve (the reference's event_idx): idx [ntraj][max_events], 0 for a scalar condition,
 * + 256 when the event terminated the trajectory's solve, -1 beyond a trajectory's event count.  Host pointer, synchronous; status as hipadj_event_states. */
int hipadj_event_components(hipadj_handle *h, int32_t *idx);
int hipadj_set_event_cotangents(hipadj_handle *h, const double *dl, const double *dr);
/* Wide runtime models — more than 8 states or more than 32 parameters (up to n = 4096 states): the workgroup-per-trajectory family
 * (csrc/hipadj_wide.hpp).  ONE workgroup of `threads` threads integrates one trajectory; the stage state, the stage adjoint and the VJP
 * output are tiles in LDS, threads own the components tid, tid + threads, ...  The model is TWO bodies of HIP C++, run by every thread
 * of the workgroup with the same arguments (SPMD) — f and the joint VJP, i.e. the reference's internal contract
 * vecjacobian!(dlam, y, lam, p, t, S; dgrad) (src/derivative_wrappers.jl:256-267) rather than the separate vjp / vjp_p of the small models:
 *   f_body    writes du[0..n)                         from u[0..n), p[0..np), t
 *   vjp_body  writes dlam[0..n) = (df/du)' lam        from lam[0..n), u, p, t;  and, inside `if (WP) { ... }`, adds w * (df/dp)' lam to the
 *             gradient: `gp[j] += w * ...` for entries owned by exactly one thread (one thread per weight of an MLP / entry of a matrix), or
 *             `acc[q] += w * ...` = this thread's partial of parameter acc_first + q (a coefficient every component feeds; nacc <= 16
 *             such parameters; the partials are summed over the workgroup once per sweep)
 * Available inside the bodies: `tid`, `T` (= threads), `N`, `NP`, `HIPADJ_W_FOR(i, count) { ... }` (i = tid, tid + T, ... < count),
 * `wg_sync()` (workgroup barrier between dependent phases, e.g. hidden layers), `wg_sum(x)` (sum of one value per thread over the workgroup,
 * returned to every thread: wavefront shuffles; every thread must call it), `wg_sum2(a, b, sa, sb)` (two such sums at once: one v_permlane32_swap level serves both),
 * `ws[0..lds_doubles)` LDS scratch.  u, lam, du, dlam, ws are
 * LDS, p is global memory.  threads: a multiple of 64 in [64, 1024], 0 = automatic.  Offered: fixed-step RK4 (loss times on the step grid;
 * Interpolating / Backsolve (checkpoints) / Gauss / GaussKronrod / QuadratureAdjoint) and adaptive Tsit5 with per-trajectory step control (stepper =
 * HIPADJ_STEPPER_TSIT5_ADAPTIVE, arbitrary loss times; all four sensealgs and GaussKronrod on both — adaptive Interpolating and Backsolve keep five np-sized rows in LDS,
 * HIPADJ_ERR_UNSUPPORTED naming GaussAdjoint when they do not fit; max_steps = 0 sizes the dense record from an 8 GiB budget, 64 ... 8192 steps);
 * discrete losses and the built-in continuous costs (cont_cost = HIPADJ_CCOST_HALF_SQ_SUM / HIPADJ_CCOST_U1SQ_PLUS_P1); parity-tested against the oracle on the reference's 30 x 50 matrix-state problem (test/Core5/size_handling_adjoint.jl:37-70)
 * and the 2 -> 50 -> 2 neural ODE of docs/src/Benchmark.md:62, with both steppers. */
int hipadj_wmodel_register(const char *name, int32_t n, int32_t np, int32_t threads, int32_t lds_doubles, int32_t nacc, int32_t acc_first,
                           const char *f_body, const char *vjp_body, int32_t *model_id);
/* Discrete loss ON THE DEVICE for a runtime-registered lane model (n <= 8) — dgdu_discrete / dgdp_discrete of adjoint_sensitivities (src/sensitivity_interface.jl:373-526,
 * evaluated by ReverseLossCallback at every loss time, src/adjoint_common.jl:771-779), selected per handle with loss_kind = HIPADJ_LOSS_MODEL:
 *   dgdu_body  writes out[0..n)  = dl_i/du   from u[0..n), p[0..np), t (= t_i), i (the 0-based index of the loss time; the reference's `i` is i + 1) and d[0..n)
 *   dgdp_body  writes out[0..np) = dl_i/dp   from the same arguments; NULL: the loss does not depend on p
 * d is the column data[trajectory][i][0..n) of the block set with hipadj_set_loss_data[_dev] (zeros when none was set).  All `double`, no global memory access.
 * The sweep adds dgdu to lam and dgdp to the parameter gradient at t_i — every sensealg (QuadratureAdjoint and GaussAdjoint add the parameter part next to their
 * quadrature; the reference's GaussAdjoint drops it: hipadj_config.reference_literal), both steppers, checkpointing, off-grid loss times. */
int hipadj_model_set_discrete_loss(int32_t model_id, const char *dgdu_body, const char *dgdp_body);
/* The same from the loss itself: l_body assigns `l` (declared `real l`) = l_i(u, p, t, i, d) with `real` locals; dl/du and dl/dp by forward-mode dual numbers, and
 * hipadj_loss_value[_dev] can then return the loss as well. */
int hipadj_model_set_discrete_loss_function(int32_t model_id, const char *l_body);
/* ... of a WIDE model (hipadj_wmodel_register), as ONE SPMD body with the conventions of its vjp body:  dloss<WP>(dlam, gp, acc, u, p, t, i, d, ws, tid)  ADDS dl_i/du into
 * dlam[0..n) (entry k from the thread that owns it: HIPADJ_W_FOR loops do) and, under `if (WP)`, dl_i/dp into gp[...] (entries owned by one thread) or acc[...] (the model's
 * reduced parameters).  u, dlam: LDS tiles; d: the trajectory's data column [n] in global memory (NULL when no block was set).  NULL / "" removes it. */
int hipadj_wmodel_set_discrete_loss(int32_t model_id, const char *dloss_body);
/* Declares that wide model `model_id` IS the dense chain Lux.Chain(x -> x.^input_power, Dense(widths[0], widths[1], act), ..., Dense(widths[L-1], widths[L])) — `act` on every
 * layer but the last, parameters in Lux's flattening order (per layer: weight [out x in] column-major, then bias) — the structure behind the neural ODEs of the reference's docs
 * (docs/src/Benchmark.md:62-96).  The bodies given to hipadj_wmodel_register stay the model's definition for the workgroup-per-trajectory family; with the declaration
 * hipadj_create selects the family ITSELF (hipadj_config.family): a chain with H x H contractions belongs on the matrix cores, the published 2-50-2 net does not.
 * widths == NULL withdraws the declaration.  Errors: widths that do not reproduce the model's n / np, an activation other than HIPADJ_ACT_TANH -> HIPADJ_ERR_INVALID_ARG. */
int hipadj_wmodel_declare_dense_chain(int32_t model_id, const int32_t *widths, int32_t nwidths, int32_t activation, int32_t input_power);

/* Compiles the forward and the InterpolatingAdjoint kernels of a registered lane model — for a wide model both forward solves and every reverse sweep of
 * the family (Interpolating, Gauss, GaussKronrod, Backsolve, Quadrature on RK4 and on adaptive Tsit5, without a cost) — for gfx950 (no device needed) so that
 * source errors surface before hipadj_create; the compiler log is available through hipadj_last_error(NULL). */
int hipadj_model_check(int32_t model_id);
/* Compiles every kernel a handle of configuration *cfg would launch (forward, reverse, quadrature, tail) — what hipadj_create
 * does lazily for a runtime-registered model — without a device; HIPADJ_OK at once for built-in models.  Lets a caller warm the
 * process-wide code cache and surface compile errors of a particular sensealg / stepper combination ahead of time. */
int hipadj_model_check_config(const hipadj_config *cfg);
/* Which compiler builds the runtime-registered models: "<path of the bound libhiprtc> [own link-map namespace]; HIP x.y.z" written to buf
 * (NUL-terminated, truncated to cap).  The library binds the hiprtc of the ROCm toolkit it was built with ($HIPADJ_HIPRTC, $ROCM_PATH/lib, the
 * build-time ROCm root) — not whatever hiprtc the host process happens to carry: a torch wheel bundles an older ROCm whose compiler
 * miscompiles wide models.  Returns HIPADJ_OK, or HIPADJ_ERR_UNSUPPORTED (with the reason in buf) when no hiprtc is found. */
int hipadj_runtime_compiler(char *buf, int32_t cap);

/* Replaces the per-call setup of ODEAdjointProblem + adjointdiffcache (src/interpolating_adjoint.jl:307-451,
 * src/backsolve_adjoint.jl:123-272, src/adjoint_common.jl:42-469): validates the configuration, allocates the
 * device workspaces (forward interpolant tiles / checkpoint tiles), creates the stream. */
int hipadj_create(const hipadj_config *cfg, hipadj_handle **out);
int hipadj_destroy(hipadj_handle *h);

/* Forward solve of the ensemble — replaces the forward `solve` in _concrete_solve_adjoint
 * (src/concrete_solve.jl:689-707) and the primal output `out = sol(ts)` (:718-770).
 * u0 [N][n]; p [np] (p_shared) or [N][np]; out [N][M][n] (may be NULL). Host pointers; synchronous. */
int hipadj_forward(hipadj_handle *h, const double *u0, const double *p, double *out);

/* Reverse pass — replaces adjoint_sensitivities(sol, alg; t, dgdu_discrete, sensealg, checkpoints)
 * (src/sensitivity_interface.jl:373-526; Quadrature: src/quadrature_adjoint.jl:510-633; Gauss:
 * src/gauss_adjoint.jl:766-870), i.e. the pullback body of src/concrete_solve.jl:955-986.
 * dLdu [N][M][n] cotangents (COTANGENT loss) or NULL (LSQ_SHIFT); du0 [N][n]; dp [np] or [N][np]. */
int hipadj_adjoint(hipadj_handle *h, const double *dLdu, double *du0, double *dp);

/* Device-pointer variants (same layouts, memory of cfg.device); asynchronous on the handle's stream.
 * Exceptions, all synchronising the stream inside the call: (i) the FIRST hipadj_adjoint_dev of a handle whose runtime-compiled reverse kernel is
 * cross-checked against its -O1 build (kernels with >= 1 KB of scratch per lane; every reverse kernel when the toolkit's compiler could not be bound:
 * hipadj_runtime_compiler) runs the pass twice and compares du0 / dp on the host — do not put that call inside a stream capture; (ii) adaptive Tsit5 with
 * max_steps = 0 reads the measured step counts back after the forward solve (and after QuadratureAdjoint's sweep). */
int hipadj_forward_dev(hipadj_handle *h, const double *d_u0, const double *d_p, double *d_out);
int hipadj_adjoint_dev(hipadj_handle *h, const double *d_dLdu, double *d_du0, double *d_dp);
/* The data block of a device-resident loss (HIPADJ_LOSS_LSQ_DATA; HIPADJ_LOSS_MODEL bodies see it as `d`): data [N][M][n] in the layout of `out`, copied into the handle
 * (the lane family keeps it transposed like its cotangents, so the reverse pass streams it coalesced and needs no per-pass transposition).  Call again to change it;
 * the _dev form is asynchronous on the handle's stream.  A handle of kind HIPADJ_LOSS_LSQ_DATA without a block refuses hipadj_adjoint with HIPADJ_ERR_STATE. */
int hipadj_set_loss_data(hipadj_handle *h, const double *data);
int hipadj_set_loss_data_dev(hipadj_handle *h, const double *d_data);
/* The loss itself, summed over the ensemble, from the primal output `out` [N][M][n] of the last forward solve: LSQ_SHIFT: sum |u - shift|^2 / 2; LSQ_DATA: loss_scale / 2 *
 * sum |u - data|^2; HIPADJ_LOSS_MODEL: sum l_i for models registered with hipadj_model_set_discrete_loss_function.  The loss time at t0 is left out under no_start.
 * Fixed summation order (bit-reproducible).  _dev: out and loss [1] in device memory, asynchronous. */
int hipadj_loss_value(hipadj_handle *h, const double *out, double *loss);
int hipadj_loss_value_dev(hipadj_handle *h, const double *d_out, double *d_loss);
/* Cotangents handed over ALREADY in the lane family's streaming layout: Delta_soa[(i_time * n + j) * ld + trajectory] with ld = hipadj_soa_stride (N rounded up to 64) —
 * the reverse pass then reads the block in place (hipadj_adjoint_dev transposes [N][M][n] into this layout first: one more launch, 2 x 8 n M N bytes of traffic).
 * Lane-per-trajectory models (both steppers); other families: HIPADJ_ERR_UNSUPPORTED. */
int hipadj_adjoint_dev_soa(hipadj_handle *h, const double *d_dLdu_soa, double *d_du0, double *d_dp);
int hipadj_soa_stride(hipadj_handle *h, int64_t *ld);
/* run on the caller's hipStream_t (e.g. torch's current stream); NULL restores the handle's own stream */
int hipadj_set_stream(hipadj_handle *h, void *hip_stream);
int hipadj_synchronize(hipadj_handle *h);
/* device timing recorded per adjoint call: 0 none, 1 dominant-kernel bracket (2 events), 2 + whole-call bracket (default).
 * Each event costs a few microseconds of queue time; throughput-critical callers use 0 or 1. */
int hipadj_set_timing(hipadj_handle *h, int level);

int hipadj_get_stats(hipadj_handle *h, hipadj_stats *stats);

/* Sharded ensembles — replaces the reference's EnsembleDistributed pattern (test/Core4/distributed.jl, docs/src/tutorials/
 * data_parallel.md:77-136: every worker solves its own trajectories, the outer loss sums them).  One process per GPU, one
 * handle per process holding a contiguous trajectory range; trajectories never interact, so the ONLY exchange is the sum of
 * dL/dp over the shards when p is shared.  A handle that carries a communicator all-reduces dp[np] over RCCL (xGMI inside a
 * node) in-stream at the end of every hipadj_adjoint / hipadj_adjoint_dev call; du0 stays sharded.  RCCL is bound with dlopen
 * at the first of these calls (a library already loaded by the process — torch's — is reused).
 *   hipadj_comm_unique_id  rank 0: a fresh 128-byte id (ncclGetUniqueId) that the host ships to the other ranks by its own
 *                          means (Julia Distributed, MPI, torch.distributed, a file)
 *   hipadj_comm_init_rank  collective over the nranks processes: ncclCommInitRank on the handle's device; owned by the handle
 *   hipadj_comm_attach     use an existing ncclComm_t of the host instead (not owned; NULL detaches)
 *   hipadj_comm_destroy    drops the communicator (hipadj_destroy does it as well)
 *   hipadj_comm_count      ranks of the handle's communicator (ncclCommCount); 0 = no communicator, dp is the shard's own sum
 *   hipadj_comm_selfcheck  collective: all-reduces a known probe exactly as dp is all-reduced and checks the sum — call once
 *                          after hipadj_comm_init_rank / _attach on every rank (synchronises the handle's stream)
 * Summation order depends on the shard count: compare results across shard counts at rtol 1e-12, not bitwise. */
#define HIPADJ_COMM_ID_BYTES 128
int hipadj_comm_unique_id(char *id /* [HIPADJ_COMM_ID_BYTES] */);
int hipadj_comm_init_rank(hipadj_handle *h, const char *id /* [HIPADJ_COMM_ID_BYTES] */, int nranks, int rank);
int hipadj_comm_attach(hipadj_handle *h, void *nccl_comm);
int hipadj_comm_destroy(hipadj_handle *h);
int hipadj_comm_count(hipadj_handle *h, int *nranks);
int hipadj_comm_selfcheck(hipadj_handle *h);
/* on = 1: the all-reduce of dp moves to a second stream of the handle: hipadj_adjoint_dev enqueues the reverse pass on the handle's stream and the all-reduce behind it on
 * that second stream, so that the NEXT reverse pass does not wait for the collective's latency (strong scaling: a 24-byte all-reduce costs 10-20 us next to a 30 us shard
 * pass).  Contract: the caller alternates between TWO dp buffers from call to call; dp of call k is complete after hipadj_synchronize (which waits for both streams) — or
 * once call k + 2 has been enqueued and the handle's stream has reached it.  on = 0 (default): in-stream, dp complete in stream order.  ABI 107. */
int hipadj_comm_overlap(hipadj_handle *h, int on);

#ifdef __cplusplus
}
#endif
#endif /* HIPADJ_H */
