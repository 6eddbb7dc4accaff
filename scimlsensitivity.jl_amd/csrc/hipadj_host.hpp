// hipadj_host.hpp — host-side state shared by the translation units of libhipadj.so: the handle, the error macros and the
// small launch helpers.  The library is built from several translation units compiled in parallel (build.py): hipadj_api.hip
// (C ABI, planning, runtime-compiled models) and one unit per kernel family / compiled-in model (hipadj_tu_*.hip), which
// instantiate the launch sequences of hipadj_host_impl.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
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
#include "hipadj_field_etd.hpp"
#include "hipadj_mlp.hpp"
#include "hipadj_mlp_grad.hpp"
#include "hipadj_adaptive.hpp"
#include "hipadj_wide.hpp"

using namespace hipadj;
#ifdef HIPADJ_WAVE_TRACE
extern unsigned long long* g_hipadj_wave_trace;      // development builds only: defined in hipadj_api.hip, set through hipadj_debug_set_trace
#endif

#define HIPADJ_WIDE_MAXSEG 32   // segments of one adaptive Gauss-Kronrod quadrature in the wide family (segment integrals are np-vectors in HBM scratch)

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
    bool mlp = false; int NQ = 0;
    double *d_c1 = nullptr;               // MLP family: the workgroups' partial gradients (Quadrature: of one launch of panels)
    // MLP family, QuadratureAdjoint: host-driven adaptive Gauss-Kronrod (mlp_quadrature): pool of panel vectors, panel list, id lists, norms
    double *d_mq_pool = nullptr, *d_mq_norm = nullptr; void* d_mq_panels = nullptr; int *d_mq_ids = nullptr;
    long mq_pool_cap = 0; int mq_chunk = 0;
    std::vector<double> qa_host, qb_host;
    MlpGeom mg{};
    FieldGeom fg{};
    bool user = false;                    // runtime-compiled right-hand side (hipadj_user.hpp)
    bool wide = false;                    // ... of the workgroup-per-trajectory family (hipadj_wide.hpp)
    WideGeom wg{}; int wide_T = 0; double* d_wscr = nullptr;
    bool wide_ts5 = false; WideAdapt wa{};       // adaptive Tsit5 of the wide family (hipadj_wide.hpp): records d_rec [N][Smax][2 + 5 n], d_nsteps [N]
    bool dae = false;      // the model carries a singular semi-explicit mass matrix (hipadj_model_set_mass_matrix): the loss jumps of the reverse sweep leave a parameter term in dp_traj
    bool has_mm = false; double minv[64] = {0};   // the model's mass matrix at create time (M^{-1}, row-major): du0 = M^{-T} nu(t0) after the sweep
    hipModule_t umod = nullptr;
    hipFunction_t uf_forward = nullptr, uf_main = nullptr, uf_tail = nullptr, uf_gk = nullptr;   // tail = k_compose_finish or k_finish
    hipModule_t umod_alt = nullptr;            // the same kernels at -O1: second opinion for reverse kernels that spill heavily (user_prepare)
    hipFunction_t uf_main_alt = nullptr;
    hipFunction_t uf_aux = nullptr;        // a fifth kernel of the model's module: QuadratureAdjoint with loss times off the step grid (uf_gk carries out = sol(ts), the GK pass sits here)
    int wide_SmaxI = 0;                    // wide models, adaptive stepper, checkpointing = true: the step capacity of the one-interval record a trajectory re-solves into
    // checkpointing = true over the reverse step list (k_offgrid_ckpt): per interval (S, q_lo, q_hi) [3][nint] ints, the last step lengths [nint], the per-lane knot tile
    bool og_ck = false; int og_nint = 0; int* d_og_i = nullptr; double* d_og_h = nullptr; dbl2* d_og_tile = nullptr;
    int rtc_selftest = 0;                      // 1: pending (first adjoint call runs both builds and compares), 2: agreed, 3: disagreed -> the -O1 build is used, 4: disagreed, the -O1 build irreproducible -> the -O3 build stays
    bool adaptive = false;                // adaptive Tsit5 (hipadj_adaptive.hpp)
    AdaptGeom ag{};
    int cbs = 0;         // k_compose_finish workgroup size: 0 = by ensemble size, 64 / 256 forced (HIPADJ_CBS; tuning study)
    bool wpb4 = false;   // k_interp in 256-thread workgroups (HIPADJ_WPB=4; tuning study)
    bool no_ops = false; // HIPADJ_NO_OPS=1: the generic vjp_u / vjp_p form of the multi-column step also for models with stage operators (A/B study)
    double *d_rec = nullptr, *d_save_t = nullptr, *d_ck_t = nullptr, *d_tstops = nullptr, *d_arec = nullptr;
    int *d_nsteps = nullptr, *d_nsteps_adj = nullptr, ntstops = 0, SmaxA = 0;
    int *d_ev_s = nullptr, *d_nev = nullptr, *d_ev_k = nullptr, maxev = 0; double *d_ev_t = nullptr, *d_ev_ul = nullptr, *d_ev_ur = nullptr, *d_ev_dl = nullptr, *d_ev_dr = nullptr;      // ContinuousCallback of a runtime model: per-trajectory event lists (AdaptGeom::ev_s / nev), capacity maxev
    bool auto_steps = false;              // max_steps == 0: record capacity sized from a counting pass of the forward solve
    long rec_cap = 0;                     // accepted steps the record buffer(s) currently hold per trajectory
    unsigned* d_ticket = nullptr;
    int* d_save_rev = nullptr;            // the reverse kernels' loss-time map: d_save_of_knot itself, or an own copy when no_start clears the jump at T (hipadj_plan.hpp)
    int *d_prev_ck = nullptr, *d_save_of_knot = nullptr, *d_ckpt_of_knot = nullptr, *d_seg_bounds = nullptr, *d_flag = nullptr;
    const double* p_dev_last = nullptr;  // device p used by the last forward (the adjoint reuses it)
    bool have_forward = false, timing_pending_fwd = false;
    int fused_final = 0;                  // 1: dp reduced in-launch by the last-arriving workgroup (HIPADJ_FUSED_FINAL)
    // one launch per reverse pass (hipadj_fused.hpp): composition tree + dp reduction inside the sweep kernel.  fused: 1 = on (default where a
    // fused kernel exists), 0 = the three-launch sequence (HIPADJ_FUSED=0: A/B and fallback)
    int fused = 1;
    int fgroup = 0;                       // G > 0: G waves (consecutive segments of a trajectory block) per workgroup, first composition level in LDS (k_interp_fused_g; 4 or 8)
    TreePlan tp{};
    double* d_tbuf = nullptr; unsigned* d_tcnt = nullptr; long tcnt_n = 0;
    int *d_fev_knot = nullptr, *d_fev_save = nullptr, *d_fev_ckpt = nullptr, nfev = 0;   // event knots of the forward solve (k_forward_ev)
    int fwd_ev = 1;                       // HIPADJ_FWD_EV=0: the per-knot form k_forward (A/B)
    bool wide_auto = false;               // wide models, adaptive, max_steps = 0 and a budget-limited record: step counts are read back after every forward solve (wide_autosize)
    int wide_KT = 0;                      // wide models, checkpointing = true: knots of one re-solve tile (longest checkpoint interval + 1)
    // four lanes per trajectory (hipadj_quad.hpp, hipadj_quad_ts5.hpp) for models with a component form — while the quads do not outnumber the SIMDs' wavefront slots: a quad
    // kernel launches 4 x the wavefronts of the lane kernel with ~0.7 x the instructions each, which pays only as long as those wavefronts find idle SIMDs.  Measured, Lorenz
    // (profiles/r4_quad_vs_lane_by_N.log): forward RK4 0.117 vs 0.164 ms at 10^4 trajectories, 0.223 vs 0.167 at 2 x 10^4; Tsit5 Interpolating sweep 1.52 vs 1.80 at 2 x 10^4,
    // 2.57 vs 2.09 at 4 x 10^4.  Forward solves: N <= 16 SIMDs (one wavefront per SIMD); reverse sweeps (two 209-register wavefronts fit a SIMD): N <= 32 SIMDs.
    // HIPADJ_QUAD=0: never, 2: always (A/B).
    int quad_fwd = 1, quad_adj = 1;
    int timing = 2;                       // 0: no events, 1: dominant-kernel bracket only, 2: + whole-call bracket (HIPADJ_TIMING)
    double ws_bytes = 0;
    double* d_gtile = nullptr; long gtile_stride = 0; bool ck_long = false;   // checkpoint intervals longer than HIPADJ_CKPT_KMAX: re-solve tiles in HBM
    bool offgrid = false;                 // fixed-step RK4 with loss times off the step grid: reverse step list on the device
    double *d_rs_t = nullptr, *d_rs_h = nullptr, *d_rs_te = nullptr; int *d_rs_save = nullptr, *d_rs_ck = nullptr; int nrs = 0, rs_save_at_start = -1;
    // device-resident discrete losses (HIPADJ_LOSS_LSQ_DATA / HIPADJ_LOSS_MODEL): the data block of hipadj_set_loss_data.  Lane family: transposed into d_cotT (the reverse
    // kernels stream it like a cotangent block); workgroup families: d_ldata [N][M][n] in the caller's layout, handed to the kernels in the cotangents' place
    double* d_ldata = nullptr; bool have_ldata = false;
    const double* cot_soa = nullptr;      // set for the duration of a hipadj_adjoint_dev_soa call
    // host-pointer calls: the [N][M][n] block (cotangents up, out = sol(ts) down) goes through a PINNED staging buffer of the handle — the caller's arrays are pageable
    // (a Julia / numpy array), and a pageable hipMemcpyAsync of 24 MB ran at 3 GB/s (profiles/r5_visit1_bench.json: 7.7 ms for the Delta of BASELINE configs[1] against 0.39 ms
    // of link time); host threads copy into the pinned block, ONE DMA moves it
    hipEvent_t xfer_ev[16] = {}; int xfer_nev = 0, xfer_chunks = 0;      // events behind the chunks of a pipelined device-to-host staging copy (hipadj_api.hip stage_down_*)
    double* h_pin = nullptr; size_t pin_count = 0, pin_map_len = 0;      // pin_map_len: bytes of the mmap region behind h_pin (guard pages included)
    hipModule_t lmod = nullptr; hipFunction_t lf_value = nullptr;   // runtime model with a discrete-loss FUNCTION: its loss-value kernel (hipadj_loss_value)
    double* d_lpart = nullptr;            // per-workgroup partials of hipadj_loss_value
    double* d_lval = nullptr;             // ... and the one double its host-pointer form (and a non-local shard of a multi handle) receives the value in
    // ONE handle over several devices (hipadj_multi.hpp): the shards are ordinary handles on contiguous trajectory ranges; everything above is unused in a multi handle
    // except cfg, n, np, N, M, the (primary) stream and err
    bool multi = false;
    std::vector<hipadj_handle*> shards; std::vector<long> shard_off; std::vector<int> dev_ids;
    std::vector<hipEvent_t> shard_ev; hipEvent_t multi_in = nullptr;
    double* d_dp_parts = nullptr;         // primary device: the shards' dp partials [G][np] + G doubles (loss values)
    std::vector<double> dp_host;          // host-pointer calls: the shards' dp partials
    // a dense chain the library runs on the FP64-MFMA family (hipadj_route.hpp): `inner` is the MLP handle whose batch columns are this handle's trajectories; everything
    // above is unused in a routed handle except cfg, n, np, N, M, S, err and the two transposition blocks
    bool route = false; hipadj_handle* inner = nullptr; double *d_rt_a = nullptr, *d_rt_b = nullptr;
    void* comm = nullptr;                 // ncclComm_t of the ensemble shards (hipadj_comm.hpp); dp is all-reduced over it
    bool comm_owned = false;
    // hipadj_comm_overlap: the all-reduce of dp on its own stream, off the critical path of the next reverse pass
    int comm_overlap = 0; hipStream_t comm_stream = nullptr; hipEvent_t comm_ready = nullptr, comm_done[2] = {nullptr, nullptr}; unsigned comm_seq = 0; long comm_test_delay = 0;
    hipadj_stats st{};
    std::string err;
};

#define HIPADJ_FAIL(h, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (h)->err = _b; return (code); } while (0)
#define HIP_TRY(h, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    HIPADJ_FAIL(h, HIPADJ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

template <class T> static int dev_alloc(hipadj_handle* h, T** p, size_t count) {
    if (count == 0) { *p = nullptr; return HIPADJ_OK; }
    HIP_TRY(h, hipMalloc((void**)p, count * sizeof(T)));
    h->ws_bytes += (double)(count * sizeof(T));
    return HIPADJ_OK;
}
#define TRY(expr) do { int _rc = (expr); if (_rc != HIPADJ_OK) return _rc; } while (0)

// does the reverse pass stream a column next to the state at the loss times — cotangents, or the data block of a device-resident loss?  (template bit 0 of the lane kernels' MODE)
static inline bool loss_streams(const hipadj_handle* h) { return h->M > 0 && h->cfg.loss_kind != HIPADJ_LOSS_LSQ_SHIFT; }

static inline void harvest_set(hipadj_handle* h, hipadj_handle::EvSet& q, bool block) {
    if (!q.pending) return;
    hipEvent_t last = q.full ? q.a1 : q.k1;
    if (block) { if (hipEventSynchronize(last) != hipSuccess) { q.pending = false; return; } }
    else if (hipEventQuery(last) != hipSuccess) return;           // still running: look again later
    float ms = 0.f;
    if (q.full && hipEventElapsedTime(&ms, q.a0, q.a1) == hipSuccess) { h->st.adjoint_ms_last = ms; h->st.adjoint_ms_total += ms; }
    if (hipEventElapsedTime(&ms, q.k0, q.k1) == hipSuccess) { h->st.adjoint_main_kernel_ms_last = ms; h->st.adjoint_main_kernel_ms_total += ms; }
    q.pending = false;
}

// ---- launch helpers --------------------------------------------------------------------------------------
static inline int launch_transpose_to_soa(hipadj_handle* h, const double* src, double* dst, int C) {
    dim3 blk(32, 8), grd((unsigned)((h->Npad + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_aos_to_soa, grd, blk, 0, h->stream, src, dst, h->N, h->Npad, C);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
static inline int launch_transpose_to_aos(hipadj_handle* h, const double* src, double* dst, int C) {
    dim3 blk(32, 8), grd((unsigned)((h->N + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_soa_to_aos, grd, blk, 0, h->stream, src, dst, h->N, h->Npad, C);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}

// ---- launch sequences per kernel family: defined in hipadj_host_impl.hpp, instantiated explicitly by the hipadj_tu_*.hip units
template <class Mo> int forward_impl(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out);
template <class Mo> int adjoint_impl(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp);
template <class Mo> int adaptive_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out);
template <class Mo> int adaptive_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp);
template <int G> int field_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out);
template <int G> int field_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp);
template <int H> int mlp_forward_launch(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out);
template <int H> int mlp_adjoint_launch(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp);
int adaptive_autosize(hipadj_handle* h);   // record capacity from the counting pass (max_steps == 0); defined in hipadj_api.hip
int adaptive_adjoint_autosize(hipadj_handle* h);   // the same for QuadratureAdjoint's dense adjoint record
