// hipadj_route.hpp — family selection INSIDE the library (ABI 109; VERDICT r5 missing 4 / next 4).
//
// A wide runtime model declared as a dense chain (hipadj_wmodel_declare_dense_chain) keeps its SPMD bodies as its definition for the workgroup-per-trajectory family; for the
// shape the FP64-MFMA family is built for — 2 -> H -> H -> 2, H in {32, 64, 128}, tanh, shared weights — hipadj_create builds the handle on THAT family instead: the N
// trajectories of the ensemble are the B = N batch columns of ONE trajectory of HIPADJ_MODEL_MLP (csrc/hipadj_mlp*.hpp: the H x H contractions on the matrix cores; measured
// 19-91 x the workgroup family at these widths, profiles/r5_dense_chain_crossover.json).  Round 5 had this decision in the Python mirror only (interface._mfma_route): a Julia
// host never reached the MFMA kernels for a registered chain.
//
// A routed handle is a thin shell around the inner MLP handle.  Shapes seen by the caller stay those of the ensemble — u0 / du0 [N][d], out / dLdu / data [N][M][d], dp [np] —
// and differ from the inner handle's only in the order of the two leading axes of the [N][M][d] blocks (the batched state at one time is [B][d] = the ensemble's [N][d] in the
// same memory order; the time axis moves in front): one small transposition kernel per block and call, on the handle's stream.  When the MLP family refuses the configuration
// (loss times off the step grid, a cost, ...) hipadj_create falls back to the registered family: routing is an optimisation, never a new way to fail.
#pragma once
#include "hipadj_host.hpp"

static int upload_block(hipadj_handle* h, double* d_dst, const double* src, size_t count);
static int adjoint_host_download(hipadj_handle* h, double* du0, double* dp);
static int download_block(hipadj_handle* h, double* dst, const double* d_src, size_t count);

// out[b][a][:] = in[a][b][:]   (A x B blocks of d doubles)
static __global__ void k_swap_leading(long A, long B, long d, const double* __restrict__ in, double* __restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A * B * d) return;
    const long j = e % d, ab = e / d, b = ab % B, a = ab / B;
    out[(b * A + a) * d + j] = in[e];
}
static int route_swap(hipadj_handle* h, long A, long B, const double* in, double* out) {
    const long tot = A * B * h->n;
    if (tot <= 0) return HIPADJ_OK;
    hipLaunchKernelGGL(k_swap_leading, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->inner->stream, A, B, (long)h->n, in, out);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}

// Is *cfg a configuration the FP64-MFMA family takes for the model's declared chain?  (the conditions of round 5's interface._mfma_route)
static bool route_eligible(const hipadj_config* cfg, int& H) {
    if (cfg->family != HIPADJ_FAMILY_AUTO || cfg->model < HIPADJ_MODEL_USER_BASE) return false;
    if (const char* e = std::getenv("HIPADJ_DENSE_CHAIN_ROUTE")) if (e[0] == '0') return false;      // A/B hook
    std::vector<int> w; int power = 1;
    if (!user_dense_chain(cfg->model, w, power)) return false;
    if (w.size() != 4 || w[0] != 2 || w[3] != 2 || w[1] != w[2] || (w[1] != 32 && w[1] != 64 && w[1] != 128) || power != 1) return false;
    if (user_has_cost(cfg->model)) return false;
    if (cfg->stepper != HIPADJ_STEPPER_RK4_FIXED || !cfg->p_shared || cfg->ntraj < 16 || cfg->ntraj % 16 != 0 || cfg->ntraj > (1L << 24)) return false;
    if (cfg->alg != HIPADJ_ALG_GAUSS && cfg->alg != HIPADJ_ALG_INTERPOLATING && cfg->alg != HIPADJ_ALG_BACKSOLVE && cfg->alg != HIPADJ_ALG_QUADRATURE) return false;
    if (cfg->loss_kind == HIPADJ_LOSS_MODEL || cfg->cont_cost != HIPADJ_CCOST_NONE || cfg->ncheckpoints > 0 || cfg->ndevices > 1) return false;
    if (cfg->checkpointing && cfg->alg != HIPADJ_ALG_BACKSOLVE) return false;
    H = w[1];
    return true;
}

static void route_free(hipadj_handle* h) {
    if (h->inner) { (void)hipSetDevice(h->inner->cfg.device); (void)hipStreamSynchronize(h->inner->stream); }
    if (h->d_rt_a) (void)hipFree(h->d_rt_a);
    if (h->d_rt_b) (void)hipFree(h->d_rt_b);
    if (h->inner) (void)hipadj_destroy(h->inner);
    h->inner = nullptr;
}

// HIPADJ_OK with *out set: routed.  HIPADJ_OK with *out == NULL: not routed (the MLP family refused; the caller continues with the registered family).
static int route_create(const hipadj_config* cfg, int H, hipadj_handle** out) {
    *out = nullptr;
    hipadj_config c = *cfg;
    c.model = HIPADJ_MODEL_MLP; c.dims[0] = 2; c.dims[1] = H; c.dims[2] = (int32_t)cfg->ntraj; c.dims[3] = 0; c.ntraj = 1; c.family = HIPADJ_FAMILY_AS_REGISTERED;
    hipadj_handle* in = nullptr;
    const int rc = hipadj_create(&c, &in);
    if (rc == HIPADJ_ERR_UNSUPPORTED || rc == HIPADJ_ERR_INVALID_ARG) return HIPADJ_OK;      // e.g. loss times off the step grid: the registered family takes those
    if (rc != HIPADJ_OK) return rc;
    auto* h = new hipadj_handle();
    h->route = true; h->inner = in; h->cfg = *cfg; h->cfg.save_times = nullptr; h->cfg.checkpoints = nullptr; h->cfg.device_ids = nullptr;
    h->N = cfg->ntraj; h->n = 2; h->np = in->np; h->M = in->M; h->S = in->S; h->stream = in->stream; h->save_times = in->save_times;
    const size_t blk = (size_t)h->N * (size_t)std::max(h->M, 1) * (size_t)h->n;
    if (hipSetDevice(cfg->device) != hipSuccess || hipMalloc((void**)&h->d_rt_a, sizeof(double) * blk) != hipSuccess || hipMalloc((void**)&h->d_rt_b, sizeof(double) * blk) != hipSuccess) {
        (void)hipGetLastError(); route_free(h); delete h; return HIPADJ_OK; }
    h->st.struct_size = sizeof(hipadj_stats);
    *out = h;
    return HIPADJ_OK;
}

static int route_fail(hipadj_handle* h, int rc) { h->err = hipadj_last_error(h->inner); return rc; }
#define ROUTE_TRY(h, call) do { const int rc_ = (call); if (rc_ != HIPADJ_OK) return route_fail((h), rc_); } while (0)

static int route_forward_dev(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    hipadj_handle* in = h->inner;
    ROUTE_TRY(h, hipadj_forward_dev(in, d_u0, d_p, (d_out && h->M > 0) ? h->d_rt_a : nullptr));      // [M][N][d]
    if (d_out && h->M > 0) TRY(route_swap(h, h->M, h->N, h->d_rt_a, d_out));
    h->have_forward = true;
    return HIPADJ_OK;
}
static int route_adjoint_dev(hipadj_handle* h, const double* d_dLdu, double* d_du0, double* d_dp) {
    hipadj_handle* in = h->inner;
    const bool cot = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0;
    if (cot && !d_dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    if (cot) TRY(route_swap(h, h->N, h->M, d_dLdu, h->d_rt_b));
    ROUTE_TRY(h, hipadj_adjoint_dev(in, cot ? h->d_rt_b : nullptr, d_du0, d_dp));
    return HIPADJ_OK;
}
static int route_forward(hipadj_handle* h, const double* u0, const double* p, double* out) {
    hipadj_handle* in = h->inner;
    HIP_TRY(h, hipSetDevice(in->cfg.device));
    ROUTE_TRY(h, upload_block(in, in->d_u0, u0, (size_t)h->N * h->n));      // (through the inner handle's staging block, like every host-pointer transfer: hipadj_api.hip)
    ROUTE_TRY(h, upload_block(in, in->d_p, p, (size_t)h->np));
    TRY(route_forward_dev(h, in->d_u0, in->d_p, out ? h->d_rt_b : nullptr));
    if (out && h->M > 0) ROUTE_TRY(h, download_block(in, out, h->d_rt_b, (size_t)h->N * h->M * h->n));
    ROUTE_TRY(h, hipadj_synchronize(in));
    return HIPADJ_OK;
}
static int route_adjoint(hipadj_handle* h, const double* dLdu, double* du0, double* dp) {
    hipadj_handle* in = h->inner;
    HIP_TRY(h, hipSetDevice(in->cfg.device));
    const bool cot = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0;
    if (cot && !dLdu) HIPADJ_FAIL(h, HIPADJ_ERR_INVALID_ARG, "dLdu required for HIPADJ_LOSS_COTANGENT");
    if (cot) ROUTE_TRY(h, upload_block(in, h->d_rt_a, dLdu, (size_t)h->N * h->M * h->n));
    TRY(route_adjoint_dev(h, cot ? h->d_rt_a : nullptr, in->d_du0, in->d_dp));
    ROUTE_TRY(h, adjoint_host_download(in, du0, dp));
    ROUTE_TRY(h, hipadj_synchronize(in));
    return HIPADJ_OK;
}
static int route_set_loss_data(hipadj_handle* h, const double* data, bool dev) {
    hipadj_handle* in = h->inner;
    HIP_TRY(h, hipSetDevice(in->cfg.device));
    if (h->M <= 0) return HIPADJ_OK;
    const double* src = data;
    if (!dev) { ROUTE_TRY(h, upload_block(in, h->d_rt_a, data, (size_t)h->N * h->M * h->n)); src = h->d_rt_a; }
    TRY(route_swap(h, h->N, h->M, src, h->d_rt_b));
    ROUTE_TRY(h, hipadj_set_loss_data_dev(in, h->d_rt_b));
    if (!dev) HIP_TRY(h, hipStreamSynchronize(in->stream));      // the caller's buffer may go away
    h->have_ldata = true;
    return HIPADJ_OK;
}
static int route_loss_value(hipadj_handle* h, const double* out, double* loss, bool dev) {
    hipadj_handle* in = h->inner;
    HIP_TRY(h, hipSetDevice(in->cfg.device));
    const double* src = out;
    if (!dev) { ROUTE_TRY(h, upload_block(in, h->d_rt_a, out, (size_t)h->N * h->M * h->n)); src = h->d_rt_a; }
    TRY(route_swap(h, h->N, h->M, src, h->d_rt_b));
    if (dev) { ROUTE_TRY(h, hipadj_loss_value_dev(in, h->d_rt_b, loss)); return HIPADJ_OK; }
    ROUTE_TRY(h, hipadj_loss_value_dev(in, h->d_rt_b, h->d_rt_a));      // (the uploaded block has been consumed by the transposition)
    ROUTE_TRY(h, download_block(in, loss, h->d_rt_a, 1));
    return HIPADJ_OK;
}
static int route_get_stats(hipadj_handle* h, hipadj_stats* st) {
    ROUTE_TRY(h, hipadj_get_stats(h->inner, st));
    st->n = h->n; st->ntraj = h->N; st->vjp_steps = (double)h->N * h->S * 4.0; st->routed_family = HIPADJ_FAMILY_MFMA;
    return HIPADJ_OK;
}
