// hipadj_field.hpp — workgroup-per-trajectory kernel family for PDE-sized states (BASELINE config 5:
// 2-D Brusselator, 32 x 32 grid, n = 2048, QuadratureAdjoint).
//
// One workgroup integrates one trajectory.  Each thread owns Q grid cells (both species) in VGPRs; the only
// coupling between cells is the periodic 5-point Laplacian, served from an LDS copy of the current stage vector
// (double-buffered: ONE s_barrier per RK stage).  The VJP of the sparse Jacobian is evaluated matrix-free:
//     (df/du)^T lam = adx * L lam  +  local 2x2 reaction block^T lam          (L symmetric under periodic BCs)
//     (df/dp)^T lam = sum over cells of (dA, dB, dalpha) partials               -> per-thread partial sums, reduced
//                                                                                 across the workgroup ONCE at the end
// so the parameter-gradient accumulators never cost a reduction inside the time loop.  dalpha uses
// sum_c (L y)_c lam_c == sum_c y_c (L lam)_c, which reuses the Laplacian the lambda equation needs anyway.
//
// What is restated (reference = SciMLSensitivity.jl):
//   right-hand side           docs/src/examples/pde/brusselator.md:98-112 (periodic wrap `limit`, forcing :85)
//   Interpolating RHS         src/interpolating_adjoint.jl:150-174     Gauss   src/gauss_adjoint.jl:118-128, 745-759, 809-851
//   Quadrature                src/quadrature_adjoint.jl:35-46, 486-502, 510-616 (block-uniform adaptive GK15)
//   loss jumps                src/adjoint_common.jl:754-821
// BacksolveAdjoint is not offered for this family: integrating a diffusion equation backward in time is
// ill-posed (the reference warns likewise, src/sensitivity_algorithms.jl:168-198).
//
// Layouts (trajectory-major, cell index fastest => coalesced per workgroup):
//   knots [N][S+1][2][n]   (u_k, f(u_k));   out / cot [N][M][n] (the caller's layout, used in place);
//   adj   [N][S][4][n]     (lam_start, lam'_start, lam_end, lam'_end) for QuadratureAdjoint.
#pragma once

#include <hip/hip_runtime.h>
#include "hipadj_lane.hpp"

namespace hipadj {

// Gauss-Kronrod (7,15) tables in constant memory (runtime-indexed by the non-unrolled panel loop)
__constant__ double c_gk_x[8] = {0.991455371120812639206854697526329, 0.949107912342758524526189684047851,
                                 0.864864423359769072789712788640926, 0.741531185599394439863864773280788,
                                 0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
                                 0.207784955007898467600689403773245, 0.0};
__constant__ double c_gk_wk[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                  0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                  0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                  0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
__constant__ double c_gk_wg[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                                  0.381830050505118944950369775488975, 0.417959183673469387755102040816327};

struct FieldGeom {
    long N;
    int S, M;
    double t0, dt, loss_shift;
    int loss_kind, no_start, p_shared;   // loss_kind: hipadj_loss 0 cotangent, 1 lsq_shift, 2 lsq_data (dgdu = lsq_w (u - data), the block in the cotangents' place)
    double lsq_w;
    int cont_cost;      // hipadj_cont_cost 0 none, 1 g = (sum u)^2 / 2, 2 g = u_1^2 + p_1 (round 5: accumulate_cost!, src/derivative_wrappers.jl:1411-1442, on the PDE family)
};

template <int G> struct Bruss {
    static constexpr int CELLS = G * G;
#ifndef HIPADJ_BRUSS_T
#define HIPADJ_BRUSS_T 1024   // 32 x 32: one cell per thread, 16 waves on the CU.  Measured against 256 threads x 4 cells (round 1): forward solve 0.88 -> 0.69 ms,
#endif                        // Interpolating 2.71 -> 1.95 us per step, Gauss 4.45 -> 3.26, Quadrature 3.47 -> 3.20 (N = 1) and 4.99 -> 4.66 (N = 256, HBM-bound)
    static constexpr int T = CELLS < HIPADJ_BRUSS_T ? CELLS : HIPADJ_BRUSS_T;   // threads per workgroup
    static constexpr int Q = CELLS / T;                   // cells per thread
    static constexpr int NS = 2 * CELLS;                  // state size n
    static_assert(CELLS % T == 0 && T % 64 == 0, "grid must tile the workgroup");
};

__device__ __forceinline__ double bruss_force(double x, double y, double t) {
    return (((x - 0.3) * (x - 0.3) + (y - 0.6) * (y - 0.6)) <= 0.01 && t >= 1.1) ? 5.0 : 0.0;
}

// per-thread neighbour table
template <int G> struct Nbr {
    int c[Bruss<G>::Q], im[Bruss<G>::Q], ip[Bruss<G>::Q], jm[Bruss<G>::Q], jp[Bruss<G>::Q];
    double x[Bruss<G>::Q], y[Bruss<G>::Q];
    __device__ __forceinline__ void init() {
        constexpr int T = Bruss<G>::T, Q = Bruss<G>::Q;
        const double dx = 1.0 / (G - 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int cc = threadIdx.x + q * T, i = cc % G, j = cc / G;
            c[q] = cc;
            im[q] = (i + G - 1) % G + j * G; ip[q] = (i + 1) % G + j * G;
            jm[q] = i + ((j + G - 1) % G) * G; jp[q] = i + ((j + 1) % G) * G;
            x[q] = i * dx; y[q] = j * dx;
        }
    }
};

// write a two-species stage vector to LDS buffer `buf` and return after the barrier
template <int G>
__device__ __forceinline__ void publish(double* __restrict__ buf, const Nbr<G>& nb, const double (&a)[Bruss<G>::Q], const double (&b)[Bruss<G>::Q]) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS;
#pragma unroll
    for (int q = 0; q < Q; ++q) { buf[nb.c[q]] = a[q]; buf[CELLS + nb.c[q]] = b[q]; }
    __syncthreads();
}
template <int G>
__device__ __forceinline__ void laplace(const double* __restrict__ buf, const Nbr<G>& nb, const double (&a)[Bruss<G>::Q], const double (&b)[Bruss<G>::Q],
                                        double (&La)[Bruss<G>::Q], double (&Lb)[Bruss<G>::Q]) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        La[q] = buf[nb.im[q]] + buf[nb.ip[q]] + buf[nb.jp[q]] + buf[nb.jm[q]] - 4.0 * a[q];
        Lb[q] = buf[CELLS + nb.im[q]] + buf[CELLS + nb.ip[q]] + buf[CELLS + nb.jp[q]] + buf[CELLS + nb.jm[q]] - 4.0 * b[q];
    }
}

struct BrussP { double A, B, alpha, adx, idx2; };
template <int G> __device__ __forceinline__ BrussP load_bruss_p(const double* __restrict__ p, int p_shared, long traj) {
    const double* pp = p_shared ? p : p + traj * 3;
    BrussP r; r.A = pp[0]; r.B = pp[1]; r.alpha = pp[2];
    const double dx = 1.0 / (G - 1);
    r.idx2 = 1.0 / (dx * dx); r.adx = r.alpha / (dx * dx);
    return r;
}

// f(u) at the thread's cells; (U,V) must already be published in `buf`
template <int G>
__device__ __forceinline__ void bruss_f(const double* __restrict__ buf, const Nbr<G>& nb, const BrussP& P, double t,
                                        const double (&U)[Bruss<G>::Q], const double (&V)[Bruss<G>::Q],
                                        double (&dU)[Bruss<G>::Q], double (&dV)[Bruss<G>::Q]) {
    constexpr int Q = Bruss<G>::Q;
    double LU[Q], LV[Q];
    laplace<G>(buf, nb, U, V, LU, LV);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        dU[q] = P.adx * LU[q] + P.B + U[q] * U[q] * V[q] - (P.A + 1.0) * U[q] + bruss_force(nb.x[q], nb.y[q], t);
        dV[q] = P.adx * LV[q] + P.A * U[q] - U[q] * U[q] * V[q];
    }
}

// (df/du)^T lam at the thread's cells (lam published in `buf`) and the per-cell parameter partials added into w[3]
template <int G, bool WITH_P>
__device__ __forceinline__ void bruss_vjp(const double* __restrict__ buf, const Nbr<G>& nb, const BrussP& P,
                                          const double (&U)[Bruss<G>::Q], const double (&V)[Bruss<G>::Q],
                                          const double (&lU)[Bruss<G>::Q], const double (&lV)[Bruss<G>::Q],
                                          double (&dlU)[Bruss<G>::Q], double (&dlV)[Bruss<G>::Q], double wgt, double (&w)[3]) {
    constexpr int Q = Bruss<G>::Q;
    double LlU[Q], LlV[Q];
    laplace<G>(buf, nb, lU, lV, LlU, LlV);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double uv2 = 2.0 * U[q] * V[q], uu = U[q] * U[q];
        dlU[q] = P.adx * LlU[q] + (uv2 - (P.A + 1.0)) * lU[q] + (P.A - uv2) * lV[q];
        dlV[q] = P.adx * LlV[q] + uu * lU[q] - uu * lV[q];
        if (WITH_P) {
            w[0] += wgt * (-U[q] * lU[q] + U[q] * lV[q]);
            w[1] += wgt * lU[q];
            w[2] += wgt * ((U[q] * LlU[q] + V[q] * LlV[q]) * P.idx2);
        }
    }
}

// workgroup sum of K per-thread values, result broadcast to every thread (fixed order: wave shuffle tree, waves in order)
template <int T, int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* __restrict__ red /* LDS [T/64][K] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double x = v[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) red[wv * K + j] = x;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j) { double s = 0.0;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) s += red[w * K + j];
        v[j] = s; }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_forward(FieldGeom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                               double* __restrict__ knots, double* __restrict__ out,
                                                               const int* __restrict__ save_of_knot) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    __shared__ double sh[2][NS];
    const long traj = blockIdx.x;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    double U[Q], V[Q], k1U[Q], k1V[Q], k2U[Q], k2V[Q], k3U[Q], k3V[Q], k4U[Q], k4V[Q], sU[Q], sV[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { U[q] = u0[traj * NS + nb.c[q]]; V[q] = u0[traj * NS + CELLS + nb.c[q]]; }
    const double dt = g.dt;
    for (int k = 0; k <= g.S; ++k) {
        const double t = g.t0 + k * dt;
        publish<G>(sh[0], nb, U, V);
        bruss_f<G>(sh[0], nb, P, t, U, V, k1U, k1V);
        if (knots) { double* kn = knots + ((traj * (g.S + 1) + k) * 2) * NS;
#pragma unroll
            for (int q = 0; q < Q; ++q) { kn[nb.c[q]] = U[q]; kn[CELLS + nb.c[q]] = V[q]; kn[NS + nb.c[q]] = k1U[q]; kn[NS + CELLS + nb.c[q]] = k1V[q]; } }
        if (out) { const int s = save_of_knot[k]; if (s >= 0) { double* o = out + (traj * g.M + s) * NS;
#pragma unroll
            for (int q = 0; q < Q; ++q) { o[nb.c[q]] = U[q]; o[CELLS + nb.c[q]] = V[q]; } } }
        if (k == g.S) break;
#pragma unroll
        for (int q = 0; q < Q; ++q) { sU[q] = U[q] + 0.5 * dt * k1U[q]; sV[q] = V[q] + 0.5 * dt * k1V[q]; }
        publish<G>(sh[1], nb, sU, sV);
        bruss_f<G>(sh[1], nb, P, t + 0.5 * dt, sU, sV, k2U, k2V);
#pragma unroll
        for (int q = 0; q < Q; ++q) { sU[q] = U[q] + 0.5 * dt * k2U[q]; sV[q] = V[q] + 0.5 * dt * k2V[q]; }
        publish<G>(sh[0], nb, sU, sV);
        bruss_f<G>(sh[0], nb, P, t + 0.5 * dt, sU, sV, k3U, k3V);
#pragma unroll
        for (int q = 0; q < Q; ++q) { sU[q] = U[q] + dt * k3U[q]; sV[q] = V[q] + dt * k3V[q]; }
        publish<G>(sh[1], nb, sU, sV);
        bruss_f<G>(sh[1], nb, P, t + dt, sU, sV, k4U, k4V);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            U[q] = U[q] + (dt / 6.0) * (k1U[q] + 2.0 * (k2U[q] + k3U[q]) + k4U[q]);
            V[q] = V[q] + (dt / 6.0) * (k1V[q] + 2.0 * (k2V[q] + k3V[q]) + k4V[q]);
        }
    }
}

// one thread's slice of a knot
template <int G> struct FKnot { double U[Bruss<G>::Q], V[Bruss<G>::Q], fU[Bruss<G>::Q], fV[Bruss<G>::Q]; };
template <int G>
__device__ __forceinline__ void load_fknot(const double* __restrict__ knots, const FieldGeom& g, long traj, int k, const Nbr<G>& nb, FKnot<G>& kn) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    const double* b = knots + ((traj * (g.S + 1) + k) * 2) * NS;
#pragma unroll
    for (int q = 0; q < Q; ++q) { kn.U[q] = b[nb.c[q]]; kn.V[q] = b[CELLS + nb.c[q]]; kn.fU[q] = b[NS + nb.c[q]]; kn.fV[q] = b[NS + CELLS + nb.c[q]]; }
}

template <int G>
__device__ __forceinline__ void field_jump(const FieldGeom& g, long traj, int s, const double* __restrict__ cot, const Nbr<G>& nb,
                                           const double (&U)[Bruss<G>::Q], const double (&V)[Bruss<G>::Q],
                                           double (&lU)[Bruss<G>::Q], double (&lV)[Bruss<G>::Q]) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    if (g.loss_kind == 0) {
        const double* c = cot + (traj * g.M + s) * NS;
#pragma unroll
        for (int q = 0; q < Q; ++q) { lU[q] += c[nb.c[q]]; lV[q] += c[CELLS + nb.c[q]]; }
    } else if (g.loss_kind == 2) {
        const double* c = cot + (traj * g.M + s) * NS;
#pragma unroll
        for (int q = 0; q < Q; ++q) { lU[q] += g.lsq_w * (U[q] - c[nb.c[q]]); lV[q] += g.lsq_w * (V[q] - c[CELLS + nb.c[q]]); }
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) { lU[q] += U[q] - g.loss_shift; lV[q] += V[q] - g.loss_shift; }
    }
}

// Continuous costs on the PDE family (round 5): the cost gradient g_u(y(t)) enters every stage next to J^T lam (dlam -= g_u in the reference's sign, V = J^T lam + g_u
// here), g_p the gradient quadrature.  HIPADJ_CCOST_HALF_SQ_SUM needs sum(y) at the three stage points of a step: the sums of (u, f) over a knot's cells are taken once
// per knot (field_knot_sums: one workgroup reduction), the Hermite midpoint of the sums is the sum of the midpoint.  HIPADJ_CCOST_U1SQ_PLUS_P1 touches cell 0 of the
// first species and parameter 1 only.
struct FieldCostSums { double S, SF; };      // sum over both species of u and of f(u) at a knot
template <int G, int CC>
__device__ __forceinline__ FieldCostSums field_knot_sums(const FKnot<G>& kn, double* __restrict__ red) {
    FieldCostSums r{0.0, 0.0};
    if constexpr (CC == 1) {
        double v[2] = {0.0, 0.0};
#pragma unroll
        for (int q = 0; q < Bruss<G>::Q; ++q) { v[0] += kn.U[q] + kn.V[q]; v[1] += kn.fU[q] + kn.fV[q]; }
        block_sum<Bruss<G>::T, 2>(v, red);
        r.S = v[0]; r.SF = v[1];
    }
    return r;
}
// V += g_u(y) at the thread's cells for the stage state (yU, yV) whose cell sum is Sy; WITH_P: w += wgt * g_p
template <int G, bool WITH_P, int CC>
__device__ __forceinline__ void field_cost_add(const Nbr<G>& nb, double Sy, const double (&yU)[Bruss<G>::Q], double (&vU)[Bruss<G>::Q], double (&vV)[Bruss<G>::Q],
                                               double wgt, double (&w)[3]) {
    if constexpr (CC == 1) {
#pragma unroll
        for (int q = 0; q < Bruss<G>::Q; ++q) { vU[q] += Sy; vV[q] += Sy; }
    } else if constexpr (CC == 2) {
#pragma unroll
        for (int q = 0; q < Bruss<G>::Q; ++q) if (nb.c[q] == 0) { vU[q] += 2.0 * yU[q]; if (WITH_P) w[0] += wgt; }
    }
}

// One reverse RK4 step of lam (and the mu partials) through [t_k, t_{k+1}].  4 LDS exchanges.
// Returns V1 = (df/du)^T lam_hi (i.e. -lam' at the start) in (v1U, v1V) for the Gauss / Quadrature records.
template <int G, bool WITH_P, int CC = 0>
__device__ __forceinline__ void field_rk4_step(double (*sh)[Bruss<G>::NS], const Nbr<G>& nb, const BrussP& P, double dt,
                                               const FKnot<G>& hi, const FKnot<G>& lo, double (&lU)[Bruss<G>::Q], double (&lV)[Bruss<G>::Q],
                                               double (&w)[3], double (&v1U)[Bruss<G>::Q], double (&v1V)[Bruss<G>::Q],
                                               const FieldCostSums& chi, const FieldCostSums& clo) {
    constexpr int Q = Bruss<G>::Q;
    double mU[Q], mV[Q], sU[Q], sV[Q], aU[Q], aV[Q], vU[Q], vV[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        mU[q] = 0.5 * (lo.U[q] + hi.U[q]) + (0.125 * dt) * (lo.fU[q] - hi.fU[q]);
        mV[q] = 0.5 * (lo.V[q] + hi.V[q]) + (0.125 * dt) * (lo.fV[q] - hi.fV[q]);
    }
    const double Smid = 0.5 * (clo.S + chi.S) + (0.125 * dt) * (clo.SF - chi.SF);
    publish<G>(sh[0], nb, lU, lV);
    bruss_vjp<G, WITH_P>(sh[0], nb, P, hi.U, hi.V, lU, lV, v1U, v1V, dt / 6.0, w);
    if constexpr (CC != 0) field_cost_add<G, WITH_P, CC>(nb, chi.S, hi.U, v1U, v1V, dt / 6.0, w);
#pragma unroll
    for (int q = 0; q < Q; ++q) { aU[q] = v1U[q]; aV[q] = v1V[q]; sU[q] = lU[q] + (0.5 * dt) * v1U[q]; sV[q] = lV[q] + (0.5 * dt) * v1V[q]; }
    publish<G>(sh[1], nb, sU, sV);
    bruss_vjp<G, WITH_P>(sh[1], nb, P, mU, mV, sU, sV, vU, vV, dt / 3.0, w);
    if constexpr (CC != 0) field_cost_add<G, WITH_P, CC>(nb, Smid, mU, vU, vV, dt / 3.0, w);
#pragma unroll
    for (int q = 0; q < Q; ++q) { aU[q] += 2.0 * vU[q]; aV[q] += 2.0 * vV[q]; sU[q] = lU[q] + (0.5 * dt) * vU[q]; sV[q] = lV[q] + (0.5 * dt) * vV[q]; }
    publish<G>(sh[0], nb, sU, sV);
    bruss_vjp<G, WITH_P>(sh[0], nb, P, mU, mV, sU, sV, vU, vV, dt / 3.0, w);
    if constexpr (CC != 0) field_cost_add<G, WITH_P, CC>(nb, Smid, mU, vU, vV, dt / 3.0, w);
#pragma unroll
    for (int q = 0; q < Q; ++q) { aU[q] += 2.0 * vU[q]; aV[q] += 2.0 * vV[q]; sU[q] = lU[q] + dt * vU[q]; sV[q] = lV[q] + dt * vV[q]; }
    publish<G>(sh[1], nb, sU, sV);
    bruss_vjp<G, WITH_P>(sh[1], nb, P, lo.U, lo.V, sU, sV, vU, vV, dt / 6.0, w);
    if constexpr (CC != 0) field_cost_add<G, WITH_P, CC>(nb, clo.S, lo.U, vU, vV, dt / 6.0, w);
#pragma unroll
    for (int q = 0; q < Q; ++q) { lU[q] = lU[q] + (dt / 6.0) * (aU[q] + vU[q]); lV[q] = lV[q] + (dt / 6.0) * (aV[q] + vV[q]); }
}

template <int G>
__device__ __forceinline__ void field_finish(const FieldGeom& g, long traj, long Npad, const Nbr<G>& nb, const double (&lU)[Bruss<G>::Q],
                                             const double (&lV)[Bruss<G>::Q], double (&w)[3], double* __restrict__ red,
                                             double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS, T = Bruss<G>::T;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        du0[traj * NS + nb.c[q]] = lU[q]; du0[traj * NS + CELLS + nb.c[q]] = lV[q];
        bad |= !(fabs(lU[q]) <= 1.79769313486231570e308) || !(fabs(lV[q]) <= 1.79769313486231570e308);
    }
    if (dp_traj) {
        block_sum<T, 3>(w, red);
        if (threadIdx.x < 3) dp_traj[(long)threadIdx.x * Npad + traj] = w[threadIdx.x];
    }
    if (bad) atomicOr(flag, 1);
}

// InterpolatingAdjoint (ALG = 0), GaussAdjoint (ALG = 2) and GaussKronrodAdjoint (ALG = 4, round 5) share the sweep; Gauss adds the FSAL exchange and the
// two Gauss-Legendre nodes per step (lam from the adjoint step's Hermite interpolant, y from the forward one), GaussKronrod the adaptive (7,15) rule instead.
template <int G, int ALG, int CC = 0>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_adjoint(FieldGeom g, long Npad, const double* __restrict__ p, const double* __restrict__ knots,
                                                               const double* __restrict__ cot, const int* __restrict__ save_of_knot,
                                                               double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag) {
    constexpr int Q = Bruss<G>::Q, NS = Bruss<G>::NS, T = Bruss<G>::T;
    __shared__ double sh[2][NS];
    __shared__ double red[(T / 64) * 3];
    __shared__ double redk[ALG == 4 ? (T / 64) * 6 : 1];
    const long traj = blockIdx.x;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    double lU[Q], lV[Q], w[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Q; ++q) { lU[q] = 0.0; lV[q] = 0.0; }
    FKnot<G> hi, lo, nx;
    load_fknot<G>(knots, g, traj, g.S, nb, hi);
    { const int s = save_of_knot[g.S]; if (s >= 0) field_jump<G>(g, traj, s, cot, nb, hi.U, hi.V, lU, lV); }
    load_fknot<G>(knots, g, traj, g.S - 1, nb, lo);
    FieldCostSums chi = field_knot_sums<G, CC>(hi, red), clo = chi;
    const double dt = g.dt;
    const double xg = 0.5773502691896257645;
    for (int k = g.S - 1; k >= 0; --k) {
        load_fknot<G>(knots, g, traj, k > 0 ? k - 1 : 0, nb, nx);        // prefetch one knot ahead
        clo = field_knot_sums<G, CC>(lo, red);
        double v1U[Q], v1V[Q];
        if (ALG == 0) {
            field_rk4_step<G, true, CC>(sh, nb, P, dt, hi, lo, lU, lV, w, v1U, v1V, chi, clo);
        } else {
            double hU[Q], hV[Q], wd[3];
#pragma unroll
            for (int q = 0; q < Q; ++q) { hU[q] = lU[q]; hV[q] = lV[q]; }
            field_rk4_step<G, false, CC>(sh, nb, P, dt, hi, lo, lU, lV, wd, v1U, v1V, chi, clo);
            double v5U[Q], v5V[Q];
            publish<G>(sh[0], nb, lU, lV);                                 // fsallast: (df/du)^T lam_new at u_k
            bruss_vjp<G, false>(sh[0], nb, P, lo.U, lo.V, lU, lV, v5U, v5V, 0.0, wd);
            if constexpr (CC != 0) field_cost_add<G, false, CC>(nb, clo.S, lo.U, v5U, v5V, 0.0, wd);
            // the integrand of the step's gradient quadrature at theta (0 at t_hi, 1 at t_lo): (df/dp)^T lam (+ g_p) with lam from the adjoint step's Hermite interpolant and y
            // from the forward one, as per-thread partials added into out[3] with weight wq
            auto node = [&](double th, double wq, double (&out)[3], int buf) {
                const double tf = 1.0 - th;
                double gU[Q], gV[Q], yU[Q], yV[Q], dU_[Q], dV_[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    gU[q] = (1.0 - th) * hU[q] + th * lU[q] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lU[q] - hU[q]) + (th - 1.0) * (-dt) * (-v1U[q]) + th * (-dt) * (-v5U[q]));
                    gV[q] = (1.0 - th) * hV[q] + th * lV[q] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lV[q] - hV[q]) + (th - 1.0) * (-dt) * (-v1V[q]) + th * (-dt) * (-v5V[q]));
                    yU[q] = (1.0 - tf) * lo.U[q] + tf * hi.U[q] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.U[q] - lo.U[q]) + (tf - 1.0) * dt * lo.fU[q] + tf * dt * hi.fU[q]);
                    yV[q] = (1.0 - tf) * lo.V[q] + tf * hi.V[q] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.V[q] - lo.V[q]) + (tf - 1.0) * dt * lo.fV[q] + tf * dt * hi.fV[q]);
                }
                publish<G>(sh[buf], nb, gU, gV);
                bruss_vjp<G, true>(sh[buf], nb, P, yU, yV, gU, gV, dU_, dV_, wq, out);
                if constexpr (CC == 2) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) if (nb.c[q] == 0) out[0] += wq;
                }
            };
            if constexpr (ALG == 4) {
                // GaussKronrodAdjoint (src/gauss_adjoint.jl:820-825; IntegratingGKSumCallback [upstream-recall], the restatement of hipadj_wide.hpp wide_gk_panels / oracle
                // gk_panel): the adaptive (7,15) rule on the step, halved — the half next to t_hi first — while ||Kronrod - Gauss||_2 over the three gradient entries exceeds
                // 1e-7 (depth <= 12), decisions uniform over the workgroup; an accepted panel's Kronrod sum is added by thread 0 (w is summed over the workgroup at the end)
                constexpr int GKD = 12;
                double pa[GKD + 2], pb[GKD + 2]; int pd[GKD + 2]; int sp = 1;
                pa[0] = 0.0; pb[0] = 1.0; pd[0] = 0;
#pragma unroll 1
                while (sp > 0) {
                    --sp;
                    const double a = pa[sp], b = pb[sp]; const int d = pd[sp];
                    const double c = 0.5 * (a + b), h = 0.5 * (b - a);
                    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
                    for (int jn = 0; jn < 15; ++jn) {
                        const int qn = jn < 7 ? jn : (jn == 7 ? 7 : 14 - jn);
                        const double x = jn < 7 ? -c_gk_x[qn] : (jn == 7 ? 0.0 : c_gk_x[qn]);
                        double f[3] = {0.0, 0.0, 0.0};
                        node(c + h * x, 1.0, f, (jn + 1) & 1);      // the first node of a panel publishes into sh[1]: sh[0] may still be read by the fsallast VJP above (no barrier in between), like the 2-node path below
                        const double wk = c_gk_wk[qn], wg = (qn & 1) ? c_gk_wg[qn >> 1] : 0.0;
#pragma unroll
                        for (int e = 0; e < 3; ++e) { acc[e] += wk * f[e]; acc[3 + e] += wg * f[e]; }
                    }
                    __syncthreads();                                     // the last node's stencil reads precede the reduction scratch / the next publication
                    block_sum<T, 6>(acc, redk);
                    const double fq = dt * h;
                    double e2 = 0.0;
#pragma unroll
                    for (int e = 0; e < 3; ++e) { const double dd = (acc[e] - acc[3 + e]) * fq; e2 += dd * dd; }
                    if (sqrt(e2) <= 1e-7 || d >= GKD) {
                        if (threadIdx.x == 0) { w[0] += fq * acc[0]; w[1] += fq * acc[1]; w[2] += fq * acc[2]; }
                    } else {
                        pa[sp] = c; pb[sp] = b; pd[sp] = d + 1; ++sp;      // the half next to t_lo: after the other one
                        pa[sp] = a; pb[sp] = c; pd[sp] = d + 1; ++sp;
                    }
                }
            } else
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                const double x = nq == 0 ? -xg : xg, th = 0.5 * (1.0 + x), tf = 1.0 - th;
                double gU[Q], gV[Q], yU[Q], yV[Q], dU_[Q], dV_[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    // adjoint-step Hermite (h = -dt, derivatives -v1 at the start, -v5 at the end)
                    gU[q] = (1.0 - th) * hU[q] + th * lU[q] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lU[q] - hU[q]) + (th - 1.0) * (-dt) * (-v1U[q]) + th * (-dt) * (-v5U[q]));
                    gV[q] = (1.0 - th) * hV[q] + th * lV[q] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lV[q] - hV[q]) + (th - 1.0) * (-dt) * (-v1V[q]) + th * (-dt) * (-v5V[q]));
                    // forward Hermite at theta_f = 1 - th on [t_k, t_{k+1}]
                    yU[q] = (1.0 - tf) * lo.U[q] + tf * hi.U[q] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.U[q] - lo.U[q]) + (tf - 1.0) * dt * lo.fU[q] + tf * dt * hi.fU[q]);
                    yV[q] = (1.0 - tf) * lo.V[q] + tf * hi.V[q] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.V[q] - lo.V[q]) + (tf - 1.0) * dt * lo.fV[q] + tf * dt * hi.fV[q]);
                }
                publish<G>(sh[1 - nq], nb, gU, gV);
                bruss_vjp<G, true>(sh[1 - nq], nb, P, yU, yV, gU, gV, dU_, dV_, 0.5 * dt, w);
                if constexpr (CC == 2) {      // + g_p in the Gauss integrand (the library's sign: Gauss == Interpolating)
#pragma unroll
                    for (int q = 0; q < Q; ++q) if (nb.c[q] == 0) w[0] += 0.5 * dt;
                }
            }
            __syncthreads();   // the next step republishes sh[0], which node 1 has just been reading
        }
        { const int s = save_of_knot[k]; if (s >= 0 && !(g.no_start && s == 0)) field_jump<G>(g, traj, s, cot, nb, lo.U, lo.V, lU, lV); }
        hi = lo; lo = nx; chi = clo;
    }
    field_finish<G>(g, traj, Npad, nb, lU, lV, w, red, du0, dp_traj, flag);
}

// QuadratureAdjoint pass 1: lambda-only sweep recording (lam_start, lam'_start, lam_end, lam'_end) per step
template <int G, int CC = 0>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_quad_adj(FieldGeom g, const double* __restrict__ p, const double* __restrict__ knots,
                                                                const double* __restrict__ cot, const int* __restrict__ save_of_knot,
                                                                double* __restrict__ adj, double* __restrict__ du0, int* __restrict__ flag) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    __shared__ double sh[2][NS];
    const long traj = blockIdx.x;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    double lU[Q], lV[Q], wd[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Q; ++q) { lU[q] = 0.0; lV[q] = 0.0; }
    __shared__ double redc[CC == 1 ? (Bruss<G>::T / 64) * 2 : 1];
    FKnot<G> hi, lo, nx;
    load_fknot<G>(knots, g, traj, g.S, nb, hi);
    { const int s = save_of_knot[g.S]; if (s >= 0) field_jump<G>(g, traj, s, cot, nb, hi.U, hi.V, lU, lV); }
    load_fknot<G>(knots, g, traj, g.S - 1, nb, lo);
    FieldCostSums chi = field_knot_sums<G, CC>(hi, redc), clo = chi;
    // First-same-as-last over the steps: lam' at the END of step k+1 (record slot 3) is -J(u_{k+1})^T lam there, which is exactly the V1 that
    // step k computes at its start - unless a loss jump changed lam at knot k+1.  So the closing exchange + VJP of a step is only done when a
    // jump follows (uniform: save_of_knot) or at the last step; otherwise the slot is filled one iteration later (4 instead of 5 LDS exchanges).
    bool pending = false;                                   // slot 3 of step k+1 still to be written (uniform)
    for (int k = g.S - 1; k >= 0; --k) {
        load_fknot<G>(knots, g, traj, k > 0 ? k - 1 : 0, nb, nx);
        double* rec = adj + ((traj * g.S + k) * 4) * NS;
        double v1U[Q], v1V[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { rec[nb.c[q]] = lU[q]; rec[CELLS + nb.c[q]] = lV[q]; }
        clo = field_knot_sums<G, CC>(lo, redc);
        field_rk4_step<G, false, CC>(sh, nb, P, g.dt, hi, lo, lU, lV, wd, v1U, v1V, chi, clo);
        if (pending) {
            double* up = rec + 4 * NS;                      // record of step k+1
#pragma unroll
            for (int q = 0; q < Q; ++q) { up[3 * NS + nb.c[q]] = -v1U[q]; up[3 * NS + CELLS + nb.c[q]] = -v1V[q]; }
        }
        const int s = save_of_knot[k];
        const bool jumps = s >= 0 && !(g.no_start && s == 0);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            rec[NS + nb.c[q]] = -v1U[q]; rec[NS + CELLS + nb.c[q]] = -v1V[q];
            rec[2 * NS + nb.c[q]] = lU[q]; rec[2 * NS + CELLS + nb.c[q]] = lV[q];
        }
        if (jumps || k == 0) {
            double v5U[Q], v5V[Q];
            publish<G>(sh[0], nb, lU, lV);
            bruss_vjp<G, false>(sh[0], nb, P, lo.U, lo.V, lU, lV, v5U, v5V, 0.0, wd);
            if constexpr (CC != 0) field_cost_add<G, false, CC>(nb, clo.S, lo.U, v5U, v5V, 0.0, wd);
            __syncthreads();       // the next step republishes sh[0]
#pragma unroll
            for (int q = 0; q < Q; ++q) { rec[3 * NS + nb.c[q]] = -v5U[q]; rec[3 * NS + CELLS + nb.c[q]] = -v5V[q]; }
            pending = false;
        } else pending = true;
        if (jumps) field_jump<G>(g, traj, s, cot, nb, lo.U, lo.V, lU, lV);
        hi = lo; lo = nx; chi = clo;
    }
    bool bad = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        du0[traj * NS + nb.c[q]] = lU[q]; du0[traj * NS + CELLS + nb.c[q]] = lV[q];
        bad |= !(fabs(lU[q]) <= 1.79769313486231570e308) || !(fabs(lV[q]) <= 1.79769313486231570e308);
    }
    if (bad) atomicOr(flag, 1);
}

// QuadratureAdjoint pass 2: workgroup (trajectory, loss interval) runs quadgk(integrand, a, b; atol, rtol) with
// block-uniform decisions: every GK15 panel is evaluated cooperatively (per-thread partial integrands, ONE
// workgroup reduction per panel), the segment list lives in LDS.
template <int G, int MAXSEG, int CC = 0>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_quad_gk(FieldGeom g, long Npad, const double* __restrict__ p, const double* __restrict__ knots,
                                                               const double* __restrict__ adj, const double* __restrict__ qa,
                                                               const double* __restrict__ qb, double atol, double rtol,
                                                               double* __restrict__ qres) {
    constexpr int Q = Bruss<G>::Q, CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS, T = Bruss<G>::T;
    __shared__ double sh[2][NS];
    __shared__ double red[(T / 64) * 6];
    __shared__ double seg_a[MAXSEG], seg_b[MAXSEG], seg_E[MAXSEG], seg_I[MAXSEG][3];
    const long traj = blockIdx.x; const int qi = blockIdx.y;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    int flip = 0;

    // integrand partials at time t (src/quadrature_adjoint.jl:486-502): y = sol(t), lam = adj_sol(t), f_p^T lam
    auto integrand = [&](double t, double (&out)[3]) {
        int k = (int)((t - g.t0) / g.dt);
        if (k < 0) k = 0;
        if (k > g.S - 1) k = g.S - 1;
        if (t < g.t0 + k * g.dt && k > 0) --k;
        if (t > g.t0 + (k + 1) * g.dt && k < g.S - 1) ++k;
        const double thf = (t - (g.t0 + k * g.dt)) / g.dt, tha = 1.0 - thf;
        const double* b0 = knots + ((traj * (g.S + 1) + k) * 2) * NS;
        const double* b1 = b0 + 2 * NS;
        const double* r = adj + ((traj * g.S + k) * 4) * NS;
        double yU[Q], yV[Q], lU[Q], lV[Q], dU_[Q], dV_[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int c = nb.c[q];
            { const double u0_ = b0[c], f0 = b0[NS + c], u1 = b1[c], f1 = b1[NS + c];
              yU[q] = (1.0 - thf) * u0_ + thf * u1 + thf * (thf - 1.0) * ((1.0 - 2.0 * thf) * (u1 - u0_) + (thf - 1.0) * g.dt * f0 + thf * g.dt * f1); }
            { const double u0_ = b0[CELLS + c], f0 = b0[NS + CELLS + c], u1 = b1[CELLS + c], f1 = b1[NS + CELLS + c];
              yV[q] = (1.0 - thf) * u0_ + thf * u1 + thf * (thf - 1.0) * ((1.0 - 2.0 * thf) * (u1 - u0_) + (thf - 1.0) * g.dt * f0 + thf * g.dt * f1); }
            { const double l0 = r[c], d0 = r[NS + c], l1 = r[2 * NS + c], d1 = r[3 * NS + c];
              lU[q] = (1.0 - tha) * l0 + tha * l1 + tha * (tha - 1.0) * ((1.0 - 2.0 * tha) * (l1 - l0) + (tha - 1.0) * (-g.dt) * d0 + tha * (-g.dt) * d1); }
            { const double l0 = r[CELLS + c], d0 = r[NS + CELLS + c], l1 = r[2 * NS + CELLS + c], d1 = r[3 * NS + CELLS + c];
              lV[q] = (1.0 - tha) * l0 + tha * l1 + tha * (tha - 1.0) * ((1.0 - 2.0 * tha) * (l1 - l0) + (tha - 1.0) * (-g.dt) * d0 + tha * (-g.dt) * d1); }
        }
        publish<G>(sh[flip], nb, lU, lV);
        out[0] = out[1] = out[2] = 0.0;
        bruss_vjp<G, true>(sh[flip], nb, P, yU, yV, lU, lV, dU_, dV_, 1.0, out);
        if constexpr (CC == 2) {       // + g_p of the continuous cost in the integrand (src/quadrature_adjoint.jl:486-502 with dgdp_continuous): one thread carries it into the block sum
#pragma unroll
            for (int q = 0; q < Q; ++q) if (nb.c[q] == 0) out[0] += 1.0;
        }
        flip ^= 1;
    };
    // one GK15 panel: returns workgroup-summed I (Kronrod) and E = |I_K - I_G|_2, identical in every thread.
    // The node loop is deliberately NOT unrolled and there is a single call site (state machine below): the
    // fully unrolled version needed 8.8 KB of scratch per lane.
    auto panel = [&](double a, double b, double (&I)[3]) -> double {
        const double c = 0.5 * (a + b), h = 0.5 * (b - a);
        double acc[6] = {0, 0, 0, 0, 0, 0}, f1[3], f2[3];
#pragma unroll 1
        for (int j = 0; j < 7; ++j) {
            const double xj = c_gk_x[j], wk = c_gk_wk[j], wg = (j & 1) ? c_gk_wg[j >> 1] : 0.0;
            integrand(c - h * xj, f1); integrand(c + h * xj, f2);
#pragma unroll
            for (int e = 0; e < 3; ++e) { const double s = f1[e] + f2[e]; acc[e] += wk * s; acc[3 + e] += wg * s; }
        }
        integrand(c, f1);
#pragma unroll
        for (int e = 0; e < 3; ++e) { acc[e] += c_gk_wk[7] * f1[e]; acc[3 + e] += c_gk_wg[3] * f1[e]; }
        block_sum<T, 6>(acc, red);
        double e2 = 0.0;
#pragma unroll
        for (int e = 0; e < 3; ++e) { I[e] = acc[e] * h; const double d = I[e] - acc[3 + e] * h; e2 += d * d; }
        return sqrt(e2);
    };

    // quadgk: bisect the worst segment until E <= max(atol, rtol |I|)   (state machine: one panel() call site)
    double Itot[3] = {0.0, 0.0, 0.0}, E = 0.0;
    double I1[3] = {0.0, 0.0, 0.0}, E1 = 0.0;
    int ns = 0, mode = 0, wi = 0;
    double pa = qa[qi], pb = qb[qi], wa = pa, wb = pb, mid = 0.0;
    for (;;) {
        double In[3];
        const double En = panel(pa, pb, In);
        if (mode == 1) {             // first half done: remember it, evaluate the second half
            I1[0] = In[0]; I1[1] = In[1]; I1[2] = In[2]; E1 = En;
            pa = mid; pb = wb; mode = 2;
            continue;
        }
        __syncthreads();
        if (mode == 0) {
            if (threadIdx.x == 0) { seg_a[0] = pa; seg_b[0] = pb; seg_E[0] = En; seg_I[0][0] = In[0]; seg_I[0][1] = In[1]; seg_I[0][2] = In[2]; }
            Itot[0] = In[0]; Itot[1] = In[1]; Itot[2] = In[2]; E = En; ns = 1;
        } else {                     // mode 2: replace segment wi by its two halves
            const double oE = seg_E[wi], o0 = seg_I[wi][0], o1 = seg_I[wi][1], o2 = seg_I[wi][2];
            Itot[0] += I1[0] + In[0] - o0; Itot[1] += I1[1] + In[1] - o1; Itot[2] += I1[2] + In[2] - o2;
            E += E1 + En - oE;
            __syncthreads();
            if (threadIdx.x == 0) {
                seg_a[wi] = wa; seg_b[wi] = mid; seg_E[wi] = E1; seg_I[wi][0] = I1[0]; seg_I[wi][1] = I1[1]; seg_I[wi][2] = I1[2];
                seg_a[ns] = mid; seg_b[ns] = wb; seg_E[ns] = En; seg_I[ns][0] = In[0]; seg_I[ns][1] = In[1]; seg_I[ns][2] = In[2];
            }
            ++ns;
        }
        __syncthreads();
        const double nrm = sqrt(Itot[0] * Itot[0] + Itot[1] * Itot[1] + Itot[2] * Itot[2]);
        const double tol = atol > rtol * nrm ? atol : rtol * nrm;
        if (E <= tol || ns + 1 > MAXSEG) break;
        wi = 0;
        for (int s2 = 1; s2 < ns; ++s2) if (seg_E[s2] > seg_E[wi]) wi = s2;
        wa = seg_a[wi]; wb = seg_b[wi]; mid = 0.5 * (wa + wb);
        if (!(mid > (wa < wb ? wa : wb) && mid < (wa < wb ? wb : wa))) break;
        pa = wa; pb = mid; mode = 1;
    }
    __syncthreads();
    if (threadIdx.x < 3) { double s = 0.0; for (int q = 0; q < ns; ++q) s += seg_I[q][threadIdx.x]; qres[((long)qi * 3 + threadIdx.x) * Npad + traj] = s; }
}

}  // namespace hipadj
