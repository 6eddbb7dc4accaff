// hipadj_kernels.hpp — __global__ wrappers (gfx950) around the lane bodies of hipadj_lane.hpp, plus the small
// utility kernels (layout transposes, segment composition, deterministic dp reduction, non-finite scan).
//
// Launch geometry: 64-thread workgroups = exactly one wavefront each, so that a 10^4-trajectory ensemble
// (157 waves) spreads over as many CUs / XCDs / L2 slices as possible; blockIdx.x indexes the wave of
// trajectories, blockIdx.y the time segment.  Consecutive blockIdx.x land on different XCDs (b % 8) which
// round-robins the HBM streams of neighbouring trajectory tiles over all 8 L2s.
#pragma once

#include <hip/hip_runtime.h>
#include "hipadj_lane.hpp"

namespace hipadj {

constexpr int WAVE = 64;

struct SegPlan {
    int nseg;              // C
    const int* bounds;     // device [C+1] knot indices, bounds[0] = 0, bounds[C] = S
};

template <class Mo>
__global__ void __launch_bounds__(WAVE) k_forward(Geom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                  dbl2* __restrict__ knots, double* __restrict__ ckpt,
                                                  const int* __restrict__ ckpt_of_knot, double* __restrict__ outT,
                                                  const int* __restrict__ save_of_knot, double* __restrict__ yT) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    forward_lane<Mo>(g, i, u0, p, knots, ckpt, ckpt_of_knot, outT, save_of_knot, yT);
}

// segbuf layout: [segment][column][N+NP][Npad]; the top segment only fills column 0.
template <class Mo, int PF>
__global__ void __launch_bounds__(WAVE) k_interp(Geom g, SegPlan sp, const double* __restrict__ p,
                                                 const dbl2* __restrict__ knots, const double* __restrict__ cotT,
                                                 const int* __restrict__ save_of_knot, double* __restrict__ segbuf) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, PF>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, PF>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * R + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * g.Npad] = mu[c][j];
        }
    }
}

// compose the segment maps top -> bottom:  lam <- A lam + c_l ; mu <- mu + B lam + c_m
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_compose(Geom g, int nseg, const double* __restrict__ segbuf,
                                                  double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N], mu[NP];
    { const double* __restrict__ src = segbuf + (long)(nseg - 1) * NC * R * g.Npad + i;
#pragma unroll
      for (int j = 0; j < N; ++j) lam[j] = src[(long)j * g.Npad];
#pragma unroll
      for (int j = 0; j < NP; ++j) mu[j] = src[(long)(N + j) * g.Npad]; }
    for (int s = nseg - 2; s >= 0; --s) {
        const double* __restrict__ src = segbuf + (long)s * NC * R * g.Npad + i;
        double nl[N], nm[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) nl[j] = src[(long)j * g.Npad];
#pragma unroll
        for (int j = 0; j < NP; ++j) nm[j] = mu[j] + src[(long)(N + j) * g.Npad];
#pragma unroll
        for (int c = 0; c < N; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) nl[j] += src[((long)(c + 1) * R + j) * g.Npad] * lam[c];
#pragma unroll
            for (int j = 0; j < NP; ++j) nm[j] += src[((long)(c + 1) * R + N + j) * g.Npad] * lam[c];
        }
#pragma unroll
        for (int j = 0; j < N; ++j) lam[j] = nl[j];
#pragma unroll
        for (int j = 0; j < NP; ++j) mu[j] = nm[j];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[j];
}

template <class Mo>
__global__ void __launch_bounds__(WAVE) k_backsolve(Geom g, const double* __restrict__ p, const double* __restrict__ yT,
                                                    const double* __restrict__ ckpt, const int* __restrict__ ckpt_of_knot,
                                                    const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                    double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N], mu[NP];
    backsolve_lane<Mo>(g, i, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[j];
}

template <class Mo, int PF>
__global__ void __launch_bounds__(WAVE) k_gauss(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N], mu[NP];
    gauss_lane<Mo, PF>(g, i, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[j];
}

template <class Mo, int PF>
__global__ void __launch_bounds__(WAVE) k_quad_adj(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                   const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                   dbl2* __restrict__ adj, double* __restrict__ du0) {
    constexpr int N = Mo::N;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N];
    quad_adj_lane<Mo, PF>(g, i, p, knots, cotT, save_of_knot, adj, lam);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
}

// one lane per (trajectory, quadrature interval); qres [interval][NP][Npad]
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_quad_gk(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                  const dbl2* __restrict__ adj, const double* __restrict__ qa,
                                                  const double* __restrict__ qb, double atol, double rtol,
                                                  double* __restrict__ qres) {
    constexpr int NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int q = blockIdx.y;
    if (i >= g.N) return;
    double res[NP];
    quad_gk_lane<Mo, 32>(g, i, p, knots, adj, qa[q], qb[q], atol, rtol, res);
#pragma unroll
    for (int j = 0; j < NP; ++j) qres[((long)q * NP + j) * g.Npad + i] = res[j];
}
// res .+= quadgk(...) in the reference's order (src/quadrature_adjoint.jl:563-616)
__global__ void __launch_bounds__(WAVE) k_quad_sum(long N, long Npad, int np, int nq, const double* __restrict__ qres,
                                                   double* __restrict__ dp_traj) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= N) return;
    for (int j = 0; j < np; ++j) {
        double s = 0.0;
        for (int q = 0; q < nq; ++q) s += qres[((long)q * np + j) * Npad + i];
        dp_traj[(long)j * Npad + i] = s;
    }
}

// ---- utilities -------------------------------------------------------------------------------------------
// AoS [N][C] (caller layout) -> SoA [C][Npad]; 32x32 LDS tile, +1 padding against bank conflicts
__global__ void k_aos_to_soa(const double* __restrict__ src, double* __restrict__ dst, long N, long Npad, int C) {
    __shared__ double tile[32][33];
    const long i0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const long i = i0 + r; const int c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < N && c < C) ? src[i * C + c] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r; const long i = i0 + threadIdx.x;
        if (c < C && i < Npad) dst[(long)c * Npad + i] = tile[threadIdx.x][r];
    }
}
__global__ void k_soa_to_aos(const double* __restrict__ src, double* __restrict__ dst, long N, long Npad, int C) {
    __shared__ double tile[32][33];
    const long i0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r; const long i = i0 + threadIdx.x;
        tile[r][threadIdx.x] = (c < C && i < N) ? src[(long)c * Npad + i] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const long i = i0 + r; const int c = c0 + threadIdx.x;
        if (i < N && c < C) dst[i * C + c] = tile[threadIdx.x][r];
    }
}

// dp[j] = sum_i dp_traj[j][i] — one workgroup per parameter, fixed-order tree => bit-reproducible for a given N.
// Also scans du0 / dp_traj for NaN/Inf (the reference's retcode check) into *flag.
__global__ void __launch_bounds__(256) k_reduce_dp(long N, long Npad, const double* __restrict__ dp_traj,
                                                   double* __restrict__ dp, int* __restrict__ flag) {
    __shared__ double sh[256];
    const int j = blockIdx.x;
    double s = 0.0; int bad = 0;
    for (long i = threadIdx.x; i < N; i += 256) { const double v = dp_traj[(long)j * Npad + i]; s += v; bad |= !(fabs(v) <= 1.79769313486231570e308); }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0 && dp) dp[j] = sh[0];
    if (bad) atomicOr(flag, 1);
}
// dp per trajectory: SoA [np][Npad] -> caller [N][np] handled by k_soa_to_aos.
__global__ void __launch_bounds__(256) k_scan_finite(long count, const double* __restrict__ v, int* __restrict__ flag) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count && !(fabs(v[i]) <= 1.79769313486231570e308)) atomicOr(flag, 2);
}

}  // namespace hipadj
