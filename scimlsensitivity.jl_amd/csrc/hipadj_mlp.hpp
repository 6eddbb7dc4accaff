// hipadj_mlp.hpp — FP64-MFMA kernel family for the neural-ODE case (BASELINE config 4: 3-layer tanh MLP, 128 hidden,
// 4096-column batch, GaussAdjoint): the model, the forward solve, and what the reverse sweeps (hipadj_mlp_grad.hpp) share with it.
// Model (oracle/adjoint_oracle.c ORC_MODEL_MLP; docs/src/Benchmark.md:62 shape):
//     X is d x B (column-major, d = 2),  f(X) = W3 tanh(W2 tanh(W1 X + b1) + b2) + b3   applied column-wise,
//     p = [W1 (H x d), b1, W2 (H x H), b2, W3 (d x H), b3], all column-major.
//
// The batch columns are independent given the weights, so ONE WORKGROUP integrates 16 columns through the whole solve; its waves split the H
// output rows of every layer, so both the MFMA work and the FP64 tanh work use all four SIMDs.  The H x H contraction runs on the matrix cores:
//     v_mfma_f64_16x16x4_f64:  A[i = l&15][k = l>>4] (one f64 per lane), B[k = l>>4][j = l&15],
//     C/D: col = l & 15, row = (l >> 4) + 4 * reg   (the f64 map, cdna_hip_programming.md §3)
// Forward solve: the batch column on j = l & 15, register r of output tile t holds row 16t + 4r + (l>>4): for fixed r the four lane groups
// hold FOUR CONSECUTIVE rows, i.e. exactly the B operand layout of a K-step of the next layer: a wave writes its rows of the activation to an
// LDS tile act[row][16] (64 consecutive doubles per store: conflict-free), one s_barrier, and every wave reads its B operands as
// act[4 st + (l>>4)][l&15] (again 64 consecutive doubles).  The A operands — the wave's own 16 x 4 blocks of W2 — stay in registers for the
// whole solve (64 VGPRs for H = 128).  tanh is fused on the accumulator; the d-sized contractions (W1, W3) are VALU work plus a two-step
// cross-lane-group reduction.  Two barriers per pass (activation exchange, d-sized reduction).
//
// The reverse sweeps put the batch columns on the M side instead (hipadj_mlp_grad.hpp): their accumulator layout is then the operand layout of
// the weight-gradient outer products, and the parameter gradient accumulates in registers.  (Round 1 wrote weighted activation records — 5.6 GB
// for config 4 — and contracted them with split-K GEMM kernels afterwards; that path was retired in round 2: profiles/r2_mlpbench_variants.log.)
#pragma once

#include <hip/hip_runtime.h>
#include "hipadj_lane.hpp"

namespace hipadj {

typedef double mlp_d4 __attribute__((ext_vector_type(4)));

struct MlpGeom {
    long N;            // trajectories (each with its own d x B state)
    int B, S, M;
    double t0, dt, loss_shift;
    int loss_kind, no_start, p_shared;   // loss_kind: hipadj_loss 0 cotangent, 1 lsq_shift, 2 lsq_data (dgdu = lsq_w (u - data), the block in the cotangents' place)
    double lsq_w;
    int NQ;            // unused (round-1 record count), kept for the launch sites' aggregate initialisation
};

template <int H> struct Mlp {
    static constexpr int D = 2, TT = H / 16;       // row tiles
#ifndef HIPADJ_MLP_MAXW
#define HIPADJ_MLP_MAXW 8     // forward solve: 2 waves per SIMD, one wave's tanh (VALU) overlaps the other's MFMAs
#endif
    static constexpr int NW = TT >= HIPADJ_MLP_MAXW ? HIPADJ_MLP_MAXW : TT;    // waves per workgroup (row split)
    static constexpr int TW = TT / NW;             // row tiles per wave
    static constexpr int NT = 64 * NW;             // threads per workgroup
    static constexpr int NPAR = H * D + H + H * H + H + D * H + D;
    static_assert(H % 16 == 0 && TT % NW == 0, "hidden width must be a multiple of 16 (and of 64 beyond 48)");
};

template <int H> struct MlpW { const double *W1, *b1, *W2, *b2, *W3, *b3; };
template <int H> __device__ __forceinline__ MlpW<H> mlp_weights(const double* __restrict__ p, int p_shared, long traj) {
    constexpr int D = Mlp<H>::D;
    const double* pp = p_shared ? p : p + traj * Mlp<H>::NPAR;
    MlpW<H> w; w.W1 = pp; w.b1 = w.W1 + H * D; w.W2 = w.b1 + H; w.b2 = w.W2 + H * H; w.W3 = w.b2 + H; w.b3 = w.W3 + D * H;
    return w;
}

// LDS of one forward-solve workgroup: the exchanged activation tile and the cross-wave reduction scratch
template <int H> struct MlpLds { double act[H * 16]; double red[Mlp<H>::NW][16][2]; };

// A operands held in REGISTERS for the whole solve: a wave only ever needs the A fragments of its own TW row tiles, H/4 doubles per lane
// and tile (H = 128, eight waves: 32 doubles = 64 VGPRs).  The MFMAs read A from the register file, B (the activation tile) from LDS.
template <int H> struct MlpWReg { double a[Mlp<H>::TW][H / 4]; };
template <int H>
__device__ __forceinline__ void mlp_load_frag(const double* __restrict__ Wm, MlpWReg<H>& r) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned li = threadIdx.x & 15u, lq = (threadIdx.x & 63u) >> 4;
    const unsigned t0 = (threadIdx.x >> 6) * (unsigned)TW;
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int st = 0; st < NK; ++st) r.a[t][st] = Wm[(16u * (t0 + (unsigned)t) + li) + (4u * (unsigned)st + lq) * (unsigned)H];
}
template <int H>
__device__ __forceinline__ void mlp_gemm_reg(const MlpWReg<H>& r, const double* __restrict__ act, mlp_d4 (&acc)[Mlp<H>::TW]) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned act_off = (lane >> 4) * 16u + (lane & 15u);
#pragma unroll
    for (int st = 0; st < NK; ++st) {
        const double b = act[act_off + (unsigned)(64 * st)];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a[t][st], b, acc[t], 0, 0, 0);
    }
}

// XOR swizzle of the LDS copy of W2 in the reverse sweeps (hipadj_mlp_grad.hpp): element (r, c) lives at r * H + (c ^ mlp_swz(r)).  The forward
// contraction reads 16 x 2 footprints per half wave (rows of a fragment on 16 distinct even bank pairs, the K index picks the odd one), the
// transposed one 2 x 16 footprints (the two rows differ in bit 4, the 16 columns fill the low four bits): ONE copy serves both without conflicts.
__device__ __forceinline__ unsigned mlp_swz(unsigned r) { return (2u * (r & 15u)) ^ (16u * (r & 1u)); }

// tanh for the activations: (1 - t) / (1 + t) with t = exp(-2|x|) in (0, 1].  The cancellation in 1 - t for small |x| is
// an ABSOLUTE error of one ulp of 1 (1e-16) in a quantity that only enters sums W h — harmless against the 1e-6 gate.
#ifndef HIPADJ_MLP_FAST_TANH
#define HIPADJ_MLP_FAST_TANH 1     // 0: library exp + IEEE division (round 1)
#endif
__device__ __forceinline__ double mlp_tanh(double x) {
#ifdef HIPADJ_MLP_DBG_NOTANH      // scripts/mlpbench.hip: what the sweep costs without the transcendental work (wrong numbers, timing only)
    return x * 0.5;
#endif
#if HIPADJ_MLP_FAST_TANH
    // t = 2^k e^r with k = rint(a log2 e), r = a - k ln 2 (two-part constant), e^r by the degree-12 Taylor polynomial (|r| <= 0.347: remainder
    // 1.7e-16); the quotient by v_rcp_f64 + two Newton steps + one residual correction (1 + t lies in [1, 2]: no scaling, no special cases).
    // 31 instructions instead of the ~48 of the library exp + IEEE division; max |difference| to libm tanh 2.2e-16 (scripts/kbench_tanh).
    const double a = fmax(-2.0 * fabs(x), -80.0);
    const double kf = __builtin_rint(a * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, a);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = __builtin_fma(p, r, 1.0 / 39916800.0); p = __builtin_fma(p, r, 1.0 / 3628800.0); p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0); p = __builtin_fma(p, r, 1.0 / 5040.0); p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0); p = __builtin_fma(p, r, 1.0 / 24.0); p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5); p = __builtin_fma(p, r, 1.0); p = __builtin_fma(p, r, 1.0);
    const double t = __builtin_amdgcn_ldexp(p, (int)kf);
    const double d = 1.0 + t, n = 1.0 - t;
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
    double q = n * y;
    q = __builtin_fma(__builtin_fma(-d, q, n), y, q);
    return __builtin_copysign(q, x);
#else
    const double t = exp(-2.0 * fabs(x));
    const double r = (1.0 - t) / (1.0 + t);
    return x < 0.0 ? -r : r;
#endif
}

__device__ __forceinline__ double group_sum4(double v) {   // sum over the four 16-lane groups (same column j)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// cross-wave sum of a per-wave (o0, o1) pair for column l&15; every lane of every wave gets the total
template <int H>
__device__ __forceinline__ void mlp_reduce2(MlpLds<H>& L, double o0, double o1, double (&out)[2]) {
    constexpr int NW = Mlp<H>::NW;
    const int wv = threadIdx.x >> 6, li = threadIdx.x & 15;
    o0 = group_sum4(o0); o1 = group_sum4(o1);
    if (((threadIdx.x & 63) >> 4) == 0) { L.red[wv][li][0] = o0; L.red[wv][li][1] = o1; }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { s0 += L.red[w][li][0]; s1 += L.red[w][li][1]; }
    out[0] = s0; out[1] = s1;
}

// f(x) for the workgroup's 16 columns: x[D] per lane (column l&15, replicated over lane groups and waves)
template <int H>
__device__ __forceinline__ void mlp_forward(const MlpW<H>& w, MlpLds<H>& L, const MlpWReg<H>& w2reg, const double (&x)[2], double (&out)[2]) {
    constexpr int TW = Mlp<H>::TW;
    const unsigned lq = (threadIdx.x & 63u) >> 4, li = threadIdx.x & 15u;
    const int t0 = (threadIdx.x >> 6) * TW;
    mlp_d4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            acc[t][r] = w.b2[row];
            L.act[row * 16u + li] = mlp_tanh(w.b1[row] + w.W1[row] * x[0] + w.W1[row + (unsigned)H] * x[1]);
        }
    }
    __syncthreads();
    mlp_gemm_reg<H>(w2reg, L.act, acc);
    double o0 = 0.0, o1 = 0.0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            const double h2 = mlp_tanh(acc[t][r]);
            o0 += w.W3[row * 2u] * h2;
            o1 += w.W3[row * 2u + 1u] * h2;
        }
    }
    mlp_reduce2<H>(L, o0, o1, out);      // barrier inside: also orders the act reads above before the next pass's writes
    out[0] += w.b3[0]; out[1] += w.b3[1];
}

// forward RK4; knots [traj][S+1][2][D][B]  (x_k then f(x_k)); out [traj][M][D*B] in the caller's layout
template <int H>
__global__ void __launch_bounds__(Mlp<H>::NT) k_mlp_forward(MlpGeom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                            double* __restrict__ knots, double* __restrict__ out, const int* __restrict__ save_of_knot) {
    constexpr int D = 2;
    __shared__ MlpLds<H> L;
    const long traj = blockIdx.y;
    const int col = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool writer = (threadIdx.x >> 4) == 0;          // wave 0, lane group 0
    const MlpW<H> w = mlp_weights<H>(p, g.p_shared, traj);
    const long nB = (long)D * g.B;
    MlpWReg<H> WR;
    mlp_load_frag<H>(w.W2, WR);                           // this wave's A fragments of W2, in registers for the whole solve
    double x[D], k1[D], k2[D], k3[D], k4[D], xs[D];
    x[0] = u0[traj * nB + (long)col * D]; x[1] = u0[traj * nB + (long)col * D + 1];
    const double dt = g.dt;
    for (int k = 0; k <= g.S; ++k) {
        mlp_forward<H>(w, L, WR, x, k1);
        if (writer) {
            double* kn = knots + ((traj * (g.S + 1) + k) * 2) * nB;
            kn[col] = x[0]; kn[g.B + col] = x[1]; kn[nB + col] = k1[0]; kn[nB + g.B + col] = k1[1];
            const int s = save_of_knot[k];
            if (out && s >= 0) { double* o = out + (traj * g.M + s) * nB; o[(long)col * D] = x[0]; o[(long)col * D + 1] = x[1]; }
        }
        if (k == g.S) break;
        xs[0] = x[0] + 0.5 * dt * k1[0]; xs[1] = x[1] + 0.5 * dt * k1[1];
        mlp_forward<H>(w, L, WR, xs, k2);
        xs[0] = x[0] + 0.5 * dt * k2[0]; xs[1] = x[1] + 0.5 * dt * k2[1];
        mlp_forward<H>(w, L, WR, xs, k3);
        xs[0] = x[0] + dt * k3[0]; xs[1] = x[1] + dt * k3[1];
        mlp_forward<H>(w, L, WR, xs, k4);
        x[0] = x[0] + (dt / 6.0) * (k1[0] + 2.0 * (k2[0] + k3[0]) + k4[0]);
        x[1] = x[1] + (dt / 6.0) * (k1[1] + 2.0 * (k2[1] + k3[1]) + k4[1]);
    }
}

}  // namespace hipadj
