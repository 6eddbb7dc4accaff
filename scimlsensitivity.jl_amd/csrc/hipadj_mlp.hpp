// hipadj_mlp.hpp — FP64-MFMA kernel family for the neural-ODE case (BASELINE config 4: 3-layer tanh MLP, 128 hidden,
// 4096-column batch, GaussAdjoint).  Model (oracle/adjoint_oracle.c ORC_MODEL_MLP; docs/src/Benchmark.md:62 shape):
//     X is d x B (column-major, d = 2),  f(X) = W3 tanh(W2 tanh(W1 X + b1) + b2) + b3   applied column-wise,
//     p = [W1 (H x d), b1, W2 (H x H), b2, W3 (d x H), b3], all column-major.
//
// The batch columns are independent given the weights, so ONE WORKGROUP integrates 16 columns through the whole
// reverse sweep; its NW = min(4, H/16) waves (one per SIMD of the CU) split the H output rows of every layer, so both
// the MFMA work and the FP64 tanh work (which dominates: ~150 VALU instructions each) use all four SIMDs.
// The two H x H contractions per VJP (W2 H1 forward, W2^T G2 backward) run on the matrix cores:
//     v_mfma_f64_16x16x4_f64:  A[i = l&15][k = l>>4] (one f64 per lane), B[k = l>>4][j = l&15],
//     C/D: col = l & 15, row = (l >> 4) + 4 * reg   (the f64 map, cdna_hip_programming.md §3)
// With the batch column on j = l & 15, register r of output tile t holds row 16t + 4r + (l>>4): for fixed r the four
// lane groups hold FOUR CONSECUTIVE rows, i.e. exactly the B operand layout of a K-step of the next layer: a wave
// writes its rows of the activation to an LDS tile act[row][16] (64 consecutive doubles per store: conflict-free),
// one s_barrier, and every wave reads its B operands as act[4 st + (l>>4)][l&15] (again 64 consecutive doubles).
// tanh is fused on the accumulator.  Two barriers per forward / backward pass (activation exchange, d-sized reduction).
// A operands (16 x 4 blocks of W2 / W2^T) are read straight from L2 (128 KB, shared by every wave); the d-sized
// contractions (W1, W3) are VALU work plus a two-step cross-lane-group reduction (__shfl_xor 16, 32).
//
// Parameter gradient: (df/dp)^T lam = sum over columns of outer products of activations (G2 H1^T, ...).  A lane
// cannot carry the 17 282 accumulators; instead the sweep writes the weighted activation records
//     X_q, w Lam_q (16-row padded), H1_q (+ ones row), H2_q (+ ones row), w G1_q, w G2_q     q = quadrature points
// (Gauss: the two Gauss-Legendre nodes of every step, w = dt/2, src/gauss_adjoint.jl:745-759, 809-851;
//  Interpolating: the four RK4 stages, w = dt/6, dt/3, dt/3, dt/6, src/interpolating_adjoint.jl:166-172)
// and k_mlp_wgrad contracts them over (q, column) with MFMA as three split-K NT-GEMMs:
//     G2 x [H1;1]^T -> dW2, db2      G1 x [X;1]^T -> dW1, db1      [Lam] x [H2;1]^T -> dW3, db3
// followed by a fixed-order reduction of the split-K partials (bit-reproducible).
#pragma once

#include <hip/hip_runtime.h>
#include "hipadj_lane.hpp"

namespace hipadj {

typedef double mlp_d4 __attribute__((ext_vector_type(4)));
#ifndef HIPADJ_MLP_KUNROLL
#define HIPADJ_MLP_KUNROLL 8    // K-steps unrolled per loop trip of the LDS-fed contractions: a full unroll (32) lets the scheduler hoist every ds_read, the sweep then
                                // needs more than its 256 registers and spills (68-276 B of scratch per lane, 6.45 ms); 8 at a time: 237-245 registers, no scratch, 5.3 ms
#endif
#define HIPADJ_MLP_PRAGMA_(x) _Pragma(#x)
#define HIPADJ_MLP_UNROLL_K HIPADJ_MLP_PRAGMA_(unroll HIPADJ_MLP_KUNROLL)

struct MlpGeom {
    long N;            // trajectories (each with its own d x B state)
    int B, S, M;
    double t0, dt, loss_shift;
    int loss_kind, no_start, p_shared;
    int NQ;            // quadrature records per step (Gauss 2, Interpolating 4)
};

template <int H> struct Mlp {
    static constexpr int D = 2, TT = H / 16;       // row tiles
#ifndef HIPADJ_MLP_MAXW
#define HIPADJ_MLP_MAXW 8     // 2 waves per SIMD: one wave's tanh (VALU) overlaps the other's MFMAs (measured 14.1 -> 11.8 ms)
#endif
    static constexpr int NW = TT >= HIPADJ_MLP_MAXW ? HIPADJ_MLP_MAXW : TT;    // waves per workgroup (row split)
    static constexpr int TW = TT / NW;             // row tiles per wave
    static constexpr int NT = 64 * NW;             // threads per workgroup
    static constexpr int NPAR = H * D + H + H * H + H + D * H + D;
    static constexpr int HP = H + 16;              // H rows + a 16-row tile whose first row is the ones row
    static_assert(H % 16 == 0 && TT % NW == 0, "hidden width must be a multiple of 16 (and of 64 beyond 48)");
};

template <int H> struct MlpW { const double *W1, *b1, *W2, *b2, *W3, *b3, *W2T; };
template <int H> __device__ __forceinline__ MlpW<H> mlp_weights(const double* __restrict__ p, const double* __restrict__ w2t, int p_shared, long traj) {
    constexpr int D = Mlp<H>::D;
    const double* pp = p_shared ? p : p + traj * Mlp<H>::NPAR;
    MlpW<H> w; w.W1 = pp; w.b1 = w.W1 + H * D; w.W2 = w.b1 + H; w.b2 = w.W2 + H * H; w.W3 = w.b2 + H; w.b3 = w.W3 + D * H;
    w.W2T = p_shared ? w2t : w2t + traj * (long)H * H;
    return w;
}

// LDS of one workgroup: the exchanged activation tile and the cross-wave reduction scratch
// wfrag (adjoint kernel only): W2^T in MFMA A-fragment order — fragment (row tile t, K-step st) is 64 consecutive doubles
// (lane l holds W2^T[16 t + (l&15)][4 st + (l>>4)]), so a wave's A operand is one conflict-free ds_read_b64.
template <int H> struct MlpLds { double act[H * 16]; double red[Mlp<H>::NW][16][2]; };
template <int H> struct MlpLdsW { double wfrag[H * H]; };

// acc[t] (+)= rows (16 (t0 + t) .. +15) of  Wm (H x H, column-major) . act   with act read from the LDS tile.
// The TW A operands of a K-step (16 x 4 blocks of Wm, L2-resident) are fetched ONE K-step ahead; a scheduling fence per
// K-step keeps hipcc from hoisting all loads to the top, and the lane offset is made opaque per call so that the address
// arithmetic is not hoisted out of the time loop as hundreds of live 64-bit VGPR pairs (both spilled KBs per lane).
template <int H>
__device__ __forceinline__ void mlp_gemm(const double* __restrict__ Wm, const double* __restrict__ act, int t0, mlp_d4 (&acc)[Mlp<H>::TW]) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned li = threadIdx.x & 15u, lq = (threadIdx.x & 63u) >> 4;
    unsigned lane_off = li + lq * (unsigned)H + 16u * (unsigned)t0;
    asm volatile("" : "+v"(lane_off));
    const unsigned act_off = lq * 16u + li;
    double a_cur[TW], a_nxt[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) a_cur[t] = Wm[lane_off + (unsigned)(16 * t)];
#pragma unroll
    for (int st = 0; st < NK; ++st) {
        if (st + 1 < NK) {
#pragma unroll
            for (int t = 0; t < TW; ++t) a_nxt[t] = Wm[lane_off + (unsigned)(16 * t + 4 * (st + 1) * H)];
        }
        const double b = act[act_off + (unsigned)(64 * st)];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TW; ++t) a_cur[t] = a_nxt[t];
    }
}

// A operands from the LDS fragment copy of the matrix (see MlpLdsW): no L2 traffic in the time loop.  With 16 batch
// columns per workgroup a weight fetched from L2 is used by only 16 columns: 512 B per MFMA per wave = 32 B/clk per CU,
// 19.7 TB/s chip-wide at the FP64-MFMA peak — beyond what the L2 delivers; LDS serves the same 32 B/clk at a quarter of
// its bandwidth.
template <int H>
__device__ __forceinline__ void mlp_gemm_lds(const double* __restrict__ wfrag, const double* __restrict__ act, int t0, mlp_d4 (&acc)[Mlp<H>::TW]) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned act_off = (lane >> 4) * 16u + (lane & 15u);
    HIPADJ_MLP_UNROLL_K
    for (int st = 0; st < NK; ++st) {
        const double b = act[act_off + (unsigned)(64 * st)];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfrag[((unsigned)(t0 + t) * NK + st) * 64u + lane], b, acc[t], 0, 0, 0);
    }
}

// A operands held in REGISTERS for the whole sweep (HIPADJ_MLP_REGW): a wave only ever needs the A fragments of its own TW row tiles,
// H/4 doubles per lane and tile (H = 128, eight waves: 32 doubles = 64 VGPRs for W2, as many for W2^T).  With two waves per SIMD each
// wave owns 256 registers, so both fit next to the activations; the MFMAs then read A from the register file, B (the activation tile) from
// LDS — half the LDS traffic of the fragment copy (which ran the LDS port at its limit, 128 B/clk, at full matrix rate) and no L2 traffic
// at all in the forward passes of the adjoint kernel (where LDS had no room for a second 128 KB copy).
#ifndef HIPADJ_MLP_REGW
#define HIPADJ_MLP_REGW 0      // 0: W2 from L2 / W2^T from LDS (round 1); 1: W2 in registers; 2: both in registers (measured slower: the sweep already needs its 256 registers, the fragments spill)
#endif
#ifndef HIPADJ_MLP_FSAL
#define HIPADJ_MLP_FSAL 1      // reuse the activations at x_lo as those at x_hi of the next step, and V5 as V1 when no loss jump intervenes
#endif
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
// ONE LDS copy of W2 for BOTH contractions of the adjoint kernel (HIPADJ_MLP_SWZ).  The forward pass needs A fragments of W2
// (lane (i = l&15, q = l>>4), tile t, K-step s: element (16t + i, 4s + q)), the backward pass A fragments of W2^T (element (4s + q, 16t + i) of
// W2): a 16 x 2 and a 2 x 16 footprint per half wave.  No row pitch serves both without bank conflicts, an XOR swizzle does: element (r, c) lives at
//     r * H + (c ^ sw(r)),   sw(r) = 2 (r & 15) ^ 16 (r & 1)
// forward: the 16 rows of a fragment land on 16 distinct even bank pairs, q picks the odd one; backward: the two rows of a half wave differ in
// bit 4, the 16 columns fill the low four bits.  128 KB for H = 128: the forward passes of the sweep no longer fetch their A operands from L2
// (19.7 TB/s chip-wide at the matrix peak - more than the L2 delivers), and no register is spent on weights.
#ifndef HIPADJ_MLP_SWZ
#define HIPADJ_MLP_SWZ 1
#endif
__device__ __forceinline__ unsigned mlp_swz(unsigned r) { return (2u * (r & 15u)) ^ (16u * (r & 1u)); }
template <int H>
__device__ __forceinline__ void mlp_fill_swz(const double* __restrict__ W2, double* __restrict__ w2s) {
    for (int e = threadIdx.x; e < H * H; e += Mlp<H>::NT) {
        const unsigned r = (unsigned)e % (unsigned)H, c = (unsigned)e / (unsigned)H;      // W2 is column-major: coalesced reads
        w2s[r * (unsigned)H + (c ^ mlp_swz(r))] = W2[e];
    }
}
// acc[t] += rows of W2 . act   (A fragment (16 (t0+t) + i, 4 st + q))
template <int H>
__device__ __forceinline__ void mlp_gemm_swz_n(const double* __restrict__ w2s, const double* __restrict__ act, int t0, mlp_d4 (&acc)[Mlp<H>::TW]) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned lane = threadIdx.x & 63u, i = lane & 15u, q = lane >> 4;
    const unsigned act_off = q * 16u + i;
    const unsigned sw = mlp_swz(i), swh = sw & ~3u;
    unsigned base[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) base[t] = (16u * (unsigned)(t0 + t) + i) * (unsigned)H + (q ^ (sw & 3u));
    HIPADJ_MLP_UNROLL_K
    for (int st = 0; st < NK; ++st) {
        const double b = act[act_off + (unsigned)(64 * st)];
        const unsigned cx = (unsigned)(4 * st) ^ swh;
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2s[base[t] + cx], b, acc[t], 0, 0, 0);
    }
}
// acc[t] += rows of W2^T . act   (A fragment = element (4 st + q, 16 (t0+t) + i) of W2)
template <int H>
__device__ __forceinline__ void mlp_gemm_swz_t(const double* __restrict__ w2s, const double* __restrict__ act, int t0, mlp_d4 (&acc)[Mlp<H>::TW]) {
    constexpr int TW = Mlp<H>::TW, NK = H / 4;
    const unsigned lane = threadIdx.x & 63u, i = lane & 15u, q = lane >> 4;
    const unsigned act_off = q * 16u + i;
    unsigned col[TW][4];                                     // (16 t + i) ^ sw(4 st + q) for the four values of st & 3
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) col[t][m] = q * (unsigned)H + ((16u * (unsigned)(t0 + t) + i) ^ (2u * q) ^ (16u * (q & 1u)) ^ (8u * (unsigned)m));
    HIPADJ_MLP_UNROLL_K
    for (int st = 0; st < NK; ++st) {
        const double b = act[act_off + (unsigned)(64 * st)];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w2s[(unsigned)(4 * st * H) + col[t][st & 3]], b, acc[t], 0, 0, 0);
    }
}
// SRC: 0 = A operands from global memory / L2 (mem), 1 = from the LDS fragment copy (mem), 2 = from registers (reg),
// 3 / 4 = from the swizzled LDS copy of W2, plain / transposed (mem)
template <int H, int SRC>
__device__ __forceinline__ void mlp_gemm_any(const double* __restrict__ mem, const MlpWReg<H>& reg, const double* __restrict__ act, int t0, mlp_d4 (&acc)[Mlp<H>::TW]) {
    if constexpr (SRC == 2) mlp_gemm_reg<H>(reg, act, acc);
    else if constexpr (SRC == 3) mlp_gemm_swz_n<H>(mem, act, t0, acc);
    else if constexpr (SRC == 4) mlp_gemm_swz_t<H>(mem, act, t0, acc);
    else if constexpr (SRC == 1) mlp_gemm_lds<H>(mem, act, t0, acc);
    else mlp_gemm<H>(mem, act, t0, acc);
}

// tanh for the activations: (1 - t) / (1 + t) with t = exp(-2|x|) in (0, 1].  The cancellation in 1 - t for small |x| is
// an ABSOLUTE error of one ulp of 1 (1e-16) in a quantity that only enters sums W h — harmless against the 1e-6 gate — and
// the formula costs one exp and one division instead of the general-purpose library tanh (the forward passes of this
// kernel are bound by these VALU instructions, not by the MFMAs).
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

// activation records for the weight-gradient GEMMs; q indexes (traj, step, point); all arrays [q][rows][B]
template <int H> struct MlpRec { double *AX, *AL, *AH1, *AH2, *AG1, *AG2; };
// Where a pass writes its record rows AS SOON AS they exist (HIPADJ_MLP_EARLY_REC): h1 before the contraction, h2 after it, g2 before the
// transposed contraction, g1 after it - four 16 KB bursts per workgroup spread over the pass instead of one 64 KB burst at its end.  All 256
// workgroups run in lockstep, so the end-of-pass form hit HBM with 16.7 MB at once and every CU waited for its share of the write bandwidth
// (about 8 us per record, a third of the GaussAdjoint step); the spread stores drain under the MFMAs of the same pass.
template <int H> struct MlpSink { MlpRec<H> R; long q; long B; int col; double wq; };
#ifndef HIPADJ_MLP_EARLY_REC
#define HIPADJ_MLP_EARLY_REC 1
#endif

// forward pass for the workgroup's 16 columns: x[D] per lane (column l&15, replicated over lane groups and waves);
// h1/h2 hold THIS WAVE's rows (tiles t0 .. t0+TW-1) in the MFMA accumulator layout
template <int H, int SRC, bool REC = false>
__device__ __forceinline__ void mlp_forward(const MlpW<H>& w, MlpLds<H>& L, const double* __restrict__ w2mem, const MlpWReg<H>& w2reg, const double (&x)[2], double (&h1)[Mlp<H>::TW][4], double (&h2)[Mlp<H>::TW][4], double (&out)[2],
                                            const MlpSink<H>* sk = nullptr) {
    constexpr int TW = Mlp<H>::TW, HP = Mlp<H>::HP;
    if (REC && (threadIdx.x >> 4) == 0) {                 // wave 0, lane group 0: the d-sized rows and the ones rows
        double* ax = sk->R.AX + sk->q * 16 * sk->B;
        ax[sk->col] = x[0]; ax[sk->B + sk->col] = x[1]; ax[2 * sk->B + sk->col] = 1.0;
        sk->R.AH1[(sk->q * HP + H) * sk->B + sk->col] = 1.0; sk->R.AH2[(sk->q * HP + H) * sk->B + sk->col] = 1.0;
    }
    const unsigned lq = (threadIdx.x & 63u) >> 4, li = threadIdx.x & 15u;
    const int t0 = (threadIdx.x >> 6) * TW;
    mlp_d4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            h1[t][r] = mlp_tanh(w.b1[row] + w.W1[row] * x[0] + w.W1[row + (unsigned)H] * x[1]);
            acc[t][r] = w.b2[row];
            L.act[row * 16u + li] = h1[t][r];
            if (REC) sk->R.AH1[(sk->q * HP + (long)row) * sk->B + sk->col] = h1[t][r];
        }
    }
    __syncthreads();
    mlp_gemm_any<H, SRC>(w2mem, w2reg, L.act, t0, acc);
    double o0 = 0.0, o1 = 0.0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            h2[t][r] = mlp_tanh(acc[t][r]);
            if (REC) sk->R.AH2[(sk->q * HP + (long)row) * sk->B + sk->col] = h2[t][r];
            o0 += w.W3[row * 2u] * h2[t][r];
            o1 += w.W3[row * 2u + 1u] * h2[t][r];
        }
    }
    mlp_reduce2<H>(L, o0, o1, out);      // barrier inside: also orders the act reads above before the next pass's writes
    out[0] += w.b3[0]; out[1] += w.b3[1];
}

// (df/du)^T lam for the workgroup's columns, given the activations of the forward pass; g1/g2 are this wave's rows of
// the layer cotangents
template <int H, int SRC, bool REC = false>
__device__ __forceinline__ void mlp_backward(const MlpW<H>& w, MlpLds<H>& L, const double* __restrict__ wfrag, const MlpWReg<H>& wtreg, const double (&lam)[2], const double (&h1)[Mlp<H>::TW][4], const double (&h2)[Mlp<H>::TW][4],
                                             double (&g1)[Mlp<H>::TW][4], double (&g2)[Mlp<H>::TW][4], double (&dlam)[2], const MlpSink<H>* sk = nullptr) {
    constexpr int TW = Mlp<H>::TW;
    if (REC && (threadIdx.x >> 4) == 0) {
        double* al = sk->R.AL + sk->q * 16 * sk->B;
        al[sk->col] = sk->wq * lam[0]; al[sk->B + sk->col] = sk->wq * lam[1];
    }
    const unsigned lq = (threadIdx.x & 63u) >> 4, li = threadIdx.x & 15u;
    const int t0 = (threadIdx.x >> 6) * TW;
    mlp_d4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            g2[t][r] = (w.W3[row * 2u] * lam[0] + w.W3[row * 2u + 1u] * lam[1]) * (1.0 - h2[t][r] * h2[t][r]);
            acc[t][r] = 0.0;
            L.act[row * 16u + li] = g2[t][r];
            if (REC) sk->R.AG2[(sk->q * H + (long)row) * sk->B + sk->col] = sk->wq * g2[t][r];
        }
    }
    __syncthreads();
    mlp_gemm_any<H, SRC>(wfrag, wtreg, L.act, t0, acc);
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = (unsigned)(16 * (t0 + t) + 4 * r) + lq;
            g1[t][r] = acc[t][r] * (1.0 - h1[t][r] * h1[t][r]);
            if (REC) sk->R.AG1[(sk->q * H + (long)row) * sk->B + sk->col] = sk->wq * g1[t][r];
            d0 += w.W1[row] * g1[t][r];
            d1 += w.W1[row + (unsigned)H] * g1[t][r];
        }
    }
    mlp_reduce2<H>(L, d0, d1, dlam);
}

// W2T[i + k*H] = W2[k + i*H]
static __global__ void k_mlp_transpose_w2(int H, int npar, int hd, const double* __restrict__ p, double* __restrict__ w2t) {
    const long traj = blockIdx.y;
    const double* W2 = p + traj * npar + hd;   // hd = H*D + H
    double* o = w2t + traj * (long)H * H;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < H * H; e += gridDim.x * blockDim.x) {
        const int i = e % H, k = e / H;
        o[i + (long)k * H] = W2[k + (long)i * H];
    }
}

// forward RK4; knots [traj][S+1][2][D][B]  (x_k then f(x_k)); out [traj][M][D*B] in the caller's layout
template <int H>
__global__ void __launch_bounds__(Mlp<H>::NT) k_mlp_forward(MlpGeom g, const double* __restrict__ u0, const double* __restrict__ p, const double* __restrict__ w2t,
                                                            double* __restrict__ knots, double* __restrict__ out, const int* __restrict__ save_of_knot) {
    constexpr int TW = Mlp<H>::TW, D = 2;
    __shared__ MlpLds<H> L;
    const long traj = blockIdx.y;
    const int col = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool writer = (threadIdx.x >> 4) == 0;          // wave 0, lane group 0
    const MlpW<H> w = mlp_weights<H>(p, w2t, g.p_shared, traj);
    const long nB = (long)D * g.B;
    MlpWReg<H> WR;
#ifndef HIPADJ_MLP_FWD_REGW
#define HIPADJ_MLP_FWD_REGW 1   // the forward solve keeps its W2 fragments in registers (226 VGPRs, no spill): 3 % faster than the LDS fragment copy, 128 KB of LDS free
#endif
#if HIPADJ_MLP_FWD_REGW >= 1
    constexpr int FS = 2;
    const double* wmem = nullptr;
    mlp_load_frag<H>(w.W2, WR);                           // this wave's A fragments of W2, in registers for the whole solve
#else
    constexpr int FS = 1;
    __shared__ MlpLdsW<H> LW;        // W2 in A-fragment order (the forward solve only needs W2)
    for (int e = threadIdx.x; e < H * H; e += Mlp<H>::NT) {
        const int l = e & 63, f = e >> 6, st = f % (H / 4), t = f / (H / 4);
        LW.wfrag[e] = w.W2[(16 * t + (l & 15)) + (4 * st + (l >> 4)) * H];
    }
    __syncthreads();
    const double* wmem = LW.wfrag;
#endif
    double x[D], k1[D], k2[D], k3[D], k4[D], xs[D], h1[TW][4], h2[TW][4];
    x[0] = u0[traj * nB + (long)col * D]; x[1] = u0[traj * nB + (long)col * D + 1];
    const double dt = g.dt;
    for (int k = 0; k <= g.S; ++k) {
        mlp_forward<H, FS>(w, L, wmem, WR, x, h1, h2, k1);
        if (writer) {
            double* kn = knots + ((traj * (g.S + 1) + k) * 2) * nB;
            kn[col] = x[0]; kn[g.B + col] = x[1]; kn[nB + col] = k1[0]; kn[nB + g.B + col] = k1[1];
            const int s = save_of_knot[k];
            if (out && s >= 0) { double* o = out + (traj * g.M + s) * nB; o[(long)col * D] = x[0]; o[(long)col * D + 1] = x[1]; }
        }
        if (k == g.S) break;
        xs[0] = x[0] + 0.5 * dt * k1[0]; xs[1] = x[1] + 0.5 * dt * k1[1];
        mlp_forward<H, FS>(w, L, wmem, WR, xs, h1, h2, k2);
        xs[0] = x[0] + 0.5 * dt * k2[0]; xs[1] = x[1] + 0.5 * dt * k2[1];
        mlp_forward<H, FS>(w, L, wmem, WR, xs, h1, h2, k3);
        xs[0] = x[0] + dt * k3[0]; xs[1] = x[1] + dt * k3[1];
        mlp_forward<H, FS>(w, L, wmem, WR, xs, h1, h2, k4);
        x[0] = x[0] + (dt / 6.0) * (k1[0] + 2.0 * (k2[0] + k3[0]) + k4[0]);
        x[1] = x[1] + (dt / 6.0) * (k1[1] + 2.0 * (k2[1] + k3[1]) + k4[1]);
    }
}


template <int H>
__device__ __forceinline__ void mlp_record(const MlpRec<H>& R, const MlpGeom& g, long q, int col, double wq, const double (&x)[2], const double (&lam)[2],
                                           const double (&h1)[Mlp<H>::TW][4], const double (&h2)[Mlp<H>::TW][4], const double (&g1)[Mlp<H>::TW][4], const double (&g2)[Mlp<H>::TW][4]) {
    constexpr int TW = Mlp<H>::TW, HP = Mlp<H>::HP;
#ifdef HIPADJ_MLP_DBG_NOREC       // scripts/mlpbench.hip: what the sweep costs without the record stores (timing only)
    return;
#endif
    const int lq = (threadIdx.x & 63) >> 4, t0 = (threadIdx.x >> 6) * TW;
    const long B = g.B;
    if ((threadIdx.x >> 4) == 0) {       // wave 0, lane group 0: the d-sized rows and the ones rows
        double* ax = R.AX + q * 16 * B; double* al = R.AL + q * 16 * B;
        ax[col] = x[0]; ax[B + col] = x[1]; ax[2 * B + col] = 1.0;
        al[col] = wq * lam[0]; al[B + col] = wq * lam[1];
        R.AH1[(q * HP + H) * B + col] = 1.0; R.AH2[(q * HP + H) * B + col] = 1.0;
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = 16 * (t0 + t) + 4 * r + lq;
            R.AH1[(q * HP + row) * B + col] = h1[t][r];
            R.AH2[(q * HP + row) * B + col] = h2[t][r];
            R.AG1[(q * H + row) * B + col] = wq * g1[t][r];
            R.AG2[(q * H + row) * B + col] = wq * g2[t][r];
        }
    }
}

// reverse sweep: ALG 0 = InterpolatingAdjoint (records at the 4 RK4 stages), ALG 2 = GaussAdjoint (records at the
// two Gauss-Legendre nodes; lam from the adjoint step's Hermite interpolant, y from the forward one)
template <int H, int ALG>
__global__ void __launch_bounds__(Mlp<H>::NT) k_mlp_adjoint(MlpGeom g, const double* __restrict__ p, const double* __restrict__ w2t, const double* __restrict__ knots,
                                                    const double* __restrict__ cot, const int* __restrict__ save_of_knot, MlpRec<H> R,
                                                    double* __restrict__ du0, int* __restrict__ flag) {
    constexpr int TW = Mlp<H>::TW, D = 2;
    __shared__ MlpLds<H> L;
    const long traj = blockIdx.y;
    const int col = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool writer = (threadIdx.x >> 4) == 0;          // wave 0, lane group 0
    const MlpW<H> w = mlp_weights<H>(p, w2t, g.p_shared, traj);
    const long nB = (long)D * g.B;
    const double dt = g.dt;
    MlpWReg<H> WF, WB;                                    // A fragments of W2 (forward passes) and W2^T (backward passes) when held in registers
#if HIPADJ_MLP_SWZ
    constexpr int FS = 3, BS = 4;
    __shared__ MlpLdsW<H> LW;
    mlp_fill_swz<H>(w.W2, LW.wfrag);                      // one swizzled copy of W2 serves both contractions
    __syncthreads();
    const double* fmem = LW.wfrag; const double* bmem = LW.wfrag;
#else
#if HIPADJ_MLP_REGW >= 1
    constexpr int FS = 2;
    const double* fmem = nullptr;
    mlp_load_frag<H>(w.W2, WF);
#else
    constexpr int FS = 0;
    const double* fmem = w.W2;
#endif
#if HIPADJ_MLP_REGW >= 2
    constexpr int BS = 2;
    const double* bmem = nullptr;
    mlp_load_frag<H>(w.W2T, WB);
#else
    constexpr int BS = 1;
    __shared__ MlpLdsW<H> LW;
    for (int e = threadIdx.x; e < H * H; e += Mlp<H>::NT) {   // W2^T -> A-fragment order, once per workgroup
        const int l = e & 63, f = e >> 6, st = f % (H / 4), t = f / (H / 4);
        LW.wfrag[e] = w.W2T[(16 * t + (l & 15)) + (4 * st + (l >> 4)) * H];
    }
    __syncthreads();
    const double* bmem = LW.wfrag;
#endif
#endif
    auto knot = [&](int k, double (&xx)[2], double (&ff)[2]) {
        const double* kn = knots + ((traj * (g.S + 1) + k) * 2) * nB;
        xx[0] = kn[col]; xx[1] = kn[g.B + col]; ff[0] = kn[nB + col]; ff[1] = kn[nB + g.B + col];
    };
    auto jump = [&](int s, const double (&xx)[2], double (&lam)[2]) {
        if (g.loss_kind == 0) { const double* c = cot + (traj * g.M + s) * nB; lam[0] += c[(long)col * D]; lam[1] += c[(long)col * D + 1]; }
        else { lam[0] += xx[0] - g.loss_shift; lam[1] += xx[1] - g.loss_shift; }
    };
    double lam[D] = {0.0, 0.0}, xh[D], fh[D], xl[D], fl[D];
    double h1[TW][4], h2[TW][4], g1[TW][4], g2[TW][4], out[D];
    knot(g.S, xh, fh);
    { const int s = save_of_knot[g.S]; if (s >= 0) jump(s, xh, lam); }
    const double xg = 0.5773502691896257645;
#if HIPADJ_MLP_FSAL
    // first-same-as-last over the steps: the activations at x_lo of a step are those at x_hi of the next one (same inputs, same code:
    // bit-identical), and GaussAdjoint's closing evaluation V5 = J(x_lo)^T lam is the next step's V1 unless a loss jump changed lam.
    double h1e[TW][4], h2e[TW][4], Vn[D] = {0.0, 0.0};
    bool have_v = false;                                   // uniform over the workgroup (save_of_knot is)
    mlp_forward<H, FS>(w, L, fmem, WF, xh, h1e, h2e, out);
#endif
    for (int k = g.S - 1; k >= 0; --k) {
        knot(k, xl, fl);
        const long qbase = (traj * g.S + k) * g.NQ;
        double xm[D], ls[D], V1[D], V2[D], V3[D], V4[D], lam_hi[D] = {lam[0], lam[1]};
        xm[0] = 0.5 * (xl[0] + xh[0]) + (0.125 * dt) * (fl[0] - fh[0]);
        xm[1] = 0.5 * (xl[1] + xh[1]) + (0.125 * dt) * (fl[1] - fh[1]);
        // stage 1 at x_hi
#if HIPADJ_MLP_FSAL
        if (ALG == 2 && have_v) { V1[0] = Vn[0]; V1[1] = Vn[1]; }
        else {
            mlp_backward<H, BS>(w, L, bmem, WB, lam, h1e, h2e, g1, g2, V1);
            if (ALG == 0) mlp_record<H>(R, g, qbase + 0, col, dt / 6.0, xh, lam, h1e, h2e, g1, g2);
        }
#else
        mlp_forward<H, FS>(w, L, fmem, WF, xh, h1, h2, out);
        mlp_backward<H, BS>(w, L, bmem, WB, lam, h1, h2, g1, g2, V1);
        if (ALG == 0) mlp_record<H>(R, g, qbase + 0, col, dt / 6.0, xh, lam, h1, h2, g1, g2);
#endif
        // stages 2, 3 at the Hermite midpoint (same activations)
        ls[0] = lam[0] + 0.5 * dt * V1[0]; ls[1] = lam[1] + 0.5 * dt * V1[1];
        mlp_forward<H, FS>(w, L, fmem, WF, xm, h1, h2, out);
        mlp_backward<H, BS>(w, L, bmem, WB, ls, h1, h2, g1, g2, V2);
        if (ALG == 0) mlp_record<H>(R, g, qbase + 1, col, dt / 3.0, xm, ls, h1, h2, g1, g2);
        ls[0] = lam[0] + 0.5 * dt * V2[0]; ls[1] = lam[1] + 0.5 * dt * V2[1];
        mlp_backward<H, BS>(w, L, bmem, WB, ls, h1, h2, g1, g2, V3);
        if (ALG == 0) mlp_record<H>(R, g, qbase + 2, col, dt / 3.0, xm, ls, h1, h2, g1, g2);
        // stage 4 at x_lo
        ls[0] = lam[0] + dt * V3[0]; ls[1] = lam[1] + dt * V3[1];
#if HIPADJ_MLP_FSAL
        mlp_forward<H, FS>(w, L, fmem, WF, xl, h1e, h2e, out);
        mlp_backward<H, BS>(w, L, bmem, WB, ls, h1e, h2e, g1, g2, V4);
        if (ALG == 0) mlp_record<H>(R, g, qbase + 3, col, dt / 6.0, xl, ls, h1e, h2e, g1, g2);
#else
        mlp_forward<H, FS>(w, L, fmem, WF, xl, h1, h2, out);
        mlp_backward<H, BS>(w, L, bmem, WB, ls, h1, h2, g1, g2, V4);
        if (ALG == 0) mlp_record<H>(R, g, qbase + 3, col, dt / 6.0, xl, ls, h1, h2, g1, g2);
#endif
        lam[0] = lam[0] + (dt / 6.0) * (V1[0] + 2.0 * (V2[0] + V3[0]) + V4[0]);
        lam[1] = lam[1] + (dt / 6.0) * (V1[1] + 2.0 * (V2[1] + V3[1]) + V4[1]);
        if (ALG == 2) {
            double V5[D];
#if HIPADJ_MLP_FSAL
            mlp_backward<H, BS>(w, L, bmem, WB, lam, h1e, h2e, g1, g2, V5);             // fsallast at x_lo (activations of stage 4)
            Vn[0] = V5[0]; Vn[1] = V5[1];
#else
            mlp_backward<H, BS>(w, L, bmem, WB, lam, h1, h2, g1, g2, V5);               // fsallast at x_lo (activations of stage 4)
#endif
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                const double x = nq == 0 ? -xg : xg, th = 0.5 * (1.0 + x), tf = 1.0 - th;
                double lg[D], yg[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    lg[j] = (1.0 - th) * lam_hi[j] + th * lam[j] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lam[j] - lam_hi[j]) + (th - 1.0) * (-dt) * (-V1[j]) + th * (-dt) * (-V5[j]));
                    yg[j] = (1.0 - tf) * xl[j] + tf * xh[j] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (xh[j] - xl[j]) + (tf - 1.0) * dt * fl[j] + tf * dt * fh[j]);
                }
                double dl[D];
#if HIPADJ_MLP_EARLY_REC
                const MlpSink<H> sk{R, qbase + nq, (long)g.B, col, 0.5 * dt};
                mlp_forward<H, FS, true>(w, L, fmem, WF, yg, h1, h2, out, &sk);
                mlp_backward<H, BS, true>(w, L, bmem, WB, lg, h1, h2, g1, g2, dl, &sk);
#else
                mlp_forward<H, FS>(w, L, fmem, WF, yg, h1, h2, out);
                mlp_backward<H, BS>(w, L, bmem, WB, lg, h1, h2, g1, g2, dl);
                mlp_record<H>(R, g, qbase + nq, col, 0.5 * dt, yg, lg, h1, h2, g1, g2);
#endif
            }
        }
        {
            const int s = save_of_knot[k];
            const bool jumped = s >= 0 && !(g.no_start && s == 0);
            if (jumped) jump(s, xl, lam);
#if HIPADJ_MLP_FSAL
            have_v = !jumped;
#endif
        }
        xh[0] = xl[0]; xh[1] = xl[1]; fh[0] = fl[0]; fh[1] = fl[1];
    }
    if (writer) {
        du0[traj * nB + (long)col * D] = lam[0]; du0[traj * nB + (long)col * D + 1] = lam[1];
        if (!(fabs(lam[0]) <= 1.79769313486231570e308) || !(fabs(lam[1]) <= 1.79769313486231570e308)) atomicOr(flag, 1);
    }
}

// split-K NT-GEMM on the records:  Cpart[grp][ks][ra][rb] = sum_{s in K-slice ks} A[ra][s] Bm[rb][s]
// A: [Q][RA][B], Bm: [Q][RB][B] (RA, RB multiples of 16); samples s = (q, column).
// One workgroup = RA/16 waves (one per 16-row tile of A) shares the B operand: every iteration the workgroup stages a
// [RB][64-sample] slab of Bm in LDS with coalesced 16-byte loads (each Bm byte leaves HBM once instead of once per A
// tile — that re-read was 8x the traffic and the whole cost of the first version), each wave reads its own A rows
// from global memory and its B fragments from LDS with ds_read_b128 (row pitch 66 doubles: the 16 rows of a fragment
// land on 16 distinct 16-byte slots).  Within a 16-sample chunk lane (i, g) takes samples 4g..4g+3 for the four
// K-steps kk = 0..3: K-step kk contracts samples {4g + kk}, the same assignment for A and B.
constexpr int WG_SAMPLES = 64, WG_PITCH = WG_SAMPLES + 2;
template <int NTB>
__global__ void __launch_bounds__(512) k_mlp_wgrad(const double* __restrict__ A, const double* __restrict__ Bm, int RA, int RB, long Qper, int B,
                                                   int ksplit, double* __restrict__ Cpart) {
    extern __shared__ __attribute__((aligned(16))) double bt[];       // [RB][WG_PITCH]
    const int ks = blockIdx.y; const long grp = blockIdx.z;
    const int ti = threadIdx.x >> 6, li = threadIdx.x & 15, lq = (threadIdx.x & 63) >> 4;
    const int nthreads = blockDim.x;
    mlp_d4 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) acc[t] = mlp_d4{0.0, 0.0, 0.0, 0.0};
    const long slabs_per_q = B / WG_SAMPLES, nslabs = Qper * slabs_per_q;
    const long c0 = nslabs * ks / ksplit, c1 = nslabs * (ks + 1) / ksplit;
    for (long c = c0; c < c1; ++c) {
        const long q = grp * Qper + c / slabs_per_q; const int s_base = (int)(c % slabs_per_q) * WG_SAMPLES;
        // stage Bm[q][0..RB)[s_base .. s_base+64) : RB rows x 32 dbl2
        for (int e = threadIdx.x; e < RB * (WG_SAMPLES / 2); e += nthreads) {
            const int row = e / (WG_SAMPLES / 2), pr = e % (WG_SAMPLES / 2);
            const dbl2 v = *reinterpret_cast<const dbl2*>(Bm + (q * RB + row) * (long)B + s_base + 2 * pr);
            *reinterpret_cast<dbl2*>(bt + row * WG_PITCH + 2 * pr) = v;
        }
        // this wave's A fragments of the slab (4 chunks x 4 samples per lane) are requested BEFORE the barrier, so their global
        // latency overlaps the LDS staging of the B slab instead of sitting in front of every chunk's MFMAs
        dbl2 afr[WG_SAMPLES / 16][2];
#pragma unroll
        for (int c4 = 0; c4 < WG_SAMPLES / 16; ++c4) {
            const double* ap = A + (q * RA + 16 * ti + li) * (long)B + s_base + 16 * c4 + 4 * lq;
            afr[c4][0] = *reinterpret_cast<const dbl2*>(ap); afr[c4][1] = *reinterpret_cast<const dbl2*>(ap + 2);
        }
        __syncthreads();
#pragma unroll
        for (int c4 = 0; c4 < WG_SAMPLES / 16; ++c4) {
            const int s0 = 16 * c4 + 4 * lq;
            const dbl2 a01 = afr[c4][0], a23 = afr[c4][1];
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const double* bp = bt + (16 * t + li) * WG_PITCH + s0;
                const dbl2 b01 = *reinterpret_cast<const dbl2*>(bp), b23 = *reinterpret_cast<const dbl2*>(bp + 2);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b01.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b01.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b23.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b23.y, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    double* cp = Cpart + ((grp * ksplit + ks) * (long)RA) * RB;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cp[(long)(16 * ti + lq + 4 * r) * RB + 16 * t + li] = acc[t][r];
    }
}

// fallback for batches that are not a multiple of 64 columns: one wave per (A row tile, K slice), 16-sample chunks
template <int NTB>
__global__ void __launch_bounds__(64) k_mlp_wgrad_small(const double* __restrict__ A, const double* __restrict__ Bm, int RA, int RB, long Qper, int B,
                                                  int ksplit, double* __restrict__ Cpart) {
    const int ti = blockIdx.x, ks = blockIdx.y; const long grp = blockIdx.z;
    const int li = threadIdx.x & 15, lq = (threadIdx.x & 63) >> 4;
    mlp_d4 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) acc[t] = mlp_d4{0.0, 0.0, 0.0, 0.0};
    const long chunks_per_q = B / 16, nchunks = Qper * chunks_per_q;
    const long c0 = nchunks * ks / ksplit, c1 = nchunks * (ks + 1) / ksplit;
    for (long c = c0; c < c1; ++c) {
        const long q = grp * Qper + c / chunks_per_q; const int s0 = (int)(c % chunks_per_q) * 16 + 4 * lq;
        const double* ap = A + (q * RA + 16 * ti + li) * (long)B + s0;
        const double a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            if (16 * t < RB) {
                const double* bp = Bm + (q * RB + 16 * t + li) * (long)B + s0;
                const double b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc[t], 0, 0, 0);
            }
        }
    }
    double* cp = Cpart + ((grp * ksplit + ks) * (long)RA) * RB;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        if (16 * t < RB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(long)(16 * ti + lq + 4 * r) * RB + 16 * t + li] = acc[t][r];
        }
    }
}

// dp = fixed-order sum of the split-K partials, scattered into the parameter layout
//   C1 = G2 x [H1;1]^T (H x HP): dW2[i + m H], db2[i] = C1[i][H];  C2 = G1 x [X;1]^T (H x 16): dW1[i + d H], db1[i] = C2[i][2]
//   C3 = Lam x [H2;1]^T (16 x HP): dW3[d + m D], db3[d] = C3[d][H]
template <int H>
__global__ void __launch_bounds__(256) k_mlp_wreduce(int ksplit, const double* __restrict__ C1, const double* __restrict__ C2, const double* __restrict__ C3,
                                                     double* __restrict__ dp) {
    constexpr int D = 2, HP = Mlp<H>::HP, NPAR = Mlp<H>::NPAR;
    const long grp = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NPAR) return;
    const double* src; long idx, stride;
    int o = e;
    if (o < H * D) { const int i = o % H, d = o / H; src = C2 + grp * ksplit * (long)H * 16; idx = (long)i * 16 + d; stride = (long)H * 16; }
    else if ((o -= H * D) < H) { src = C2 + grp * ksplit * (long)H * 16; idx = (long)o * 16 + D; stride = (long)H * 16; }
    else if ((o -= H) < H * H) { const int i = o % H, m = o / H; src = C1 + grp * ksplit * (long)H * HP; idx = (long)i * HP + m; stride = (long)H * HP; }
    else if ((o -= H * H) < H) { src = C1 + grp * ksplit * (long)H * HP; idx = (long)o * HP + H; stride = (long)H * HP; }
    else if ((o -= H) < D * H) { const int d = o % D, m = o / D; src = C3 + grp * ksplit * (long)16 * HP; idx = (long)d * HP + m; stride = (long)16 * HP; }
    else { o -= D * H; src = C3 + grp * ksplit * (long)16 * HP; idx = (long)o * HP + H; stride = (long)16 * HP; }
    double s = 0.0;
    for (int k = 0; k < ksplit; ++k) s += src[idx + k * stride];
    dp[grp * NPAR + e] = s;
}

}  // namespace hipadj
