// hipadj_kernels.hpp — __global__ wrappers (gfx950) around the lane bodies of hipadj_lane.hpp, plus the small
// utility kernels (layout transposes, segment composition, deterministic dp reduction, non-finite scan).
//
// Launch geometry: 64-thread workgroups = exactly one wavefront each, so that a 10^4-trajectory ensemble
// (157 waves) spreads over as many CUs / XCDs / L2 slices as possible; blockIdx.x indexes the wave of
// trajectories, blockIdx.y the time segment.  Consecutive blockIdx.x land on different XCDs (b % 8) which
// round-robins the HBM streams of neighbouring trajectory tiles over all 8 L2s.
#pragma once

#if !defined(__HIPCC_RTC__)   // hiprtc (runtime-compiled user models, hipadj_user.hpp) provides the runtime header implicitly
#include <hip/hip_runtime.h>
#endif
#include "hipadj_lane.hpp"
#include "hipadj_quad.hpp"
#include "hipadj_fused.hpp"
#if !defined(__HIPCC_RTC__)
#include "hipadj_plan.hpp"
#endif

namespace hipadj {

constexpr int WAVE = 64;

struct SegPlan {
    int nseg;              // C
    const int* bounds;     // device [C+1] knot indices, bounds[0] = 0, bounds[C] = S
};
// (round 6 measured the bounds INSIDE the kernarg segment — a 65-entry table read with the scalar loads of the other arguments instead of a dependent global load: 0.1 us
// faster at 1250 trajectories, 1 us SLOWER at 2500 / 5000 / 10^4, profiles/r6_shard_time_bounds_ab.jsonl; the wave timeline shows why: the bounds are 0.5-0.9 us of a 28 us
// pass either way, profiles/r6_wave_trace_1250_*.jsonl — not kept)

template <class Mo>
__global__ void __launch_bounds__(WAVE) k_forward(Geom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                  dbl2* __restrict__ knots, double* __restrict__ ckpt,
                                                  const int* __restrict__ ckpt_of_knot, double* __restrict__ outT,
                                                  const int* __restrict__ save_of_knot, double* __restrict__ yT) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    forward_lane<Mo>(g, i, u0, p, knots, ckpt, ckpt_of_knot, outT, save_of_knot, yT);
}

// the forward solve as a tight loop between event knots (forward_lane_ev); padding lanes of the last wave repeat its last trajectory
// (their stores hit their own padded columns), so the wave runs without an exec mask
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_forward_ev(Geom g, const double* __restrict__ u0, const double* __restrict__ p, FwdEvents ev,
                                                     dbl2* __restrict__ knots, double* __restrict__ ckpt, double* __restrict__ outT, double* __restrict__ yT) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    forward_lane_ev<Mo>(g, i, u0, p, ev, knots, ckpt, outT, yT);
}

// the same forward solve with four lanes per trajectory (hipadj_quad.hpp): a wavefront = 16 trajectories, lane c of a quad = state component c; lanes
// beyond the model's n components and beyond the ensemble leave at once (the DPP permutations of a component form never read them)
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_forward_quad(Geom g, const double* __restrict__ u0, const double* __restrict__ p, FwdEvents ev,
                                                       dbl2* __restrict__ knots, double* __restrict__ ckpt, double* __restrict__ outT, double* __restrict__ yT) {
    const long i = (long)blockIdx.x * (WAVE / 4) + (threadIdx.x >> 2);
    const int c = threadIdx.x & 3;
    if (i >= g.N || c >= Mo::N) return;
    if constexpr (QuadForm<Mo>::value) forward_quad_ev<Mo>(g, i, c, u0, p, ev, knots, ckpt, outT, yT);
}

// segbuf layout: [segment][column][N+NP][Npad]; the top segment only fills column 0.
// SEG = false (runtime models whose (1 + n)(n + np) segment columns do not fit the register file: the planner keeps them
// sequential in time) compiles ONLY the one-column path: the unused (1 + n)-column code would otherwise set the kernel's
// register allocation (n = 7: 256 VGPRs + 256 AGPRs + 2.7 KB scratch, and a wrong result on the device).
// WPB = waves per workgroup.  1: grid (wave blocks, segments), one wave per workgroup.  4: a 1-D grid of 256-thread
// workgroups whose four waves take four consecutive (wave block, segment) items — the hardware spreads the waves of ONE
// workgroup over the four SIMDs of its CU, which single-wave workgroups do not guarantee (a SIMD that happens to receive
// three of a CU's eight waves finishes 1.5x later than the kernel needs).
// PSH = true: launched only when the parameters are shared (g.p_shared): lets models with stage operators keep their (p, dt)
// constants in SGPRs (interp_lane).
#ifndef HIPADJ_KINTERP_ATTR      // development hook (scripts/kbench.hip): e.g. __attribute__((amdgpu_waves_per_eu(3, 3)))
#define HIPADJ_KINTERP_ATTR
#endif
template <class Mo, int PF, int LOSS, bool SEG = true, int WPB = 1, bool PSH = false>
__global__ void HIPADJ_KINTERP_ATTR __launch_bounds__(WAVE * WPB) k_interp(Geom g, SegPlan sp, const double* __restrict__ p,
                                                       const dbl2* __restrict__ knots, const double* __restrict__ cotT,
                                                       const int* __restrict__ save_of_knot, double* __restrict__ segbuf) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    long i; int seg;
    if constexpr (WPB == 1) {
        i = (long)blockIdx.x * WAVE + threadIdx.x;
        seg = sp.nseg - 1 - (int)blockIdx.y;         // longest (top, 1-column) segment is dispatched first
    } else {
        const int nwb = (int)(g.Npad / WAVE);
        const int item = (int)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform (SGPR): loop bounds stay scalar
        if (item >= nwb * sp.nseg) return;
        seg = sp.nseg - 1 - item / nwb;
        i = (long)(item % nwb) * WAVE + (threadIdx.x & (WAVE - 1));
    }
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (!SEG || seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, PF, LOSS, 0, PSH>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * R + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * g.Npad] = mu[c][j];
        }
    }
}

// The same sweep as ONE launch per reverse pass: every wave hands its segment map to the composition tree of hipadj_fused.hpp
// instead of a segment buffer for k_compose_finish* / k_reduce_final.  Grid (wave blocks, segments), 64-thread workgroups; lanes
// beyond the ensemble (the padding of the last block) run on the padded tiles and are masked where results leave the wave.
// PSH (the stage-operator form of the compiled-in Lorenz model, the headline kernels): at least two wavefronts per SIMD, i.e. at most 256 VGPRs — the column-streaming
// instantiation came out at 260 once the loss gradient became la u + lb c (round 5), which halves the residency for four registers
template <class Mo, int PF, int LOSS, bool SEG = true, bool PSH = false>
__global__ void HIPADJ_KINTERP_ATTR __attribute__((amdgpu_waves_per_eu(PSH ? 2 : 1))) __launch_bounds__(WAVE) k_interp_fused(Geom g, SegPlan sp, TreePlan tp, const double* __restrict__ p,
                                                       const dbl2* __restrict__ knots, const double* __restrict__ cotT,
                                                       const int* __restrict__ save_of_knot, double* __restrict__ du0,
                                                       double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i_raw = (long)blockIdx.x * WAVE + threadIdx.x;
    const long i = i_raw < g.N ? i_raw : g.N - 1;                    // padding lanes of the last block repeat its last trajectory
    const int rank = (int)blockIdx.y, seg = sp.nseg - 1 - rank;      // rank 0 = the top (longest, 1-column) segment: dispatched first
    HIPADJ_TP(HIPADJ_GTRACE(g), 0, 0);                               // wave entry
#if defined(HIPADJ_WAVE_TRACE) && defined(__HIP_DEVICE_COMPILE__)
    if (HIPADJ_GTRACE(g) && threadIdx.x == 0) {                      // where the wave runs: (XCC_ID << 32) | HW_ID (SIMD, CU, SH, SE)
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        HIPADJ_GTRACE(g)[((long)blockIdx.y * gridDim.x + blockIdx.x) * 32 + 24] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    HIPADJ_TP(HIPADJ_GTRACE(g), 1, k_lo + k_hi);                     // segment bounds loaded
    if constexpr (!SEG) {   // one segment (models whose segment columns do not fit the registers): the wave is its block's root, no map is ever built
        double lam[1][N], mu[1][NP], v[R];
        interp_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) v[N + j] = mu[0][j];
        fused_root<N, NP>(v, tp, g.N, (long)gridDim.x, (long)blockIdx.x, du0, dp_rows, dp_sum, flag);
        return;
    }
    double m[NC * R];
    if (rank == 0) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int e = 0; e < NC * R; ++e) m[e] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) m[N + j] = mu[0][j];
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, PF, LOSS, 0, PSH>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) m[c * R + j] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) m[c * R + N + j] = mu[c][j];
        }
    }
    HIPADJ_TP(HIPADJ_GTRACE(g), 3, m[0]);                            // sweep done: the segment's map is in registers
    fused_tail<N, NP>(m, tp, g.N, (long)gridDim.x, (long)blockIdx.x, rank, du0, dp_rows, dp_sum, flag);
}

// The one-launch pass with G waves per workgroup (hipadj_fused.hpp, "GROUPED form"): grid (wave blocks, ceil(C / G)), wave w of workgroup y sweeps the segment of rank
// y G + w; the group's maps are composed through LDS, the group's first wave enters the HBM tree as leaf y.  Segmented sweeps of models with stage operators (the headline /
// shard kernels) only; G = 4: one wave per SIMD of the workgroup's CU, G = 8: two.
template <class Mo, int PF, int LOSS, int G>
__global__ void HIPADJ_KINTERP_ATTR __attribute__((amdgpu_waves_per_eu(2))) __launch_bounds__(WAVE * G) k_interp_fused_g(Geom g, SegPlan sp, TreePlan tp, const double* __restrict__ p,
                                                       const dbl2* __restrict__ knots, const double* __restrict__ cotT,
                                                       const int* __restrict__ save_of_knot, double* __restrict__ du0,
                                                       double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    __shared__ double lds[(G - 1) * NC * R * WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long i_raw = (long)blockIdx.x * WAVE + lane;
    const long i = i_raw < g.N ? i_raw : g.N - 1;                    // padding lanes of the last block repeat its last trajectory
    const int rank = (int)blockIdx.y * G + wv;
    const int nvalid = sp.nseg - (int)blockIdx.y * G < G ? sp.nseg - (int)blockIdx.y * G : G;
    double m[NC * R];
#pragma unroll
    for (int e = 0; e < NC * R; ++e) m[e] = 0.0;
#ifdef HIPADJ_PRIO_TOGGLE
    g.prio_phase = (2 * rank >= sp.nseg) ? 1 : 0;      // the second half of the ranks is dispatched behind the first: the younger wave of every SIMD
#endif
    if (rank < sp.nseg) {
        const int seg = sp.nseg - 1 - rank;
        HIPADJ_TP(HIPADJ_GTRACE(g), 0, 0);
        const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
        HIPADJ_TP(HIPADJ_GTRACE(g), 1, k_lo + k_hi);
        if (rank == 0) {
            double lam[1][N], mu[1][NP];
            interp_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
            for (int j = 0; j < N; ++j) m[j] = lam[0][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) m[N + j] = mu[0][j];
        } else {
            double lam[NC][N], mu[NC][NP];
            interp_lane<Mo, NC, PF, LOSS, 0, true>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int j = 0; j < N; ++j) m[c * R + j] = lam[c][j];
#pragma unroll
                for (int j = 0; j < NP; ++j) m[c * R + N + j] = mu[c][j];
            }
        }
        HIPADJ_TP(HIPADJ_GTRACE(g), 3, m[0]);
    }
    fused_tail_group<N, NP, G>(m, lds, tp, g.N, (long)gridDim.x, (long)blockIdx.x, (int)blockIdx.y, wv, nvalid, du0, dp_rows, dp_sum, flag);
}

// checkpointing=true variants: checkpoint tiles in HBM, interval re-solve tile in LDS ([step][component][lane])
// GT = true: checkpoint intervals longer than HIPADJ_CKPT_KMAX steps — the re-solve tile of a wave is a slice of an HBM scratch
// buffer (gtile + wave * gtile_stride, same [step][component][lane] layout: 512 B rows, coalesced) instead of LDS.
template <class Mo, int LOSS, bool GT = false>
__global__ void __launch_bounds__(WAVE) k_interp_ckpt(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ ckpt,
                                                      const int* __restrict__ ckpt_of_knot, const int* __restrict__ prev_ck,
                                                      const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                      double* __restrict__ segbuf, double* __restrict__ gtile, long gtile_stride) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP, KM = HIPADJ_CKPT_KMAX;
    __shared__ double tile_lds[GT ? 1 : (KM + 1) * N * WAVE];
    double* tile = GT ? gtile + ((long)blockIdx.y * gridDim.x + blockIdx.x) * gtile_stride : tile_lds;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    const CkptSrc C{ckpt, ckpt_of_knot, prev_ck, tile, WAVE, (int)threadIdx.x};
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * R + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * g.Npad] = mu[c][j];
        }
    }
}

// ---- finishing stage -------------------------------------------------------------------------------------
// Per 256-thread workgroup: (optionally) compose the segment maps of each trajectory, scan for NaN/Inf (the
// reference's retcode check), write du0 / per-trajectory dp, and reduce mu over the workgroup's trajectories in a
// FIXED order (shuffle tree per wave, then waves 0..3) into partial[block][NP]; k_reduce_final then sums the
// partials in block order => dp is bit-reproducible for a given N.
constexpr int FIN = 256;

template <int NP, int BS = FIN>
__device__ __forceinline__ void block_partial(const double (&mu)[NP], bool valid, double* __restrict__ partial) {
    __shared__ double sh[BS / WAVE][NP];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double v = valid ? mu[j] : 0.0;
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
        if (lane == 0) sh[wv][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < NP) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < BS / WAVE; ++w) t += sh[w][threadIdx.x];
        partial[(long)blockIdx.x * NP + threadIdx.x] = t;
    }
}

__device__ __forceinline__ bool finite_d(double v) { return fabs(v) <= 1.79769313486231570e308; }

// In-launch final reduction (replaces a separate k_reduce_final launch): the workgroup that draws the last arrival
// ticket sums all partials in block order.  Placement-independent hand-off per cdna_hip_programming.md §6 G16:
// plain stores -> __syncthreads -> one lane: agent-scope release + asm vmcnt(0) -> relaxed agent ticket;
// last arriver: agent-scope acquire -> plain loads.  The counter is reset by the last arriver (zeroed at create).
template <int NP>
__device__ __forceinline__ void final_reduce_last_arriver(const double* __restrict__ partial, int nblocks, unsigned* __restrict__ ticket_ctr,
                                                          double* __restrict__ dp) {
    __syncthreads();                       // this workgroup's partial[] stores are issued
    if (threadIdx.x >= WAVE) return;       // wave 0 does the hand-off
    const int lane = threadIdx.x;
    unsigned t = 0;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t = __shfl(t, 0, WAVE);
    if (t != (unsigned)nblocks - 1u) return;
    if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // lane l sums blocks l, l+64, ... in increasing order, then a fixed shuffle tree: deterministic for a given N
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double v = 0.0;
        for (int b = lane; b < nblocks; b += WAVE) v += partial[(long)b * NP + j];
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
        if (lane == 0) dp[j] = v;
    }
    if (lane == 0) __hip_atomic_store(ticket_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Composition of the C segment maps of each trajectory, top -> bottom:  lam <- A lam + c_l ; mu <- mu + B lam + c_m.
// A plain loop over the maps is a chain of C dependent load rounds (13 us for C = 13).  Instead FOUR lanes share a
// trajectory: each takes a contiguous quarter of the lower maps, fetches them in ONE round of independent loads and
// folds them into a single affine map (map o map), the first quarter applying its maps to the top segment's vector;
// three shuffle hops then push the vector through the other quarters' composed maps.
template <class Mo, int BS = FIN>   // BS = workgroup size: BS / 4 trajectories per workgroup (BS = 64: 4x the workgroups for small ensembles)
__global__ void __launch_bounds__(BS) k_compose_finish(Geom g, int nseg, const double* __restrict__ segbuf,
                                                        double* __restrict__ du0, double* __restrict__ dp_rows,
                                                        double* __restrict__ partial, int* __restrict__ flag,
                                                        unsigned* __restrict__ ticket_ctr, double* __restrict__ dp_sum) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP, CHT = 3;
    const long i_raw = (long)blockIdx.x * (BS / 4) + (threadIdx.x >> 2);
    const int part = threadIdx.x & 3;
    const bool tvalid = i_raw < g.N;
    const long i = tvalid ? i_raw : g.N - 1;
    const int L = nseg - 1;                                  // lower maps, rank 0 = segment nseg-2 ... rank L-1 = segment 0
    const int r0 = (L * part) / 4, r1 = (L * (part + 1)) / 4;
    // group map G (identity) or, for part 0, the running vector starting from the top segment's (c_l, c_m)
    double A[N][N], Bm[NP][N], cl[N], cm[NP];
#pragma unroll
    for (int a = 0; a < N; ++a) { cl[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) A[a][b] = (a == b) ? 1.0 : 0.0; }
#pragma unroll
    for (int a = 0; a < NP; ++a) { cm[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) Bm[a][b] = 0.0; }
    if (part == 0) {
        const double* __restrict__ src = segbuf + (long)(nseg - 1) * NC * R * g.Npad + i;
#pragma unroll
        for (int j = 0; j < N; ++j) cl[j] = src[(long)j * g.Npad];
#pragma unroll
        for (int j = 0; j < NP; ++j) cm[j] = src[(long)(N + j) * g.Npad];
    }
    for (int rb = r0; rb < r1; rb += CHT) {
        double m[CHT][NC * R];
#pragma unroll
        for (int q = 0; q < CHT; ++q) {
            const int rk = rb + q < r1 ? rb + q : r1 - 1;
            const double* __restrict__ src = segbuf + (long)(nseg - 2 - rk) * NC * R * g.Npad + i;
#pragma unroll
            for (int e = 0; e < NC * R; ++e) m[q][e] = src[(long)e * g.Npad];
        }
#pragma unroll
        for (int q = 0; q < CHT; ++q) {
            if (rb + q < r1) {
                // lower map Lm: c_l = m[j], c_m = m[N+j], A[:,c] = m[(c+1)R + j], B[:,c] = m[(c+1)R + N + j]
                double ncl[N], ncm[NP];
#pragma unroll
                for (int j = 0; j < N; ++j) { ncl[j] = m[q][j];
#pragma unroll
                    for (int c = 0; c < N; ++c) ncl[j] += m[q][(c + 1) * R + j] * cl[c]; }
#pragma unroll
                for (int j = 0; j < NP; ++j) { ncm[j] = cm[j] + m[q][N + j];
#pragma unroll
                    for (int c = 0; c < N; ++c) ncm[j] += m[q][(c + 1) * R + N + j] * cl[c]; }
                if (part != 0) {   // full map o map for the quarters that do not know their input vector yet
                    double nA[N][N], nB[NP][N];
#pragma unroll
                    for (int a = 0; a < N; ++a)
#pragma unroll
                        for (int b = 0; b < N; ++b) { double v = 0.0;
#pragma unroll
                            for (int c = 0; c < N; ++c) v += m[q][(c + 1) * R + a] * A[c][b];
                            nA[a][b] = v; }
#pragma unroll
                    for (int a = 0; a < NP; ++a)
#pragma unroll
                        for (int b = 0; b < N; ++b) { double v = Bm[a][b];
#pragma unroll
                            for (int c = 0; c < N; ++c) v += m[q][(c + 1) * R + N + a] * A[c][b];
                            nB[a][b] = v; }
#pragma unroll
                    for (int a = 0; a < N; ++a)
#pragma unroll
                        for (int b = 0; b < N; ++b) A[a][b] = nA[a][b];
#pragma unroll
                    for (int a = 0; a < NP; ++a)
#pragma unroll
                        for (int b = 0; b < N; ++b) Bm[a][b] = nB[a][b];
                }
#pragma unroll
                for (int j = 0; j < N; ++j) cl[j] = ncl[j];
#pragma unroll
                for (int j = 0; j < NP; ++j) cm[j] = ncm[j];
            }
        }
    }
    // push the vector through quarters 1..3: v <- G_q(v)
    double lam[N], mu[NP];
#pragma unroll
    for (int j = 0; j < N; ++j) lam[j] = cl[j];
#pragma unroll
    for (int j = 0; j < NP; ++j) mu[j] = cm[j];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        double li[N], mi[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) li[j] = __shfl(lam[j], (threadIdx.x & ~3) + q - 1, WAVE);
#pragma unroll
        for (int j = 0; j < NP; ++j) mi[j] = __shfl(mu[j], (threadIdx.x & ~3) + q - 1, WAVE);
        if (part == q) {
#pragma unroll
            for (int j = 0; j < N; ++j) { double v = cl[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += A[j][c] * li[c];
                lam[j] = v; }
#pragma unroll
            for (int j = 0; j < NP; ++j) { double v = mi[j] + cm[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += Bm[j][c] * li[c];
                mu[j] = v; }
        }
    }
    const bool owner = tvalid && part == 3;
    if (owner) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < N; ++j) { du0[i * N + j] = lam[j]; bad |= !finite_d(lam[j]); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { bad |= !finite_d(mu[j]); if (dp_rows) dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(flag, 1);
    }
    block_partial<NP, BS>(mu, owner, partial);
    if (dp_sum) final_reduce_last_arriver<NP>(partial, (int)gridDim.x, ticket_ctr, dp_sum);
}

// The same composition with SIXTEEN lanes per trajectory (one DPP row), for ensembles that leave the chip idle during this kernel (< 65 536
// trajectories: the 4-lane form runs 625 single-wave workgroups at 10^4 and spends its time in ONE round of 72 loads per lane).  Lane q folds
// its contiguous chunk of the lower maps (one map at 13 segments, four at 62) into an affine map; the vector then walks down the row:
// step s moves it one lane to the right with two DPP row_shr:1 moves per double — VALU only, no LDS crossbar — and lane s applies its map.
// 3x the loads in flight and a chain of 15 cheap steps instead of three shuffle hops after a long load round.
__device__ __forceinline__ double dpp_row_shr1(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true));
}
template <class Mo>
__global__ void __launch_bounds__(FIN) k_compose_finish16(Geom g, int nseg, const double* __restrict__ segbuf,
                                                          double* __restrict__ du0, double* __restrict__ dp_rows,
                                                          double* __restrict__ partial, int* __restrict__ flag,
                                                          unsigned* __restrict__ ticket_ctr, double* __restrict__ dp_sum) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i_raw = (long)blockIdx.x * (FIN / 16) + (threadIdx.x >> 4);
    const int part = threadIdx.x & 15;
    const bool tvalid = i_raw < g.N;
    const long i = tvalid ? i_raw : g.N - 1;
    const int L = nseg - 1;                                  // lower maps, rank 0 = segment nseg-2 ... rank L-1 = segment 0
    const int r0 = (L * part) / 16, r1 = (L * (part + 1)) / 16;
    double A[N][N], Bm[NP][N], cl[N], cm[NP];                // this lane's composed map: lam <- A lam + cl ; mu <- mu + Bm lam + cm
#pragma unroll
    for (int a = 0; a < N; ++a) { cl[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) A[a][b] = (a == b) ? 1.0 : 0.0; }
#pragma unroll
    for (int a = 0; a < NP; ++a) { cm[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) Bm[a][b] = 0.0; }
    for (int rk = r0; rk < r1; ++rk) {
        double m[NC * R];
        const double* __restrict__ src = segbuf + (long)(nseg - 2 - rk) * NC * R * g.Npad + i;
#pragma unroll
        for (int e = 0; e < NC * R; ++e) m[e] = src[(long)e * g.Npad];
        // G <- m o G   (m: c_l = m[j], c_m = m[N+j], A[:,c] = m[(c+1)R + j], B[:,c] = m[(c+1)R + N + j])
        double nA[N][N], nB[NP][N], ncl[N], ncm[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) { ncl[j] = m[j];
#pragma unroll
            for (int c = 0; c < N; ++c) ncl[j] += m[(c + 1) * R + j] * cl[c]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) { ncm[j] = cm[j] + m[N + j];
#pragma unroll
            for (int c = 0; c < N; ++c) ncm[j] += m[(c + 1) * R + N + j] * cl[c]; }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) { double v = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) v += m[(c + 1) * R + a] * A[c][b];
                nA[a][b] = v; }
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) { double v = Bm[a][b];
#pragma unroll
                for (int c = 0; c < N; ++c) v += m[(c + 1) * R + N + a] * A[c][b];
                nB[a][b] = v; }
#pragma unroll
        for (int a = 0; a < N; ++a) { cl[a] = ncl[a];
#pragma unroll
            for (int b = 0; b < N; ++b) A[a][b] = nA[a][b]; }
#pragma unroll
        for (int a = 0; a < NP; ++a) { cm[a] = ncm[a];
#pragma unroll
            for (int b = 0; b < N; ++b) Bm[a][b] = nB[a][b]; }
    }
    // the vector: lane 0 starts from the top segment's (c_l, c_m) and applies its own map; then one lane to the right per step
    double lam[N], mu[NP];
    {
        const double* __restrict__ src = segbuf + (long)(nseg - 1) * NC * R * g.Npad + i;
        double tl[N], tm[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) tl[j] = part == 0 ? src[(long)j * g.Npad] : 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) tm[j] = part == 0 ? src[(long)(N + j) * g.Npad] : 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) { double v = cl[j];
#pragma unroll
            for (int c = 0; c < N; ++c) v += A[j][c] * tl[c];
            lam[j] = v; }
#pragma unroll
        for (int j = 0; j < NP; ++j) { double v = tm[j] + cm[j];
#pragma unroll
            for (int c = 0; c < N; ++c) v += Bm[j][c] * tl[c];
            mu[j] = v; }
    }
#pragma unroll
    for (int s = 1; s < 16; ++s) {
        double li[N], mi[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) li[j] = dpp_row_shr1(lam[j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) mi[j] = dpp_row_shr1(mu[j]);
        if (part == s) {
#pragma unroll
            for (int j = 0; j < N; ++j) { double v = cl[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += A[j][c] * li[c];
                lam[j] = v; }
#pragma unroll
            for (int j = 0; j < NP; ++j) { double v = mi[j] + cm[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += Bm[j][c] * li[c];
                mu[j] = v; }
        }
    }
    const bool owner = tvalid && part == 15;
    if (owner) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < N; ++j) { du0[i * N + j] = lam[j]; bad |= !finite_d(lam[j]); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { bad |= !finite_d(mu[j]); if (dp_rows) dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(flag, 1);
    }
    block_partial<NP, FIN>(mu, owner, partial);
    if (dp_sum) final_reduce_last_arriver<NP>(partial, (int)gridDim.x, ticket_ctr, dp_sum);
}

// The composition with ONE LANE per trajectory and the lower maps spread over the W waves of the workgroup.  Every load is a wave-wide 512-byte row
// of the segment buffer (the 4- and 16-lane forms above put lanes of one trajectory on different maps: 64 cache lines per load instruction, and the
// kernel was latency-bound at 11-14 us whatever the ensemble size); wave w folds its contiguous chunk of the maps into one affine map with the next
// map's loads in flight, then the running vector (lam, mu) goes down the waves through a 3 KB LDS tile, one barrier per hand-over.
template <class Mo, int W>
__global__ void __launch_bounds__(WAVE * W) k_compose_finish_w(Geom g, int nseg, const double* __restrict__ segbuf,
                                                               double* __restrict__ du0, double* __restrict__ dp_rows,
                                                               double* __restrict__ partial, int* __restrict__ flag,
                                                               unsigned* __restrict__ ticket_ctr, double* __restrict__ dp_sum) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    __shared__ double vec[R][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const long i_raw = (long)blockIdx.x * WAVE + lane;
    const bool tvalid = i_raw < g.N;
    const long i = tvalid ? i_raw : g.N - 1;
    const int L = nseg - 1;                                  // lower maps, rank 0 = segment nseg-2 ... rank L-1 = segment 0
    const int r0 = (L * wv) / W, r1 = (L * (wv + 1)) / W;
    double A[N][N], Bm[NP][N], cl[N], cm[NP];                // this wave's composed map: lam <- A lam + cl ; mu <- mu + Bm lam + cm
#pragma unroll
    for (int a = 0; a < N; ++a) { cl[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) A[a][b] = (a == b) ? 1.0 : 0.0; }
#pragma unroll
    for (int a = 0; a < NP; ++a) { cm[a] = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) Bm[a][b] = 0.0; }
    double m[NC * R], mn[NC * R];
    if (r0 < r1) {
        const double* __restrict__ src = segbuf + (long)(nseg - 2 - r0) * NC * R * g.Npad + i;
#pragma unroll
        for (int e = 0; e < NC * R; ++e) mn[e] = src[(long)e * g.Npad];
    }
    for (int rk = r0; rk < r1; ++rk) {
#pragma unroll
        for (int e = 0; e < NC * R; ++e) m[e] = mn[e];
        if (rk + 1 < r1) {                                   // the next map's loads are in flight while this one is folded in
            const double* __restrict__ src = segbuf + (long)(nseg - 3 - rk) * NC * R * g.Npad + i;
#pragma unroll
            for (int e = 0; e < NC * R; ++e) mn[e] = src[(long)e * g.Npad];
        }
        // G <- m o G   (m: c_l = m[j], c_m = m[N+j], A[:,c] = m[(c+1)R + j], B[:,c] = m[(c+1)R + N + j])
        double nA[N][N], nB[NP][N], ncl[N], ncm[NP];
#pragma unroll
        for (int j = 0; j < N; ++j) { ncl[j] = m[j];
#pragma unroll
            for (int c = 0; c < N; ++c) ncl[j] += m[(c + 1) * R + j] * cl[c]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) { ncm[j] = cm[j] + m[N + j];
#pragma unroll
            for (int c = 0; c < N; ++c) ncm[j] += m[(c + 1) * R + N + j] * cl[c]; }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) { double v = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) v += m[(c + 1) * R + a] * A[c][b];
                nA[a][b] = v; }
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) { double v = Bm[a][b];
#pragma unroll
                for (int c = 0; c < N; ++c) v += m[(c + 1) * R + N + a] * A[c][b];
                nB[a][b] = v; }
#pragma unroll
        for (int a = 0; a < N; ++a) { cl[a] = ncl[a];
#pragma unroll
            for (int b = 0; b < N; ++b) A[a][b] = nA[a][b]; }
#pragma unroll
        for (int a = 0; a < NP; ++a) { cm[a] = ncm[a];
#pragma unroll
            for (int b = 0; b < N; ++b) Bm[a][b] = nB[a][b]; }
    }
    // the vector: wave 0 starts from the top segment's (c_l, c_m), every wave applies its own map and hands over through LDS
    double lam[N], mu[NP];
#pragma unroll
    for (int j = 0; j < N; ++j) lam[j] = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) mu[j] = 0.0;
#pragma unroll 1
    for (int s = 0; s < W; ++s) {
        if (wv == s) {
            double tl[N], tm[NP];
            if (s == 0) {
                const double* __restrict__ src = segbuf + (long)(nseg - 1) * NC * R * g.Npad + i;
#pragma unroll
                for (int j = 0; j < N; ++j) tl[j] = src[(long)j * g.Npad];
#pragma unroll
                for (int j = 0; j < NP; ++j) tm[j] = src[(long)(N + j) * g.Npad];
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) tl[j] = vec[j][lane];
#pragma unroll
                for (int j = 0; j < NP; ++j) tm[j] = vec[N + j][lane];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) { double v = cl[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += A[j][c] * tl[c];
                lam[j] = v; }
#pragma unroll
            for (int j = 0; j < NP; ++j) { double v = tm[j] + cm[j];
#pragma unroll
                for (int c = 0; c < N; ++c) v += Bm[j][c] * tl[c];
                mu[j] = v; }
            if (s < W - 1) {
#pragma unroll
                for (int j = 0; j < N; ++j) vec[j][lane] = lam[j];
#pragma unroll
                for (int j = 0; j < NP; ++j) vec[N + j][lane] = mu[j];
            }
        }
        if (s < W - 1) __syncthreads();
    }
    const bool owner = tvalid && wv == W - 1;
    if (owner) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < N; ++j) { du0[i * N + j] = lam[j]; bad |= !finite_d(lam[j]); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { bad |= !finite_d(mu[j]); if (dp_rows) dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(flag, 1);
    }
    block_partial<NP, WAVE * W>(mu, owner, partial);
    if (dp_sum) final_reduce_last_arriver<NP>(partial, (int)gridDim.x, ticket_ctr, dp_sum);
}

// finishing stage for kernels that already wrote du0 [N][n] and dp_traj [NP][Npad]
template <int N, int NP>
__global__ void __launch_bounds__(FIN) k_finish(long Ntraj, long Npad, const double* __restrict__ du0,
                                                const double* __restrict__ dp_traj, double* __restrict__ dp_rows,
                                                double* __restrict__ partial, int* __restrict__ flag,
                                                unsigned* __restrict__ ticket_ctr, double* __restrict__ dp_sum) {
    const long i = (long)blockIdx.x * FIN + threadIdx.x;
    const bool valid = i < Ntraj;
    double mu[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) mu[j] = 0.0;
    if (valid) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < N; ++j) bad |= !finite_d(du0[i * N + j]);
#pragma unroll
        for (int j = 0; j < NP; ++j) { mu[j] = dp_traj[(long)j * Npad + i]; bad |= !finite_d(mu[j]); if (dp_rows) dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(flag, 1);
    }
    block_partial<NP>(mu, valid, partial);
    if (dp_sum) final_reduce_last_arriver<NP>(partial, (int)gridDim.x, ticket_ctr, dp_sum);
}

// finishing stage for the one-segment (SEG = false) sweeps: the top map's column 0 of segbuf IS (lam(t0), mu(t0))
template <int N, int NP>
__global__ void __launch_bounds__(FIN) k_finish_map(long Ntraj, long Npad, const double* __restrict__ segbuf, double* __restrict__ du0,
                                                    double* __restrict__ dp_rows, double* __restrict__ partial, int* __restrict__ flag,
                                                    unsigned* __restrict__ ticket_ctr, double* __restrict__ dp_sum) {
    const long i = (long)blockIdx.x * FIN + threadIdx.x;
    const bool valid = i < Ntraj;
    double mu[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) mu[j] = 0.0;
    if (valid) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < N; ++j) { const double v = segbuf[(long)j * Npad + i]; du0[i * N + j] = v; bad |= !finite_d(v); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { mu[j] = segbuf[(long)(N + j) * Npad + i]; bad |= !finite_d(mu[j]); if (dp_rows) dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(flag, 1);
    }
    block_partial<NP>(mu, valid, partial);
    if (dp_sum) final_reduce_last_arriver<NP>(partial, (int)gridDim.x, ticket_ctr, dp_sum);
}

// dp[j] = sum over workgroup partials in block order (one workgroup per parameter)
static __global__ void __launch_bounds__(FIN) k_reduce_final(int nblocks, int np, const double* __restrict__ partial, double* __restrict__ dp) {
    __shared__ double sh[FIN];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += FIN) s += partial[(long)b * np + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = FIN / 2; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) dp[j] = sh[0];
}

// BacksolveAdjoint, segmented at checkpoint knots; writes the segment maps like k_interp
template <class Mo, int CC, bool SEG = true>
__global__ void __launch_bounds__(WAVE) k_backsolve(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ yT,
                                                    const double* __restrict__ ckpt, const int* __restrict__ ckpt_of_knot,
                                                    const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                    double* __restrict__ segbuf) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (!SEG || seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        backsolve_lane<Mo, 1, CC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        backsolve_lane<Mo, NC, CC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * R + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * g.Npad] = mu[c][j];
        }
    }
}

template <class Mo, int NC>
__device__ __forceinline__ void store_segment_map(double* __restrict__ dst, long Npad, const double (&lam)[NC][Mo::N], const double (&mu)[NC][Mo::NP]) {
    constexpr int N = Mo::N, NP = Mo::NP, R = N + NP;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int j = 0; j < N; ++j) dst[((long)c * R + j) * Npad] = lam[c][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * Npad] = mu[c][j];
    }
}

// GaussAdjoint, time-segmented like k_interp (segment maps -> k_compose_finish)
template <class Mo, int PF, int LOSS, bool GKR = false, bool SEG = true>
__global__ void __launch_bounds__(WAVE) k_gauss(Geom g, SegPlan sp, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                double* __restrict__ segbuf) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (!SEG || seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        gauss_lane<Mo, 1, PF, LOSS, 0, GKR>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        store_segment_map<Mo, 1>(dst, g.Npad, lam, mu);
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        gauss_lane<Mo, NC, PF, LOSS, 0, GKR>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        store_segment_map<Mo, NC>(dst, g.Npad, lam, mu);
    }
}

// a wave's segment result as a map for the composition tree of hipadj_fused.hpp: NCV = 1 (top / only segment) leaves the basis columns zero
template <class Mo, int NCV>
__device__ __forceinline__ void segment_map_regs(double (&m)[(1 + Mo::N) * (Mo::N + Mo::NP)], const double (&lam)[NCV][Mo::N], const double (&mu)[NCV][Mo::NP]) {
    constexpr int N = Mo::N, NP = Mo::NP, R = N + NP;
#pragma unroll
    for (int e = 0; e < (1 + N) * R; ++e) m[e] = 0.0;
#pragma unroll
    for (int c = 0; c < NCV; ++c) {
#pragma unroll
        for (int j = 0; j < N; ++j) m[c * R + j] = lam[c][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) m[c * R + N + j] = mu[c][j];
    }
}

// GaussAdjoint and BacksolveAdjoint (segmented at checkpoint knots) as ONE launch per reverse pass, like k_interp_fused
template <class Mo, int PF, int LOSS, bool GKR = false, bool SEG = true>
__global__ void __launch_bounds__(WAVE) k_gauss_fused(Geom g, SegPlan sp, TreePlan tp, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                      const double* __restrict__ cotT, const int* __restrict__ save_of_knot, double* __restrict__ du0,
                                                      double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i_raw = (long)blockIdx.x * WAVE + threadIdx.x;
    const long i = i_raw < g.N ? i_raw : g.N - 1;
    const int rank = (int)blockIdx.y, seg = sp.nseg - 1 - rank;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    if constexpr (!SEG) {
        double lam[1][N], mu[1][NP], v[R];
        gauss_lane<Mo, 1, PF, LOSS, 0, GKR>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) v[N + j] = mu[0][j];
        fused_root<N, NP>(v, tp, g.N, (long)gridDim.x, (long)blockIdx.x, du0, dp_rows, dp_sum, flag);
        return;
    }
    double m[NC * R];
    if (rank == 0) {
        double lam[1][N], mu[1][NP];
        gauss_lane<Mo, 1, PF, LOSS, 0, GKR>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_map_regs<Mo, 1>(m, lam, mu);
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        gauss_lane<Mo, NC, PF, LOSS, 0, GKR>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_map_regs<Mo, NC>(m, lam, mu);
    }
    fused_tail<N, NP>(m, tp, g.N, (long)gridDim.x, (long)blockIdx.x, rank, du0, dp_rows, dp_sum, flag);
}

template <class Mo, int CC, bool SEG = true>
__global__ void __launch_bounds__(WAVE) k_backsolve_fused(Geom g, SegPlan sp, TreePlan tp, const double* __restrict__ p, const double* __restrict__ yT,
                                                          const double* __restrict__ ckpt, const int* __restrict__ ckpt_of_knot,
                                                          const double* __restrict__ cotT, const int* __restrict__ save_of_knot, double* __restrict__ du0,
                                                          double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const long i_raw = (long)blockIdx.x * WAVE + threadIdx.x;
    const long i = i_raw < g.N ? i_raw : g.N - 1;
    const int rank = (int)blockIdx.y, seg = sp.nseg - 1 - rank;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    if constexpr (!SEG) {
        double lam[1][N], mu[1][NP], v[R];
        backsolve_lane<Mo, 1, CC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) v[N + j] = mu[0][j];
        fused_root<N, NP>(v, tp, g.N, (long)gridDim.x, (long)blockIdx.x, du0, dp_rows, dp_sum, flag);
        return;
    }
    double m[NC * R];
    if (rank == 0) {
        double lam[1][N], mu[1][NP];
        backsolve_lane<Mo, 1, CC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
        segment_map_regs<Mo, 1>(m, lam, mu);
    } else if constexpr (SEG) {
        double lam[NC][N], mu[NC][NP];
        backsolve_lane<Mo, NC, CC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
        segment_map_regs<Mo, NC>(m, lam, mu);
    }
    fused_tail<N, NP>(m, tp, g.N, (long)gridDim.x, (long)blockIdx.x, rank, du0, dp_rows, dp_sum, flag);
}

template <class Mo, int LOSS, bool GT = false>
__global__ void __launch_bounds__(WAVE) k_gauss_ckpt(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ ckpt,
                                                     const int* __restrict__ ckpt_of_knot, const int* __restrict__ prev_ck,
                                                     const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                     double* __restrict__ segbuf, double* __restrict__ gtile, long gtile_stride) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP, KM = HIPADJ_CKPT_KMAX;
    __shared__ double tile_lds[GT ? 1 : (KM + 1) * N * WAVE];
    double* tile = GT ? gtile + ((long)blockIdx.y * gridDim.x + blockIdx.x) * gtile_stride : tile_lds;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    const CkptSrc C{ckpt, ckpt_of_knot, prev_ck, tile, WAVE, (int)threadIdx.x};
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        gauss_lane<Mo, 1, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        store_segment_map<Mo, 1>(dst, g.Npad, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        gauss_lane<Mo, NC, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        store_segment_map<Mo, NC>(dst, g.Npad, lam, mu);
    }
}

// InterpolatingAdjoint with loss times off the step grid: one lane per trajectory, sequential in time (hipadj_lane.hpp)
template <class Mo, int MODE>
__global__ void __launch_bounds__(WAVE) k_interp_offgrid(Geom g, RevSteps R, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                         const double* __restrict__ cotT, double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[1][N], mu[1][NP];
    interp_offgrid_lane<Mo, MODE>(g, i, p, knots, cotT, R, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[0][j];
}
template <class Mo, int MODE, bool GKR = false>   // GKR: GaussKronrodAdjoint (the adaptive (7,15) rule per reverse step)
__global__ void __launch_bounds__(WAVE) k_gauss_offgrid(Geom g, RevSteps R, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                        const double* __restrict__ cotT, double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[1][N], mu[1][NP];
    gauss_offgrid_lane<Mo, MODE, 1, GKR>(g, i, p, knots, cotT, R, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[0][j];
}
// Interpolating / Gauss / GaussKronrod (ALG 0 / 2 / 4) with checkpointing = true over the reverse step list (offgrid_ckpt_lane): one lane per trajectory, sequential in time;
// `tile` is the per-lane knot tile of one checkpoint interval [(longest interval) + 1][N pairs][Npad], written and read by the same lane
template <class Mo, int MODE, int ALG>
__global__ void __launch_bounds__(WAVE) k_offgrid_ckpt(Geom g, RevSteps R, OgIntervals I, const double* __restrict__ p, const double* __restrict__ ckpt, dbl2* tile,
                                                       const double* __restrict__ cotT, double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[1][N], mu[1][NP];
    offgrid_ckpt_lane<Mo, MODE, ALG>(g, i, p, ckpt, tile, cotT, R, I, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[0][j];
}
// Time-segmented off-grid sweeps: the reverse step list (host-planned, the same for every trajectory) is cut into sp.nseg pieces;
// bounds are in "position" r = R.n - q (increasing with time like knot indices): segment s covers steps q in [R.n - bounds[s + 1],
// R.n - bounds[s]).  Segment maps in the k_interp layout, composed by k_compose_finish.  GAUSS = true: gauss_offgrid_lane.
template <class Mo, int MODE, bool GAUSS>
__global__ void __launch_bounds__(WAVE) k_offgrid_seg(Geom g, RevSteps R, SegPlan sp, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                      const double* __restrict__ cotT, double* __restrict__ segbuf) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, RR = N + NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;         // the (longer, 1-column) top segment is dispatched first
    if (i >= g.N) return;
    const int q_lo = R.n - sp.bounds[seg + 1], q_hi = R.n - sp.bounds[seg];
    double* __restrict__ dst = segbuf + (long)seg * NC * RR * g.Npad + i;
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        if constexpr (GAUSS) gauss_offgrid_lane<Mo, MODE, 1>(g, i, p, knots, cotT, R, lam, mu, q_lo, q_hi);
        else interp_offgrid_lane<Mo, MODE, 1>(g, i, p, knots, cotT, R, lam, mu, q_lo, q_hi);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else {
        double lam[NC][N], mu[NC][NP];
        if constexpr (GAUSS) gauss_offgrid_lane<Mo, MODE, NC>(g, i, p, knots, cotT, R, lam, mu, q_lo, q_hi);
        else interp_offgrid_lane<Mo, MODE, NC>(g, i, p, knots, cotT, R, lam, mu, q_lo, q_hi);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * RR + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * RR + N + j) * g.Npad] = mu[c][j];
        }
    }
}
template <class Mo, int CC>
__global__ void __launch_bounds__(WAVE) k_backsolve_offgrid(Geom g, RevSteps R, const double* __restrict__ p, const double* __restrict__ yT, const double* __restrict__ ckpt,
                                                            const double* __restrict__ cotT, double* __restrict__ du0, double* __restrict__ dp_traj) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[1][N], mu[1][NP];
    backsolve_offgrid_lane<Mo, CC>(g, i, p, yT, ckpt, cotT, R, lam, mu);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[0][j];
#pragma unroll
    for (int j = 0; j < NP; ++j) dp_traj[(long)j * g.Npad + i] = mu[0][j];
}
// sol(times[0..nt)) -> dst [nt][n][Npad]: the primal output at off-grid save times, or Backsolve's checkpoint states
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_out_offgrid(Geom g, const dbl2* __restrict__ knots, const double* __restrict__ times, int nt, double* __restrict__ dst) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    out_offgrid_lane<Mo>(g, i, knots, times, nt, dst);
}

template <class Mo, int PF, int LOSS>
__global__ void __launch_bounds__(WAVE) k_quad_adj(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                   const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                   dbl2* __restrict__ adj, double* __restrict__ du0, double* __restrict__ gpd_out) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N], gpo[NP];
    quad_adj_lane<Mo, PF, LOSS>(g, i, p, knots, cotT, save_of_knot, adj, lam, gpo);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
    if (gpd_out) {   // [NP][Npad]: sum of dgdp_discrete over the loss times, one more term of k_quad_sum (models with discrete-loss bodies)
#pragma unroll
        for (int j = 0; j < NP; ++j) gpd_out[(long)j * g.Npad + i] = gpo[j];
    }
}

// one lane per (trajectory, quadrature interval); qres [interval][NP][Npad]
template <class Mo, int CC = 0>
__global__ void __launch_bounds__(WAVE) k_quad_gk(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                  const dbl2* __restrict__ adj, const double* __restrict__ qa,
                                                  const double* __restrict__ qb, double atol, double rtol,
                                                  double* __restrict__ qres) {
    constexpr int NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int q = blockIdx.y;
    if (i >= g.N) return;
    double res[NP];
    quad_gk_lane<Mo, 128, CC>(g, i, p, knots, adj, qa[q], qb[q], atol, rtol, res);
#pragma unroll
    for (int j = 0; j < NP; ++j) qres[((long)q * NP + j) * g.Npad + i] = res[j];
}
// QuadratureAdjoint with loss times off the step grid (hipadj_lane.hpp): pass 1 sequential over the reverse step list, pass 2 one lane per
// (trajectory, loss interval)
template <class Mo, int MODE>
__global__ void __launch_bounds__(WAVE) k_quad_adj_offgrid(Geom g, RevSteps R, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                           const double* __restrict__ cotT, dbl2* __restrict__ adj, double* __restrict__ du0, double* __restrict__ gpd_out) {
    constexpr int N = Mo::N, NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N], gpo[NP];
    quad_adj_offgrid_lane<Mo, MODE>(g, i, p, knots, cotT, R, adj, lam, gpo);
#pragma unroll
    for (int j = 0; j < N; ++j) du0[i * N + j] = lam[j];
    if (gpd_out) {
#pragma unroll
        for (int j = 0; j < NP; ++j) gpd_out[(long)j * g.Npad + i] = gpo[j];
    }
}
template <class Mo, int CC = 0>
__global__ void __launch_bounds__(WAVE) k_quad_gk_offgrid(Geom g, RevSteps R, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                          const dbl2* __restrict__ adj, const double* __restrict__ qa, const double* __restrict__ qb,
                                                          double atol, double rtol, double* __restrict__ qres) {
    constexpr int NP = Mo::NP;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int q = blockIdx.y;
    if (i >= g.N) return;
    double res[NP];
    quad_gk_offgrid_lane<Mo, 128, CC>(g, i, p, knots, adj, R, qa[q], qb[q], atol, rtol, res);
#pragma unroll
    for (int j = 0; j < NP; ++j) qres[((long)q * NP + j) * g.Npad + i] = res[j];
}
// res .+= quadgk(...) in the reference's order (src/quadrature_adjoint.jl:563-616)
// add = 1: dp_traj already holds the sum of dgdp_discrete over the loss times (src/quadrature_adjoint.jl:545-552, 601-605), left there by pass 1 for a model with discrete-loss bodies
static __global__ void __launch_bounds__(WAVE) k_quad_sum(long N, long Npad, int np, int nq, const double* __restrict__ qres,
                                                   double* __restrict__ dp_traj, int add = 0) {
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= N) return;
    for (int j = 0; j < np; ++j) {
        double s = add ? dp_traj[(long)j * Npad + i] : 0.0;
        for (int q = 0; q < nq; ++q) s += qres[((long)q * np + j) * Npad + i];
        dp_traj[(long)j * Npad + i] = s;
    }
}

// ---- utilities -------------------------------------------------------------------------------------------
// AoS [N][C] (caller layout) -> SoA [C][Npad]; 32x32 LDS tile, +1 padding against bank conflicts
static __global__ void k_aos_to_soa(const double* __restrict__ src, double* __restrict__ dst, long N, long Npad, int C) {
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
static __global__ void k_soa_to_aos(const double* __restrict__ src, double* __restrict__ dst, long N, long Npad, int C) {
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

// ---- the loss itself (hipadj_loss_value): built-in kinds from the primal output out [N][M][n] (caller layout) and the data block [N][M][n] --------------------------
//   kind 1: sum |u - shift|^2 / 2;  kind 2: scale / 2 * sum |u - data|^2.  m0 = 1 leaves the loss time at t0 out (no_start).
// Two levels in a fixed order: block b sums elements [b * LV_CHUNK, (b + 1) * LV_CHUNK) (thread-strided partials, then a fixed tree), k_sum_fixed adds the block partials.
constexpr int LV_CHUNK = 16384;
static __global__ void __launch_bounds__(256) k_loss_value(long total, int M, int n, int m0, int kind, double shift, double scale, const double* __restrict__ out,
                                                           const double* __restrict__ data, double* __restrict__ part) {
    __shared__ double sh[256];
    const long b0 = (long)blockIdx.x * LV_CHUNK, b1 = b0 + LV_CHUNK < total ? b0 + LV_CHUNK : total;
    double s = 0.0;
    for (long e = b0 + threadIdx.x; e < b1; e += 256) {
        const int m = (int)((e / n) % M);
        if (m < m0) continue;
        const double r = out[e] - (kind == 2 ? data[e] : shift);
        s += r * r;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = 0.5 * (kind == 2 ? scale : 1.0) * sh[0];
}
static __global__ void __launch_bounds__(256) k_sum_fixed(long count, const double* __restrict__ part, double* __restrict__ res) {
    __shared__ double sh[256];
    double s = 0.0;
    for (long e = threadIdx.x; e < count; e += 256) s += part[e];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) res[0] = sh[0];
}
// ... of a runtime model registered with hipadj_model_set_discrete_loss_function: part[i] = sum over the loss times of l_i(u(t_i), p, t_i, i, d_i) for trajectory i
template <class Mo, class = void> struct model_has_lvalue { static constexpr bool value = false; };
template <class Mo> struct model_has_lvalue<Mo, decltype((void)Mo::HAS_LVALUE)> { static constexpr bool value = Mo::HAS_LVALUE; };
template <class Mo>
__global__ void __launch_bounds__(256) k_user_loss_value(long N, int M, int m0, long ldp, const double* __restrict__ out, const double* __restrict__ data, const double* __restrict__ p,
                                                         const double* __restrict__ save_t, double* __restrict__ part) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
    if constexpr (model_has_lvalue<Mo>::value) {
        double pv[Mo::NP], u[Mo::N], d[Mo::N];
#pragma unroll
        for (int j = 0; j < Mo::NP; ++j) pv[j] = p[i * ldp + j];
        for (int m = m0; m < M; ++m) {
#pragma unroll
            for (int j = 0; j < Mo::N; ++j) { u[j] = out[(i * M + m) * Mo::N + j]; d[j] = data ? data[(i * M + m) * Mo::N + j] : 0.0; }
            s += Mo::l_disc(u, pv, save_t[m], m, d);
        }
    }
    part[i] = s;
}

// ---- DiscreteCallback affects of runtime models (hipadj_model_set_affect; src/callback_tracking.jl:232-470) --------------------------------
// (u_out[i], p_out[i]) = a(u[i], p, t): the affect applied to every trajectory's state (and, for affects that edit `pn`, its parameters) at an
// event time.  ldp = 0: shared parameters, NP: per trajectory; p_out is always per trajectory [N][NP].
template <class Mo>
__global__ void __launch_bounds__(256) k_user_affect(long N, long ldp, const double* __restrict__ u, const double* __restrict__ p, double t, double* __restrict__ out,
                                                     double* __restrict__ p_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double uu[Mo::N], pp[Mo::NP], un[Mo::N], pn[Mo::NP];
#pragma unroll
    for (int j = 0; j < Mo::N; ++j) uu[j] = u[i * Mo::N + j];
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) pp[j] = p[i * ldp + j];
    Mo::affect(un, pn, uu, pp, t);
#pragma unroll
    for (int j = 0; j < Mo::N; ++j) out[i * Mo::N + j] = un[j];
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) p_out[i * Mo::NP + j] = pn[j];
}
// the reverse callback at an event (:330-452), for the map (u, p) -> (un, pn) at the LEFT state:
//   lam_out = (dun/du)^T lam + (dpn/du)^T gp,   gp_out = (dun/dp)^T lam + (dpn/dp)^T gp
// gp [N][NP]: the gradient with respect to the parameters AFTER the event (of everything later in time); an affect that leaves p alone has
// dpn/dp = I, dpn/du = 0, i.e. gp_out = (dun/dp)^T lam + gp.
template <class Mo>
__global__ void __launch_bounds__(256) k_user_affect_vjp(long N, long ldp, const double* __restrict__ u, const double* __restrict__ p, double t, const double* __restrict__ lam,
                                                         const double* __restrict__ gp, double* __restrict__ lam_out, double* __restrict__ gp_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double uu[Mo::N], pp[Mo::NP], ll[Mo::N], gg[Mo::NP], lo[Mo::N], go[Mo::NP];
#pragma unroll
    for (int j = 0; j < Mo::N; ++j) { uu[j] = u[i * Mo::N + j]; ll[j] = lam[i * Mo::N + j]; }
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) { pp[j] = p[i * ldp + j]; gg[j] = gp[i * Mo::NP + j]; }
    Mo::affect_vjp(lo, go, ll, gg, uu, pp, t);
#pragma unroll
    for (int j = 0; j < Mo::N; ++j) lam_out[i * Mo::N + j] = lo[j];
#pragma unroll
    for (int j = 0; j < Mo::NP; ++j) gp_out[i * Mo::NP + j] = go[j];
}

}  // namespace hipadj
