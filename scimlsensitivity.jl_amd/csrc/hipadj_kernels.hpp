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
#include "hipadj_plan.hpp"

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

// ---- time-segmented kernels: segment maps + in-launch composition ---------------------------------------------
// Every (trajectory tile, segment) wave integrates its segment and publishes the affine map
//     segbuf[segment][column][N+NP][Npad]           (the top segment only fills column 0)
// The wave that draws the LAST arrival ticket of its trajectory tile composes the C maps top -> bottom
//     lam <- A lam + c_l ;  mu <- mu + B lam + c_m
// right there (no separate composition launch), scans for NaN/Inf (the reference's retcode check), writes du0 / the
// per-trajectory dp rows and a wave-level partial sum of mu; the tile that draws the last GLOBAL ticket sums the tile
// partials in tile order => dp is bit-reproducible for a given N whatever the arrival order.
// Hand-off form (cdna_hip_programming.md §6 G16, "8-byte agent-scope atomics on both sides"): the maps and partials
// are stored with relaxed agent-scope 8-byte atomic stores (write-through, sc1), every storing wave drains
// `s_waitcnt vmcnt(0)` before lane 0 takes its relaxed agent-scope ticket, and the composing wave reads with relaxed
// agent-scope atomic loads (bypass L1).  Placement-independent; counters are reset by the last arriver (zeroed at create).
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct SegEpilogue {
    double* segbuf;
    unsigned* tile_ctr;      // [ntiles]
    unsigned* global_ctr;    // [1]
    double* partial;         // [ntiles][NP]
    double* du0;             // [N][n]
    double* dp_rows;         // [N][np] or nullptr (shared p)
    double* dp_sum;          // [np] or nullptr
    int* flag;
};

template <class Mo, int NC>
__device__ __forceinline__ void segment_epilogue(const Geom& g, int nseg, const SegEpilogue& E, int seg, long i, bool valid,
                                                 const double (&lam_s)[NC][Mo::N], const double (&mu_s)[NC][Mo::NP]) {
    constexpr int N = Mo::N, NP = Mo::NP, NCM = 1 + N, R = N + NP;
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned tile = blockIdx.x, ntiles = gridDim.x;
    {   // publish this segment's map
        double* dst = E.segbuf + (long)seg * NCM * R * g.Npad + i;
        if (valid) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int j = 0; j < N; ++j) st_agent(dst + ((long)c * R + j) * g.Npad, lam_s[c][j]);
#pragma unroll
                for (int j = 0; j < NP; ++j) st_agent(dst + ((long)c * R + N + j) * g.Npad, mu_s[c][j]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(E.tile_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __shfl(t, 0, WAVE);
    if (t != (unsigned)nseg - 1u) return;
    // ---- last arriver of this trajectory tile: compose
    double lam[N], mu[NP];
    {
        const double* src = E.segbuf + (long)(nseg - 1) * NCM * R * g.Npad + i;
#pragma unroll
        for (int j = 0; j < N; ++j) lam[j] = ld_agent(src + (long)j * g.Npad);
#pragma unroll
        for (int j = 0; j < NP; ++j) mu[j] = ld_agent(src + (long)(N + j) * g.Npad);
    }
    constexpr int CH = 2;     // maps of CH segments in flight together, composed in order (kept small: registers)
    for (int sb = nseg - 2; sb >= 0; sb -= CH) {
        double m[CH][NCM * R];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int sq = sb - q > 0 ? sb - q : 0;
            const double* src = E.segbuf + (long)sq * NCM * R * g.Npad + i;
#pragma unroll
            for (int e = 0; e < NCM * R; ++e) m[q][e] = ld_agent(src + (long)e * g.Npad);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            if (sb - q >= 0) {
                double nl[N], nm[NP];
#pragma unroll
                for (int j = 0; j < N; ++j) nl[j] = m[q][j];
#pragma unroll
                for (int j = 0; j < NP; ++j) nm[j] = mu[j] + m[q][N + j];
#pragma unroll
                for (int c = 0; c < N; ++c) {
#pragma unroll
                    for (int j = 0; j < N; ++j) nl[j] += m[q][(c + 1) * R + j] * lam[c];
#pragma unroll
                    for (int j = 0; j < NP; ++j) nm[j] += m[q][(c + 1) * R + N + j] * lam[c];
                }
#pragma unroll
                for (int j = 0; j < N; ++j) lam[j] = nl[j];
#pragma unroll
                for (int j = 0; j < NP; ++j) mu[j] = nm[j];
            }
        }
    }
    bool bad = false;
    if (valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) { E.du0[i * N + j] = lam[j]; bad |= !(fabs(lam[j]) <= 1.79769313486231570e308); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { bad |= !(fabs(mu[j]) <= 1.79769313486231570e308); if (E.dp_rows) E.dp_rows[i * NP + j] = mu[j]; }
        if (bad) atomicOr(E.flag, 1);
    }
    if (lane == 0) __hip_atomic_store(E.tile_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!E.dp_sum) return;
    // ---- wave partial of mu (fixed shuffle tree), then the last tile sums the partials in tile order
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double v = valid ? mu[j] : 0.0;
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
        if (lane == 0) st_agent(E.partial + (long)tile * NP + j, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned tg = 0;
    if (lane == 0) tg = __hip_atomic_fetch_add(E.global_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tg = __shfl(tg, 0, WAVE);
    if (tg != ntiles - 1u) return;
    if (lane < NP) {
        double s = 0.0;
        for (unsigned b = 0; b < ntiles; ++b) s += ld_agent(E.partial + (long)b * NP + lane);
        E.dp_sum[lane] = s;
    }
    if (lane == 0) __hip_atomic_store(E.global_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// common prologue of the segmented kernels: lane -> (trajectory, segment); padded lanes redo the last trajectory
// (their stores are masked) so that the whole wave reaches the epilogue's shuffles
#define HIPADJ_SEG_PROLOGUE()                                                                          \
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N;                                                  \
    const long i_raw = (long)blockIdx.x * WAVE + threadIdx.x;                                          \
    const bool valid = i_raw < g.N;                                                                    \
    const long i = valid ? i_raw : g.N - 1;                                                            \
    const int seg = sp.nseg - 1 - (int)blockIdx.y;   /* longest (top, 1-column) segment dispatched first */ \
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1]

template <class Mo, int PF, int LOSS>
__global__ void __launch_bounds__(WAVE) k_interp(Geom g, SegPlan sp, const double* __restrict__ p,
                                                 const dbl2* __restrict__ knots, const double* __restrict__ cotT,
                                                 const int* __restrict__ save_of_knot, SegEpilogue E) {
    HIPADJ_SEG_PROLOGUE();
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, 1>(g, sp.nseg, E, seg, i, valid, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, NC>(g, sp.nseg, E, seg, i, valid, lam, mu);
    }
}

// checkpointing=true variants: checkpoint tiles in HBM, interval re-solve tile in LDS ([step][component][lane])
template <class Mo, int LOSS>
__global__ void __launch_bounds__(WAVE) k_interp_ckpt(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ ckpt,
                                                      const int* __restrict__ ckpt_of_knot, const int* __restrict__ prev_ck,
                                                      const double* __restrict__ cotT, const int* __restrict__ save_of_knot, SegEpilogue E) {
    constexpr int KM = HIPADJ_CKPT_KMAX;
    __shared__ double tile[(KM + 1) * Mo::N * WAVE];
    HIPADJ_SEG_PROLOGUE();
    const CkptSrc C{ckpt, ckpt_of_knot, prev_ck, tile, WAVE, (int)threadIdx.x};
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        segment_epilogue<Mo, 1>(g, sp.nseg, E, seg, i, valid, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        segment_epilogue<Mo, NC>(g, sp.nseg, E, seg, i, valid, lam, mu);
    }
}

template <class Mo, int PF, int LOSS>
__global__ void __launch_bounds__(WAVE) k_gauss(Geom g, SegPlan sp, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                const double* __restrict__ cotT, const int* __restrict__ save_of_knot, SegEpilogue E) {
    HIPADJ_SEG_PROLOGUE();
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        gauss_lane<Mo, 1, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, 1>(g, sp.nseg, E, seg, i, valid, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        gauss_lane<Mo, NC, PF, LOSS>(g, i, k_lo, k_hi, p, knots, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, NC>(g, sp.nseg, E, seg, i, valid, lam, mu);
    }
}

template <class Mo, int LOSS>
__global__ void __launch_bounds__(WAVE) k_gauss_ckpt(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ ckpt,
                                                     const int* __restrict__ ckpt_of_knot, const int* __restrict__ prev_ck,
                                                     const double* __restrict__ cotT, const int* __restrict__ save_of_knot, SegEpilogue E) {
    constexpr int KM = HIPADJ_CKPT_KMAX;
    __shared__ double tile[(KM + 1) * Mo::N * WAVE];
    HIPADJ_SEG_PROLOGUE();
    const CkptSrc C{ckpt, ckpt_of_knot, prev_ck, tile, WAVE, (int)threadIdx.x};
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        gauss_lane<Mo, 1, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        segment_epilogue<Mo, 1>(g, sp.nseg, E, seg, i, valid, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        gauss_lane<Mo, NC, 1, LOSS, KM>(g, i, k_lo, k_hi, p, nullptr, cotT, save_of_knot, lam, mu, &C);
        segment_epilogue<Mo, NC>(g, sp.nseg, E, seg, i, valid, lam, mu);
    }
}

// BacksolveAdjoint, segmented at checkpoint knots
template <class Mo>
__global__ void __launch_bounds__(WAVE) k_backsolve(Geom g, SegPlan sp, const double* __restrict__ p, const double* __restrict__ yT,
                                                    const double* __restrict__ ckpt, const int* __restrict__ ckpt_of_knot,
                                                    const double* __restrict__ cotT, const int* __restrict__ save_of_knot, SegEpilogue E) {
    HIPADJ_SEG_PROLOGUE();
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        backsolve_lane<Mo, 1>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, 1>(g, sp.nseg, E, seg, i, valid, lam, mu);
    } else {
        double lam[NC][N], mu[NC][NP];
        backsolve_lane<Mo, NC>(g, i, k_lo, k_hi, p, yT, ckpt, ckpt_of_knot, cotT, save_of_knot, lam, mu);
        segment_epilogue<Mo, NC>(g, sp.nseg, E, seg, i, valid, lam, mu);
    }
}

// ---- finishing stage -------------------------------------------------------------------------------------
// Per 256-thread workgroup: (optionally) compose the segment maps of each trajectory, scan for NaN/Inf (the
// reference's retcode check), write du0 / per-trajectory dp, and reduce mu over the workgroup's trajectories in a
// FIXED order (shuffle tree per wave, then waves 0..3) into partial[block][NP]; k_reduce_final then sums the
// partials in block order => dp is bit-reproducible for a given N.
constexpr int FIN = 256;

template <int NP>
__device__ __forceinline__ void block_partial(const double (&mu)[NP], bool valid, double* __restrict__ partial) {
    __shared__ double sh[FIN / WAVE][NP];
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
        for (int w = 0; w < FIN / WAVE; ++w) t += sh[w][threadIdx.x];
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
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)nblocks - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double s[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) s[j] = 0.0;
            for (int b = 0; b < nblocks; ++b) {
#pragma unroll
                for (int j = 0; j < NP; ++j) s[j] += partial[(long)b * NP + j];
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) dp[j] = s[j];
            __hip_atomic_store(ticket_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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

template <class Mo, int PF, int LOSS>
__global__ void __launch_bounds__(WAVE) k_quad_adj(Geom g, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                   const double* __restrict__ cotT, const int* __restrict__ save_of_knot,
                                                   dbl2* __restrict__ adj, double* __restrict__ du0) {
    constexpr int N = Mo::N;
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    if (i >= g.N) return;
    double lam[N];
    quad_adj_lane<Mo, PF, LOSS>(g, i, p, knots, cotT, save_of_knot, adj, lam);
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
    quad_gk_lane<Mo, 128>(g, i, p, knots, adj, qa[q], qb[q], atol, rtol, res);
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

}  // namespace hipadj
