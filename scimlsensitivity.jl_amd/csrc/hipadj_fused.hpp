// hipadj_fused.hpp — the reverse pass as ONE launch: the time-segmented sweep kernels finish the pass themselves.
//
// A segment wave leaves an affine map  (lam, mu) -> (A lam + c_l, mu + B lam + c_m)  of its time segment in registers
// (hipadj_lane.hpp, "TIME SEGMENTATION").  Until round 3 the maps went to HBM and two more kernels composed them per trajectory
// (k_compose_finish*) and summed dp (k_reduce_final): three dependent launches, whose fixed cost — 8.5 + 4.4 us of kernels and
// ~5 us of dispatch gaps — was 13 % of the 10^4-trajectory pass and half of a 1250-trajectory shard's.
//
// Here the waves of one trajectory block (64 trajectories x C segments = C waves, on C different CUs) compose their maps as a
// TREE through HBM — composition of affine maps is associative, so any bracketing gives the sequential result up to roundoff:
//   level 0: every wave publishes its map (write-through stores), then takes a ticket on its parent node (RADIX children);
//   the LAST arriver of a node loads the node's children, folds them in rank order (upper segment first) and carries the node's
//   map one level up; the wave that completes the root holds (lam(t0), mu(t0)) of its 64 trajectories: it writes du0, scans for
//   NaN/Inf, reduces mu over its lanes and takes a ticket on the ensemble; the last block sums the per-block partials in block order.
// The bracketing depends on (C, RADIX) only and every sum runs in a fixed order, never on arrival order: dp is bit-reproducible
// for a given N, as before.  A pass costs ceil(log_RADIX C) + 1 hand-offs (~2.5-3 us each on a draining chip) instead of two
// kernel boundaries + two kernels.
//
// Visibility across CUs / XCDs (cdna_hip_programming.md §6 Guideline 16, MI355X_MICROARCH.md "inter-workgroup visibility"):
// payload by 16-byte sc1 (write-through) buffer stores — no release fence needed — -> the storing wave drains vmcnt -> ONE lane's
// relaxed agent-scope fetch_add on the node's counter; the last arriver reads the children with 16-byte sc1 buffer loads (served
// below the CU's L1, which is never refreshed by other CUs' stores).  8-byte agent-scope atomics did the same at 2.7x the fabric
// writes per byte (first form of this file: +6.5 us of kernel time at 10^4 trajectories, +10-20 us on the shards).  Every hand-off here is
// wave -> wave (64-thread workgroups), so no workgroup barrier is involved.  Counters are zeroed when the handle is created and
// before every forward solve, and each is reset by its own last arriver.
#pragma once

#include "hipadj_lane.hpp"

namespace hipadj {

constexpr int HIPADJ_TREE_MAXLEV = 8;

struct TreePlan {
    int radix;                              // children per node (4: one batch of loads per level; 8 / 16: two / four batches, fewer levels)
    int nlev;                               // levels above the leaves; count[0] = C (segments), count[nlev] = 1
    int count[HIPADJ_TREE_MAXLEV + 1];      // nodes per trajectory block on level l
    long map_off[HIPADJ_TREE_MAXLEV + 1];   // first map slot of level l in tbuf (slots of one level: [block][node])
    int cnt_off[HIPADJ_TREE_MAXLEV + 1];    // first arrival counter of level l (l >= 1) in cnt ([block][node])
    double* tbuf;                           // map slots: ceil(MAPSZ / 2) rows of 64 x 16 bytes each (entries 2r, 2r + 1 of lane l in row r)
    long tbuf_bytes;                        // < 2^31 (buffer descriptor range; larger trees fall back to the three-launch sequence)
    unsigned* cnt;
    double* partial;                        // [blocks][NP] per-block sums of mu
    unsigned* ticket;                       // ensemble ticket (one word)
#ifdef HIPADJ_WAVE_TRACE
    unsigned long long* trace;              // development builds only (see HIPADJ_TP, hipadj_lane.hpp)
#endif
};
#ifdef HIPADJ_WAVE_TRACE
#define HIPADJ_TTRACE(T) ((T).trace)
#else
#define HIPADJ_TTRACE(T) ((unsigned long long*)nullptr)
#endif

// Host side: the shape of the tree for C leaves (hipadj_plan.hpp / hipadj_api.hip allocate by it).
inline void tree_plan_shape(int C, int radix, long blocks, TreePlan& T, long* map_slots, long* counters) {
    T.radix = radix; T.nlev = 0; T.count[0] = C; T.map_off[0] = 0; T.cnt_off[0] = 0;
    long slots = (long)C * blocks, ctr = 0;
    while (T.count[T.nlev] > 1) {
        const int l = T.nlev + 1;
        T.count[l] = (T.count[l - 1] + radix - 1) / radix;
        T.map_off[l] = slots; T.cnt_off[l] = (int)ctr;
        slots += (long)T.count[l] * blocks; ctr += (long)T.count[l] * blocks;
        T.nlev = l;
    }
    if (map_slots) *map_slots = slots;
    if (counters) *counters = ctr > 0 ? ctr : 1;
}

// O = L o U for segment maps stored as m[c * R + j]: column c = 0 the affine column (c_l, c_m), columns 1..N the images of the
// basis vectors (A, B); rows j < N lambda, j >= N mu.  U acts first (the upper time segment), L second.
//   lam: A_O = A_L A_U, c_O = A_L c_U + c_L;   mu: B_O = B_U + B_L A_U, cm_O = cm_U + B_L c_U + cm_L.
template <int N, int NP>
HIPADJ_HD void map_compose(const double (&U)[(1 + N) * (N + NP)], const double (&L)[(1 + N) * (N + NP)], double (&O)[(1 + N) * (N + NP)]) {
    constexpr int R = N + NP;
#pragma unroll
    for (int c = 0; c <= N; ++c) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double v = c == 0 ? L[j] : 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) v += L[(k + 1) * R + j] * U[c * R + k];
            O[c * R + j] = v;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            double v = U[c * R + N + j] + (c == 0 ? L[N + j] : 0.0);
#pragma unroll
            for (int k = 0; k < N; ++k) v += L[(k + 1) * R + N + j] * U[c * R + k];
            O[c * R + N + j] = v;
        }
    }
}

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)

__device__ __forceinline__ void map_store_agent(double* __restrict__ dst, double v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double map_load_agent(const double* __restrict__ src) { return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte sc1 accesses through a buffer descriptor over the whole slot array (compiler-tracked vmcnt, unlike inline asm): voffset = the
// lane's 16 bytes inside a 1 KB row, soffset = the row (wave-uniform).  aux 16 = sc1.
typedef unsigned int hipadj_u4 __attribute__((ext_vector_type(4)));
template <class RS>
__device__ __forceinline__ void pair_store_sc1(RS rs, int voff, int soff, double a, double b) {
    hipadj_u4 v; v.x = __double2loint(a); v.y = __double2hiint(a); v.z = __double2loint(b); v.w = __double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 16);
}
template <class RS>
__device__ __forceinline__ void pair_load_sc1(RS rs, int voff, int soff, double& a, double& b) {
    const hipadj_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
    a = __hiloint2double((int)v.y, (int)v.x); b = __hiloint2double((int)v.w, (int)v.z);
}

#ifndef HIPADJ_TREE_ACQREL
#define HIPADJ_TREE_ACQREL 0      // 1: the conforming acquire-release ticket (A/B builds: scripts/r5/ab_variants.sh measures its cost)
#endif
// one wave: ticket on `ctr`; true on every lane iff this wave is the last of `expected` arrivers (then the counter is reset).
// The caller has issued its payload stores; they are drained here before the ticket is drawn.
__device__ __forceinline__ bool tree_arrive_last(unsigned* __restrict__ ctr, unsigned expected, unsigned long long* tr = nullptr, int slot = 0) {
    (void)tr; (void)slot;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through payload stores have left the CU
    HIPADJ_TP(tr, slot, 0);                                     // stores drained
    unsigned old = 0;
#if HIPADJ_TREE_ACQREL
    // the language-level form: an agent-scope acquire-release RMW — the compiler adds the L2 write-back in front and the L1 invalidate behind it
    if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
    if ((threadIdx.x & 63) == 0) old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    // compiler-level ordering: the last arriver's sc1 loads of the children's payloads sit behind the control dependency on `old` AND behind this
    // barrier, so no compiler version may hoist them above the ticket; the hardware side is the guide's recipe (16-byte sc1 stores drained before the ticket,
    // sc1 loads after it: cdna_hip_programming.md section 6 G16).  Runtime models under an UNTRUSTED hiprtc keep the three-launch pass (hipadj_api.hip).
    asm volatile("" ::: "memory");
    HIPADJ_TP(tr, slot + 1, old);                               // ticket returned
    if (old != expected - 1u) return false;
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next pass
    return true;
}

// The root of a trajectory block: v = (lam(t0)[N], mu(t0)[NP]) of the wave's 64 trajectories.  Writes du0 (caller layout) and the dp rows, scans for
// NaN / Inf, and — shared parameters — sums mu over the lanes and, as the last block of the ensemble, over the blocks (fixed orders).
// One-segment kernels (SEG = false, or C = 1) call this directly; segmented ones reach it through fused_tail.
template <int N, int NP>
__device__ __forceinline__ void fused_root(const double (&m)[N + NP], const TreePlan& T, long ntraj, long blocks, long block,
                                           double* __restrict__ du0, double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const long i = block * 64 + lane;
    const bool valid = i < ntraj;
    bool bad = false;
    if (valid) {
#pragma unroll
        for (int j = 0; j < N; ++j) { du0[i * N + j] = m[j]; bad |= !(fabs(m[j]) <= 1.79769313486231570e308); }
#pragma unroll
        for (int j = 0; j < NP; ++j) { bad |= !(fabs(m[N + j]) <= 1.79769313486231570e308); if (dp_rows) dp_rows[i * NP + j] = m[N + j]; }
        if (bad) atomicOr(flag, 1);
    }
    if (!dp_sum) return;
    double s[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double v = valid ? m[N + j] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);   // fixed tree over the lanes
        s[j] = v;
    }
    if (blocks == 1) {
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NP; ++j) dp_sum[j] = s[j];
        }
        return;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) map_store_agent(T.partial + block * NP + j, s[j]);
    }
    HIPADJ_TP(HIPADJ_TTRACE(T), 20, s[0]);                      // du0 written, mu reduced over the lanes, the block's partial issued
    if (!tree_arrive_last(T.ticket, (unsigned)blocks, HIPADJ_TTRACE(T), 21)) return;
    // last block: lane l sums blocks l, l + 64, ... in increasing order, then the same fixed tree over the lanes.  The loads of Q rounds (64 Q blocks) x NP entries are
    // issued together: one memory round trip instead of a dependent chain of ceil(blocks / 64) x NP (round 6: 2.8-3.4 us of the 10^4-trajectory pass' tail, 157 blocks,
    // profiles/r6_wave_trace_10000*.jsonl "root: ticket->dp written")
    constexpr int Q = NP <= 4 ? 4 : (NP <= 8 ? 2 : 1);      // rounds in flight: Q x NP registers (runtime models carry up to 32 parameters: one round, the old order)
    double acc[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = 0.0;
    for (long b0 = lane; b0 < blocks; b0 += 64 * Q) {
        double x[Q][NP];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const long b = b0 + 64 * q;
#pragma unroll
            for (int j = 0; j < NP; ++j) x[q][j] = b < blocks ? map_load_agent(T.partial + b * NP + j) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[j] += x[q][j];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double v = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) dp_sum[j] = v;
        if (j == NP - 1) HIPADJ_TP(HIPADJ_TTRACE(T), 23, v);   // the ensemble's dp summed
    }
}


// The tail of a segment wave.  m: this wave's map (NCOL = 1 + N columns; a top-segment / single-segment wave passes its vector in
// column 0 and zeros elsewhere).  block: trajectory block (64 trajectories), rank: 0 = top segment ... C-1 = the segment at t0.
// du0 [N_traj][N] (caller layout), dp_rows [N_traj][NP] or null, dp_sum [NP] or null (shared parameters), flag: non-finite marker.
template <int N, int NP>
__device__ __forceinline__ void fused_tail(double (&m)[(1 + N) * (N + NP)], const TreePlan& T, long ntraj, long blocks, long block, int rank,
                                           double* __restrict__ du0, double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int R = N + NP, MAPSZ = (1 + N) * R, MAPP = (MAPSZ + 1) / 2;      // a slot: MAPP rows of 64 x 16 bytes (entries 2r, 2r + 1 of the lane)
    constexpr int SLOTB = MAPP * 1024;
    const int lane = threadIdx.x & 63, RADIX = T.radix, voff = lane * 16;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(T.tbuf, 0, (int)T.tbuf_bytes, 0x00020000);
    int idx = rank;
    for (int l = 0; l < T.nlev; ++l) {
        const int count = T.count[l], parent = idx / RADIX;
        const int first = parent * RADIX, nchild = (count - first) < RADIX ? (count - first) : RADIX;
        if (nchild > 1) {
            const int slot0 = (int)(T.map_off[l] + block * count) + first;       // the node's first child
            const int so = (slot0 + (idx - first)) * SLOTB;
#pragma unroll
            for (int r = 0; r < MAPP; ++r) pair_store_sc1(rs, voff, so + r * 1024, m[2 * r], 2 * r + 1 < MAPSZ ? m[2 * r + 1] : 0.0);
            HIPADJ_TP(HIPADJ_TTRACE(T), 4 + 4 * l, 0);            // level l: payload stores issued
            if (!tree_arrive_last(T.cnt + T.cnt_off[l + 1] + block * T.count[l + 1] + parent, (unsigned)nchild, HIPADJ_TTRACE(T), 5 + 4 * l)) return;
            // last arriver of the node: fold its children, upper segment (lower rank) first; children in batches of four
            // (4 x MAPSZ doubles in flight fit the register file next to m; RADIX = 8 takes two batches)
            for (int c0 = 0; c0 < nchild; c0 += 4) {
                double ch[4][2 * MAPP];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cc = c0 + c < nchild ? c0 + c : nchild - 1;        // clamped: uniform addresses, all loads issued before the first use
#pragma unroll
                    for (int r = 0; r < MAPP; ++r) pair_load_sc1(rs, voff, (slot0 + cc) * SLOTB + r * 1024, ch[c][2 * r], ch[c][2 * r + 1]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c0 + c == 0) {
#pragma unroll
                        for (int e = 0; e < MAPSZ; ++e) m[e] = ch[0][e];
                    } else if (c0 + c < nchild) {
                        double lo_[MAPSZ], o[MAPSZ];
#pragma unroll
                        for (int e = 0; e < MAPSZ; ++e) lo_[e] = ch[c][e];
                        map_compose<N, NP>(m, lo_, o);
#pragma unroll
                        for (int e = 0; e < MAPSZ; ++e) m[e] = o[e];
                    }
                }
            }
            HIPADJ_TP(HIPADJ_TTRACE(T), 7 + 4 * l, m[0]);         // level l: children loaded and folded
        }
        idx = parent;
    }
    // root: column 0 holds (lam(t0), mu(t0)) of the block's trajectories
    double v[N + NP];
#pragma unroll
    for (int j = 0; j < N + NP; ++j) v[j] = m[j];
    fused_root<N, NP>(v, T, ntraj, blocks, block, du0, dp_rows, dp_sum, flag);
}

// GROUPED form (round 6): G consecutive segments of one trajectory block share a WORKGROUP of G waves, and the first level of the composition happens in LDS instead of HBM —
// every wave but the first publishes its map into the workgroup's LDS block ([wave][entry][lane]: 512-byte rows, conflict-free), one barrier, the first wave folds them in rank
// order and carries the group's map into the tree of hipadj_fused.hpp as leaf `group` of ceil(C / G) leaves.  The wave timeline of a 1250-trajectory shard
// (profiles/r6_wave_trace_1250_*.jsonl) prices a level of the HBM tree at 3.7-4.9 us (payload stores 0.6, drain 0.45, ticket 0.45, sibling loads + fold 2-3.2) against
// ~12.7 us of sweep: three levels + the root are as long as the sweep itself.  `nvalid`: waves of this group that carry a segment (the last group of a block may be short).
template <int N, int NP, int G>
__device__ __forceinline__ void fused_tail_group(double (&m)[(1 + N) * (N + NP)], double* __restrict__ lds, const TreePlan& T, long ntraj, long blocks, long block, int group,
                                                 int wv, int nvalid, double* __restrict__ du0, double* __restrict__ dp_rows, double* __restrict__ dp_sum, int* __restrict__ flag) {
    constexpr int MAPSZ = (1 + N) * (N + NP);
    const int lane = threadIdx.x & 63;
    if (wv > 0 && wv < nvalid) {
#pragma unroll
        for (int e = 0; e < MAPSZ; ++e) lds[((wv - 1) * MAPSZ + e) * 64 + lane] = m[e];
    }
    __syncthreads();
    if (wv != 0) return;
    for (int c = 1; c < nvalid; ++c) {      // upper segment (lower rank) first, like the tree's fold
        double lo_[MAPSZ], o[MAPSZ];
#pragma unroll
        for (int e = 0; e < MAPSZ; ++e) lo_[e] = lds[((c - 1) * MAPSZ + e) * 64 + lane];
        map_compose<N, NP>(m, lo_, o);
#pragma unroll
        for (int e = 0; e < MAPSZ; ++e) m[e] = o[e];
    }
    fused_tail<N, NP>(m, T, ntraj, blocks, block, group, du0, dp_rows, dp_sum, flag);
}

#endif  // device code

}  // namespace hipadj
