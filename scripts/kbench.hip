// kbench.hip — standalone micro-benchmark and wave tracer of the headline kernel (k_interp<ModelLorenz, PF, LSQ>) on the
// BASELINE configs[1] workload.  Development tooling, not product: it includes the library's own kernel headers, builds its own
// plan with make_plan, and launches the same kernels the library would (forward solve, segmented sweep, composition, reduction).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scimlsensitivity.jl_amd/csrc scripts/kbench.hip -o scripts/kbench   [-DKB_NO_OPS] [-DKB_PF=8]
//   scripts/kbench [ntraj=10000] [segments=0 (auto)] [reps=40] [mode: bench | trace] [wtop]
//
// bench: back-to-back reverse passes (sweep + composition + reduction) timed with HIP events: per-kernel and whole-pass times,
//        dp printed for comparison with the library / oracle (-456767.77032422, 27053040.0294687, -3421830.96163125 at 10^4).
// trace: one launch of a copy of the sweep kernel that records, per wave, s_memtime at entry / exit, the 100 MHz real-time counter,
//        HW_ID and XCC_ID; prints the distribution of wave lifetimes, start skew and waves per SIMD — the placement picture behind
//        "SQ_WAVE_CYCLES / SQ_WAVES = 67 % of the kernel duration" (profiles/r1_rocprofv3_pmc_sq.txt).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#ifdef KB_NO_OPS
#define HIPADJ_DISABLE_OPS 1
#endif
#include "hipadj_kernels.hpp"

using namespace hipadj;
#ifndef KB_PF
#define KB_PF 8
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using Mo = ModelLorenz;

template <int PF>
__global__ void __launch_bounds__(WAVE) k_interp_traced(Geom g, SegPlan sp, const double* __restrict__ p, const dbl2* __restrict__ knots,
                                                        const int* __restrict__ save_of_knot, double* __restrict__ segbuf, unsigned long long* __restrict__ trace) {
    constexpr int N = Mo::N, NP = Mo::NP, NC = 1 + N, R = N + NP;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    const long i = (long)blockIdx.x * WAVE + threadIdx.x;
    const int seg = sp.nseg - 1 - (int)blockIdx.y;
    if (i >= g.N) return;
    const int k_lo = sp.bounds[seg], k_hi = sp.bounds[seg + 1];
    double* __restrict__ dst = segbuf + (long)seg * NC * R * g.Npad + i;
    if (seg == sp.nseg - 1) {
        double lam[1][N], mu[1][NP];
        interp_lane<Mo, 1, PF, 1>(g, i, k_lo, k_hi, p, knots, nullptr, save_of_knot, lam, mu);
#pragma unroll
        for (int j = 0; j < N; ++j) dst[(long)j * g.Npad] = lam[0][j];
#pragma unroll
        for (int j = 0; j < NP; ++j) dst[(long)(N + j) * g.Npad] = mu[0][j];
    } else {
        double lam[NC][N], mu[NC][NP];
        interp_lane<Mo, NC, PF, 1, 0, true>(g, i, k_lo, k_hi, p, knots, nullptr, save_of_knot, lam, mu);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < N; ++j) dst[((long)c * R + j) * g.Npad] = lam[c][j];
#pragma unroll
            for (int j = 0; j < NP; ++j) dst[((long)c * R + N + j) * g.Npad] = mu[c][j];
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* t = trace + 6 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
        t[0] = c0; t[1] = c1; t[2] = r0; t[3] = r1; t[4] = ((unsigned long long)xcc << 32) | hwid; t[5] = (unsigned long long)seg;
    }
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 10000;
    const int segs = argc > 2 ? atoi(argv[2]) : 0;
    const int reps = argc > 3 ? atoi(argv[3]) : 40;
    const std::string mode = argc > 4 ? argv[4] : "bench";
    if (argc > 5) setenv("HIPADJ_WTOP", argv[5], 1);
    const double T = 10.0, dt = 0.01;
    std::vector<double> ts(101);
    for (int i = 0; i <= 100; ++i) ts[i] = 0.1 * i;
    ts[100] = T;
    hipadj_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.model = HIPADJ_MODEL_LORENZ; cfg.alg = HIPADJ_ALG_INTERPOLATING; cfg.stepper = HIPADJ_STEPPER_RK4_FIXED;
    cfg.ntraj = N; cfg.t0 = 0.0; cfg.t1 = T; cfg.dt = dt; cfg.nsave = 101; cfg.save_times = ts.data();
    cfg.loss_kind = HIPADJ_LOSS_LSQ_SHIFT; cfg.loss_shift = 2.0; cfg.p_shared = 1; cfg.time_segments = segs;
    Plan P; std::string err;
    if (make_plan(&cfg, P, err) != HIPADJ_OK) { fprintf(stderr, "plan: %s\n", err.c_str()); return 1; }
    const int n = 3, np = 3, S = P.S, C = P.nseg; const long Np = P.Npad;
    Geom g{N, Np, S, P.M, 0.0, dt, 2.0, 1, 0, 1, -1, dt};
    // inputs: a fixed seed (not numpy's stream: dp differs from bench.py's unless N matches and u0 is loaded from a file)
    std::vector<double> u0((size_t)N * 3);
    { FILE* f = fopen("/tmp/kbench_u0.bin", "rb");
      if (f && fread(u0.data(), 8, u0.size(), f) == u0.size()) { fclose(f); printf("u0 from /tmp/kbench_u0.bin (bench.py's ensemble)\n"); }
      else { if (f) fclose(f); std::mt19937_64 rng(20240601); std::normal_distribution<double> nd(0.0, 1.0);
             for (long i = 0; i < N; ++i) { u0[3 * i] = 1.0 + 0.1 * nd(rng); u0[3 * i + 1] = 0.1 * nd(rng); u0[3 * i + 2] = 0.1 * nd(rng); } } }
    const double ph[3] = {10.0, 28.0, 8.0 / 3.0};
    double *d_u0, *d_p, *d_yT = nullptr, *d_segbuf, *d_du0, *d_dp, *d_partial; dbl2* d_knots; int *d_save, *d_bounds, *d_flag; unsigned* d_ticket;
    CK(hipMalloc(&d_u0, u0.size() * 8)); CK(hipMalloc(&d_p, 24)); CK(hipMalloc(&d_knots, (size_t)(S + 1) * n * Np * sizeof(dbl2)));
    CK(hipMalloc(&d_segbuf, (size_t)C * (1 + n) * (n + np) * Np * 8)); CK(hipMalloc(&d_du0, (size_t)N * 3 * 8)); CK(hipMalloc(&d_dp, 24));
    CK(hipMalloc(&d_partial, (size_t)((N + 15) / 16) * np * 8)); CK(hipMalloc(&d_save, (S + 1) * 4)); CK(hipMalloc(&d_bounds, (C + 1) * 4));
    CK(hipMalloc(&d_flag, 4)); CK(hipMalloc(&d_ticket, 4)); CK(hipMemset(d_flag, 0, 4)); CK(hipMemset(d_ticket, 0, 4));
    CK(hipMemcpy(d_u0, u0.data(), u0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_p, ph, 24, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_save, P.save_of_knot_rev.data(), (S + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bounds, P.seg_bounds.data(), (C + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_knots, 0, (size_t)(S + 1) * n * Np * sizeof(dbl2)));
    hipStream_t st; CK(hipStreamCreate(&st));
    const unsigned waves = (unsigned)(Np / WAVE);
    hipLaunchKernelGGL((k_forward<Mo>), dim3(waves), dim3(WAVE), 0, st, g, d_u0, d_p, d_knots, (double*)nullptr, (const int*)nullptr, (double*)nullptr, (const int*)d_save, d_yT);
    CK(hipStreamSynchronize(st));
    SegPlan sp{C, d_bounds};
    printf("N=%ld Npad=%ld S=%d segments=%d  bounds:", N, Np, S, C);
    for (int s = 0; s <= C; ++s) printf(" %d", P.seg_bounds[s]);
    printf("\n");
    const unsigned compose_blocks = (unsigned)((N + 15) / 16);
    auto pass = [&](hipEvent_t k0, hipEvent_t k1) {
        if (C > 1) {
            hipExtLaunchKernelGGL((k_interp<Mo, KB_PF, 1, true, 1, true>), dim3(waves, (unsigned)C), dim3(WAVE), 0, st, k0, k1, 0, g, sp, (const double*)d_p, (const dbl2*)d_knots, (const double*)nullptr, (const int*)d_save, d_segbuf);
        } else {
            hipExtLaunchKernelGGL((k_interp<Mo, KB_PF, 1, true, 1, true>), dim3(waves, 1u), dim3(WAVE), 0, st, k0, k1, 0, g, sp, (const double*)d_p, (const dbl2*)d_knots, (const double*)nullptr, (const int*)d_save, d_segbuf);
        }
        if (getenv("KB_FUSED_FINAL")) {
            hipLaunchKernelGGL((k_compose_finish<Mo, 64>), dim3(compose_blocks), dim3(64), 0, st, g, C, (const double*)d_segbuf, d_du0, (double*)nullptr, d_partial, d_flag, d_ticket, d_dp);
        } else {
            hipLaunchKernelGGL((k_compose_finish<Mo, 64>), dim3(compose_blocks), dim3(64), 0, st, g, C, (const double*)d_segbuf, d_du0, (double*)nullptr, d_partial, d_flag, d_ticket, (double*)nullptr);
            hipLaunchKernelGGL(k_reduce_final, dim3(3u), dim3(FIN), 0, st, (int)compose_blocks, 3, (const double*)d_partial, d_dp);
        }
    };
    if (mode == "graph") {     // the three launches of a pass captured once and replayed
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        pass(nullptr, nullptr);
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        CK(hipEventRecord(a0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(a1, st));
        CK(hipStreamSynchronize(st));
        float tot; CK(hipEventElapsedTime(&tot, a0, a1));
        double dp[3]; CK(hipMemcpy(dp, d_dp, 24, hipMemcpyDeviceToHost));
        printf("{\"mode\": \"graph\", \"fused_final\": %d, \"ntraj\": %ld, \"segments\": %d, \"pass_ms\": %.5f, \"dp\": [%.17g, %.17g, %.17g]}\n", getenv("KB_FUSED_FINAL") ? 1 : 0, N, C, tot / reps, dp[0], dp[1], dp[2]);
        return 0;
    }
    if (mode == "bench") {
        std::vector<hipEvent_t> e0(reps), e1(reps);
        for (int r = 0; r < reps; ++r) { CK(hipEventCreate(&e0[r])); CK(hipEventCreate(&e1[r])); }
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        for (int r = 0; r < 5; ++r) pass(nullptr, nullptr);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a0, st));
        for (int r = 0; r < reps; ++r) pass(e0[r], e1[r]);
        CK(hipEventRecord(a1, st));
        CK(hipStreamSynchronize(st));
        float tot; CK(hipEventElapsedTime(&tot, a0, a1));
        std::vector<float> km(reps);
        for (int r = 0; r < reps; ++r) CK(hipEventElapsedTime(&km[r], e0[r], e1[r]));
        { double f5 = 0, l5 = 0; for (int r = 0; r < 5 && r < reps; ++r) { f5 += km[r]; l5 += km[reps - 1 - r]; }
          printf("k_interp ms in launch order: first5 mean %.5f last5 mean %.5f |", f5 / 5, l5 / 5); for (int r = 0; r < reps && r < 12; ++r) printf(" %.4f", km[r]); printf("\n"); }
        std::sort(km.begin(), km.end());
        double mean = 0; for (float v : km) mean += v; mean /= reps;
        double dp[3]; CK(hipMemcpy(dp, d_dp, 24, hipMemcpyDeviceToHost));
        const double bytes = (double)N * (S + 1) * 16.0 * n + (double)N * 8.0 * (n + np);
        printf("{\"ntraj\": %ld, \"segments\": %d, \"pf\": %d, \"ops\": %d, \"pass_ms\": %.5f, \"k_interp_ms_mean\": %.5f, \"k_interp_ms_min\": %.5f, \"k_interp_ms_med\": %.5f, "
               "\"frac_hbm_mean\": %.4f, \"whole_pass_frac\": %.4f, \"dp\": [%.17g, %.17g, %.17g]}\n",
               N, C, KB_PF, (int)model_has_ops<Mo>::value, tot / reps, mean, km[0], km[reps / 2], bytes / (mean * 1e-3) / 8e12, bytes / (tot / reps * 1e-3) / 8e12, dp[0], dp[1], dp[2]);
        return 0;
    }
    // ---- trace
    unsigned long long* d_tr; const size_t nw = (size_t)waves * C;
    CK(hipMalloc(&d_tr, nw * 6 * 8)); CK(hipMemset(d_tr, 0, nw * 6 * 8));
    for (int r = 0; r < 3; ++r) pass(nullptr, nullptr);
    CK(hipStreamSynchronize(st));
    hipEvent_t k0, k1; CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    hipExtLaunchKernelGGL((k_interp_traced<KB_PF>), dim3(waves, (unsigned)C), dim3(WAVE), 0, st, k0, k1, 0, g, sp, (const double*)d_p, (const dbl2*)d_knots, (const int*)d_save, d_segbuf, d_tr);
    CK(hipStreamSynchronize(st));
    float kms; CK(hipEventElapsedTime(&kms, k0, k1));
    std::vector<unsigned long long> tr(nw * 6); CK(hipMemcpy(tr.data(), d_tr, nw * 6 * 8, hipMemcpyDeviceToHost));
    unsigned long long cmin = ~0ull, cmax = 0, rmin = ~0ull, rmax = 0;
    for (size_t w = 0; w < nw; ++w) { cmin = std::min(cmin, tr[6 * w]); cmax = std::max(cmax, tr[6 * w + 1]); rmin = std::min(rmin, tr[6 * w + 2]); rmax = std::max(rmax, tr[6 * w + 3]); }
    printf("traced kernel %.4f ms; span: %llu shader-clock ticks, %llu real-time ticks (100 MHz => %.4f ms) => shader clock %.3f GHz\n", kms, cmax - cmin, rmax - rmin,
           (rmax - rmin) / 1e5, (cmax - cmin) / ((rmax - rmin) * 10.0));
    // per-SIMD occupancy and lifetimes
    std::map<unsigned long long, std::vector<size_t>> simd;
    std::vector<double> life(nw), start(nw);
    for (size_t w = 0; w < nw; ++w) {
        const unsigned hw = (unsigned)(tr[6 * w + 4] & 0xffffffffu), xcc = (unsigned)(tr[6 * w + 4] >> 32) & 0xf;
        const unsigned simd_id = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        simd[((unsigned long long)xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | simd_id].push_back(w);
        life[w] = (double)(tr[6 * w + 3] - tr[6 * w + 2]) * 10.0;  // ns
        start[w] = (double)(tr[6 * w + 2] - rmin) * 10.0;
    }
    std::map<size_t, int> hist; double busy_max = 0;
    for (auto& kv : simd) { hist[kv.second.size()]++; double last = 0; for (size_t w : kv.second) last = std::max(last, start[w] + life[w]); busy_max = std::max(busy_max, last); }
    printf("SIMDs used: %zu; waves-per-SIMD histogram:", simd.size());
    for (auto& kv : hist) printf("  %zu waves: %d SIMDs", kv.first, kv.second);
    printf("\n");
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("wave lifetime us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f | start skew us: med %.1f p90 %.1f max %.1f | last wave ends at %.1f us\n",
           pct(life, 0) / 1e3, pct(life, .1) / 1e3, pct(life, .5) / 1e3, pct(life, .9) / 1e3, pct(life, 1) / 1e3, pct(start, .5) / 1e3, pct(start, .9) / 1e3, pct(start, 1) / 1e3, busy_max / 1e3);
    if (const char* dump = getenv("KB_TRACE_DUMP")) {   // raw per-wave records for offline analysis: seg wave_block start_ns life_ns xcc se sh cu simd
        FILE* f = fopen(dump, "w");
        for (size_t w = 0; w < nw; ++w) {
            const unsigned hw = (unsigned)(tr[6 * w + 4] & 0xffffffffu), xcc = (unsigned)(tr[6 * w + 4] >> 32) & 0xf;
            fprintf(f, "%d %zu %.0f %.0f %u %u %u %u %u\n", (int)tr[6 * w + 5], w % waves, start[w], life[w], xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3);
        }
        fclose(f);
    }
    // lifetime by segment kind and by co-residency
    std::vector<double> ltop, llow;
    for (size_t w = 0; w < nw; ++w) ((int)tr[6 * w + 5] == C - 1 ? ltop : llow).push_back(life[w]);
    if (!ltop.empty() && !llow.empty())
        printf("top-segment waves (1 column): med %.1f us max %.1f | lower-segment waves (4 columns): med %.1f us max %.1f\n", pct(ltop, .5) / 1e3, pct(ltop, 1) / 1e3, pct(llow, .5) / 1e3, pct(llow, 1) / 1e3);
    for (auto& hk : hist) {
        std::vector<double> l, e;
        for (auto& kv : simd) if (kv.second.size() == hk.first) for (size_t w : kv.second) { l.push_back(life[w]); e.push_back(start[w] + life[w]); }
        printf("  SIMDs with %zu waves: wave lifetime med %.1f us, end time med %.1f max %.1f us\n", hk.first, pct(l, .5) / 1e3, pct(e, .5) / 1e3, pct(e, 1) / 1e3);
    }
    return 0;
}
