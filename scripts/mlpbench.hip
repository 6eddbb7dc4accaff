// mlpbench.hip — standalone micro-benchmark of the FP64-MFMA kernel family (BASELINE configs[3]: tanh MLP 2 -> 128 -> 128 -> 2, 4096 batch
// columns, GaussAdjoint, RK4 dt = 0.01, 150 steps).  Development tooling, not product: includes the library's own headers and launches
// the same kernels (forward solve, reverse sweep with the in-register parameter gradient, fixed-order partial sum) so that kernel variants
// can be A/B-compared by -D flags without rebuilding the library.  (profiles/r2_mlpbench_*.log were produced by earlier revisions of this
// file, which also drove the retired record path.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scimlsensitivity.jl_amd/csrc scripts/mlpbench.hip -o scripts/kbench_mlp [-DHIPADJ_MLPG_...]
//   scripts/kbench_mlp [B=4096] [S=150] [reps=3] [alg: 2 = gauss | 0 = interpolating | 1 = backsolve]
// Prints one JSON line: forward / sweep ms and checksums of du0, dp and the knots (to compare variants).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "hipadj_mlp_grad.hpp"

using namespace hipadj;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#ifndef MB_H
#define MB_H 128
#endif
constexpr int H = MB_H;

__global__ void k_checksum(const double* __restrict__ a, long n, double* out) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i] * (1.0 + 1e-3 * (double)(i % 97));
    atomicAdd(out, s);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, S = argc > 2 ? atoi(argv[2]) : 150, reps = argc > 3 ? atoi(argv[3]) : 3, alg = argc > 4 ? atoi(argv[4]) : 2;
    constexpr int D = 2, NPAR = Mlp<H>::NPAR;
    const int M = S / 5;
    std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> p(NPAR), u0((size_t)D * B), cot((size_t)M * D * B);
    { size_t o = 0;
      for (int i = 0; i < H * D; ++i) p[o++] = nd(rng) / sqrt((double)D);
      for (int i = 0; i < H; ++i) p[o++] = 0.1 * nd(rng);
      for (int i = 0; i < H * H; ++i) p[o++] = nd(rng) / sqrt((double)H);
      for (int i = 0; i < H; ++i) p[o++] = 0.1 * nd(rng);
      for (int i = 0; i < D * H; ++i) p[o++] = nd(rng) / sqrt((double)H);
      for (int i = 0; i < D; ++i) p[o++] = 0.1 * nd(rng); }
    for (auto& v : u0) v = nd(rng);
    for (auto& v : cot) v = nd(rng);
    std::vector<int> save(S + 1, -1);
    for (int k = 5; k <= S; k += 5) save[k] = k / 5 - 1;
    MlpGeom g{}; g.N = 1; g.B = B; g.S = S; g.M = M; g.t0 = 0.0; g.dt = 0.01; g.loss_shift = 0.0; g.loss_kind = 0; g.no_start = 0; g.p_shared = 1; g.NQ = 0;
    double *d_p, *d_u0, *d_cot, *d_knots, *d_du0, *d_sum, *d_part, *d_dp; int *d_save, *d_flag;
    CK(hipMalloc(&d_p, sizeof(double) * NPAR)); CK(hipMalloc(&d_u0, sizeof(double) * D * B));
    CK(hipMalloc(&d_cot, sizeof(double) * cot.size())); CK(hipMalloc(&d_knots, sizeof(double) * (size_t)(S + 1) * 2 * D * B)); CK(hipMalloc(&d_du0, sizeof(double) * D * B));
    CK(hipMalloc(&d_save, sizeof(int) * (S + 1))); CK(hipMalloc(&d_flag, sizeof(int))); CK(hipMalloc(&d_sum, sizeof(double) * 8));
    CK(hipMalloc(&d_part, sizeof(double) * (size_t)(B / 16) * NPAR)); CK(hipMalloc(&d_dp, sizeof(double) * NPAR));
    CK(hipMemcpy(d_p, p.data(), sizeof(double) * NPAR, hipMemcpyHostToDevice)); CK(hipMemcpy(d_u0, u0.data(), sizeof(double) * D * B, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cot, cot.data(), sizeof(double) * cot.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_save, save.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice));
    CK(hipMemset(d_flag, 0, sizeof(int))); CK(hipMemset(d_sum, 0, sizeof(double) * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)(B / 16), 1u);
    float fwd_ms = 0, gms = 0, gbest = 1e30f; double gtot = 0;
    for (int r = 0; r < 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mlp_forward<H>), grid, dim3(Mlp<H>::NT), 0, 0, g, (const double*)d_u0, (const double*)d_p, d_knots, (double*)nullptr, (const int*)d_save);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&fwd_ms, e0, e1));
    }
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0));
        if (alg == 2) hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 2>), grid, dim3(MlpG<H>::NT), 0, 0, g, (const double*)d_p, (const double*)d_knots, (const double*)d_cot, (const int*)d_save, (const int*)nullptr, d_part, (double*)nullptr, d_du0, d_flag);
        else if (alg == 1) hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 1>), grid, dim3(MlpG<H>::NT), 0, 0, g, (const double*)d_p, (const double*)d_knots, (const double*)d_cot, (const int*)d_save, (const int*)nullptr, d_part, (double*)nullptr, d_du0, d_flag);
        else hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 0>), grid, dim3(MlpG<H>::NT), 0, 0, g, (const double*)d_p, (const double*)d_knots, (const double*)d_cot, (const int*)d_save, (const int*)nullptr, d_part, (double*)nullptr, d_du0, d_flag);
        hipLaunchKernelGGL(k_mlp_grad_reduce, dim3((NPAR + 255) / 256, 1), dim3(256), 0, 0, (int)NPAR, (long)(B / 16), (const double*)d_part, d_dp);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&gms, e0, e1));
        if (r > 0) { gtot += gms; if (gms < gbest) gbest = gms; }
    }
    hipLaunchKernelGGL(k_checksum, dim3(256), dim3(256), 0, 0, (const double*)d_du0, (long)D * B, d_sum);
    hipLaunchKernelGGL(k_checksum, dim3(64), dim3(256), 0, 0, (const double*)d_dp, (long)NPAR, d_sum + 1);
    hipLaunchKernelGGL(k_checksum, dim3(256), dim3(256), 0, 0, (const double*)d_knots, (long)(S + 1) * 2 * D * B, d_sum + 2);
    double cs[8]; CK(hipMemcpy(cs, d_sum, sizeof(cs), hipMemcpyDeviceToHost));
    printf("{\"H\": %d, \"B\": %d, \"S\": %d, \"alg\": %d, \"forward_ms\": %.4f, \"grad_sweep_ms_mean\": %.4f, \"grad_sweep_ms_min\": %.4f, \"cs_du0\": %.15g, \"cs_dp\": %.15g, \"cs_knots\": %.15g}\n",
           H, B, S, alg, fwd_ms, gtot / reps, gbest, cs[0], cs[1], cs[2]);
    return 0;
}
