// fp64_issue.hip — how fast does one MI355X SIMD issue FP64 VALU work?  Development micro-benchmark behind DESIGN.md §4.1:
// W waves per SIMD (1024 SIMDs x W single-wave workgroups), each running CH independent chains of dependent v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 scripts/fp64_issue.hip -o scripts/kbench_fp64 && scripts/kbench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CH, int VG>
__global__ void __launch_bounds__(64) k_fma(double* out, int iters, double a, double b) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    // VG extra live registers: forces the VGPR allocation (hence the waves per SIMD) without adding instructions
    double pad[VG > 0 ? VG : 1];
#pragma unroll
    for (int q = 0; q < VG; ++q) pad[q] = out[q];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
#pragma unroll
    for (int q = 0; q < VG; ++q) s += pad[q];
    if (s == 12345.678) out[0] = s;
}

// the matrix pipe: CH independent accumulators of v_mfma_f64_16x16x4_f64 (2 * 16 * 16 * 4 = 2048 flop per wave-instruction)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void __launch_bounds__(64) k_mfma(double* out, int iters, double a, double b) {
    d4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = d4{(double)c, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[0] = s;
}
template <int CH>
void run_mfma(int waves_per_simd, double* d) {
    const int iters = 2000, blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mfma<CH>), dim3(blocks), dim3(64), 0, 0, d, 10, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_mfma<CH>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)iters * 8 * CH * waves_per_simd;
    printf("v_mfma_f64_16x16x4_f64           chains %2d  waves/SIMD %d  %.3f ms  ns per wave-instruction per SIMD %.3f;  TFLOP/s %.1f\n", CH, waves_per_simd, ms,
           ms * 1e6 / inst_per_simd, inst_per_simd * 1024 * 2048.0 / (ms * 1e-3) / 1e12);
}

template <int CH, int VG>
void run(int waves_per_simd, double* d, const char* tag) {
    const int iters = 2000, blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_fma<CH, VG>), dim3(blocks), dim3(64), 0, 0, d, 10, 1.0000001, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_fma<CH, VG>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_wave = (double)iters * 16 * CH;
    const double inst_per_simd = inst_per_wave * waves_per_simd;
    printf("%-28s chains %2d  waves/SIMD %d  %.3f ms  ns per wave-instruction per SIMD %.3f  => %.2f cycles at 2.4 GHz;  TFLOP/s %.1f\n", tag, CH, waves_per_simd, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, inst_per_simd * 1024 * 128.0 / (ms * 1e-3) / 1e12);
}

int main() {
    double* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    run<1, 0>(1, d, "dependent chain");
    run<2, 0>(1, d, "2 chains");
    run<3, 0>(1, d, "3 chains");
    run<4, 0>(1, d, "4 chains");
    run<8, 0>(1, d, "8 chains");
    run<12, 0>(1, d, "12 chains");
    run<1, 0>(2, d, "dependent chain");
    run<3, 0>(2, d, "3 chains");
    run<12, 0>(2, d, "12 chains");
    run<1, 0>(4, d, "dependent chain");
    run<3, 0>(4, d, "3 chains");
    run<12, 0>(4, d, "12 chains");
    run<12, 0>(8, d, "12 chains");
    run_mfma<1>(1, d); run_mfma<2>(1, d); run_mfma<4>(1, d); run_mfma<8>(1, d);
    run_mfma<4>(2, d); run_mfma<8>(2, d); run_mfma<4>(4, d);
    return 0;
}
