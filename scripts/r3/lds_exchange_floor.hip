// lds_exchange_floor.hip — what ONE workgroup on ONE CU pays per LDS exchange (publish a value per thread, barrier, read the five stencil
// neighbours) and per bare barrier: the latency floor of the workgroup-per-trajectory kernels at N = 1 (Brusselator 32 x 32: 1024 threads,
// 4-5 exchanges per reverse RK4 step; wide runtime models: two barriers per joint-VJP call).  VERDICT r2 weak 9 asked for this model.
//   hipcc --offload-arch=gfx950 -O3 scripts/r3/lds_exchange_floor.hip -o /tmp/lds_floor && /tmp/lds_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int T, int MODE>   // MODE 0: barrier only; 1: write + barrier + 5 reads + 6 dependent FMAs (one stencil stage); 2: the same, double-buffered (one barrier per exchange)
__global__ void __launch_bounds__(T) k_exch(int iters, double* out) {
    __shared__ double sh[2][2048];
    const int t = threadIdx.x, G = 32;
    const int i = t % G, j = (t / G) % G;
    const int im = (i + G - 1) % G + j * G, ip = (i + 1) % G + j * G, jm = i + ((j + G - 1) % G) * G, jp = i + ((j + 1) % G) * G;
    double v = 1.0 + 1e-3 * t, w = 0.5;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { __syncthreads(); v = v * 1.0000001 + 1e-9; }
        else if (MODE == 1) {
            sh[0][t] = v; sh[0][1024 + t] = w;
            __syncthreads();
            const double L = sh[0][im] + sh[0][ip] + sh[0][jm] + sh[0][jp] - 4.0 * v, M = sh[0][1024 + im] + sh[0][1024 + ip] + sh[0][1024 + jm] + sh[0][1024 + jp] - 4.0 * w;
            v = v + 1e-6 * (L + v * v * w - 2.0 * v); w = w + 1e-6 * (M + v - v * v * w);
            __syncthreads();
        } else {
            double* b = sh[it & 1];
            b[t] = v; b[1024 + t] = w;
            __syncthreads();
            const double L = b[im] + b[ip] + b[jm] + b[jp] - 4.0 * v, M = b[1024 + im] + b[1024 + ip] + b[1024 + jm] + b[1024 + jp] - 4.0 * w;
            v = v + 1e-6 * (L + v * v * w - 2.0 * v); w = w + 1e-6 * (M + v - v * v * w);
        }
    }
    out[t] = v + w;
}

template <int T, int MODE> int run(const char* label) {
    double* d; CK(hipMalloc(&d, 2048 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 200000;
    hipLaunchKernelGGL((k_exch<T, MODE>), dim3(1), dim3(T), 0, 0, 1000, d);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL((k_exch<T, MODE>), dim3(1), dim3(T), 0, 0, iters, d); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-70s threads %4d  %.1f ns per iteration\n", label, T, ms * 1e6 / iters);
    return 0;
}
int main() {
    run<1024, 0>("barrier only (16 waves)");
    run<256, 0>("barrier only (4 waves)");
    run<64, 0>("barrier only (1 wave)");
    run<1024, 1>("publish 2 values + barrier + 10 stencil reads + update + barrier");
    run<1024, 2>("the same, double-buffered: one barrier per exchange");
    run<256, 2>("the same, double-buffered, 256 threads");
    return 0;
}
