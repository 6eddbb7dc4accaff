// quad_fwd.hip — development micro-benchmark (round 4): RK4 forward solve of the 10^4-trajectory Lorenz ensemble with knot stores,
//   (a) one lane per trajectory (the library's k_forward_ev mapping: 157 lone wavefronts),
//   (b) four lanes per trajectory, one state component per lane, neighbours' components through DPP quad_perm (628 wavefronts),
//   (c) as (b) with two trajectories per quad (ILP 2).
// hipcc --offload-arch=gfx950 -O3 scripts/r4/quad_fwd.hip -o scripts/r4/quad_fwd && scripts/r4/quad_fwd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct dbl2 { double x, y; };

__global__ void __launch_bounds__(64) k_lane(long N, long Npad, int S, double dt, const double* __restrict__ u0, double s, double r, double b, dbl2* __restrict__ knots) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= N) return;
    double u[3] = {u0[i * 3], u0[i * 3 + 1], u0[i * 3 + 2]}, k1[3];
    auto f = [&](double (&du)[3], const double (&v)[3]) { du[0] = s * (v[1] - v[0]); du[1] = v[0] * (r - v[2]) - v[1]; du[2] = v[0] * v[1] - b * v[2]; };
    f(k1, u);
    const double hh = 0.5 * dt, h6 = dt / 6.0;
    for (int k = 0; k < S; ++k) {
        double k2[3], k3[3], k4[3], us[3];
        for (int j = 0; j < 3; ++j) knots[((long)k * 3 + j) * Npad + i] = dbl2{u[j], k1[j]};
        for (int j = 0; j < 3; ++j) us[j] = u[j] + hh * k1[j];
        f(k2, us);
        for (int j = 0; j < 3; ++j) us[j] = u[j] + hh * k2[j];
        f(k3, us);
        for (int j = 0; j < 3; ++j) us[j] = u[j] + dt * k3[j];
        f(k4, us);
        for (int j = 0; j < 3; ++j) u[j] = u[j] + h6 * (k1[j] + 2.0 * (k2[j] + k3[j]) + k4[j]);
        f(k1, u);
    }
    for (int j = 0; j < 3; ++j) knots[((long)S * 3 + j) * Npad + i] = dbl2{u[j], k1[j]};
}

template <int CTRL> __device__ __forceinline__ double dppq(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// quad_perm control: lane q of a quad reads lane sel[q]
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
// Lorenz in component form: du_c = e1 * (A + B * e2) + C * s_c with e1 = s[perm1[c]], e2 = s[perm2[c]]
//   c = 0: sigma (y - x)      e1 = y, A = sigma, B = 0, C = -sigma
//   c = 1: x (rho - z) - y    e1 = x, e2 = z, A = rho, B = -1, C = -1
//   c = 2: x y - beta z       e1 = x, e2 = y, A = 0, B = 1, C = -beta
__device__ __forceinline__ double fq(double sc, double A, double B, double C) {
    const double e1 = dppq<QP(1, 0, 0, 3)>(sc), e2 = dppq<QP(0, 2, 1, 3)>(sc);
    return fma(e1, fma(B, e2, A), C * sc);
}
template <int ILP>
__global__ void __launch_bounds__(64) k_quad(long N, long Npad, int S, double dt, const double* __restrict__ u0, double s, double r, double b, dbl2* __restrict__ knots) {
    const int c = threadIdx.x & 3;
    const long q0 = ((long)blockIdx.x * 16 + (threadIdx.x >> 2)) * ILP;        // first trajectory of this quad
    const double A = c == 0 ? s : (c == 1 ? r : 0.0), B = c == 0 ? 0.0 : (c == 1 ? -1.0 : 1.0), C = c == 0 ? -s : (c == 1 ? -1.0 : (c == 2 ? -b : 0.0));
    double u[ILP], k1[ILP];
    bool live[ILP];
#pragma unroll
    for (int v = 0; v < ILP; ++v) { const long i = q0 + v; live[v] = i < N && c < 3; u[v] = live[v] ? u0[i * 3 + c] : 0.0; k1[v] = fq(u[v], A, B, C); }
    const double hh = 0.5 * dt, h6 = dt / 6.0;
    for (int k = 0; k < S; ++k) {
#pragma unroll
        for (int v = 0; v < ILP; ++v) if (live[v]) knots[((long)k * 3 + c) * Npad + q0 + v] = dbl2{u[v], k1[v]};
        double k2[ILP], k3[ILP], k4[ILP];
#pragma unroll
        for (int v = 0; v < ILP; ++v) k2[v] = fq(fma(hh, k1[v], u[v]), A, B, C);
#pragma unroll
        for (int v = 0; v < ILP; ++v) k3[v] = fq(fma(hh, k2[v], u[v]), A, B, C);
#pragma unroll
        for (int v = 0; v < ILP; ++v) k4[v] = fq(fma(dt, k3[v], u[v]), A, B, C);
#pragma unroll
        for (int v = 0; v < ILP; ++v) { u[v] = fma(h6, k1[v] + 2.0 * (k2[v] + k3[v]) + k4[v], u[v]); k1[v] = fq(u[v], A, B, C); }
    }
#pragma unroll
    for (int v = 0; v < ILP; ++v) if (live[v]) knots[((long)S * 3 + c) * Npad + q0 + v] = dbl2{u[v], k1[v]};
}

int main() {
    const long N = 10000, Npad = 10048; const int S = 1000; const double dt = 0.01;
    std::vector<double> u0(N * 3);
    srand(1);
    for (long i = 0; i < N; ++i) { u0[i * 3] = 1.0 + 0.1 * (rand() / (double)RAND_MAX - 0.5); u0[i * 3 + 1] = 0.1 * (rand() / (double)RAND_MAX - 0.5); u0[i * 3 + 2] = 0.1 * (rand() / (double)RAND_MAX - 0.5); }
    double* d_u0; dbl2 *d_k[2];
    const size_t kb = sizeof(dbl2) * (size_t)(S + 1) * 3 * Npad;
    CK(hipMalloc(&d_u0, sizeof(double) * N * 3)); CK(hipMalloc(&d_k[0], kb)); CK(hipMalloc(&d_k[1], kb));
    CK(hipMemcpy(d_u0, u0.data(), sizeof(double) * N * 3, hipMemcpyHostToDevice));
    CK(hipMemset(d_k[0], 0, kb)); CK(hipMemset(d_k[1], 0, kb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* tag, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int r = 0; r < 10; ++r) { CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms; }
        printf("%-28s best %.4f ms  mean %.4f ms  = %.2f TB/s of knot writes\n", tag, best, sum / 10, kb / (best * 1e-3) / 1e12);
    };
    time("lane per trajectory", [&] { hipLaunchKernelGGL(k_lane, dim3((N + 63) / 64), dim3(64), 0, 0, N, Npad, S, dt, d_u0, 10.0, 28.0, 8.0 / 3.0, d_k[0]); });
    time("quad, 1 trajectory / quad", [&] { hipLaunchKernelGGL(k_quad<1>, dim3((N + 15) / 16), dim3(64), 0, 0, N, Npad, S, dt, d_u0, 10.0, 28.0, 8.0 / 3.0, d_k[1]); });
    std::vector<dbl2> a((size_t)(S + 1) * 3 * Npad), bq(a.size());
    CK(hipMemcpy(a.data(), d_k[0], kb, hipMemcpyDeviceToHost)); CK(hipMemcpy(bq.data(), d_k[1], kb, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (long i = 0; i < N; ++i) for (int j = 0; j < 3; ++j) { const size_t o = ((size_t)S * 3 + j) * Npad + i; worst = fmax(worst, fabs(a[o].x - bq[o].x)); scale = fmax(scale, fabs(a[o].x)); }
    printf("quad vs lane, u(T): max abs diff %.3e (scale %.3e)\n", worst, scale);
    time("quad, 2 trajectories / quad", [&] { hipLaunchKernelGGL(k_quad<2>, dim3((N / 2 + 15) / 16), dim3(64), 0, 0, N, Npad, S, dt, d_u0, 10.0, 28.0, 8.0 / 3.0, d_k[1]); });
    time("quad, 4 trajectories / quad", [&] { hipLaunchKernelGGL(k_quad<4>, dim3((N / 4 + 15) / 16), dim3(64), 0, 0, N, Npad, S, dt, d_u0, 10.0, 28.0, 8.0 / 3.0, d_k[1]); });
    CK(hipMemcpy(bq.data(), d_k[1], kb, hipMemcpyDeviceToHost));
    worst = 0;
    for (long i = 0; i < N; ++i) for (int j = 0; j < 3; ++j) { const size_t o = ((size_t)S * 3 + j) * Npad + i; worst = fmax(worst, fabs(a[o].x - bq[o].x)); }
    printf("quad ILP 4 vs lane, u(T): max abs diff %.3e\n", worst);
    return 0;
}
