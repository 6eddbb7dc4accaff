// hipadj_field_etd.hpp — the stiff stepper of the PDE family (round 5): exponential time differencing, ETDRK4, for the semilinear Brusselator
//     u' = (alpha / dx^2) L u + N(u, t),          L = periodic 5-point Laplacian (both species), N = the local reaction + forcing terms.
//
// Why: the reference integrates this example over (0, 11.5) with an implicit stepper (docs/src/examples/pde/brusselator.md:115, FBDF) and runs the adjoint with the
// stepper of the forward solve (src/sensitivity_interface.jl:487-491).  The explicit RK4 of hipadj_field.hpp is bound by the diffusion limit dt <= 2.8 dx^2 / (8 alpha):
// 460 000 steps and 45 GB for that horizon.  L is diagonal in the 2-D DFT basis and 32 x 32 complex numbers fit one workgroup, so the stiff part is integrated EXACTLY
// (Cox & Matthews, J. Comput. Phys. 176 (2002), eqs. 26-29 — the scheme OrdinaryDiffEq ships as ETDRK4) and the step is limited by the reaction terms only: 10^3-10^4
// steps for the documented horizon.  Forward and reverse use the same scheme; the reverse pass is the CONTINUOUS adjoint lam' = -(alpha/dx^2) L lam - R(y(t))^T lam
// (L symmetric) integrated with h = -dt, y(t) from the cubic Hermite interpolant of the forward knots exactly as the RK4 sweeps do, the parameter gradient as the
// quadrature the classic RK4 weights give on the stage values (on a component without linear part ETDRK4 IS the classic RK4).
//
// One workgroup per trajectory, ONE grid cell per thread (G = 8, 16, 32).  The two species are packed into one complex field z = U + i V: the operator has the same real
// eigenvalue for both species, so every spectral operation is a real factor on the complex transform and one complex FFT serves both.  The 2-D FFT is
//     rows:     radix-2 butterflies across the lanes of a row (ds_swizzle xor exchanges, 5 stages at G = 32, twiddles from a 16-entry LDS table),
//     columns:  a transposition through LDS (one barrier), then the same row transform,
// forward as decimation in frequency (natural in, bit-reversed out), inverse as decimation in time (bit-reversed in, natural out): no reordering pass, the thread of lane r
// and row q holds mode (k, l) = (rev q, rev r) of the transposed spectrum, and since the eigenvalues are symmetric in (k, l) the coefficients never notice.
// Eight 2-D transforms per forward step, nine per reverse step.
//
// Forcing: piecewise constant in time (it switches on at t = 1.1).  Inside an exponential step it is evaluated at the step's MIDPOINT time for all stages, and the knot
// derivative f(u_k) takes the forcing of the step that starts at the knot: a step that ends exactly at the switch sees the forcing of its interior, as with a tstop
// (oracle: tls_force_t, oracle/adjoint_oracle.c section 2b).
//
// Restated: right-hand side docs/src/examples/pde/brusselator.md:98-112; Interpolating RHS src/interpolating_adjoint.jl:150-174; Quadrature pass 1
// src/quadrature_adjoint.jl:35-46, 527-530; loss jumps src/adjoint_common.jl:754-821.  Oracle: ORC_STEPPER_ETDRK4.
#pragma once

#include "hipadj_field.hpp"

namespace hipadj {

#ifndef HIPADJ_ETD_PIN
#define HIPADJ_ETD_PIN(ALG) false            // true: that reverse kernel's transforms read their twiddles from LDS per level instead of holding them in registers (A/B;
                                              // registers win everywhere: Interpolating 116 -> 92 ms with 18 spilled registers, profiles/r5_etd_twiddles_ab.jsonl)
#endif
template <int G> struct EtdShape {
    static_assert(G == 8 || G == 16 || G == 32, "the exponential stepper holds one grid cell per thread: G = 8, 16 or 32");
    static_assert(Bruss<G>::Q == 1, "one cell per thread");
    static constexpr int LOGG = G == 8 ? 3 : (G == 16 ? 4 : 5);
    static constexpr int T = G * G, PITCH = G + 1;
};
template <int G> struct EtdLds {
    double sh[1][2 * G * G];                    // published stage vector (stencil reads).  ONE buffer (static LDS is capped at 64 KB): every pair of publications has a
                                                // barrier of a transposition between the reads of the first and the writes of the second, except where noted below
    double tr_re[2][G * (G + 1)], tr_im[2][G * (G + 1)];   // transposition tiles, double-buffered
    double tw_re[G / 2], tw_im[G / 2];          // exp(-2 pi i k / G)
    double red[(G * G / 64) * 3];
};

// phi_k(z) = sum_j z^j / (j + k)!  (k = 1, 2, 3) and e^z: Taylor below |z| = 1 (the closed forms cancel), recurrences above — the oracle's etd_phi
__device__ __forceinline__ void etd_phi(double z, double& ez, double& p1, double& p2, double& p3) {
    ez = exp(z);
    if (fabs(z) < 1.0) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0, zj = 1.0, fact = 1.0;
        for (int j = 0; j < 22; ++j) {
            if (j > 0) { zj *= z; fact *= j; }
            const double term = zj / fact;
            a1 += term / (j + 1.0);
            a2 += term / ((j + 1.0) * (j + 2.0));
            a3 += term / ((j + 1.0) * (j + 2.0) * (j + 3.0));
        }
        p1 = a1; p2 = a2; p3 = a3;
    } else { p1 = (ez - 1.0) / z; p2 = (p1 - 1.0) / z; p3 = (p2 - 0.5) / z; }
}
struct EtdCoef { double E2, Q, f1, f2, f3; };      // e^{hM} is taken as (e^{hM/2})^2 where it is used (two registers less in a kernel that has 128; the oracle does the same)
__device__ __forceinline__ int etd_rev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }
// coefficients of this thread's mode for dz/dt = coef * L z + N, signed step h
template <int G>
__device__ __forceinline__ EtdCoef etd_coefs(double coef, double h) {
    constexpr int LOGG = EtdShape<G>::LOGG;
    const int r = threadIdx.x & (G - 1), row = threadIdx.x / G;
    const int k = etd_rev(row, LOGG), l = etd_rev(r, LOGG);
    const double pi = 3.14159265358979323846;
    const double eig = coef * (2.0 * cos(2.0 * pi * k / G) + 2.0 * cos(2.0 * pi * l / G) - 4.0);
    const double z = h * eig;
    double ez, p1, p2, p3, ezh, q1, q2, q3;
    etd_phi(z, ez, p1, p2, p3); etd_phi(0.5 * z, ezh, q1, q2, q3);
    EtdCoef c;
    (void)ez; c.E2 = ezh; c.Q = 0.5 * h * q1;
    c.f1 = h * (p1 - 3.0 * p2 + 4.0 * p3); c.f2 = h * (p2 - 2.0 * p3); c.f3 = h * (4.0 * p3 - p2);
    return c;
}

// the value of lane ^ M.  M = 1, 2, 8 are single DPP moves on the vector pipe (quad_perm, row_ror:8), M = 4 is two (row_half_mirror = lane ^ 7, then quad_perm
// [3, 2, 1, 0] = lane ^ 3); only M = 16 goes through the LDS crossbar (ds_swizzle, bit mode) — with sixteen wavefronts per CU the swizzles of a transform had the LDS pipe
// as busy as the vector pipe
template <int M> __device__ __forceinline__ int etd_xor32(int v) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    else if constexpr (M == 4) return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true), 0x1B, 0xf, 0xf, true);
    else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
    else return __builtin_amdgcn_ds_swizzle(v, (M << 10) | 0x1F);
}
template <int M> __device__ __forceinline__ double etd_xor(double v) {
    return __hiloint2double(etd_xor32<M>(__double2hiint(v)), etd_xor32<M>(__double2loint(v)));
}

template <int G, bool PIN_TWIDDLES = false> struct EtdFft {      // PIN_TWIDDLES: keep the twiddle reads inside the transforms (kernels at their register limit)
    EtdLds<G>* L; int par;
    int r, row;
    double twr[EtdShape<G>::LOGG], twi[EtdShape<G>::LOGG];      // !PIN_TWIDDLES: this lane's factor of every butterfly level — (1, 0) on the lower lane of a pair — in registers: the
                                                                // same for the row and the column pass (the lane index is the position in both)
    __device__ __forceinline__ void init(EtdLds<G>* lds) {
        L = lds; par = 0; r = threadIdx.x & (G - 1); row = threadIdx.x / G;
        const double pi = 3.14159265358979323846;
        if ((int)threadIdx.x < G / 2) { L->tw_re[threadIdx.x] = cos(2.0 * pi * threadIdx.x / G); L->tw_im[threadIdx.x] = -sin(2.0 * pi * threadIdx.x / G); }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < EtdShape<G>::LOGG; ++l) {
            const int M = 1 << l;
            const bool up = (r & M) != 0;
            const int k = (r & (M - 1)) * (G / (2 * M));
            twr[l] = up ? L->tw_re[k] : 1.0; twi[l] = up ? L->tw_im[k] : 0.0;
        }
    }
    // one butterfly level of span M.  DIF (forward): exchange, then the upper lane multiplies by w^k; DIT (inverse): the upper lane multiplies by conj(w)^k, then exchange
    template <int M, bool INV> __device__ __forceinline__ void level(double& re, double& im) const {
        const bool up = (r & M) != 0;
        constexpr int LV = M == 1 ? 0 : (M == 2 ? 1 : (M == 4 ? 2 : (M == 8 ? 3 : 4)));
        double wr, wi;
        if constexpr (!PIN_TWIDDLES) { wr = twr[LV]; wi = INV ? -twi[LV] : twi[LV]; }
        else {
        int k = (r & (M - 1)) * (G / (2 * M));
        asm volatile("" : "+v"(k));      // keeps the twiddle reads where they are: hoisted out of the time loop, the ten pairs of a 2-D transform cost the
                                                                     // Interpolating kernel (128 registers per lane at 1024 threads) 58 spilled registers; the forward and the lambda-only
                                                                     // kernels have the room and let the compiler keep them in registers
        wr = up ? L->tw_re[k] : 1.0; wi = up ? (INV ? -L->tw_im[k] : L->tw_im[k]) : 0.0;
        }
        const double sg = up ? -1.0 : 1.0;
        if (INV) { const double a = re * wr - im * wi, b = re * wi + im * wr; re = a; im = b; }
        const double sr = etd_xor<M>(re), si = etd_xor<M>(im);
        const double dr = sr + sg * re, di = si + sg * im;
        if (INV) { re = dr; im = di; }
        else { re = dr * wr - di * wi; im = dr * wi + di * wr; }
    }
    template <bool INV> __device__ __forceinline__ void rows(double& re, double& im) const {
        if (!INV) {
            if constexpr (G >= 32) level<16, false>(re, im);
            if constexpr (G >= 16) level<8, false>(re, im);
            level<4, false>(re, im); level<2, false>(re, im); level<1, false>(re, im);
        } else {
            level<1, true>(re, im); level<2, true>(re, im); level<4, true>(re, im);
            if constexpr (G >= 16) level<8, true>(re, im);
            if constexpr (G >= 32) level<16, true>(re, im);
        }
    }
    __device__ __forceinline__ void transpose(double& re, double& im) {
        constexpr int P = EtdShape<G>::PITCH;
        double* tr = L->tr_re[par]; double* ti = L->tr_im[par];
        tr[r * P + row] = re; ti[r * P + row] = im;
        __syncthreads();
        re = tr[row * P + r]; im = ti[row * P + r];
        par ^= 1;      // the next transposition uses the other tile: a wave reaches it only through this barrier, after which nobody writes this tile before the one after next
    }
    __device__ __forceinline__ void fwd(double& re, double& im) { rows<false>(re, im); transpose(re, im); rows<false>(re, im); }
    __device__ __forceinline__ void inv(double& re, double& im) {
        rows<true>(re, im); transpose(re, im); rows<true>(re, im);
        const double s = 1.0 / (double)(G * G); re *= s; im *= s;
    }
};

// reaction + forcing of the forward equation at the thread's cell (force evaluated at time tf)
template <int G>
__device__ __forceinline__ void etd_react(const Nbr<G>& nb, const BrussP& P, double tf, double U, double V, double& nU, double& nV) {
    nU = P.B + U * U * V - (P.A + 1.0) * U + bruss_force(nb.x[0], nb.y[0], tf);
    nV = P.A * U - U * U * V;
}

// ---- forward solve ---------------------------------------------------------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_forward_etd(FieldGeom g, const double* __restrict__ u0, const double* __restrict__ p,
                                                                   double* __restrict__ knots, double* __restrict__ out, const int* __restrict__ save_of_knot) {
    constexpr int CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    __shared__ EtdLds<G> L;
    const long traj = blockIdx.x;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    const double dt = g.dt;
    EtdFft<G> F; F.init(&L);
    const EtdCoef c = etd_coefs<G>(P.adx, dt);
    const int cell = nb.c[0];
    double U[1] = {u0[traj * NS + cell]}, V[1] = {u0[traj * NS + CELLS + cell]};
    double zr = U[0], zi = V[0];
    F.fwd(zr, zi);                                           // the spectral image of the state is carried from step to step
    for (int k = 0; k <= g.S; ++k) {
        const double tmid = g.t0 + k * dt + 0.5 * dt;
        // knot k: (u_k, f(u_k)) with the FULL right-hand side (the Hermite data of the reverse pass), forcing of the step that starts here
        double LU[1], LV[1], n1U, n1V;
        publish<G>(L.sh[0], nb, U, V);
        laplace<G>(L.sh[0], nb, U, V, LU, LV);
        etd_react<G>(nb, P, tmid, U[0], V[0], n1U, n1V);
        if (knots) { double* kn = knots + ((traj * (g.S + 1) + k) * 2) * NS;
            kn[cell] = U[0]; kn[CELLS + cell] = V[0]; kn[NS + cell] = P.adx * LU[0] + n1U; kn[NS + CELLS + cell] = P.adx * LV[0] + n1V; }
        if (out) { const int s = save_of_knot[k]; if (s >= 0) { double* o = out + (traj * g.M + s) * NS; o[cell] = U[0]; o[CELLS + cell] = V[0]; } }
        if (k == g.S) break;
        double n1r = n1U, n1i = n1V; F.fwd(n1r, n1i);
        const double ar = c.E2 * zr + c.Q * n1r, ai = c.E2 * zi + c.Q * n1i;
        double sr = ar, si = ai; F.inv(sr, si);
        double n2r, n2i; etd_react<G>(nb, P, tmid, sr, si, n2r, n2i); F.fwd(n2r, n2i);
        sr = c.E2 * zr + c.Q * n2r; si = c.E2 * zi + c.Q * n2i; F.inv(sr, si);
        double n3r, n3i; etd_react<G>(nb, P, tmid, sr, si, n3r, n3i); F.fwd(n3r, n3i);
        sr = c.E2 * ar + c.Q * (2.0 * n3r - n1r); si = c.E2 * ai + c.Q * (2.0 * n3i - n1i); F.inv(sr, si);
        double n4r, n4i; etd_react<G>(nb, P, tmid, sr, si, n4r, n4i); F.fwd(n4r, n4i);
        zr = (c.E2 * c.E2) * zr + c.f1 * n1r + 2.0 * c.f2 * (n2r + n3r) + c.f3 * n4r;
        zi = (c.E2 * c.E2) * zi + c.f1 * n1i + 2.0 * c.f2 * (n2i + n3i) + c.f3 * n4i;
        sr = zr; si = zi; F.inv(sr, si);
        U[0] = sr; V[0] = si;
    }
}

// ---- reverse pass: InterpolatingAdjoint (ALG = 0: lam and the gradient partials), GaussAdjoint (ALG = 2: lam only + two Gauss nodes per step) and QuadratureAdjoint
// pass 1 (ALG = 3: lam only, dense record for k_bruss_quad_gk) ----
// publish (a, b) and take the 5-point stencil of both at this thread's cell.  NOT inlined: four inlined copies per reverse step (one per stage) cost the Interpolating kernel
// 70 registers — 56 of them spilled at the 128 a lane has with 1024 threads (kernel-resource-usage remarks) — and twice the time (23.3 instead of 11.9 us per step)
template <int G>
__device__ __attribute__((noinline)) void etd_publish_laplace(double* __restrict__ buf, int c, int im, int ip, int jm, int jp, double a, double b, double& La, double& Lb,
                                                              bool lead_sync = false) {
    constexpr int CELLS = Bruss<G>::CELLS;
    if (lead_sync) __syncthreads();      // uniform: the previous reader of `buf` had no barrier after it (two publications with no transform in between)
    buf[c] = a; buf[CELLS + c] = b;
    __syncthreads();
    La = buf[im] + buf[ip] + buf[jp] + buf[jm] - 4.0 * a;
    Lb = buf[CELLS + im] + buf[CELLS + ip] + buf[CELLS + jp] + buf[CELLS + jm] - 4.0 * b;
}
// N(lam; y) = -R(y)^T lam (the reaction block of the transposed Jacobian, negated: lam' = M lam + N with M = -(alpha/dx^2) L); WITH_P: the gradient partials
// w += wgt * (df/dp)^T lam
template <int G, bool WITH_P>
__device__ __forceinline__ void etd_adj_react(const BrussP& P, double yU, double yV, double LyU, double LyV, double lU, double lV,
                                              double& nU, double& nV, double wgt, double (&w)[3]) {
    const double uv2 = 2.0 * yU * yV, uu = yU * yU;
    nU = -((uv2 - (P.A + 1.0)) * lU + (P.A - uv2) * lV);
    nV = -(uu * lU - uu * lV);
    if (WITH_P) {      // sum_c lam_c (L y)_c == sum_c y_c (L lam)_c (L symmetric): the alpha entry takes the Laplacian of the STAGE STATE, which is a Hermite combination of the
                       // knots' Laplacians (LyU, LyV) — one stencil pass per knot instead of one per stage on lam
        w[0] += wgt * (-yU * lU + yU * lV);
        w[1] += wgt * lU;
        w[2] += wgt * ((lU * LyU + lV * LyV) * P.idx2);
    }
}
// the full -lam' = J(y)^T lam at the thread's cell (Quadrature record), lam published in buffer `buf`
template <int G>
__device__ __forceinline__ void etd_adj_full(EtdLds<G>& L, int /*unused*/, const Nbr<G>& nb, const BrussP& P, double yU, double yV, double lU, double lV, double& vU, double& vV) {
    double a[1] = {lU}, b[1] = {lV}, y1[1] = {yU}, y2[1] = {yV}, d1[1], d2[1], wd[3] = {0.0, 0.0, 0.0};
    publish<G>(L.sh[0], nb, a, b);
    bruss_vjp<G, false>(L.sh[0], nb, P, y1, y2, a, b, d1, d2, 0.0, wd);
    vU = d1[0]; vV = d2[0];
}

template <int G, int ALG>
__global__ void __launch_bounds__(Bruss<G>::T) k_bruss_adjoint_etd(FieldGeom g, long Npad, const double* __restrict__ p, const double* __restrict__ knots,
                                                                   const double* __restrict__ cot, const int* __restrict__ save_of_knot, double* __restrict__ adj,
                                                                   double* __restrict__ du0, double* __restrict__ dp_traj, int* __restrict__ flag) {
    constexpr int CELLS = Bruss<G>::CELLS, NS = Bruss<G>::NS;
    constexpr bool WP = ALG == 0, REC = ALG == 3, GAUSS = ALG == 2;
    __shared__ EtdLds<G> L;
    const long traj = blockIdx.x;
    Nbr<G> nb; nb.init();
    const BrussP P = load_bruss_p<G>(p, g.p_shared, traj);
    const double dt = g.dt;
    EtdFft<G, HIPADJ_ETD_PIN(ALG)> F; F.init(&L);
    const EtdCoef c = etd_coefs<G>(-P.adx, -dt);              // lam' = -(alpha/dx^2) L lam + N, stepped with h = -dt
    const int cell = nb.c[0];
    double lU[1] = {0.0}, lV[1] = {0.0}, w[3] = {0.0, 0.0, 0.0};
    FKnot<G> hi, lo;
    load_fknot<G>(knots, g, traj, g.S, nb, hi);
    { const int s = save_of_knot[g.S]; if (s >= 0) field_jump<G>(g, traj, s, cot, nb, hi.U, hi.V, lU, lV); }
    double zr = lU[0], zi = lV[0];
    F.fwd(zr, zi);
    // Interpolating: the Laplacians of the knot (u, f) of both species — of the upper knot carried from step to step, of the lower one taken at the top of the step
    double LhU = 0.0, LhV = 0.0, LhfU = 0.0, LhfV = 0.0, LlU = 0.0, LlV = 0.0, LlfU = 0.0, LlfV = 0.0;
    if (WP) {
        etd_publish_laplace<G>(L.sh[0], nb.c[0], nb.im[0], nb.ip[0], nb.jm[0], nb.jp[0], hi.U[0], hi.V[0], LhU, LhV);
        etd_publish_laplace<G>(L.sh[0], nb.c[0], nb.im[0], nb.ip[0], nb.jm[0], nb.jp[0], hi.fU[0], hi.fV[0], LhfU, LhfV, true);
    }
    for (int k = g.S - 1; k >= 0; --k) {
        load_fknot<G>(knots, g, traj, k, nb, lo);              // no knot prefetch: at 1024 threads a wave has 128 registers and the step's chain of transforms hides the load
        const double mU = 0.5 * (lo.U[0] + hi.U[0]) + (0.125 * dt) * (lo.fU[0] - hi.fU[0]);      // Hermite midpoint of the forward knots
        const double mV = 0.5 * (lo.V[0] + hi.V[0]) + (0.125 * dt) * (lo.fV[0] - hi.fV[0]);
        double LmU = 0.0, LmV = 0.0;
        if (WP) {
            etd_publish_laplace<G>(L.sh[0], nb.c[0], nb.im[0], nb.ip[0], nb.jm[0], nb.jp[0], lo.U[0], lo.V[0], LlU, LlV, true);
            etd_publish_laplace<G>(L.sh[0], nb.c[0], nb.im[0], nb.ip[0], nb.jm[0], nb.jp[0], lo.fU[0], lo.fV[0], LlfU, LlfV, true);
            LmU = 0.5 * (LlU + LhU) + (0.125 * dt) * (LlfU - LhfU); LmV = 0.5 * (LlV + LhV) + (0.125 * dt) * (LlfV - LhfV);
        }
        double* rec = REC ? adj + ((traj * g.S + k) * 4) * NS : nullptr;
        double hU = 0.0, hV = 0.0, v1U = 0.0, v1V = 0.0;          // GaussAdjoint: lam and J^T lam at the start of the step (the Hermite data of lam over the step)
        if (REC || GAUSS) {
            etd_adj_full<G>(L, 0, nb, P, hi.U[0], hi.V[0], lU[0], lV[0], v1U, v1V);
            if (REC) { rec[cell] = lU[0]; rec[CELLS + cell] = lV[0]; rec[NS + cell] = -v1U; rec[NS + CELLS + cell] = -v1V; }
            hU = lU[0]; hV = lV[0];
        }
        double n1r, n1i; etd_adj_react<G, WP>(P, hi.U[0], hi.V[0], LhU, LhV, lU[0], lV[0], n1r, n1i, dt / 6.0, w); F.fwd(n1r, n1i);
        const double ar = c.E2 * zr + c.Q * n1r, ai = c.E2 * zi + c.Q * n1i;
        double sr = ar, si = ai; F.inv(sr, si);
        double n2r, n2i; etd_adj_react<G, WP>(P, mU, mV, LmU, LmV, sr, si, n2r, n2i, dt / 3.0, w); F.fwd(n2r, n2i);
        sr = c.E2 * zr + c.Q * n2r; si = c.E2 * zi + c.Q * n2i; F.inv(sr, si);
        double n3r, n3i; etd_adj_react<G, WP>(P, mU, mV, LmU, LmV, sr, si, n3r, n3i, dt / 3.0, w); F.fwd(n3r, n3i);
        sr = c.E2 * ar + c.Q * (2.0 * n3r - n1r); si = c.E2 * ai + c.Q * (2.0 * n3i - n1i); F.inv(sr, si);
        double n4r, n4i; etd_adj_react<G, WP>(P, lo.U[0], lo.V[0], LlU, LlV, sr, si, n4r, n4i, dt / 6.0, w); F.fwd(n4r, n4i);
        zr = (c.E2 * c.E2) * zr + c.f1 * n1r + 2.0 * c.f2 * (n2r + n3r) + c.f3 * n4r;
        zi = (c.E2 * c.E2) * zi + c.f1 * n1i + 2.0 * c.f2 * (n2i + n3i) + c.f3 * n4i;
        sr = zr; si = zi; F.inv(sr, si);
        lU[0] = sr; lV[0] = si;
        if (REC || GAUSS) {
            double vU, vV;
            etd_adj_full<G>(L, 1, nb, P, lo.U[0], lo.V[0], lU[0], lV[0], vU, vV);
            if (REC) { rec[2 * NS + cell] = lU[0]; rec[2 * NS + CELLS + cell] = lV[0]; rec[3 * NS + cell] = -vU; rec[3 * NS + CELLS + cell] = -vV; }
            if (GAUSS) {
                // GaussAdjoint (src/gauss_adjoint.jl:745-759, 809-851): two Gauss-Legendre nodes per step, lam from the Hermite interpolant of the adjoint step
                // (h = -dt, slopes -v1 at its start and -v at its end), y from the forward one — as k_bruss_adjoint<G, 2> does on the RK4 grid
                const double xg = 0.5773502691896257645;
#pragma unroll 1
                for (int nq = 0; nq < 2; ++nq) {
                    const double x = nq == 0 ? -xg : xg, th = 0.5 * (1.0 + x), tf = 1.0 - th;
                    const double gU = (1.0 - th) * hU + th * lU[0] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lU[0] - hU) + (th - 1.0) * (-dt) * (-v1U) + th * (-dt) * (-vU));
                    const double gV = (1.0 - th) * hV + th * lV[0] + th * (th - 1.0) * ((1.0 - 2.0 * th) * (lV[0] - hV) + (th - 1.0) * (-dt) * (-v1V) + th * (-dt) * (-vV));
                    const double yU = (1.0 - tf) * lo.U[0] + tf * hi.U[0] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.U[0] - lo.U[0]) + (tf - 1.0) * dt * lo.fU[0] + tf * dt * hi.fU[0]);
                    const double yV = (1.0 - tf) * lo.V[0] + tf * hi.V[0] + tf * (tf - 1.0) * ((1.0 - 2.0 * tf) * (hi.V[0] - lo.V[0]) + (tf - 1.0) * dt * lo.fV[0] + tf * dt * hi.fV[0]);
                    double La, Lb;
                    etd_publish_laplace<G>(L.sh[0], nb.c[0], nb.im[0], nb.ip[0], nb.jm[0], nb.jp[0], gU, gV, La, Lb, true);
                    w[0] += (0.5 * dt) * (-yU * gU + yU * gV);
                    w[1] += (0.5 * dt) * gU;
                    w[2] += (0.5 * dt) * ((yU * La + yV * Lb) * P.idx2);
                }
            }
            __syncthreads();                                     // the next step's opening publication goes into the same buffer with no transposition in between
        }
        { const int s = save_of_knot[k];
          if (s >= 0 && !(g.no_start && s == 0)) {              // uniform over the workgroup
              field_jump<G>(g, traj, s, cot, nb, lo.U, lo.V, lU, lV);
              zr = lU[0]; zi = lV[0]; F.fwd(zr, zi);            // the jump changed lam in real space: refresh its spectral image
          } }
        hi = lo; LhU = LlU; LhV = LlV; LhfU = LlfU; LhfV = LlfV;
    }
    if (REC) {
        du0[traj * NS + cell] = lU[0]; du0[traj * NS + CELLS + cell] = lV[0];
        if (!(fabs(lU[0]) <= 1.79769313486231570e308) || !(fabs(lV[0]) <= 1.79769313486231570e308)) atomicOr(flag, 1);
    } else field_finish<G>(g, traj, Npad, nb, lU, lV, w, L.red, du0, dp_traj, flag);
}

}  // namespace hipadj
