// hipadj_quad.hpp — four lanes per trajectory (round 4): lane c of a quad owns state component c, the neighbours' components arrive through DPP
// `quad_perm` moves (no LDS), a wavefront holds 16 trajectories.
//
// Why.  At 10^4 trajectories the lane-per-trajectory forward solves are 157 lone wavefronts on 1024 SIMDs, and a lone wavefront issues ONE instruction
// of any kind every ~5.5 cycles (scripts/r4/quad_fwd.hip: giving a lane two or four independent trajectories lengthens the kernel in proportion — the
// bound is the length of one wave's instruction stream, not latency).  The only lever is fewer instructions per trajectory-step and wave: with one
// component per lane the stage algebra, the knot store and the error norm cost one instruction where the lane form needs n, and the right-hand side of
// a model in COMPONENT FORM (QuadForm below) is a handful of FMAs on operands fetched with two DPP moves each.  Lorenz RK4 step: 57 -> 40 instructions,
// 628 wavefronts instead of 157 (profiles/r4_quad_fwd_microbench.log: 0.1285 -> 0.1023 ms on the bare loops).
//
// Same data layout as the lane family (knots [S+1][n pairs][Npad], pair j = (u_j, f_j); outT / ckpt / yT component-major), so every reverse kernel reads
// what this writes.  The arithmetic of a component form is NOT expression-for-expression the lane form's (e.g. Lorenz du_0 = y sigma - sigma x instead of
// sigma (y - x)): results agree to roundoff, and the parity gate (rtol 1e-6 against the oracle) is unchanged.
#pragma once
#include "hipadj_lane.hpp"

namespace hipadj {

#define HIPADJ_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))   // quad_perm control: lane q of a quad reads lane sel[q]
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL> __device__ __forceinline__ double quad_perm(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#elif defined(HIPADJ_QUAD_EMU)
// TEST-ONLY host emulation (tests/emu/quad_emu.cpp): four host threads run the four lanes of a quad in lockstep and meet here
double hipadj_quad_emu_exchange(int ctrl, double x);
template <int CTRL> inline double quad_perm(double x) { return hipadj_quad_emu_exchange(CTRL, x); }
#else
template <int CTRL> inline double quad_perm(double x) { return x; }   // host pass of hipcc: never executed
#endif

// Component form of a compiled-in model: `consts(c, p)` = what lane c needs of the parameters, `f(k, s, t)` = component c of f at the stage state whose
// component c is `s` (every lane of the quad calls it together).  Lanes c >= N are switched off by the kernels and are never read by the permutations.
template <class Mo> struct QuadForm { static constexpr bool value = false; };

// Lorenz-63: du_c = e1 (A + B e2) + C s_c with e1 = s[perm1[c]], e2 = s[perm2[c]]
//   c = 0: sigma (y - x)      e1 = y,          A = sigma, B = 0,  C = -sigma
//   c = 1: x (rho - z) - y    e1 = x, e2 = z,  A = rho,   B = -1, C = -1
//   c = 2: x y - beta z       e1 = x, e2 = y,  A = 0,     B = 1,  C = -beta
template <> struct QuadForm<ModelLorenz> {
    static constexpr bool value = true;
    struct K { double A, B, C; };
    HIPADJ_HD static K consts(int c, const double (&p)[3]) {
        K k;
        k.A = c == 0 ? p[0] : (c == 1 ? p[1] : 0.0);
        k.B = c == 0 ? 0.0 : (c == 1 ? -1.0 : 1.0);
        k.C = c == 0 ? -p[0] : (c == 1 ? -1.0 : -p[2]);
        return k;
    }
    HIPADJ_HD static double f(const K& k, double s, double) {
        const double e1 = quad_perm<HIPADJ_QP(1, 0, 0, 3)>(s), e2 = quad_perm<HIPADJ_QP(0, 2, 1, 3)>(s);
        return fma(e1, fma(k.B, e2, k.A), k.C * s);
    }
};
// Lotka-Volterra: du_c = s_c (A + B e2), e2 = the other component:  c = 0: x (p1 - p2 y);  c = 1: y (-p3 + p4 x)
template <> struct QuadForm<ModelLV> {
    static constexpr bool value = true;
    struct K { double A, B; };
    HIPADJ_HD static K consts(int c, const double (&p)[4]) { K k; k.A = c == 0 ? p[0] : -p[2]; k.B = c == 0 ? -p[1] : p[3]; return k; }
    HIPADJ_HD static double f(const K& k, double s, double) { return s * fma(k.B, quad_perm<HIPADJ_QP(1, 0, 3, 2)>(s), k.A); }
};
// time-dependent Lotka-Volterra `fb`: du_c = s_c (A + t B e2)
template <> struct QuadForm<ModelLVT> {
    static constexpr bool value = true;
    struct K { double A, B; };
    HIPADJ_HD static K consts(int c, const double (&p)[4]) { K k; k.A = c == 0 ? p[0] : -p[2]; k.B = c == 0 ? -p[1] : p[3]; return k; }
    HIPADJ_HD static double f(const K& k, double s, double t) { return s * fma(t * k.B, quad_perm<HIPADJ_QP(1, 0, 3, 2)>(s), k.A); }
};

// Forward RK4 between event knots (the quad counterpart of forward_lane_ev, hipadj_lane.hpp): lane (i, c) integrates component c of trajectory i.
// Called by all live lanes of a quad together; lanes with c >= N or i >= g.N have left the kernel.
template <class Mo>
HIPADJ_HD void forward_quad_ev(const Geom& g, long i, int c, const double* __restrict__ u0, const double* __restrict__ p, const FwdEvents ev,
                               dbl2* __restrict__ knots, double* __restrict__ ckpt, double* __restrict__ outT, double* __restrict__ yT) {
    constexpr int N = Mo::N;
    using Q = QuadForm<Mo>;
    double pv[Mo::NP]; load_p<Mo>(p, g, i, pv);
    const typename Q::K kc = Q::consts(c, pv);
    double u = u0[i * N + c];
    double k1 = Q::f(kc, u, g.t0);
    dbl2* __restrict__ kn = knots ? knots + (long)c * g.Npad + i : nullptr;      // this lane's pair of knot 0
    const long kstep = (long)N * g.Npad;
    int k = 0;
    // the NEXT event's knot and slots are fetched before the run of steps in front of the current one, so that the scalar loads (a dependent L2 round trip,
    // ~0.3 us with one wave per SIMD: 100 events were a quarter of the kernel) complete under the step loop
    int kn_next = ev.nev > 0 ? ev.knot[0] : g.S, ck_next = ev.nev > 0 ? ev.ckpt[0] : -1, sv_next = ev.nev > 0 ? ev.save[0] : -1;
    for (int e = 0; e <= ev.nev; ++e) {
        const int kn_end = kn_next, ck = ck_next, sv = sv_next;    // run of plain steps up to the next event (or to the end)
        if (e + 1 < ev.nev) { kn_next = ev.knot[e + 1]; ck_next = ev.ckpt[e + 1]; sv_next = ev.save[e + 1]; } else kn_next = g.S;
        const double dt = (k == g.S - 1) ? g.h_last : g.dt, hh = 0.5 * dt, h6 = dt / 6.0;
        const bool rag = k == g.S - 1 && g.h_last != g.dt;          // the shortened last step of a span that is not a multiple of dt (a run of its own): its end is T, not t0 + S dt
        if (knots) {
#pragma unroll 2
            for (; k < kn_end; ++k) {
                const double t = g.t0 + k * g.dt;
                { dbl2 d; d.x = u; d.y = k1; *kn = d; kn += kstep; }
                const double k2 = Q::f(kc, fma(hh, k1, u), t + hh);
                const double k3 = Q::f(kc, fma(hh, k2, u), t + hh);
                const double k4 = Q::f(kc, fma(dt, k3, u), t + dt);
                u = fma(h6, k1 + 2.0 * (k2 + k3) + k4, u);
                k1 = Q::f(kc, u, rag ? t + dt : g.t0 + (k + 1) * g.dt);   // first-same-as-last: the slope stored with knot k + 1
            }
        } else {
            for (; k < kn_end; ++k) {
                const double t = g.t0 + k * g.dt;
                const double k2 = Q::f(kc, fma(hh, k1, u), t + hh);
                const double k3 = Q::f(kc, fma(hh, k2, u), t + hh);
                const double k4 = Q::f(kc, fma(dt, k3, u), t + dt);
                u = fma(h6, k1 + 2.0 * (k2 + k3) + k4, u);
                k1 = Q::f(kc, u, rag ? t + dt : g.t0 + (k + 1) * g.dt);
            }
        }
        if (e < ev.nev) {                                          // the event AT knot kn_end (k == kn_end now)
            if (ckpt && ck >= 0) ckpt[((long)ck * N + c) * g.Npad + i] = u;
            if (outT && sv >= 0) outT[((long)sv * N + c) * g.Npad + i] = u;
        }
    }
    if (knots) { dbl2 d; d.x = u; d.y = k1; *kn = d; }
    if (yT) yT[(long)c * g.Npad + i] = u;
}

}  // namespace hipadj
