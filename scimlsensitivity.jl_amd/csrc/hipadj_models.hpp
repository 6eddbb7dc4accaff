// hipadj_models.hpp — compile-time model registry for the lane-per-trajectory kernel family.
//
// Each model provides device-inlined f, (df/du)^T lam and (df/dp)^T lam with the argument meaning of the
// reference's user-VJP seam  ODEFunction(f!; vjp = (dlam, lam, u, p, t), vjp_p = (dgrad, lam, u, p, t))
// (src/derivative_wrappers.jl:284-359; exercised by test/Core3/user_vjp.jl:14-38): both VJPs are
// UN-negated — the adjoint RHS negates afterwards (src/interpolating_adjoint.jl:169-170).
// Hand-derived; no runtime AD on the device.
#pragma once

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define HIPADJ_HD __host__ __device__ __forceinline__
#else
#define HIPADJ_HD inline
#endif
// Scheduling fence between unrolled RK4 steps (device code only): stops hipcc from hoisting the lambda-independent
// part of LATER steps (Hermite midpoints of knots that are still in flight) above the current step, which would
// turn the counted `s_waitcnt vmcnt(N)` of the software prefetch into a full drain once per block.
#if defined(__HIP_DEVICE_COMPILE__)
#define HIPADJ_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HIPADJ_STEP_FENCE() ((void)0)
#endif

namespace hipadj {

constexpr int HIPADJ_CKPT_KMAX = 16;   // longest checkpoint interval (steps) the in-kernel re-solve tile holds

// Lotka-Volterra (test/Core3/user_vjp.jl:6-10): du1 = p1 u1 - p2 u1 u2 ; du2 = -p3 u2 + p4 u1 u2
struct ModelLV {
    static constexpr int N = 2, NP = 4;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1];
        du[1] = -p[2] * u[1] + p[3] * u[0] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        dl[0] = (p[0] - p[1] * u[1]) * l[0] + p[3] * u[1] * l[1];
        dl[1] = -p[1] * u[0] * l[0] + (-p[2] + p[3] * u[0]) * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        const double xy = u[0] * u[1];
        dg[0] = u[0] * l[0]; dg[1] = -xy * l[0]; dg[2] = -u[1] * l[1]; dg[3] = xy * l[1];
    }
};

// time-dependent LV `fb` (test/Core3/adjoint.jl:8-12, Jacobian :18-25)
struct ModelLVT {
    static constexpr int N = 2, NP = 4;
    static constexpr bool TIME_DEP = true;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double t) {
        du[0] = p[0] * u[0] - p[1] * u[0] * u[1] * t;
        du[1] = -p[2] * u[1] + t * p[3] * u[0] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double t) {
        dl[0] = (p[0] - p[1] * u[1] * t) * l[0] + t * u[1] * p[3] * l[1];
        dl[1] = -p[1] * u[0] * t * l[0] + (-p[2] + t * u[0] * p[3]) * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double t) {
        const double xyt = u[0] * u[1] * t;
        dg[0] = u[0] * l[0]; dg[1] = -xyt * l[0]; dg[2] = -u[1] * l[1]; dg[3] = xyt * l[1];
    }
};

// Lorenz-63 (test/Core3/adjoint.jl:1160-1166), p = (sigma, rho, beta)
struct ModelLorenz {
    static constexpr int N = 3, NP = 3;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * (u[1] - u[0]);
        du[1] = u[0] * (p[1] - u[2]) - u[1];
        du[2] = u[0] * u[1] - p[2] * u[2];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&u)[N], const double (&p)[NP], double) {
        dl[0] = -p[0] * l[0] + (p[1] - u[2]) * l[1] + u[1] * l[2];
        dl[1] = p[0] * l[0] - l[1] + u[0] * l[2];
        dl[2] = -u[0] * l[1] - p[2] * l[2];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        dg[0] = (u[1] - u[0]) * l[0]; dg[1] = u[0] * l[1]; dg[2] = -u[2] * l[2];
    }
};

// u' = p .* u (test/Core1/sparse_adjoint.jl:6-8: jac = diag(p), paramjac = diag(u))
struct ModelLinDiag {
    static constexpr int N = 2, NP = 2;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = p[0] * u[0]; du[1] = p[1] * u[1];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&p)[NP], double) {
        dl[0] = p[0] * l[0]; dl[1] = p[1] * l[1];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&u)[N], const double (&)[NP], double) {
        dg[0] = u[0] * l[0]; dg[1] = u[1] * l[1];
    }
};

// falling mass (test/Core7/physical_ode_regression.jl:20-23): u' = [u2, -g], p = (g, m)
struct ModelFallMass {
    static constexpr int N = 2, NP = 2;
    static constexpr bool TIME_DEP = false;
    HIPADJ_HD static void f(double (&du)[N], const double (&u)[N], const double (&p)[NP], double) {
        du[0] = u[1]; du[1] = -p[0];
    }
    HIPADJ_HD static void vjp_u(double (&dl)[N], const double (&l)[N], const double (&)[N], const double (&)[NP], double) {
        dl[0] = 0.0; dl[1] = l[0];
    }
    HIPADJ_HD static void vjp_p(double (&dg)[NP], const double (&l)[N], const double (&)[N], const double (&)[NP], double) {
        dg[0] = -l[1]; dg[1] = 0.0;
    }
};

}  // namespace hipadj
