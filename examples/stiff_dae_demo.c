/* examples/stiff_dae_demo.c — a stiff semi-explicit DAE differentiated through the C ABI of libhipadj.so from plain C (no Python, no torch): the reference's singular-mass-matrix
 * test problem, test/Core3/adjoint.jl:1434-1530 —
 *
 *     rober(du, u, p, t):  du1 = -k1 y1 + k3 y2 y3;  du2 = k1 y1 - k2 y2^2 - k3 y2 y3;  du3 = y1 + y2 + y3 - 1        M = diag(1, 1, 0)
 *     p = [0.04, 3e7, 1e4], u0 = [1, 0, 1] (inconsistent: the initialisation moves y3 to 0), tspan (0, 100), G = y3(50) + y3(100)
 *
 * as a Julia host would drive it through `ccall`: hipadj_model_register (the model as text, compiled with hiprtc at hipadj_create), hipadj_model_set_mass_matrix (singular,
 * semi-explicit: accepted as a DAE), a handle on HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE — the one stepper that integrates it —, hipadj_forward, hipadj_adjoint.
 *
 *   gcc -std=c99 -Wall -Werror -Iinclude examples/stiff_dae_demo.c -o stiff_dae_demo -Lscimlsensitivity.jl_amd -lhipadj -Wl,-rpath,$PWD/scimlsensitivity.jl_amd -lm
 *   ./stiff_dae_demo [ntraj = 4] [sensealg = 0 Interpolating | 2 Gauss | 3 Quadrature | 4 GaussKronrod]
 *
 * Trajectory i > 0 runs with the rates scaled by (1 + 0.01 i).  Prints one line per result; tests/test_gpu_stiff.py runs the binary and compares the numbers of trajectory 0 with
 * the independent Radau gradient of tests/golden/stiff_adjoints.json (dG/dp = [9.587251, 5.515170e-09, -3.309056e-05]).  Without a GPU the create call fails loudly (status -2). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hipadj.h"

#define CHECK(call, h)                                                                                                       \
    do {                                                                                                                     \
        int rc_ = (call);                                                                                                    \
        if (rc_ != HIPADJ_OK) {                                                                                              \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, hipadj_status_string(rc_), hipadj_last_error(h));              \
            return 1;                                                                                                        \
        }                                                                                                                    \
    } while (0)

int main(int argc, char **argv) {
    enum { n = 3, np = 3, M = 2 };
    const long N = argc > 1 ? atol(argv[1]) : 4;
    const int alg = argc > 2 ? atoi(argv[2]) : HIPADJ_ALG_INTERPOLATING;
    int32_t id = 0;
    /* the model: f and its two VJP bodies (out = (df/du)' lam, out = (df/dp)' lam); NULL VJPs would ask for dual-number VJPs */
    CHECK(hipadj_model_register("rober_dae_demo", n, np,
                                "du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = u[0] + u[1] + u[2] - 1.0;",
                                "out[0] = -p[0]*lam[0] + p[0]*lam[1] + lam[2];"
                                "out[1] = p[2]*u[2]*lam[0] + (-2.0*p[1]*u[1] - p[2]*u[2])*lam[1] + lam[2];"
                                "out[2] = p[2]*u[1]*lam[0] - p[2]*u[1]*lam[1] + lam[2];",
                                "out[0] = -u[0]*lam[0] + u[0]*lam[1]; out[1] = -u[1]*u[1]*lam[1]; out[2] = u[1]*u[2]*lam[0] - u[1]*u[2]*lam[1];", &id), NULL);
    const double mass[n * n] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0};      /* row-major; the zero row (and column) marks y3 as algebraic */
    CHECK(hipadj_model_set_mass_matrix(id, mass), NULL);

    const double ts[M] = {50.0, 100.0};
    hipadj_config cfg;
    hipadj_handle *h = NULL;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.model = id;
    cfg.alg = alg;
    cfg.stepper = HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE;
    cfg.ntraj = N;
    cfg.t0 = 0.0; cfg.t1 = 100.0; cfg.dt = 0.0;
    cfg.abstol = 1e-10; cfg.reltol = 1e-8;
    cfg.quad_abstol = 1e-14; cfg.quad_reltol = 1e-8;
    cfg.nsave = M; cfg.save_times = ts;
    cfg.loss_kind = HIPADJ_LOSS_COTANGENT;
    cfg.p_shared = 0;                                                              /* every trajectory its own rates and its own gradient */
    CHECK(hipadj_create(&cfg, &h), NULL);

    double *u0 = malloc(sizeof(double) * N * n), *p = malloc(sizeof(double) * N * np), *out = malloc(sizeof(double) * N * M * n);
    double *dLdu = calloc((size_t)N * M * n, sizeof(double)), *du0 = malloc(sizeof(double) * N * n), *dp = malloc(sizeof(double) * N * np);
    for (long i = 0; i < N; ++i) {
        u0[i * n + 0] = 1.0; u0[i * n + 1] = 0.0; u0[i * n + 2] = 1.0;             /* the reference's inconsistent start */
        const double s = 1.0 + 0.01 * (double)i;
        p[i * np + 0] = 0.04 * s; p[i * np + 1] = 3.0e7 * s; p[i * np + 2] = 1.0e4 * s;
        for (int m = 0; m < M; ++m) dLdu[(i * M + m) * n + 2] = 1.0;                /* dg = e_3 at both loss times */
    }
    CHECK(hipadj_forward(h, u0, p, out), h);
    CHECK(hipadj_adjoint(h, dLdu, du0, dp), h);

    printf("y_at_50 %.15e %.15e %.15e\n", out[0], out[1], out[2]);
    printf("y_at_100 %.15e %.15e %.15e\n", out[3], out[4], out[5]);
    printf("constraint_residual_max");
    { double r = 0.0; for (long i = 0; i < N * M; ++i) { double s = out[i * n] + out[i * n + 1] + out[i * n + 2] - 1.0; if (s < 0) s = -s; if (s > r) r = s; } printf(" %.3e\n", r); }
    printf("dp %.15e %.15e %.15e\n", dp[0], dp[1], dp[2]);
    printf("du0 %.15e %.15e %.15e\n", du0[0], du0[1], du0[2]);
    printf("dp_last %.15e %.15e %.15e\n", dp[(N - 1) * np], dp[(N - 1) * np + 1], dp[(N - 1) * np + 2]);

    /* the explicit steppers refuse the model by name */
    {
        hipadj_handle *h2 = NULL;
        cfg.stepper = HIPADJ_STEPPER_TSIT5_ADAPTIVE;
        const int rc = hipadj_create(&cfg, &h2);
        printf("tsit5_on_the_dae %d\n", rc);                                        /* HIPADJ_ERR_UNSUPPORTED */
        if (rc == HIPADJ_OK) hipadj_destroy(h2);
    }
    hipadj_destroy(h);
    free(u0); free(p); free(out); free(dLdu); free(du0); free(dp);
    return 0;
}
