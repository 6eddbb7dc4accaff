/* examples/bouncing_ball_demo.c — a ContinuousCallback differentiated through the C ABI of libhipadj.so from plain C (no Python, no torch): the reference's bouncing ball,
 * test/Callbacks2/continuous_callbacks.jl:10-24, 212-217 —
 *
 *     fiip(du, u, p, t):  du1 = u2;  du2 = -p1          u0 = [5, 0], tspan (0, 2.5), p = [9.8, 0.8], saveat 0.5, abstol = reltol = 1e-12, g = sum(sol)
 *     condition(u, t, integrator) = u[1]                affect!(integrator): u[2] = -p[2] u[2]
 *
 * as a Julia host would drive it through `ccall`: hipadj_model_register (f as text; NULL VJP bodies: dual numbers), hipadj_model_set_continuous_callback (condition and affect
 * as text), a handle on HIPADJ_STEPPER_TSIT5_ADAPTIVE, hipadj_forward, hipadj_event_counts, hipadj_adjoint.
 *
 *   gcc -std=c99 -Wall -Werror -Iinclude examples/bouncing_ball_demo.c -o bouncing_ball_demo -Lscimlsensitivity.jl_amd -lhipadj -Wl,-rpath,$PWD/scimlsensitivity.jl_amd -lm
 *   ./bouncing_ball_demo [ntraj = 4] [sensealg = 0 Interpolating | 2 Gauss | 4 GaussKronrod]
 *
 * Trajectory i is dropped from 5 + i and has restitution 0.8 - 0.02 i.  Prints one line per result; tests/test_gpu_continuous_callbacks.py runs the binary and compares trajectory
 * 0 with the closed-form gradient of tests/golden/continuous_callbacks.json (du0 = [11.2371818, 8.0451763], dp = [-6.1927644, 59.0954544]).  Without a GPU the create call
 * fails loudly (status -2). */
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
    enum { n = 2, np = 2, M = 6 };
    const long N = argc > 1 ? atol(argv[1]) : 4;
    const int alg = argc > 2 ? atoi(argv[2]) : HIPADJ_ALG_INTERPOLATING;
    int32_t id = 0;
    CHECK(hipadj_model_register("bouncing_ball_demo", n, np, "du[0] = u[1]; du[1] = -p[0];", NULL, NULL, &id), NULL);
    CHECK(hipadj_model_set_continuous_callback(id, "c = u[0];", "un[1] = -p[1] * u[1];", 0), NULL);

    const double ts[M] = {0.0, 0.5, 1.0, 1.5, 2.0, 2.5};
    hipadj_config cfg;
    hipadj_handle *h = NULL;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.model = id;
    cfg.alg = alg;
    cfg.stepper = HIPADJ_STEPPER_TSIT5_ADAPTIVE;
    cfg.ntraj = N;
    cfg.t0 = 0.0; cfg.t1 = 2.5; cfg.dt = 0.0;
    cfg.abstol = 1e-12; cfg.reltol = 1e-12;
    cfg.nsave = M; cfg.save_times = ts;
    cfg.loss_kind = HIPADJ_LOSS_COTANGENT;
    cfg.p_shared = 0;                                                              /* every trajectory its own parameters and its own gradient */
    CHECK(hipadj_create(&cfg, &h), NULL);

    double *u0 = malloc(sizeof(double) * N * n), *p = malloc(sizeof(double) * N * np), *out = malloc(sizeof(double) * N * M * n);
    double *dLdu = malloc(sizeof(double) * N * M * n), *du0 = malloc(sizeof(double) * N * n), *dp = malloc(sizeof(double) * N * np);
    int32_t *ne = malloc(sizeof(int32_t) * N);
    for (long i = 0; i < N; ++i) {
        u0[i * n + 0] = 5.0 + (double)i; u0[i * n + 1] = 0.0;
        p[i * np + 0] = 9.8; p[i * np + 1] = 0.8 - 0.02 * (double)i;
    }
    for (long i = 0; i < N * M * n; ++i) dLdu[i] = 1.0;                             /* dg = 1: g = sum(sol) */
    CHECK(hipadj_forward(h, u0, p, out), h);
    CHECK(hipadj_event_counts(h, ne), h);
    CHECK(hipadj_adjoint(h, dLdu, du0, dp), h);

    printf("u_at_2.5 %.15e %.15e\n", out[(M - 1) * n], out[(M - 1) * n + 1]);
    printf("events"); for (long i = 0; i < N; ++i) printf(" %d", (int)ne[i]); printf("\n");
    printf("du0 %.15e %.15e\n", du0[0], du0[1]);
    printf("dp %.15e %.15e\n", dp[0], dp[1]);
    printf("dp_last %.15e %.15e\n", dp[(N - 1) * np], dp[(N - 1) * np + 1]);

    /* the fixed-step stepper refuses the model by name: there is no dense output to search between its knots */
    {
        hipadj_handle *h2 = NULL;
        cfg.stepper = HIPADJ_STEPPER_RK4_FIXED; cfg.dt = 0.01;
        const int rc = hipadj_create(&cfg, &h2);
        printf("rk4_on_the_callback %d\n", rc);                                     /* HIPADJ_ERR_UNSUPPORTED */
        if (rc == HIPADJ_OK) hipadj_destroy(h2);
    }
    hipadj_destroy(h);
    free(u0); free(p); free(out); free(dLdu); free(du0); free(dp); free(ne);
    return 0;
}
