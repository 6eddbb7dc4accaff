/* tests/c/julia_seam.c — the call sequence of julia/HIPAdj/src/HIPAdj.jl (Handle / forward! / adjoint!, as driven by
 * julia/ext/SciMLSensitivityHIPAdjExt.jl's _concrete_solve_adjoint method) replayed from C, so that the ABI side of the Julia glue is
 * checked in an image without Julia:
 *   - the configuration is NOT filled through the C struct but byte by byte at the offsets Julia's `fieldoffset(HipadjConfig, i)`
 *     yields for the isbits struct of HIPAdj.jl (table HIPAdj.CONFIG_OFFSETS); _Static_asserts tie that table to offsetof() of
 *     include/hipadj.h;
 *   - every call passes what the `ccall` passes: Ref{HipadjConfig} = pointer to the 176 bytes, Ref{Ptr{Cvoid}} = pointer to the handle
 *     slot, Julia (n, N) / (n, M, N) column-major arrays = the ABI's [N][n] / [N][M][n] unchanged;
 *   - cotangent loss (the AD path: Delta = cotangent of `out`), shared p, saveat = 0.1, RK4 dt = 0.01, InterpolatingAdjoint.
 * Prints the numbers tests/test_julia_seam.py compares with the oracle on the GPU; without a device it must fail loudly.
 *   gcc -std=c11 -Wall -Wextra -Werror -Iinclude tests/c/julia_seam.c -o julia_seam -L<libdir> -lhipadj -Wl,-rpath,<libdir> */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hipadj.h"

/* HIPAdj.CONFIG_OFFSETS / CONFIG_SIZE (julia/HIPAdj/src/HIPAdj.jl) */
enum { O_struct_size = 0, O_model = 4, O_alg = 8, O_stepper = 12, O_dims = 16, O_ntraj = 32, O_t0 = 40, O_t1 = 48, O_dt = 56, O_nsave = 64,
       O_save_times = 72, O_loss_kind = 80, O_loss_shift = 88, O_checkpointing = 96, O_ckpt_stride = 100, O_quad_abstol = 104,
       O_quad_reltol = 112, O_no_start = 120, O_p_shared = 124, O_device = 128, O_time_segments = 132, O_cont_cost = 136,
       O_max_steps = 140, O_abstol = 144, O_reltol = 152, O_ncheckpoints = 160, O_checkpoints = 168, O_loss_scale = 176, O_ndevices = 184,
       O_device_ids = 192, O_reference_literal = 200, O_family = 204, CONFIG_SIZE = 208 };
#define TIE(f) _Static_assert(offsetof(hipadj_config, f) == O_##f, "HIPAdj.CONFIG_OFFSETS disagrees with include/hipadj.h at " #f)
TIE(struct_size); TIE(model); TIE(alg); TIE(stepper); TIE(dims); TIE(ntraj); TIE(t0); TIE(t1); TIE(dt); TIE(nsave); TIE(save_times);
TIE(loss_kind); TIE(loss_shift); TIE(checkpointing); TIE(ckpt_stride); TIE(quad_abstol); TIE(quad_reltol); TIE(no_start); TIE(p_shared);
TIE(device); TIE(time_segments); TIE(cont_cost); TIE(max_steps); TIE(abstol); TIE(reltol); TIE(ncheckpoints); TIE(checkpoints);
TIE(loss_scale); TIE(ndevices); TIE(device_ids); TIE(reference_literal); TIE(family);
_Static_assert(sizeof(hipadj_config) == CONFIG_SIZE, "HIPAdj.CONFIG_SIZE disagrees with include/hipadj.h");

#define PUT(buf, off, type, val) do { type v_ = (type)(val); memcpy((buf) + (off), &v_, sizeof(type)); } while (0)

static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53) - 0.5;
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 96;
    const int alg = argc > 2 ? atoi(argv[2]) : HIPADJ_ALG_INTERPOLATING;
    const int G = argc > 3 ? atoi(argv[3]) : 0;           /* > 1: Handle(...; devices = fill(0, G)) — ONE handle over G (virtual) shards, hipadj_config.device_ids (ABI 108) */
    int32_t devs[64] = {0};
    enum { n = 3, np = 3, M = 11 };
    const double p[np] = {10.0, 28.0, 8.0 / 3.0};
    double ts[M];
    unsigned long long s = 20240926ULL;
    unsigned char cfg[CONFIG_SIZE];
    hipadj_handle *h = NULL;                      /* Ref{Ptr{Cvoid}}(C_NULL) */
    int32_t dims[4] = {0, 0, 0, 0}, nn = 0, npp = 0;
    double *u0 = (double *)malloc(sizeof(double) * n * (size_t)N);          /* Matrix{Float64}(undef, n, N) */
    double *out = (double *)malloc(sizeof(double) * n * M * (size_t)N);     /* Array{Float64}(undef, n, M, N) */
    double *delta = (double *)malloc(sizeof(double) * n * M * (size_t)N);
    double *du0 = (double *)malloc(sizeof(double) * n * (size_t)N);
    double dp[np];
    int rc;
    if (!u0 || !out || !delta || !du0 || N < 1) return 2;
    for (int i = 0; i < M; ++i) ts[i] = 0.1 * i;                            /* collect(t0:saveat:t1) */
    ts[M - 1] = 1.0;
    for (long j = 0; j < N; ++j) { u0[n * j] = 1.0 + 0.1 * lcg(&s); u0[n * j + 1] = 0.1 * lcg(&s); u0[n * j + 2] = 0.1 * lcg(&s); }
    for (long q = 0; q < (long)n * M * N; ++q) delta[q] = lcg(&s);

    printf("version %d\n", hipadj_version());                               /* HIPAdj.lib(): version gate */
    rc = hipadj_model_sizes(HIPADJ_MODEL_LORENZ, dims, &nn, &npp);          /* builtin_model(:lorenz) */
    if (rc != HIPADJ_OK || nn != n || npp != np) { fprintf(stderr, "model_sizes -> %d (%d, %d)\n", rc, nn, npp); return 3; }

    memset(cfg, 0, sizeof cfg);                                             /* HipadjConfig(...) positional constructor */
    PUT(cfg, O_struct_size, uint32_t, CONFIG_SIZE);
    PUT(cfg, O_model, int32_t, HIPADJ_MODEL_LORENZ);
    PUT(cfg, O_alg, int32_t, alg);
    PUT(cfg, O_stepper, int32_t, HIPADJ_STEPPER_RK4_FIXED);
    PUT(cfg, O_ntraj, int64_t, N);
    PUT(cfg, O_t0, double, 0.0); PUT(cfg, O_t1, double, 1.0); PUT(cfg, O_dt, double, 0.01);
    PUT(cfg, O_nsave, int32_t, M);
    PUT(cfg, O_save_times, const double *, ts);
    PUT(cfg, O_loss_kind, int32_t, HIPADJ_LOSS_COTANGENT);
    PUT(cfg, O_checkpointing, int32_t, alg == HIPADJ_ALG_BACKSOLVE);        /* ischeckpointing(inner) */
    PUT(cfg, O_quad_abstol, double, 1e-6); PUT(cfg, O_quad_reltol, double, 1e-3);
    PUT(cfg, O_p_shared, int32_t, 1);
    PUT(cfg, O_abstol, double, 1e-6); PUT(cfg, O_reltol, double, 1e-3);
    PUT(cfg, O_checkpoints, const double *, NULL);
    if (G > 1 && G <= 64) { PUT(cfg, O_ndevices, int32_t, G); PUT(cfg, O_device_ids, const int32_t *, devs); }

    rc = hipadj_create((const hipadj_config *)(const void *)cfg, &h);       /* ccall(:hipadj_create, Cint, (Ref{HipadjConfig}, Ref{Ptr{Cvoid}}), ...) */
    if (rc != HIPADJ_OK) { fprintf(stderr, "hipadj status %d: %s\n", rc, hipadj_last_error(NULL)); return 1; }   /* HIPAdj.check */
    rc = hipadj_forward(h, u0, p, out);                                     /* forward!(h, u0, p) */
    if (rc != HIPADJ_OK) { fprintf(stderr, "hipadj status %d: %s\n", rc, hipadj_last_error(h)); return 1; }
    rc = hipadj_adjoint(h, delta, du0, dp);                                 /* adjoint!(h, buf) */
    if (rc != HIPADJ_OK) { fprintf(stderr, "hipadj status %d: %s\n", rc, hipadj_last_error(h)); return 1; }
    hipadj_destroy(h);                                                      /* finalizer */
    printf("dp %.17g %.17g %.17g\n", dp[0], dp[1], dp[2]);
    printf("du0_first %.17g %.17g %.17g\n", du0[0], du0[1], du0[2]);
    printf("du0_last %.17g %.17g %.17g\n", du0[n * (N - 1)], du0[n * (N - 1) + 1], du0[n * (N - 1) + 2]);
    printf("out_last %.17g %.17g %.17g\n", out[(size_t)n * M * N - 3], out[(size_t)n * M * N - 2], out[(size_t)n * M * N - 1]);
    free(u0); free(out); free(delta); free(du0);
    return 0;
}
