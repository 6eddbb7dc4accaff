/* examples/host_demo.c — the C ABI of libhipadj.so used from plain C (no Python, no torch): what a non-Python host such as the
 * Julia glue of INTEGRATION.md does through `ccall`.
 *
 *   gcc -std=c99 -Wall -Werror -Iinclude examples/host_demo.c -o host_demo -Lscimlsensitivity.jl_amd -lhipadj \
 *       -Wl,-rpath,$PWD/scimlsensitivity.jl_amd -lm
 *   ./host_demo [ntraj]
 *
 * The reference's Lorenz adjoint test (test/Core3/adjoint.jl:1157-1172: dg = u - 2 at t = 0:0.1:T) on an ensemble, fixed-step
 * RK4: forward solve, InterpolatingAdjoint reverse pass, then the same ensemble as two shards whose dp are summed by hand (the
 * sum the RCCL communicator of hipadj_comm_* performs across processes).  Prints one line per result; tests/test_gpu_parity.py
 * runs the binary and compares the numbers with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hipadj.h"

#define CHECK(call, h)                                                                              \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_ != HIPADJ_OK) {                                                                     \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, hipadj_status_string(rc_), hipadj_last_error(h)); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

/* deterministic inputs the test can reproduce: u0_i = (1, 0, 0) + 0.1 * (a_i, b_i, c_i) from a small LCG */
static double lcg(unsigned long long *s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53) - 0.5;
}

static int run(long n_lo, long n_hi, const double *u0, const double *p, const double *ts, int m, double t1, double dt,
               double *du0, double *dp, double *out) {
    hipadj_config cfg;
    hipadj_handle *h = NULL;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.model = HIPADJ_MODEL_LORENZ;
    cfg.alg = HIPADJ_ALG_INTERPOLATING;
    cfg.stepper = HIPADJ_STEPPER_RK4_FIXED;
    cfg.ntraj = n_hi - n_lo;
    cfg.t0 = 0.0; cfg.t1 = t1; cfg.dt = dt;
    cfg.nsave = m; cfg.save_times = ts;
    cfg.loss_kind = HIPADJ_LOSS_LSQ_SHIFT; cfg.loss_shift = 2.0;
    cfg.p_shared = 1; cfg.device = 0;
    CHECK(hipadj_create(&cfg, &h), NULL);
    CHECK(hipadj_forward(h, u0 + 3 * n_lo, p, out ? out + 3 * (long)m * n_lo : NULL), h);
    CHECK(hipadj_adjoint(h, NULL, du0 + 3 * n_lo, dp), h);
    CHECK(hipadj_destroy(h), NULL);
    return 0;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 200;
    const double t1 = 1.0, dt = 0.01, p[3] = {10.0, 28.0, 8.0 / 3.0};
    double ts[11], dp[3], dpa[3], dpb[3];
    double *u0 = (double *)malloc(sizeof(double) * 3 * (size_t)n), *du0 = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    double *du0s = (double *)malloc(sizeof(double) * 3 * (size_t)n), *out = (double *)malloc(sizeof(double) * 3 * 11 * (size_t)n);
    unsigned long long s = 20240601ULL;
    int32_t nn = 0, npar = 0;
    if (!u0 || !du0 || !du0s || !out || n < 2) return 2;
    for (int i = 0; i <= 10; ++i) ts[i] = 0.1 * i;
    ts[10] = t1;
    for (long i = 0; i < n; ++i) {
        u0[3 * i + 0] = 1.0 + 0.1 * lcg(&s); u0[3 * i + 1] = 0.1 * lcg(&s); u0[3 * i + 2] = 0.1 * lcg(&s);
    }
    printf("version %d\n", hipadj_version());
    if (hipadj_model_sizes(HIPADJ_MODEL_LORENZ, NULL, &nn, &npar) != HIPADJ_OK || nn != 3 || npar != 3) return 3;
    if (run(0, n, u0, p, ts, 11, t1, dt, du0, dp, out)) return 1;
    printf("dp %.17g %.17g %.17g\n", dp[0], dp[1], dp[2]);
    printf("du0_first %.17g %.17g %.17g\n", du0[0], du0[1], du0[2]);
    printf("du0_last %.17g %.17g %.17g\n", du0[3 * (n - 1)], du0[3 * (n - 1) + 1], du0[3 * (n - 1) + 2]);
    printf("out_last %.17g %.17g %.17g\n", out[3 * 11 * n - 3], out[3 * 11 * n - 2], out[3 * 11 * n - 1]);
    /* two shards: contiguous trajectory ranges, one handle each (one process per GPU in production); dp = sum of the shards */
    if (run(0, n / 2, u0, p, ts, 11, t1, dt, du0s, dpa, NULL) || run(n / 2, n, u0, p, ts, 11, t1, dt, du0s, dpb, NULL)) return 1;
    printf("dp_shards %.17g %.17g %.17g\n", dpa[0] + dpb[0], dpa[1] + dpb[1], dpa[2] + dpb[2]);
    printf("du0_shards_equal %d\n", memcmp(du0, du0s, sizeof(double) * 3 * (size_t)n) == 0);
    free(u0); free(du0); free(du0s); free(out);
    return 0;
}
