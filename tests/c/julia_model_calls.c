/* The ccall sequence of julia/HIPAdj/src/HIPAdj.jl's register_model(...; mass_matrix), set_affect!, affect_apply and affect_vjp, replayed from C with the
 * argument types the Julia `ccall`s declare (the build image has no Julia).  CPU: compiles with -Werror, links, registers the model, compiles it
 * for gfx950 (hipadj_model_check) and fails loudly at the first call that needs a device; GPU: prints the numbers tests/test_julia_seam.py checks. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "hipadj.h"

int main(void) {
    enum { n = 2, np = 4, N = 3 };
    int32_t id = 0;
    /* register_model("lv_from_c", 2, 4; f = ..., vjp_u = nothing, vjp_p = nothing): dual-number VJPs */
    int rc = hipadj_model_register("lv_from_c", n, np, "du[0] = p[0]*u[0] - p[1]*u[0]*u[1]; du[1] = -p[2]*u[1] + p[3]*u[0]*u[1];", NULL, NULL, &id);
    if (rc != HIPADJ_OK) { fprintf(stderr, "register -> %d: %s\n", rc, hipadj_last_error(NULL)); return 2; }
    /* set_mass_matrix!(id, n, M): Julia hands permutedims(M) (column-major M' == row-major M) */
    const double M_rowmajor[n * n] = {2.0, 0.5, 0.0, 1.5};
    rc = hipadj_model_set_mass_matrix(id, M_rowmajor);
    if (rc != HIPADJ_OK) { fprintf(stderr, "mass matrix -> %d: %s\n", rc, hipadj_last_error(NULL)); return 2; }
    const double singular[n * n] = {1.0, 1.0, 1.0, 1.0};
    rc = hipadj_model_set_mass_matrix(id, singular);
    printf("singular %d\n", rc);                                            /* HIPADJ_ERR_UNSUPPORTED: refused, the previous matrix stays */
    const double semi_explicit[n * n] = {1.0, 0.0, 0.0, 0.0};                /* [Md 0; 0 0]: a DAE for HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE (round 6) */
    rc = hipadj_model_set_mass_matrix(id, semi_explicit);
    printf("semi_explicit %d\n", rc);
    rc = hipadj_model_set_mass_matrix(id, NULL);                             /* set_mass_matrix!(m, nothing) */
    if (rc != HIPADJ_OK) return 2;
    /* set_affect!(m, "un[0] += 2.0; pn[1] = 1.1 * p[1];") */
    rc = hipadj_model_set_affect(id, "un[0] += 2.0 * p[3]; pn[1] = 1.1 * p[1];");
    if (rc != HIPADJ_OK) { fprintf(stderr, "affect -> %d: %s\n", rc, hipadj_last_error(NULL)); return 2; }
    rc = hipadj_model_check(id);                                             /* check_now = true */
    if (rc != HIPADJ_OK) { fprintf(stderr, "check -> %d: %s\n", rc, hipadj_last_error(NULL)); return 2; }
    printf("version %d model %d\n", hipadj_version(), (int)id);
    /* affect_apply(m, u (n, N), p::Vector, t) */
    const double u[N * n] = {1.0, 2.0, 3.0, 4.0, 5.0, 6.0}, p[np] = {1.5, 1.0, 3.0, 0.5};
    double out[N * n], pout[N * np];
    rc = hipadj_affect_apply(id, 0, N, u, p, 1, 5.0, out, pout);
    if (rc != HIPADJ_OK) { fprintf(stderr, "hipadj status %d: %s\n", rc, hipadj_last_error(NULL)); return 1; }   /* HIPAdj.check */
    printf("out %.17g %.17g %.17g %.17g\n", out[0], out[1], out[4], out[5]);
    printf("pout %.17g %.17g %.17g %.17g\n", pout[0], pout[1], pout[2], pout[3]);
    /* affect_vjp(m, u, p, t, lam (n, N), gp (np, N)) */
    const double lam[N * n] = {1.0, -1.0, 0.5, 0.25, 2.0, 3.0};
    double gp[N * np], lam_out[N * n], gp_out[N * np];
    for (int i = 0; i < N * np; ++i) gp[i] = 0.1 * (i + 1);
    rc = hipadj_affect_vjp(id, 0, N, u, p, 1, 5.0, lam, gp, lam_out, gp_out);
    if (rc != HIPADJ_OK) { fprintf(stderr, "hipadj status %d: %s\n", rc, hipadj_last_error(NULL)); return 1; }
    printf("lam_out %.17g %.17g\n", lam_out[0], lam_out[5]);
    printf("gp_out %.17g %.17g %.17g %.17g\n", gp_out[0], gp_out[1], gp_out[2], gp_out[3]);
    return 0;
}
