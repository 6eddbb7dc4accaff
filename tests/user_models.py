"""HIP C++ bodies of the runtime-registered test models (hipadj_model_register) and their oracle counterparts.
The oracle implements the same right-hand sides in C (ORC_MODEL_ROBER, ORC_MODEL_RING) as the checker."""

ROBER = dict(  # Robertson kinetics `rober`, test/Core3/adjoint.jl:1434-1441
    n=3, np=3,
    f="du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = p[1]*u[1]*u[1];",
    vjp=("out[0] = -p[0]*lam[0] + p[0]*lam[1];"
         "out[1] = p[2]*u[2]*lam[0] + (-2.0*p[1]*u[1] - p[2]*u[2])*lam[1] + 2.0*p[1]*u[1]*lam[2];"
         "out[2] = p[2]*u[1]*lam[0] - p[2]*u[1]*lam[1];"),
    vjp_p=("out[0] = -u[0]*lam[0] + u[0]*lam[1]; out[1] = -u[1]*u[1]*lam[1] + u[1]*u[1]*lam[2];"
           "out[2] = u[1]*u[2]*lam[0] - u[1]*u[2]*lam[1];"))


def ring(n):
    """synthetic ring: du_i = p_i (u_{i+1} - u_i) + p_n sin(u_{i-1}), indices mod n; np = n + 1"""
    f = "".join(f"du[{i}] = p[{i}]*(u[{(i + 1) % n}] - u[{i}]) + p[{n}]*sin(u[{(i - 1) % n}]);" for i in range(n))
    vjp = "".join(f"out[{j}] = -p[{j}]*lam[{j}] + p[{(j - 1) % n}]*lam[{(j - 1) % n}] + p[{n}]*cos(u[{j}])*lam[{(j + 1) % n}];" for j in range(n))
    vjp_p = "".join(f"out[{k}] = lam[{k}]*(u[{(k + 1) % n}] - u[{k}]);" for k in range(n))
    vjp_p += f"out[{n}] = " + " + ".join(f"lam[{k}]*sin(u[{(k - 1) % n}])" for k in range(n)) + ";"
    return dict(n=n, np=n + 1, f=f, vjp=vjp, vjp_p=vjp_p)


LV = dict(  # the built-in `lv` re-entered through the runtime path (test/Core3/user_vjp.jl:6-38)
    n=2, np=4,
    f="du[0] = p[0]*u[0] - p[1]*u[0]*u[1]; du[1] = -p[2]*u[1] + p[3]*u[0]*u[1];",
    vjp="out[0] = (p[0] - p[1]*u[1])*lam[0] + p[3]*u[1]*lam[1]; out[1] = -p[1]*u[0]*lam[0] + (-p[2] + p[3]*u[0])*lam[1];",
    vjp_p="const double xy = u[0]*u[1]; out[0] = u[0]*lam[0]; out[1] = -xy*lam[0]; out[2] = -u[1]*lam[1]; out[3] = xy*lam[1];")


AFFINE3 = dict(  # `foo` of the mass-matrix test, test/Core3/adjoint.jl:1315-1321: du = A u + p; du[2] += sum(p)   (oracle: ORC_MODEL_AFFINE3)
    n=3, np=3,
    f=("du[0] = 1.0*u[0] + 2.0*u[1] + 3.0*u[2] + p[0];"
       "du[1] = 4.0*u[0] + 5.0*u[1] + 6.0*u[2] + p[1] + (p[0] + p[1] + p[2]);"
       "du[2] = 7.0*u[0] + 8.0*u[1] + 9.0*u[2] + p[2];"),
    vjp=("out[0] = 1.0*lam[0] + 4.0*lam[1] + 7.0*lam[2]; out[1] = 2.0*lam[0] + 5.0*lam[1] + 8.0*lam[2];"
         "out[2] = 3.0*lam[0] + 6.0*lam[1] + 9.0*lam[2];"),
    vjp_p="out[0] = lam[0] + lam[1]; out[1] = 2.0*lam[1]; out[2] = lam[2] + lam[1];")
AFFINE3_MM = [[-1.0, -2.0, -4.0], [-2.0, -3.0, -7.0], [-1.0, -3.0, -41.0]]   # mm = -[1 2 4; 2 3 7; 1 3 41], test/Core3/adjoint.jl:1322


ROBERDAE = dict(  # `rober` exactly as test/Core3/adjoint.jl:1434-1441 writes it: the third row is the conservation constraint; mass matrix diag(1, 1, 0) (:1450-1454) — oracle: ORC_MODEL_ROBERDAE
    n=3, np=3,
    f="du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = u[0] + u[1] + u[2] - 1.0;",
    vjp=("out[0] = -p[0]*lam[0] + p[0]*lam[1] + lam[2];"
         "out[1] = p[2]*u[2]*lam[0] + (-2.0*p[1]*u[1] - p[2]*u[2])*lam[1] + lam[2];"
         "out[2] = p[2]*u[1]*lam[0] - p[2]*u[1]*lam[1] + lam[2];"),
    vjp_p=("out[0] = -u[0]*lam[0] + u[0]*lam[1]; out[1] = -u[1]*u[1]*lam[1];"
           "out[2] = u[1]*u[2]*lam[0] - u[1]*u[2]*lam[1];"))
ROBERDAE_MM = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]


def roberdae_kappa(kappa=5.0):
    """ROBERDAE with the constraint y1 + y2 + y3 = 1 + kappa (p1 - 0.04) (NOT from the reference; oracle: ROBERDAE with dims[0] = kappa): a constraint that depends on a parameter"""
    m = dict(ROBERDAE)
    m["f"] = ROBERDAE["f"].replace("du[2] = u[0] + u[1] + u[2] - 1.0;", f"du[2] = u[0] + u[1] + u[2] - 1.0 - {kappa!r}*(p[0] - 0.04);")
    m["vjp_p"] = ROBERDAE["vjp_p"].replace("out[0] = -u[0]*lam[0] + u[0]*lam[1];", f"out[0] = -u[0]*lam[0] + u[0]*lam[1] - {kappa!r}*lam[2];")
    assert m["f"] != ROBERDAE["f"] and m["vjp_p"] != ROBERDAE["vjp_p"]
    return m


ROBERDAE_MIX_MD = [[2.0, 0.3], [0.1, 0.5]]
ROBERDAE_MIX_MM = [[2.0, 0.3, 0.0], [0.1, 0.5, 0.0], [0.0, 0.0, 0.0]]
ROBERDAE_MIX_F = ("const real a = -p[0]*u[0] + p[2]*u[1]*u[2]; const real b = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2];"
                  "du[0] = 2.0*a + 0.3*b; du[1] = 0.1*a + 0.5*b; du[2] = u[0] + u[1] + u[2] - 1.0 - 5.0*(p[0] - 0.04);")
"""roberdae_kappa(5) with its differential rows mixed by Md = [2 0.3; 0.1 0.5] and the mass matrix [Md 0; 0 0]: the same trajectory, a non-trivial M'[diff, diff] in the loss jumps
and off-diagonal mass entries in W (oracle: ROBERDAE with dims = (5, 1)).  f only: VJPs by dual numbers."""

# ContinuousCallback problems (oracle: event_kind of oracle/adjoint_oracle.h; test/Callbacks2/continuous_callbacks.jl)
BALL = dict(  # `fiip` of test/Callbacks2/continuous_callbacks.jl:10-14 (oracle: ORC_MODEL_FALLMASS); p2 enters through the affect only
    n=2, np=2,
    f="du[0] = u[1]; du[1] = -p[0];",
    vjp="out[0] = 0.0; out[1] = lam[0];",
    vjp_p="out[0] = -lam[1]; out[1] = 0.0;")
RELAX = dict(  # `f` of the "Re-compile tape" testset, :320 (oracle: ORC_MODEL_RELAX)
    n=1, np=2,
    f="du[0] = p[0] - u[0];",
    vjp="out[0] = -lam[0];",
    vjp_p="out[0] = lam[0]; out[1] = 0.0;")
EVENTS = {  # event_kind -> (model, condition body, affect body)
    1: (BALL, "c = u[0];", "un[1] = -p[1] * u[1];"),                               # :212-217
    2: (BALL, "c = u[0];", "un[0] = u[0] + 3.0; un[1] = u[1] * u[1];"),            # :243-250
    3: (RELAX, "c = u[0] - 0.75 * p[0];", "un[0] = u[0] + p[1];"),                 # :324-327
    7: (BALL, "c = u[0];", "un[1] = -p[1] * u[1]; terminate = true;"),           # :226-236: the affect of event 1 with terminate!(integrator)
    4: (BALL, "c = u[0] - 0.3 * t;", "un[1] = -p[1] * (u[1] - 0.3) + 0.3 + 0.1 * t;"),   # NOT from the reference: explicit t in both
}
BALL2D = dict(  # `f` of test/Callbacks2/vector_continuous_callbacks.jl:10-16 (oracle: ORC_MODEL_BALL2D)
    n=4, np=2,
    f="du[0] = u[1]; du[1] = -p[0]; du[2] = u[3]; du[3] = 0.0;",
    vjp="out[0] = 0.0; out[1] = lam[0]; out[2] = 0.0; out[3] = lam[2];",
    vjp_p="out[0] = -lam[1]; out[1] = 0.0;")
VECTOR_EVENTS = {  # event_kind -> (model, ncond, condition body, affect body)
    5: (BALL2D, 2, "out[0] = u[0]; out[1] = (u[2] - 10.0) * u[2];", "if (idx == 0) un[1] = -p[1] * u[1]; else un[3] = -p[1] * u[3];"),      # :80-96
    6: (BALL2D, 2, "out[0] = sin(t); out[1] = cos(t);", "un[0] = 0.5; un[1] = 1.0; un[2] = 0.0; un[3] = 0.0;"),                          # :100-116
}
PENDULUM = dict(  # `pendulum_eom` of test/Core7/adjoint_param.jl:6-10 (oracle: ORC_MODEL_PENDULUM); p3 enters through the affect of event 8 only
    n=2, np=3,
    f="du[0] = p[0] * u[1]; du[1] = -sin(u[0]) + (-p[0] * sin(u[0]) + p[1] * u[1]);")
EVENTS[8] = (PENDULUM, "c = u[0];", "un[1] = p[2] * u[1];")      # NOT from the reference: an oscillating state, the condition crossed in both directions (the problem for `direction`)
