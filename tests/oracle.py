"""ctypes binding of oracle/liboracle_adjoint.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package never does).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle_adjoint.so")

MODEL = dict(LV=0, LVT=1, LORENZ=2, LINDIAG=3, FALLMASS=4, MLP=5, BRUSS=6, ROBER=7, RING=8, AFFINE3=9, IDXAFF=10, MLP1=11, DENSELIN=12, PENDULUM=13, LIN1P=14, ROBERDAE=15, RELAX=16, BALL2D=17)
ALG = dict(INTERPOLATING=0, BACKSOLVE=1, GAUSS=2, QUADRATURE=3, GAUSS_KRONROD=4)
STEPPER = dict(RK4=0, TSIT5=1, ETDRK4=2, ROS23=3)
LOSS = dict(COTANGENT=0, LSQ_SHIFT=1, LSQ_DATA=2, TEST=3)


class OrcConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("alg", C.c_int), ("stepper", C.c_int),
        ("dims", C.c_int * 4),
        ("t0", C.c_double), ("t1", C.c_double), ("dt", C.c_double),
        ("abstol", C.c_double), ("reltol", C.c_double),
        ("nsave", C.c_int), ("save_times", C.POINTER(C.c_double)),
        ("loss_kind", C.c_int), ("loss_shift", C.c_double),
        ("checkpointing", C.c_int), ("nckpt", C.c_int), ("checkpoints", C.POINTER(C.c_double)),
        ("quad_abstol", C.c_double), ("quad_reltol", C.c_double),
        ("no_start", C.c_int), ("cont_cost", C.c_int),
        ("loss_scale", C.c_double), ("dloss_id", C.c_int), ("reference_literal", C.c_int), ("event_kind", C.c_int),
        ("ev_max", C.c_int), ("ev_dl", C.POINTER(C.c_double)), ("ev_dr", C.POINTER(C.c_double)), ("event_dir", C.c_int),
    ]


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "adjoint_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_model_sizes.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_forward.argtypes = [C.POINTER(OrcConfig), dp, dp, dp, C.POINTER(C.c_long)]
        L.orc_adjoint.argtypes = [C.POINTER(OrcConfig), dp, dp, dp, dp, dp, dp, C.POINTER(C.c_long)]
        L.orc_adjoint_ensemble.argtypes = [C.POINTER(OrcConfig), C.c_long, dp, dp, C.c_int, dp, dp, dp, dp,
                                           C.c_int, dp, dp]
        L.orc_model_f.argtypes = [C.c_int, C.POINTER(C.c_int), dp, dp, C.c_double, dp]
        L.orc_model_vjp.argtypes = [C.c_int, C.POINTER(C.c_int), dp, dp, dp, C.c_double, dp, dp]
        L.orc_set_mass_matrix.argtypes = [C.c_int, dp]
        L.orc_test_quadgk_poly.restype = C.c_double
        L.orc_test_quadgk_poly.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.POINTER(C.c_long)]
        L.orc_test_tsit5_order_residual.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class mass_matrix:
    """with oracle.mass_matrix(M): ...  — every oracle solve inside runs M u' = f (orc_set_mass_matrix; process-wide)."""

    def __init__(self, M):
        self.M = _arr(M)

    def __enter__(self):
        n = self.M.shape[0]
        rc = lib().orc_set_mass_matrix(n, _p(self.M))
        if rc:
            raise ValueError("singular (or oversized) mass matrix")
        return self

    def __exit__(self, *a):
        lib().orc_set_mass_matrix(0, None)


def model_sizes(model, dims=(0, 0, 0, 0)):
    n, npar = C.c_int(), C.c_int()
    d = (C.c_int * 4)(*dims)
    rc = lib().orc_model_sizes(MODEL[model], d, C.byref(n), C.byref(npar))
    if rc:
        raise ValueError("bad model")
    return n.value, npar.value


class Problem:
    """Keyword mirror of adjoint_sensitivities(sol, alg; t, dgdu_discrete, sensealg, checkpoints, ...)."""

    def __init__(self, model, alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=1.0, dt=0.01, abstol=1e-6,
                 reltol=1e-3, save_times=(), loss="COTANGENT", loss_shift=0.0, checkpointing=False,
                 checkpoints=None, quad_abstol=1e-6, quad_reltol=1e-3, no_start=False, dims=(0, 0, 0, 0), cont_cost=0, loss_scale=0.0, dloss_id=0,
                 reference_literal=False, event_kind=0, event_dir=0):
        self.model = model
        self.dims = tuple(dims)
        self.n, self.np = model_sizes(model, dims)
        self._save = _arr(np.asarray(save_times, dtype=np.float64))
        self._ck = _arr(np.asarray(checkpoints, dtype=np.float64)) if checkpoints is not None else None
        c = OrcConfig()
        c.model, c.alg, c.stepper = MODEL[model], ALG[alg], STEPPER[stepper]
        for i in range(4):
            c.dims[i] = int(dims[i])
        c.t0, c.t1, c.dt, c.abstol, c.reltol = t0, t1, dt, abstol, reltol
        c.nsave = len(self._save)
        c.save_times = _p(self._save)
        c.loss_kind, c.loss_shift = LOSS[loss], loss_shift
        c.checkpointing = int(checkpointing)
        c.nckpt = 0 if self._ck is None else len(self._ck)
        c.checkpoints = _p(self._ck) if self._ck is not None else None
        c.quad_abstol, c.quad_reltol = quad_abstol, quad_reltol
        c.no_start = int(no_start)
        c.cont_cost = int(cont_cost)
        c.loss_scale, c.dloss_id, c.reference_literal = float(loss_scale), int(dloss_id), int(bool(reference_literal))
        c.event_kind = int(event_kind)      # ContinuousCallback of adjoint_oracle.h
        c.event_dir = int(event_dir)        # 0 both directions, +1 upcrossings only, -1 downcrossings only
        self.cfg = c

    def set_event_cotangents(self, dl=None, dr=None):
        """cotangents of a loss on the saved event states (save_positions = (true, true)): dl / dr [ev_max][n] at the state before / after the affect of event k"""
        self._evd = (None if dl is None else _arr(np.asarray(dl, dtype=np.float64)), None if dr is None else _arr(np.asarray(dr, dtype=np.float64)))
        k = [x.shape[0] for x in self._evd if x is not None]
        self.cfg.ev_max = min(k) if k else 0
        self.cfg.ev_dl = _p(self._evd[0]) if self._evd[0] is not None else None
        self.cfg.ev_dr = _p(self._evd[1]) if self._evd[1] is not None else None
        return self

    def event_states(self, u0, p, cap=64):
        u0, p = _arr(u0), _arr(p)
        t, ul, ur = np.zeros(cap), np.zeros((cap, self.n)), np.zeros((cap, self.n))
        ne = lib().orc_event_states(C.byref(self.cfg), _p(u0), _p(p), cap, _p(t), _p(ul), _p(ur))
        if ne < 0:
            raise RuntimeError(f"orc_event_states rc={ne}")
        return t[:ne], ul[:ne], ur[:ne]

    @property
    def M(self):
        return len(self._save)

    def forward(self, u0, p):
        u0, p = _arr(u0), _arr(p)
        out = np.zeros((self.M, self.n))
        ns = C.c_long()
        rc = lib().orc_forward(C.byref(self.cfg), _p(u0), _p(p), _p(out), C.byref(ns))
        if rc:
            raise RuntimeError(f"orc_forward rc={rc}")
        return out, ns.value

    def adjoint(self, u0, p, dLdu=None):
        u0, p, dLdu = _arr(u0), _arr(p), _arr(dLdu)
        du0, dp, out = np.zeros(self.n), np.zeros(self.np), np.zeros((self.M, self.n))
        nr = C.c_long()
        rc = lib().orc_adjoint(C.byref(self.cfg), _p(u0), _p(p), _p(dLdu), _p(du0), _p(dp), _p(out), C.byref(nr))
        if rc:
            raise RuntimeError(f"orc_adjoint rc={rc}")
        return du0, dp, out

    def adjoint_ensemble(self, u0, p, dLdu=None, nthreads=0, want_out=True):
        u0, p, dLdu = _arr(u0), _arr(p), _arr(dLdu)
        N = u0.shape[0]
        p_shared = int(p.ndim == 1)
        du0 = np.zeros((N, self.n))
        dp = np.zeros(self.np) if p_shared else np.zeros((N, self.np))
        out = np.zeros((N, self.M, self.n)) if want_out else None
        tf, tr = C.c_double(), C.c_double()
        rc = lib().orc_adjoint_ensemble(C.byref(self.cfg), N, _p(u0), _p(p), p_shared, _p(dLdu), _p(du0), _p(dp),
                                        _p(out), nthreads, C.byref(tf), C.byref(tr))
        if rc:
            raise RuntimeError(f"orc_adjoint_ensemble rc={rc}")
        return du0, dp, out, dict(forward_s=tf.value, reverse_s=tr.value)


def model_f(model, u, p, t=0.0, dims=(0, 0, 0, 0)):
    u, p = _arr(u), _arr(p)
    du = np.zeros_like(u)
    lib().orc_model_f(MODEL[model], (C.c_int * 4)(*dims), _p(u), _p(p), t, _p(du))
    return du


def model_vjp(model, lam, u, p, t=0.0, dims=(0, 0, 0, 0)):
    lam, u, p = _arr(lam), _arr(u), _arr(p)
    dlam, dgrad = np.zeros_like(u), np.zeros_like(p)
    lib().orc_model_vjp(MODEL[model], (C.c_int * 4)(*dims), _p(lam), _p(u), _p(p), t, _p(dlam), _p(dgrad))
    return dlam, dgrad
