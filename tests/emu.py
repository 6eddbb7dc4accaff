"""ctypes binding of tests/emu/liblane_emu.so — TEST-ONLY host emulation of the device lane bodies
(see tests/emu/lane_emu.cpp).  Never imported by the product package."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, _ROOT)
_SRC = os.path.join(_HERE, "emu", "lane_emu.cpp")
_LIB = os.path.join(_HERE, "emu", "liblane_emu.so")
_CSRC = os.path.join(_ROOT, "scimlsensitivity.jl_amd", "csrc")


def build(force=False):
    """Seventeen translation units (-DEMU_UNIT=0..16: entry points + one unit per model) compiled in parallel, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in ("hipadj_lane.hpp", "hipadj_models.hpp", "hipadj_plan.hpp", "hipadj_adaptive.hpp")]
    def stale():
        return force or not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps)
    if not stale():
        return _LIB
    import fcntl
    objdir = os.path.join(_HERE, "emu", "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lk:      # pytest-xdist workers: one builds, the others wait and find the library fresh
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not stale():
            return _LIB

        def unit(k):
            obj = os.path.join(objdir, f"unit{k}.o")
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-DEMU_UNIT={k}", "-c", _SRC, "-o", obj])
            return obj
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            objs = list(pool.map(unit, range(17)))
        subprocess.check_call(["g++", "-shared", "-fPIC", "-o", _LIB + ".tmp"] + objs)
        os.replace(_LIB + ".tmp", _LIB)
    return _LIB


_lib = None
_EMU_ONLY = {"emu_ring4": 4, "emu_ring5mm": 105, "emu_rober": 203, "emu_roberdae": 204, "emu_roberdae_kappa": 205, "emu_roberdae_mix": 206,
             "emu_ball": 301, "emu_relax": 303, "emu_ball_moving": 304, "emu_ball2d": 305, "emu_ball_terminate": 307}      # test-only models of lane_emu.cpp (offset from MODEL_USER_BASE)


def ring_mm_inverse(n):
    """M^{-1} of lane_emu.cpp's EmuRingMM<n>."""
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    return 0.8 * (i == j) + 0.15 * np.sin(1.0 + 3.0 * i + 7.0 * j)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.emu_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def make_config(model, alg, ntraj, t0, t1, dt, save_times, loss_kind=0, loss_shift=0.0, checkpointing=False,
                ckpt_stride=0, quad_abstol=1e-6, quad_reltol=1e-3, no_start=False, p_shared=True, time_segments=1, cont_cost=0,
                stepper=0, abstol=1e-6, reltol=1e-3, max_steps=0, checkpoints=None, loss_scale=0.0, reference_literal=False):
    from scimlsensitivity_jl_amd import _lib as PL
    save = np.ascontiguousarray(np.asarray(save_times, dtype=np.float64))
    c = PL.HipadjConfig()
    c.struct_size = C.sizeof(PL.HipadjConfig)
    c.model, c.alg, c.stepper = _EMU_ONLY[model] + PL.MODEL_USER_BASE if model in _EMU_ONLY else PL.MODEL[model], PL.ALG[alg], stepper   # emu_ring4: test-only model of lane_emu.cpp
    c.ntraj = ntraj
    c.t0, c.t1, c.dt = t0, t1, dt
    c.nsave = len(save)
    c.save_times = save.ctypes.data_as(C.POINTER(C.c_double)) if len(save) else None
    c.loss_kind, c.loss_shift, c.loss_scale = loss_kind, loss_shift, loss_scale
    c.reference_literal = int(bool(reference_literal))
    c.checkpointing, c.ckpt_stride = int(checkpointing), ckpt_stride
    c.quad_abstol, c.quad_reltol = quad_abstol, quad_reltol
    c.no_start, c.p_shared, c.device, c.time_segments = int(no_start), int(p_shared), 0, time_segments
    c.cont_cost = cont_cost
    c.max_steps, c.abstol, c.reltol = max_steps, abstol, reltol
    ck = None if checkpoints is None else np.ascontiguousarray(np.asarray(checkpoints, dtype=np.float64))
    c.ncheckpoints = 0 if ck is None else len(ck)
    c.checkpoints = ck.ctypes.data_as(C.POINTER(C.c_double)) if c.ncheckpoints else None
    c._keep = (save, ck)
    return c


EMU_MAXEV = 16      # capacity of the emulator's event lists (lane_emu.cpp)


def set_event_cotangents(dl=None, dr=None):
    """test hook of lane_emu.cpp: cotangents at the saved event states for the following forward_adjoint calls, [N][EMU_MAXEV][n] each (None = zero); returns the arrays to keep alive"""
    keep = tuple(None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (dl, dr))
    P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L = lib(); L.emu_set_event_cotangents.argtypes = [C.c_void_p, C.c_void_p]; L.emu_set_event_cotangents.restype = None
    L.emu_set_event_cotangents(P(keep[0]), P(keep[1]))
    return keep


def set_event_output(N=None, n=None):
    """test hook: the following forward_adjoint calls write [N][EMU_MAXEV][1 + 2 n] = (t, u-, u+) per event into the returned array (None: switch off)"""
    L = lib(); L.emu_set_event_output.argtypes = [C.c_void_p]; L.emu_set_event_output.restype = None
    if N is None:
        L.emu_set_event_output(None); return None
    out = np.zeros((N, EMU_MAXEV, 1 + 2 * n))
    L.emu_set_event_output(out.ctypes.data_as(C.c_void_p))
    return out


def forward_adjoint(cfg, n, npar, u0, p, dLdu=None):
    u0 = np.ascontiguousarray(u0, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    N, M = u0.shape[0], cfg.nsave
    du0 = np.zeros((N, n))
    dp = np.zeros(npar if cfg.p_shared else (N, npar))
    out = np.zeros((N, M, n))
    P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    if dLdu is not None:
        dLdu = np.ascontiguousarray(dLdu, dtype=np.float64)
    rc = lib().emu_forward_adjoint(C.byref(cfg), P(u0), P(p), P(dLdu), P(du0), P(dp), P(out))
    if rc:
        raise RuntimeError(f"emu rc={rc}: {lib().emu_last_error().decode()}")
    return du0, dp, out
