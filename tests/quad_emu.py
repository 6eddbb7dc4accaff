"""ctypes binding of tests/emu/libquad_emu*.so — TEST-ONLY host emulation of the four-lanes-per-trajectory bodies (tests/emu/quad_emu.cpp: four host threads per quad,
DPP quad_perm as a barrier exchange).  Never imported by the product package."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, _ROOT)
_SRC = os.path.join(_HERE, "emu", "quad_emu.cpp")
_CSRC = os.path.join(_ROOT, "scimlsensitivity.jl_amd", "csrc")
_libs = {}


def lib(gauss_nz=2):
    """gauss_nz = 1 builds the one-component Gauss instantiation (-DHIPADJ_QUAD_GAUSS_NZ=1) the device returned wrong numbers from."""
    if gauss_nz not in _libs:
        path = os.path.join(_HERE, "emu", f"libquad_emu_nz{gauss_nz}.so")
        deps = [_SRC] + [os.path.join(_CSRC, f) for f in ("hipadj_quad.hpp", "hipadj_quad_ts5.hpp", "hipadj_adaptive.hpp", "hipadj_plan.hpp", "hipadj_lane.hpp", "hipadj_models.hpp")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-shared", f"-DHIPADJ_QUAD_GAUSS_NZ={gauss_nz}",
                                   "-I" + os.path.join(_ROOT, "include"), _SRC, "-o", path + ".tmp"])
            os.replace(path + ".tmp", path)
        L = C.CDLL(path)
        L.quad_emu_last_error.restype = C.c_char_p
        assert L.quad_emu_gauss_nz() == gauss_nz
        _libs[gauss_nz] = L
    return _libs[gauss_nz]


def forward_adjoint(cfg, u0, p, dLdu=None, gauss_nz=2):
    """(du0 [N][n], dp, out [N][M][n], forward step counts) of a lorenz / lv / lvt ensemble through the quad bodies."""
    L = lib(gauss_nz)
    u0 = np.ascontiguousarray(u0, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    N, n, M = u0.shape[0], u0.shape[1], cfg.nsave
    npar = p.shape[-1]
    du0 = np.zeros((N, n)); dp = np.zeros(npar if cfg.p_shared else (N, npar)); out = np.zeros((N, M, n)); ns = np.zeros(N, dtype=np.int32)
    d = None if dLdu is None else np.ascontiguousarray(dLdu, dtype=np.float64)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = L.quad_emu_forward_adjoint(C.byref(cfg), P(u0), P(p), P(d) if d is not None else None, P(du0), P(dp), P(out), ns.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise RuntimeError(f"quad emulator rc={rc}: {L.quad_emu_last_error().decode()}")
    return du0, dp, out, ns


def forward_rk4(cfg, u0, p):
    """(knots [N][S + 1][2][3] = (u_k, f(u_k)), out [N][M][3], y(T) [N][3]) of the fixed-step forward solve through forward_quad_ev."""
    L = lib(2)
    u0 = np.ascontiguousarray(u0, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    N, M = u0.shape[0], cfg.nsave
    S = int(round((cfg.t1 - cfg.t0) / cfg.dt))
    knots = np.zeros((N, S + 1, 2, 3)); out = np.zeros((N, M, 3)); yT = np.zeros((N, 3))
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = L.quad_emu_forward_rk4(C.byref(cfg), P(u0), P(p), P(knots), P(out), P(yT))
    if rc:
        raise RuntimeError(f"quad emulator rc={rc}: {L.quad_emu_last_error().decode()}")
    return knots, out, yT
