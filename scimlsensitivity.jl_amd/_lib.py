"""ctypes binding of libhipadj.so (include/hipadj.h).  Fails loudly when the HIP library is missing:
there is no CPU or PyTorch fallback behind this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPADJ_LIBRARY") or os.path.join(HERE, "libhipadj.so")   # HIPADJ_LIBRARY: another build of the same ABI (A/B runs; the Julia binding reads the same variable)

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NONFINITE, ERR_STATE, ERR_UNSUPPORTED, ERR_MAXITERS, ERR_RCCL = 0, -1, -2, -3, -4, -5, -6, -7, -8
COMM_ID_BYTES = 128
MODEL_USER_BASE = 1000

MODEL = dict(lv=0, lvt=1, lorenz=2, lindiag=3, fallmass=4, mlp=5, bruss=6)
ALG = dict(interpolating=0, backsolve=1, gauss=2, quadrature=3, gausskronrod=4)
LOSS_COTANGENT, LOSS_LSQ_SHIFT, LOSS_LSQ_DATA, LOSS_MODEL = 0, 1, 2, 3
CCOST_NONE, CCOST_HALF_SQ_SUM, CCOST_U1SQ_PLUS_P1, CCOST_MODEL = 0, 1, 2, 3

DECLARED_SYMBOLS = (
    "hipadj_version", "hipadj_status_string", "hipadj_last_error", "hipadj_model_sizes", "hipadj_create",
    "hipadj_destroy", "hipadj_forward", "hipadj_adjoint", "hipadj_forward_dev", "hipadj_adjoint_dev",
    "hipadj_set_stream", "hipadj_synchronize", "hipadj_set_timing", "hipadj_get_stats",
    "hipadj_model_register", "hipadj_wmodel_register", "hipadj_model_check", "hipadj_model_check_config", "hipadj_runtime_compiler", "hipadj_model_set_cost", "hipadj_model_set_cost_function", "hipadj_wmodel_set_cost", "hipadj_model_set_mass_matrix", "hipadj_model_set_affect", "hipadj_wmodel_set_affect", "hipadj_affect_apply", "hipadj_affect_vjp",
    "hipadj_comm_unique_id", "hipadj_comm_init_rank", "hipadj_comm_attach", "hipadj_comm_destroy",
    "hipadj_comm_count", "hipadj_comm_selfcheck", "hipadj_comm_overlap",
    "hipadj_model_set_discrete_loss", "hipadj_model_set_discrete_loss_function", "hipadj_wmodel_set_discrete_loss", "hipadj_set_loss_data", "hipadj_set_loss_data_dev",
    "hipadj_loss_value", "hipadj_loss_value_dev", "hipadj_adjoint_dev_soa", "hipadj_soa_stride", "hipadj_device_count", "hipadj_wmodel_declare_dense_chain",
    "hipadj_model_set_continuous_callback", "hipadj_model_set_vector_continuous_callback", "hipadj_model_set_callback_direction", "hipadj_event_counts", "hipadj_event_states", "hipadj_event_components", "hipadj_set_event_cotangents",
)


class HipadjConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("model", C.c_int32), ("alg", C.c_int32), ("stepper", C.c_int32),
        ("dims", C.c_int32 * 4), ("ntraj", C.c_int64),
        ("t0", C.c_double), ("t1", C.c_double), ("dt", C.c_double),
        ("nsave", C.c_int32), ("save_times", C.POINTER(C.c_double)),
        ("loss_kind", C.c_int32), ("loss_shift", C.c_double),
        ("checkpointing", C.c_int32), ("ckpt_stride", C.c_int32),
        ("quad_abstol", C.c_double), ("quad_reltol", C.c_double),
        ("no_start", C.c_int32), ("p_shared", C.c_int32), ("device", C.c_int32), ("time_segments", C.c_int32),
        ("cont_cost", C.c_int32), ("max_steps", C.c_int32),
        ("abstol", C.c_double), ("reltol", C.c_double),
        ("ncheckpoints", C.c_int32), ("checkpoints", C.POINTER(C.c_double)),
        ("loss_scale", C.c_double), ("ndevices", C.c_int32), ("device_ids", C.POINTER(C.c_int32)),
        ("reference_literal", C.c_int32), ("family", C.c_int32),
    ]


class HipadjStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n", C.c_int32), ("np", C.c_int32),
        ("ntraj", C.c_int64), ("nsteps", C.c_int64), ("time_segments", C.c_int32),
        ("forward_ms_last", C.c_double), ("adjoint_ms_last", C.c_double),
        ("forward_ms_total", C.c_double), ("adjoint_ms_total", C.c_double),
        ("forward_calls", C.c_int64), ("adjoint_calls", C.c_int64),
        ("adjoint_main_kernel_ms_last", C.c_double), ("adjoint_main_kernel_ms_total", C.c_double),
        ("adjoint_algorithmic_bytes", C.c_double), ("vjp_steps", C.c_double), ("workspace_bytes", C.c_double),
        ("launches_per_pass", C.c_int32), ("routed_family", C.c_int32),
    ]


class HipadjError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"hipadj status {status}: {message}")
        self.status = status


_lib = None


def load():
    """dlopen libhipadj.so.  Raises (never falls back) when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the gfx950 extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "This package has no CPU fallback.")
    # torch ships its own HIP runtime: when torch is going to be used for device memory / streams it must
    # initialise first, otherwise the process ends up with a second runtime that sees no GPU.
    # HIPADJ_NO_TORCH=1: a host without torch in the process (what the Julia glue would be): libhipadj.so then runs on the HIP
    # runtime / hiprtc / RCCL of the ROCm installation alone.
    if os.environ.get("HIPADJ_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    L.hipadj_version.restype = C.c_int
    L.hipadj_status_string.restype = C.c_char_p
    L.hipadj_status_string.argtypes = [C.c_int]
    L.hipadj_last_error.restype = C.c_char_p
    L.hipadj_last_error.argtypes = [vp]
    L.hipadj_model_sizes.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.hipadj_create.argtypes = [C.POINTER(HipadjConfig), C.POINTER(vp)]
    L.hipadj_destroy.argtypes = [vp]
    L.hipadj_forward.argtypes = [vp, dp, dp, dp]
    L.hipadj_adjoint.argtypes = [vp, dp, dp, dp]
    L.hipadj_forward_dev.argtypes = [vp, vp, vp, vp]
    L.hipadj_adjoint_dev.argtypes = [vp, vp, vp, vp]
    L.hipadj_set_stream.argtypes = [vp, vp]
    L.hipadj_synchronize.argtypes = [vp]
    L.hipadj_set_timing.argtypes = [vp, C.c_int]
    L.hipadj_get_stats.argtypes = [vp, C.POINTER(HipadjStats)]
    L.hipadj_model_register.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int32)]
    L.hipadj_wmodel_register.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.POINTER(C.c_int32)]
    L.hipadj_model_check.argtypes = [C.c_int32]
    L.hipadj_wmodel_declare_dense_chain.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32]
    L.hipadj_model_check_config.argtypes = [C.POINTER(HipadjConfig)]
    L.hipadj_runtime_compiler.argtypes = [C.c_char_p, C.c_int32]
    L.hipadj_model_set_cost.argtypes = [C.c_int32, C.c_char_p, C.c_char_p]
    L.hipadj_model_set_cost_function.argtypes = [C.c_int32, C.c_char_p]
    L.hipadj_wmodel_set_cost.argtypes = [C.c_int32, C.c_char_p]
    L.hipadj_model_set_mass_matrix.argtypes = [C.c_int32, C.POINTER(C.c_double)]
    L.hipadj_model_set_affect.argtypes = [C.c_int32, C.c_char_p]
    L.hipadj_model_set_continuous_callback.argtypes = [C.c_int32, C.c_char_p, C.c_char_p, C.c_int32]
    L.hipadj_event_counts.argtypes = [C.c_void_p, C.c_void_p]
    L.hipadj_model_set_vector_continuous_callback.argtypes = [C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32]
    L.hipadj_model_set_callback_direction.argtypes = [C.c_int32, C.c_int32]
    L.hipadj_event_states.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hipadj_set_event_cotangents.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hipadj_event_components.argtypes = [C.c_void_p, C.c_void_p]
    L.hipadj_wmodel_set_affect.argtypes = [C.c_int32, C.c_char_p, C.c_char_p]
    L.hipadj_affect_apply.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.hipadj_affect_vjp.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_double, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.hipadj_comm_unique_id.argtypes = [C.c_char_p]
    L.hipadj_comm_init_rank.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.hipadj_comm_attach.argtypes = [vp, vp]
    L.hipadj_comm_destroy.argtypes = [vp]
    L.hipadj_comm_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.hipadj_comm_selfcheck.argtypes = [vp]
    L.hipadj_comm_overlap.argtypes = [vp, C.c_int]
    L.hipadj_model_set_discrete_loss.argtypes = [C.c_int32, C.c_char_p, C.c_char_p]
    L.hipadj_model_set_discrete_loss_function.argtypes = [C.c_int32, C.c_char_p]
    L.hipadj_wmodel_set_discrete_loss.argtypes = [C.c_int32, C.c_char_p]
    L.hipadj_set_loss_data.argtypes = [vp, dp]
    L.hipadj_set_loss_data_dev.argtypes = [vp, vp]
    L.hipadj_loss_value.argtypes = [vp, dp, dp]
    L.hipadj_loss_value_dev.argtypes = [vp, vp, vp]
    L.hipadj_adjoint_dev_soa.argtypes = [vp, vp, vp, vp]
    L.hipadj_soa_stride.argtypes = [vp, C.POINTER(C.c_int64)]
    L.hipadj_device_count.restype = C.c_int
    _lib = L
    return L


def model_sizes(model, dims=(0, 0, 0, 0)):
    n, npar = C.c_int32(), C.c_int32()
    rc = load().hipadj_model_sizes(MODEL[model], (C.c_int32 * 4)(*dims), C.byref(n), C.byref(npar))
    if rc != OK:
        raise HipadjError(rc, f"unknown model {model!r} / dims {dims}")
    return n.value, npar.value


def register_model(name, n, npar, f, vjp=None, vjp_p=None, check=False):
    """hipadj_model_register: runtime ingestion of a right-hand side and its two VJPs (HIP C++ bodies, see include/hipadj.h).
    Adds `name` to MODEL and returns the model id; check=True compiles for gfx950 immediately (no device needed)."""
    L = load()
    mid = C.c_int32()
    enc = lambda b: None if b is None else b.encode()
    rc = L.hipadj_model_register(name.encode(), int(n), int(npar), f.encode(), enc(vjp), enc(vjp_p), C.byref(mid))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())
    MODEL[name] = mid.value
    if check:
        check_model(mid.value)
    return mid.value


def register_wide_model(name, n, npar, f, vjp, threads=0, lds_doubles=0, nacc=0, acc_first=0, check=False):
    """hipadj_wmodel_register: a model of the workgroup-per-trajectory family (more than 8 states or 32 parameters; include/hipadj.h):
    the SPMD bodies of f and of the joint VJP.  Adds `name` to MODEL and returns the model id."""
    L = load()
    mid = C.c_int32()
    rc = L.hipadj_wmodel_register(name.encode(), int(n), int(npar), int(threads), int(lds_doubles), int(nacc), int(acc_first), f.encode(), vjp.encode(), C.byref(mid))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())
    MODEL[name] = mid.value
    if check:
        check_model(mid.value)
    return mid.value


FAMILY_AUTO, FAMILY_AS_REGISTERED, FAMILY_MFMA, ACT_TANH = 0, 1, 3, 1      # hipadj_family / hipadj_activation


def declare_dense_chain(model_id, widths, input_power=1):
    """hipadj_wmodel_declare_dense_chain: the wide model IS this tanh chain — hipadj_create then selects the kernel family itself (widths = None withdraws the declaration)."""
    L = load()
    if widths is None:
        rc = L.hipadj_wmodel_declare_dense_chain(int(model_id), None, 0, ACT_TANH, 1)
    else:
        w = (C.c_int32 * len(widths))(*[int(x) for x in widths])
        rc = L.hipadj_wmodel_declare_dense_chain(int(model_id), w, len(widths), ACT_TANH, int(input_power))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def runtime_compiler():
    """hipadj_runtime_compiler: '<libhiprtc path> [own link-map namespace]; HIP x.y.z' — the compiler of the runtime-registered models
    (the build toolkit's, also inside a torch process whose wheel bundles an older ROCm)."""
    L = load()
    buf = C.create_string_buffer(1024)
    rc = L.hipadj_runtime_compiler(buf, 1024)
    if rc != OK:
        raise HipadjError(rc, buf.value.decode())
    return buf.value.decode()


def check_model(model_id):
    """hipadj_model_check: compile the forward and InterpolatingAdjoint kernels of a registered model for gfx950 (no device needed)."""
    L = load()
    rc = L.hipadj_model_check(int(model_id))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_model_affect(model_id, body):
    """hipadj_model_set_affect: the DiscreteCallback affect of a runtime-registered model (None removes it)."""
    L = load()
    rc = L.hipadj_model_set_affect(int(model_id), None if body is None else body.encode())
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_model_callback_direction(model_id, direction):
    """hipadj_model_set_callback_direction: +1 only upcrossings fire (affect_neg! = nothing), -1 only downcrossings, 0 both."""
    L = load()
    rc = L.hipadj_model_set_callback_direction(int(model_id), int(direction))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_model_continuous_callback(model_id, condition, affect, max_events=0, ncond=1):
    """hipadj_model_set_continuous_callback / hipadj_model_set_vector_continuous_callback (ncond > 1): the callback of a runtime lane model (None, None removes it)."""
    L = load()
    enc = lambda s: None if s is None else s.encode()
    if int(ncond) > 1:
        rc = L.hipadj_model_set_vector_continuous_callback(int(model_id), int(ncond), enc(condition), enc(affect), int(max_events))
    else:
        rc = L.hipadj_model_set_continuous_callback(int(model_id), enc(condition), enc(affect), int(max_events))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_wide_model_affect(model_id, body, vjp_body):
    """hipadj_wmodel_set_affect: the affect of a wide model and its reverse callback, both as serial text (None, None removes them)."""
    L = load()
    rc = L.hipadj_wmodel_set_affect(int(model_id), None if body is None else body.encode(), None if vjp_body is None else vjp_body.encode())
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def affect_apply(model_id, u, p, t, npar, device=0):
    """hipadj_affect_apply: (u_out[i], p_out[i]) = a(u[i], p, t) on the device; u [N][n], p [np] or [N][np] (host arrays in and out; p_out [N][np])."""
    import numpy as np
    L = load()
    u = np.ascontiguousarray(u, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(u); pout = np.empty((u.shape[0], int(npar)))
    dp_ = C.POINTER(C.c_double)
    rc = L.hipadj_affect_apply(int(model_id), int(device), u.shape[0], u.ctypes.data_as(dp_), p.ctypes.data_as(dp_), int(p.ndim == 1), float(t), out.ctypes.data_as(dp_),
                               pout.ctypes.data_as(dp_))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())
    return out, pout


def affect_vjp(model_id, u, p, t, lam, gp, device=0):
    """hipadj_affect_vjp: the reverse callback of the map (u, p) -> (un, pn) at the left state: (lam_out, gp_out) from (lam, gp [N][np])."""
    import numpy as np
    L = load()
    u = np.ascontiguousarray(u, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64); lam = np.ascontiguousarray(lam, dtype=np.float64)
    gp = np.ascontiguousarray(gp, dtype=np.float64)
    lo = np.empty_like(u); g = np.empty_like(gp)
    dp_ = C.POINTER(C.c_double)
    rc = L.hipadj_affect_vjp(int(model_id), int(device), u.shape[0], u.ctypes.data_as(dp_), p.ctypes.data_as(dp_), int(p.ndim == 1), float(t), lam.ctypes.data_as(dp_),
                             gp.ctypes.data_as(dp_), lo.ctypes.data_as(dp_), g.ctypes.data_as(dp_))
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())
    return lo, g


MASS = {}     # model id -> mass matrix (numpy [n][n]) or absent: the host mirror needs it where it chains pieces (events.py)


def set_model_mass_matrix(model_id, n, M):
    """hipadj_model_set_mass_matrix: ODEFunction(f; mass_matrix = M) for a runtime-registered model (constant, non-singular; None removes)."""
    L = load()
    if M is None:
        rc = L.hipadj_model_set_mass_matrix(int(model_id), None)
        if rc == OK:
            MASS.pop(int(model_id), None)
    else:
        import numpy as np
        A = np.ascontiguousarray(M, dtype=np.float64)
        if A.shape != (n, n):
            raise ValueError(f"mass_matrix must be {n} x {n}, got {A.shape}")
        rc = L.hipadj_model_set_mass_matrix(int(model_id), A.ctypes.data_as(C.POINTER(C.c_double)))
        if rc == OK:
            MASS[int(model_id)] = A.copy()
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_model_discrete_loss(model_id, dgdu=None, dgdp=None, l=None, wide=None):
    """hipadj_model_set_discrete_loss[_function] / hipadj_wmodel_set_discrete_loss: the discrete loss of a runtime-registered model as device text —
    its gradient bodies (dgdu, optional dgdp), the loss itself (l: gradients by dual numbers, and hipadj_loss_value can return the loss), or, for a
    wide model, one SPMD body."""
    L = load()
    if wide is not None:
        rc = L.hipadj_wmodel_set_discrete_loss(int(model_id), wide.encode() if wide else None)
    elif l is not None:
        rc = L.hipadj_model_set_discrete_loss_function(int(model_id), l.encode())
    else:
        rc = L.hipadj_model_set_discrete_loss(int(model_id), None if dgdu is None else dgdu.encode(), None if dgdp is None else dgdp.encode())
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())


def set_model_cost(model_id, dgdu=None, dgdp=None, g=None):
    """hipadj_model_set_cost / hipadj_model_set_cost_function: attach a continuous cost to a runtime-registered model, either
    through its gradient bodies or through the cost itself (gradients by dual numbers)."""
    L = load()
    if g is not None:
        rc = L.hipadj_model_set_cost_function(int(model_id), g.encode())
    else:
        rc = L.hipadj_model_set_cost(int(model_id), dgdu.encode(), dgdp.encode())
    if rc != OK:
        raise HipadjError(rc, L.hipadj_last_error(None).decode())
