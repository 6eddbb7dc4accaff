"""Engine: owner of one hipadj handle (include/hipadj.h).  Host (numpy) and device (torch) entry points.
torch is used for device memory and streams only."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import HipadjConfig, HipadjStats, HipadjError


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Engine:
    def __init__(self, model, alg, ntraj, t0, t1, dt, save_times=(), loss_kind=_lib.LOSS_COTANGENT, loss_shift=0.0,
                 checkpointing=False, ckpt_stride=0, quad_abstol=1e-6, quad_reltol=1e-3, no_start=False,
                 p_shared=True, device=0, time_segments=0, dims=(0, 0, 0, 0), cont_cost=0,
                 stepper=0, abstol=1e-6, reltol=1e-3, max_steps=0, checkpoints=None, loss_scale=0.0, devices=None, reference_literal=False, family=0):
        L = _lib.load()
        self._L = L
        self._save = np.ascontiguousarray(np.asarray(save_times, dtype=np.float64))
        c = HipadjConfig()
        c.struct_size = C.sizeof(HipadjConfig)
        c.model, c.alg, c.stepper = _lib.MODEL[model], _lib.ALG[alg], int(stepper)
        for i in range(4):
            c.dims[i] = int(dims[i])
        c.ntraj = int(ntraj)
        c.t0, c.t1, c.dt = float(t0), float(t1), float(dt)
        c.nsave = len(self._save)
        c.save_times = _dptr(self._save) if len(self._save) else None
        c.loss_kind, c.loss_shift = int(loss_kind), float(loss_shift)
        c.checkpointing, c.ckpt_stride = int(bool(checkpointing)), int(ckpt_stride)
        c.quad_abstol, c.quad_reltol = float(quad_abstol), float(quad_reltol)
        c.no_start, c.p_shared, c.device, c.time_segments = int(bool(no_start)), int(bool(p_shared)), int(device), int(time_segments)
        c.cont_cost = int(cont_cost)
        c.max_steps, c.abstol, c.reltol = int(max_steps), float(abstol), float(reltol)
        self._ck = None if checkpoints is None else np.ascontiguousarray(np.asarray(checkpoints, dtype=np.float64))
        c.ncheckpoints = 0 if self._ck is None else len(self._ck)
        c.checkpoints = _dptr(self._ck) if c.ncheckpoints else None
        c.loss_scale = float(loss_scale)
        self._devs = None if devices is None else np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
        c.ndevices = 0 if self._devs is None else len(self._devs)
        c.device_ids = self._devs.ctypes.data_as(C.POINTER(C.c_int32)) if c.ndevices else None
        c.reference_literal = int(bool(reference_literal))
        c.family = int(family)      # hipadj_family: 0 = the library selects the kernel family (a declared dense chain 2-H-H-2 runs on the FP64-MFMA family), 1 = as registered
        self.cfg = c
        self.model, self.alg = model, alg
        self.N, self.M = int(ntraj), len(self._save)
        self.p_shared = bool(p_shared)
        self.device = int(device)
        h = C.c_void_p()
        rc = L.hipadj_create(C.byref(c), C.byref(h))
        if rc != _lib.OK:
            raise HipadjError(rc, L.hipadj_last_error(None).decode())
        self._h = h
        self.forward_generation = 0     # bumped by every forward solve: the handle holds ONE forward solution (interface.EnsembleAdjoint checks it)
        st = self.stats()
        self.n, self.np = st["n"], st["np"]

    # ---- lifetime -------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.hipadj_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != _lib.OK:
            raise HipadjError(rc, self._L.hipadj_last_error(self._h).decode())

    # ---- host-pointer API -----------------------------------------------------------------------------
    def forward(self, u0, p, want_out=True):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        if u0.shape != (self.N, self.n):
            raise ValueError(f"u0 must be [{self.N}][{self.n}], got {u0.shape}")
        if p.shape != ((self.np,) if self.p_shared else (self.N, self.np)):
            raise ValueError(f"p has shape {p.shape}")
        out = np.empty((self.N, self.M, self.n)) if (want_out and self.M) else None
        self.forward_generation += 1
        self._check(self._L.hipadj_forward(self._h, _dptr(u0), _dptr(p), _dptr(out) if out is not None else None))
        return out

    def adjoint(self, dLdu=None):
        if dLdu is not None:
            dLdu = np.ascontiguousarray(dLdu, dtype=np.float64)
            if dLdu.shape != (self.N, self.M, self.n):
                raise ValueError(f"dLdu must be [{self.N}][{self.M}][{self.n}], got {dLdu.shape}")
        du0 = np.empty((self.N, self.n))
        dp = np.empty(self.np if self.p_shared else (self.N, self.np))
        self._check(self._L.hipadj_adjoint(self._h, _dptr(dLdu) if dLdu is not None else None, _dptr(du0), _dptr(dp)))
        from . import problems
        M = problems.WIDE_MASS_MATRICES.get(self.model)
        if M is not None:      # a traced wide model with a mass matrix: the device integrated nu = M' lam (problems.py from_callable); the reference returns lam(t0)
            du0 = np.linalg.solve(M.T, du0.T).T.copy()
        return du0, dp

    # ---- device-resident discrete losses (include/hipadj.h: hipadj_set_loss_data, hipadj_loss_value) ------
    def set_loss_data(self, data):
        """The data block [N][M][n] of HIPADJ_LOSS_LSQ_DATA / of a model's discrete-loss bodies (host array; copied into the handle)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.shape != (self.N, self.M, self.n):
            raise ValueError(f"data must be [{self.N}][{self.M}][{self.n}], got {data.shape}")
        self._check(self._L.hipadj_set_loss_data(self._h, _dptr(data)))

    def set_loss_data_dev(self, data):
        self._check(self._L.hipadj_set_loss_data_dev(self._h, C.c_void_p(data.data_ptr())))

    def loss_value(self, out):
        """The loss summed over the ensemble from the primal output `out` [N][M][n] (hipadj_loss_value)."""
        out = np.ascontiguousarray(out, dtype=np.float64)
        if out.shape != (self.N, self.M, self.n):
            raise ValueError(f"out must be [{self.N}][{self.M}][{self.n}], got {out.shape}")
        v = np.zeros(1)
        self._check(self._L.hipadj_loss_value(self._h, _dptr(out), _dptr(v)))
        return float(v[0])

    def loss_value_dev(self, out, loss):
        self._check(self._L.hipadj_loss_value_dev(self._h, C.c_void_p(out.data_ptr()), C.c_void_p(loss.data_ptr())))

    def soa_stride(self):
        ld = C.c_int64(0)
        self._check(self._L.hipadj_soa_stride(self._h, C.byref(ld)))
        return int(ld.value)

    def adjoint_dev_soa(self, dLdu_soa, du0, dp):
        """Cotangents already in the lane family's streaming layout [M][n][soa_stride()] (hipadj_adjoint_dev_soa): no transposition launch."""
        self._check(self._L.hipadj_adjoint_dev_soa(self._h, C.c_void_p(dLdu_soa.data_ptr()), C.c_void_p(du0.data_ptr()), C.c_void_p(dp.data_ptr())))

    # ---- device-pointer API (torch tensors on cuda:<device>) -------------------------------------------
    def use_torch_stream(self):
        import torch
        self._check(self._L.hipadj_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def forward_dev(self, u0, p, out=None):
        """Asynchronous on the handle's stream; errors raised ON the device (non-finite values, Tsit5 step capacity) surface at the
        next synchronize() — call it before consuming du0 / dp."""
        self.forward_generation += 1
        self._check(self._L.hipadj_forward_dev(self._h, C.c_void_p(u0.data_ptr()), C.c_void_p(p.data_ptr()),
                                               C.c_void_p(out.data_ptr()) if out is not None else None))

    def adjoint_dev(self, dLdu, du0, dp):
        self._check(self._L.hipadj_adjoint_dev(self._h, C.c_void_p(dLdu.data_ptr()) if dLdu is not None else None,
                                               C.c_void_p(du0.data_ptr()), C.c_void_p(dp.data_ptr())))
        from . import problems
        M = problems.WIDE_MASS_MATRICES.get(self.model)
        if M is not None:      # a traced wide model with a mass matrix: the same map as the host-pointer entry point (adjoint above), on the device
            import torch
            self.synchronize()                                   # the handle's stream need not be torch's current one
            MT = torch.as_tensor(np.ascontiguousarray(M.T), dtype=torch.float64, device=du0.device)
            du0.copy_(torch.linalg.solve(MT, du0.T).T)

    # ---- sharded ensembles: dL/dp all-reduce over RCCL inside the library (include/hipadj.h, hipadj_comm_*) ----
    def comm_init_rank(self, unique_id, nranks, rank):
        """Collective: joins the RCCL communicator named by the 128-byte `unique_id` (comm_unique_id() of rank 0, shipped by the
        host); from then on every adjoint call all-reduces dp over the ranks in-stream."""
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.COMM_ID_BYTES} bytes")
        self._check(self._L.hipadj_comm_init_rank(self._h, C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES), int(nranks), int(rank)))

    def comm_attach(self, nccl_comm):
        """Use an existing ncclComm_t (integer address) of the host; None detaches."""
        self._check(self._L.hipadj_comm_attach(self._h, C.c_void_p(nccl_comm)))

    def comm_destroy(self):
        self._check(self._L.hipadj_comm_destroy(self._h))

    def comm_count(self):
        """Ranks of the handle's RCCL communicator (ncclCommCount); 0 = none (dp is this shard's own sum)."""
        n = C.c_int(0)
        self._check(self._L.hipadj_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def comm_selfcheck(self):
        """Collective: all-reduces a known probe the way dp is all-reduced and verifies the sum (raises HipadjError otherwise)."""
        self._check(self._L.hipadj_comm_selfcheck(self._h))

    def comm_overlap(self, on=True):
        """The all-reduce of dp on the handle's second stream, off the next pass's critical path (include/hipadj.h hipadj_comm_overlap): alternate two dp buffers."""
        self._check(self._L.hipadj_comm_overlap(self._h, 1 if on else 0))

    def set_timing(self, level):
        """0: no device events, 1: dominant-kernel bracket only, 2: + whole-call bracket (default)."""
        self._check(self._L.hipadj_set_timing(self._h, int(level)))

    def synchronize(self):
        self._check(self._L.hipadj_synchronize(self._h))

    def event_counts(self):
        """Events per trajectory of the last forward solve (a model with a ContinuousCallback: DeviceFunction.set_continuous_callback)."""
        out = np.zeros(self.N, dtype=np.int32)
        self._check(self._L.hipadj_event_counts(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def event_states(self, max_events=64):
        """save_positions = (true, true): (t [N][max_events], u_left, u_right [N][max_events][n], counts [N]) of the last forward solve — the event times and the states just
        before / after the affect (zero beyond a trajectory's count).  `max_events` as given to set_continuous_callback (0 / default: 64)."""
        me = int(max_events) or 64
        t = np.zeros((self.N, me)); ul = np.zeros((self.N, me, self.n)); ur = np.zeros((self.N, me, self.n))
        self._check(self._L.hipadj_event_states(self._h, t.ctypes.data_as(C.c_void_p), ul.ctypes.data_as(C.c_void_p), ur.ctypes.data_as(C.c_void_p)))
        return t, ul, ur, self.event_counts()

    def event_components(self, max_events=64):
        """Which component of a VectorContinuousCallback fired at each event (the reference's event_idx): [N][max_events] ints, 0 for a scalar condition, + 256 when the event
        terminated the trajectory, -1 beyond a trajectory's count."""
        idx = np.zeros((self.N, int(max_events) or 64), dtype=np.int32)
        self._check(self._L.hipadj_event_components(self._h, idx.ctypes.data_as(C.c_void_p)))
        return idx

    def set_event_cotangents(self, dl=None, dr=None):
        """Cotangents of the caller's loss at the saved event states, [N][max_events][n] each (None = zero; both None removes them), for the following adjoint calls."""
        self._ev_cot = tuple(None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (dl, dr))
        P = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._check(self._L.hipadj_set_event_cotangents(self._h, P(self._ev_cot[0]), P(self._ev_cot[1])))

    def stats(self):
        st = HipadjStats()
        st.struct_size = C.sizeof(HipadjStats)
        self._check(self._L.hipadj_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in HipadjStats._fields_}
