"""Host-side mirror of the reference's entry points for the adjoint hot path.

    solve(ensprob, RK4() | Tsit5(); dt, saveat, sensealg, ...)   forward solve that keeps what the reverse pass needs
                                                                 (src/concrete_solve.jl:689-707)
    adjoint_sensitivities(sol, alg; t, dgdu_discrete, sensealg)  -> (du0, dp)   (src/sensitivity_interface.jl:373-526)
    concrete_solve_adjoint(prob, alg, sensealg, u0, p; ...)      -> (out, pullback)  (src/concrete_solve.jl:523-1042)
    EnsembleAdjoint (torch.autograd.Function)                    outer AD = torch autograd in the role Zygote plays
                                                                 for the reference

Everything numeric happens behind the C ABI (include/hipadj.h) on the GPU.
"""
import numpy as np

from . import _lib
from .engine import Engine
from .problems import (RK4, ETDRK4, Tsit5, Rosenbrock23, ODEProblem, EnsembleProblem, EnsembleSolution, LsqShift, LsqData, ModelLoss, HalfSquaredSum,
                       FirstStateSquaredPlusFirstParam, ModelCost, ContinuousCallback)

_attached_callbacks = {}      # model id -> the ContinuousCallback solve(...) attached last
from .sensitivity_algorithms import (AbstractAdjointSensitivityAlgorithm, InterpolatingAdjoint, BacksolveAdjoint,
                                     QuadratureAdjoint, GaussAdjoint, GaussKronrodAdjoint, ischeckpointing)


_COSTS = {HalfSquaredSum: _lib.CCOST_HALF_SQ_SUM, FirstStateSquaredPlusFirstParam: _lib.CCOST_U1SQ_PLUS_P1, ModelCost: _lib.CCOST_MODEL}


def _save_times(tspan, saveat, dt, save_everystep=False, save_start=True, save_end=True):
    """The times `ts` of the primal output / loss jumps, as the forward pass of _concrete_solve_adjoint forms them
    (src/concrete_solve.jl:713-770):
      saveat a number   ts = t0:saveat:T, and T appended when the span is not a multiple of saveat (fix_endpoints, :725, 2827-2831);
                        save_start / save_end do not trim this list (the solve is dense and `out = sol(ts)`)
      saveat a list     sorted(saveat) (:752)
      saveat empty      every step of the forward solve (save_everystep), minus the first / last point for save_start = false /
                        save_end = false (:740-750); fixed-step only here (an adaptive solve's step times are not known up front)
    `saveat=None` without `save_everystep` means "no discrete loss" (a continuous cost only)."""
    t0, t1 = float(tspan[0]), float(tspan[1])
    if saveat is None:
        if not save_everystep:
            return np.zeros(0)
        if not dt:
            raise ValueError("save_everystep without saveat needs the fixed step dt (RK4())")
        ts = t0 + dt * np.arange(int(round((t1 - t0) / dt)) + 1)
        ts[-1] = t1
        return np.ascontiguousarray(ts[(0 if save_start else 1):(len(ts) if save_end else len(ts) - 1)])
    if np.isscalar(saveat):
        step = abs(float(saveat))
        m = int(np.floor((t1 - t0) / step * (1.0 + 4e-16) + 1e-12))        # length of the range t0:step:T
        ts = t0 + step * np.arange(m + 1)
        if abs(ts[-1] - t1) <= 1e-12 * max(1.0, abs(t1)):
            ts[-1] = t1
        else:
            ts = np.append(ts, t1)
        return ts
    return np.ascontiguousarray(np.sort(np.asarray(saveat, dtype=np.float64)))


def _engine_kwargs(sensealg, checkpoints, dt, t0, adaptive=False, t1=None):
    """sensealg options -> handle configuration.  `checkpoints` (adjoint_sensitivities' keyword, default sol.t; src/sensitivity_interface.jl:
    484-486, src/backsolve_adjoint.jl:132): any strictly ascending list — on the step grid for RK4(), arbitrary for Tsit5(); an equally
    spaced grid list is handed over as a stride (the planner's cheaper form), anything else as the explicit list (ABI 102)."""
    kw = {}
    if isinstance(sensealg, (BacksolveAdjoint, InterpolatingAdjoint, GaussAdjoint, GaussKronrodAdjoint)):
        kw["checkpointing"] = sensealg.checkpointing
        if checkpoints is not None and sensealg.checkpointing:
            ck = np.sort(np.asarray(checkpoints, dtype=np.float64))
            if len(ck) > 1 and np.any(np.diff(ck) <= 0):
                raise ValueError("checkpoints must be distinct")
            if not adaptive and len(ck) > 1:
                ks = np.rint((ck - t0) / dt).astype(np.int64)
                d = np.diff(ks)
                # the stride form means {0, s, 2s, ...} + the end point T: only a list that IS that set may take it ([0, 2, 4, 5] on [0, 10]
                # is not stride 2 — its checkpoints are {0, 2, 4, 5, 10}, not {0, 2, ..., 10})
                S = None if t1 is None else int(np.rint((t1 - t0) / dt))
                on_grid = np.all(np.abs((ck - t0) / dt - ks) < 1e-6)
                if on_grid and S is not None and ks[0] == 0 and np.all(d[:-1] == d[0]) and d[-1] <= d[0]:
                    s_ = int(d[0])
                    if set(ks.tolist()) | {S} == set(range(0, S + 1, s_)) | {S}:
                        kw["ckpt_stride"] = s_
                        return kw
            kw["checkpoints"] = ck
    elif isinstance(sensealg, QuadratureAdjoint):
        kw["quad_abstol"], kw["quad_reltol"] = sensealg.abstol, sensealg.reltol
    return kw


def solve(ensprob, alg=RK4(), *, dt=None, saveat=None, sensealg=InterpolatingAdjoint(), dgdu_discrete=None, checkpoints=None,
          device=0, time_segments=0, no_start=None, want_out=True, g=None, abstol=1e-6, reltol=1e-3, max_steps=0, save_idxs=None,
          save_start=True, save_end=True, save_everystep=False, callback=None, devices=None, reference_literal=False, mfma=None):
    """Forward solve of an EnsembleProblem on the device.  The returned solution owns the device-resident
    interpolant tiles (Interpolating/Gauss/Quadrature) or checkpoints (Backsolve) the reverse pass consumes.
    `dgdu_discrete` may be given here already (LsqShift or None = cotangents) because the fused reverse kernel
    is specialised on it at handle creation; likewise the continuous cost `g` (HalfSquaredSum() or None).
    alg = RK4(): fixed step `dt` (required).  alg = Tsit5() or Rosenbrock23() (the stiff stepper of the lane family): adaptive, `abstol`/`reltol` are used for the forward AND
    the reverse solve (src/sensitivity_interface.jl:432), `dt` is the optional initial-step hint, `saveat` may hold
    arbitrary ascending times, `max_steps` bounds the accepted steps per trajectory (0 = sized automatically by a counting pass of the forward solve).
    `save_idxs` (src/concrete_solve.jl:733-736, 774-824): only those state components appear in `sol.u`; cotangents handed to
    adjoint_sensitivities then have that shape and the other components receive zero (`_out[_save_idxs] .= ...`).
    `save_start` / `save_end` / `save_everystep`: the forward solve's saving flags as _concrete_solve_adjoint reads them
    (`_save_times`); `no_start` defaults to the reference's `!save_start && t0 in ts` (src/concrete_solve.jl:962), which
    suppresses the loss jump at t0.
    `dgdu_discrete` = LsqData(data, scale) or ModelLoss(data): the loss stays on the device (include/hipadj.h HIPADJ_LOSS_LSQ_DATA / HIPADJ_LOSS_MODEL) — the data
    block is handed over once, the reverse pass takes no cotangents and `sol.loss_value()` returns the loss.
    `devices` = a list of HIP ordinals: one handle over several devices (contiguous trajectory ranges; hipadj_config.device_ids).
    `reference_literal`: reproduce the reference's lines where the library deliberately deviates (hipadj_config.reference_literal).
    `mfma`: None — the LIBRARY selects the kernel family (a WideDeviceFunction.dense_chain of shape (2, H, H, 2), H in (32, 64, 128), runs on the FP64-MFMA family:
    hipadj_config.family, csrc/hipadj_route.hpp; `sol.extra["mfma_routed"]` says so); False — the family the model was registered for; True — raise unless routed."""
    if mfma not in (None, True, False):
        raise ValueError("mfma must be None (the library selects the kernel family), True (insist on the FP64-MFMA family) or False (the family the model was registered for)")
    if isinstance(callback, ContinuousCallback):      # a property of the model on the device: events located per trajectory inside the kernels
        mid = _lib.MODEL[ensprob.prob.f]
        if mid < _lib.MODEL_USER_BASE:
            raise ValueError("ContinuousCallback needs a runtime lane model (DeviceFunction): condition and affect are compiled next to its right-hand side")
        if _attached_callbacks.get(mid) != callback:      # (re-attaching bumps the model's revision: a new code object)
            _lib.set_model_continuous_callback(mid, callback.condition, callback.affect, callback.max_events, callback.ncond)
            if callback.direction:
                _lib.set_model_callback_direction(mid, callback.direction)
            _attached_callbacks[mid] = callback
        callback = None
    if callback is not None:       # DiscreteCallback at preset times: a chain of ordinary pieces (events.py)
        from . import events
        return events.solve_with_events(solve, _save_times, ensprob, alg, callback, dt=dt, saveat=saveat, sensealg=sensealg, dgdu_discrete=dgdu_discrete, checkpoints=checkpoints,
                                        device=device, time_segments=time_segments, g=g, abstol=abstol, reltol=reltol, max_steps=max_steps, save_idxs=save_idxs,
                                        save_start=save_start, save_end=save_end, save_everystep=save_everystep)
    adaptive = isinstance(alg, (Tsit5, Rosenbrock23))
    if not adaptive and not isinstance(alg, (RK4, ETDRK4)):
        raise ValueError("alg must be RK4() / ETDRK4() (fixed step) or Tsit5() / Rosenbrock23() (adaptive)")
    if not adaptive and dt is None:
        raise ValueError("a fixed-step alg (RK4(), ETDRK4()) needs dt")
    if dt is None:
        dt = 0.0
    if not isinstance(sensealg, AbstractAdjointSensitivityAlgorithm):
        raise TypeError("sensealg must be one of InterpolatingAdjoint/BacksolveAdjoint/GaussAdjoint/QuadratureAdjoint")
    if isinstance(ensprob, ODEProblem):
        ensprob = EnsembleProblem(ensprob, ensprob.u0[None, :])
    prob = ensprob.prob
    ts = _save_times(prob.tspan, saveat, dt, save_everystep, save_start, save_end)
    if no_start is None:
        no_start = (not save_start) and len(ts) > 0 and bool(np.any(np.abs(ts - prob.tspan[0]) <= 1e-12 * max(1.0, abs(prob.tspan[0]))))
    if g is not None and not isinstance(g, tuple(_COSTS)):
        raise ValueError("g must be a registered continuous cost (HalfSquaredSum(), FirstStateSquaredPlusFirstParam(), ModelCost()) or None")
    loss_kind, shift, scale = _lib.LOSS_COTANGENT, 0.0, 0.0
    if isinstance(dgdu_discrete, LsqShift):
        loss_kind, shift = _lib.LOSS_LSQ_SHIFT, dgdu_discrete.shift
    elif isinstance(dgdu_discrete, LsqData):
        loss_kind, scale = _lib.LOSS_LSQ_DATA, dgdu_discrete.scale
    elif isinstance(dgdu_discrete, ModelLoss):
        loss_kind = _lib.LOSS_MODEL
    if isinstance(dgdu_discrete, (LsqData, ModelLoss)) and save_idxs is not None:
        raise ValueError("save_idxs with a device-resident loss is ambiguous: hand the cotangents instead")
    eng = Engine(prob.f, sensealg.name, ensprob.u0.shape[0], prob.tspan[0], prob.tspan[1], dt, save_times=ts,
                 loss_kind=loss_kind, loss_shift=shift, loss_scale=scale, devices=devices, reference_literal=reference_literal, p_shared=(ensprob.p.ndim == 1), device=device,
                 time_segments=time_segments, no_start=no_start, dims=prob.dims, cont_cost=(_COSTS[type(g)] if g is not None else 0),
                 stepper=(3 if isinstance(alg, Rosenbrock23) else (1 if adaptive else (2 if isinstance(alg, ETDRK4) else 0))), abstol=abstol, reltol=reltol, max_steps=max_steps,
                 family=(_lib.FAMILY_AS_REGISTERED if mfma is False else _lib.FAMILY_AUTO),
                 **_engine_kwargs(sensealg, checkpoints, dt, prob.tspan[0], adaptive, t1=prob.tspan[1]))
    routed = hasattr(eng, "stats") and eng.stats().get("routed_family") == _lib.FAMILY_MFMA      # hipadj_create put a declared dense chain on the FP64-MFMA family (csrc/hipadj_route.hpp)
    if mfma is True and not routed:
        eng.close()
        raise ValueError("mfma=True: the library did not take this problem to the FP64-MFMA family (a weight-shared dense_chain (2, H, H, 2), H in (32, 64, 128), on RK4 "
                         "with an ensemble size that is a multiple of 16, loss times on the step grid)")
    if isinstance(dgdu_discrete, (LsqData, ModelLoss)) and dgdu_discrete.data is not None and eng.M > 0:
        eng.set_loss_data(np.asarray(dgdu_discrete.data, dtype=np.float64).reshape(eng.N, eng.M, eng.n))
    out = eng.forward(ensprob.u0, ensprob.p, want_out=want_out)
    idxs = None
    if save_idxs is not None:
        idxs = np.atleast_1d(np.asarray(save_idxs, dtype=np.int64))
        if idxs.min() < 0 or idxs.max() >= eng.n:
            raise ValueError(f"save_idxs out of range for a {eng.n}-state model")
        if isinstance(dgdu_discrete, LsqShift):
            raise ValueError("save_idxs with the fused LsqShift loss is ambiguous: hand the cotangents instead")
        if out is not None:
            out = np.ascontiguousarray(out[:, :, idxs])
    return EnsembleSolution(engine=eng, u=out, t=ts, prob=ensprob, alg=alg, dt=dt,
                            extra=dict(sensealg=sensealg, dgdu_discrete=dgdu_discrete, g=g, save_idxs=idxs, mfma_routed=routed,
                                       checkpoints=(None if checkpoints is None else np.asarray(checkpoints, dtype=np.float64))))


def _dgdp_sum(sol, dgdp, shared):
    """sum_i dl_i/dp over the save times (and over the ensemble when p is shared)"""
    eng = sol.engine
    N, M, npar = eng.N, eng.M, eng.np
    if callable(dgdp):
        if sol.u is None or sol.extra.get("save_idxs") is not None:
            raise ValueError("a callable dgdp_discrete needs the full saved states (solve(..., want_out=True) without save_idxs)")
        p = sol.prob.p
        rows = [np.asarray(dgdp(sol.u[:, i, :], p, sol.t[i], i), dtype=np.float64).reshape(N, npar) for i in range(M)]
        tot = np.sum(rows, axis=0) if rows else np.zeros((N, npar))
    else:
        tot = np.asarray(dgdp, dtype=np.float64).reshape(N, M, npar).sum(axis=1)
    return tot.sum(axis=0) if shared else tot


def adjoint_sensitivities(sol, alg=RK4(), *, t=None, dgdu_discrete=None, dgdp_discrete=None, sensealg=None, checkpoints=None, g=None, **kwargs):
    """(du0, dp) for the loss  sum_i l_i(u(t_i))  with dl_i/du = dgdu_discrete at the times `t`
    (src/sensitivity_interface.jl:373-526).  `dgdu_discrete`: LsqShift(c) or an array [N][M][n] of cotangents
    (the AD path hands `Delta[:, i]`, src/concrete_solve.jl:842-851).  Returns du0 [N][n] and dp: [np] row
    (sum over the ensemble when p is shared) or [N][np].
    `dgdp_discrete`: the direct parameter derivative of a loss that also depends on p at the save times — an array
    [N][M][np] of dl_i/dp, or a callable (u_i [N][n], p, t_i, i) -> [N][np].  The reference adds it to the parameter part
    of the augmented state inside ReverseLossCallback (src/adjoint_common.jl:775-779); that part obeys mu' = -f_p^T lam,
    which does not contain mu, so the jumps commute with the integration and their sum is added to dp once, exactly."""
    from . import events
    if isinstance(sol, events.EventSolution):     # DiscreteCallback problem: the pieces' reverse passes chained by the reverse callbacks
        return events.adjoint_sensitivities_events(adjoint_sensitivities, sol, alg, t=t, dgdu_discrete=dgdu_discrete, dgdp_discrete=dgdp_discrete, g=g,
                                                   sensealg=sensealg, checkpoints=checkpoints, **kwargs)
    if kwargs:
        raise TypeError(f"unsupported keyword(s) {sorted(kwargs)} (callbacks: SURVEY.md §8f)")
    if dgdp_discrete is not None:
        du0, dp = adjoint_sensitivities(sol, alg, t=t, dgdu_discrete=dgdu_discrete, sensealg=sensealg, checkpoints=checkpoints, g=g)
        return du0, dp + _dgdp_sum(sol, dgdp_discrete, dp.ndim == 1)
    if g is not None and sol.extra.get("g") != g:
        raise ValueError("pass the continuous cost g to solve(...) as well: the reverse kernel is specialised on it")
    eng = sol.engine
    want_alg = (sensealg or sol.extra["sensealg"])
    if want_alg.name != eng.alg:
        raise ValueError(f"solution was prepared for sensealg={eng.alg!r}; re-run solve(...; sensealg={want_alg!r})")
    # the handle was configured at solve time (checkpointing flag, Quadrature tolerances, checkpoint list): anything different
    # handed here would be silently dropped, so it is an error like a different g / LsqShift (the reference builds the adjoint
    # problem from THESE arguments, src/sensitivity_interface.jl:426-526)
    if sensealg is not None and sensealg != sol.extra["sensealg"]:
        raise ValueError(f"sensealg={sensealg!r} differs from the one the forward solve was prepared with ({sol.extra['sensealg']!r}); "
                         "pass it to solve(...)")
    if checkpoints is not None:
        ck0 = sol.extra.get("checkpoints")
        if ck0 is None or not (len(ck0) == len(checkpoints) and np.allclose(np.asarray(checkpoints, dtype=np.float64), ck0, rtol=0, atol=1e-12)):
            raise ValueError("checkpoints differ from the list the forward solve was prepared with; pass checkpoints=... to solve(...)")
    if t is not None and not (len(t) == len(sol.t) and np.allclose(np.asarray(t, dtype=np.float64), sol.t, rtol=0, atol=1e-12)):
        raise ValueError("t must equal the save times the forward solve was run with")
    if isinstance(dgdu_discrete, LsqShift):
        if eng.cfg.loss_kind != _lib.LOSS_LSQ_SHIFT or eng.cfg.loss_shift != dgdu_discrete.shift:
            raise ValueError("pass dgdu_discrete=LsqShift(...) to solve(...) as well: the reverse kernel is specialised on it")
        return eng.adjoint(None)
    if isinstance(dgdu_discrete, (LsqData, ModelLoss)):
        if sol.extra.get("dgdu_discrete") != dgdu_discrete:
            raise ValueError("pass the same device-resident loss (LsqData / ModelLoss) to solve(...) as well: the handle is configured with it and owns its data block")
        return eng.adjoint(None)
    if dgdu_discrete is None:
        if eng.cfg.loss_kind != _lib.LOSS_COTANGENT or eng.M == 0:
            return eng.adjoint(None)
        raise ValueError("dgdu_discrete required")
    if eng.cfg.loss_kind != _lib.LOSS_COTANGENT:
        raise ValueError("solution was prepared with a loss that stays on the device (LsqShift / LsqData / ModelLoss); cotangents need dgdu_discrete=None at solve time")
    idxs = sol.extra.get("save_idxs")
    delta = pack_cotangent(dgdu_discrete, eng.N, eng.M, eng.n if idxs is None else len(idxs))
    if idxs is not None:                      # cotangent of the saved components only: zero elsewhere (src/concrete_solve.jl:790-824)
        full = np.zeros((eng.N, eng.M, eng.n))
        full[:, :, idxs] = delta.reshape(eng.N, eng.M, len(idxs))
        delta = full
    return eng.adjoint(delta)


def pack_cotangent(delta, N, M, nsave, only_end=False):
    """The cotangent of the saved solution in any form the reference's pullback accepts (src/concrete_solve.jl:776-869) as the dense block
    [N][M][nsave] the C ABI takes:
      * a dense array with N * M * nsave elements (the `Array(sol)` form; any shape that reshapes to [N][M][nsave]);
      * a sequence of M per-time cotangents, each [N][nsave] — the vector-of-arrays / VectorOfArray forms (:818-867); an entry that is
        None stands for NoTangent / ZeroTangent and contributes zeros;
      * only_end (a single save time == T, :716, 783-814): additionally the bare state cotangent [N][nsave] ("user did sol[end]")."""
    if delta is None:
        return np.zeros((N, M, nsave))
    if isinstance(delta, (list, tuple)):
        if len(delta) != M:
            raise ValueError(f"a vector-of-arrays cotangent needs one entry per save time ({M}), got {len(delta)}")
        out = np.zeros((N, M, nsave))
        for i, x in enumerate(delta):
            if x is not None:
                out[:, i, :] = np.asarray(x, dtype=np.float64).reshape(N, nsave)
        return out
    d = np.asarray(delta, dtype=np.float64)
    if only_end and d.size == N * nsave:
        return d.reshape(N, 1, nsave)
    if d.size != N * M * nsave:
        raise ValueError(f"cotangent has {d.size} elements, expected {N} x {M} x {nsave}")
    return d.reshape(N, M, nsave)


def concrete_solve_adjoint(prob, alg, sensealg, u0, p, *, dt=None, saveat, **kw):
    """(out, pullback) — the contract of SciMLBase._concrete_solve_adjoint (src/concrete_solve.jl:523-1042):
    `out` = sol(ts) [N][M][n]; pullback(Delta) -> (du0, dp) runs the reverse pass with Delta as dgdu_discrete."""
    ens = EnsembleProblem(ODEProblem(prob.f, prob.u0, prob.tspan, np.asarray(p if np.ndim(p) == 1 else prob.p), prob.dims),
                          np.atleast_2d(u0), np.asarray(p))
    sol = solve(ens, alg, dt=dt, saveat=saveat, sensealg=sensealg, **kw)

    N, M, nsave = sol.u.shape                  # sol.u already has the save_idxs shape
    only_end = bool(M == 1 and abs(sol.t[0] - sol.prob.prob.tspan[1]) <= 1e-12 * max(1.0, abs(sol.prob.prob.tspan[1])))   # src/concrete_solve.jl:716

    def pullback(delta):
        return adjoint_sensitivities(sol, alg, dgdu_discrete=pack_cotangent(delta, N, M, nsave, only_end))
    pullback.only_end = only_end
    return sol.u, pullback


def make_autograd_function():
    """torch.autograd.Function running forward and reverse passes on device tensors through the C ABI
    (hipadj_forward_dev / hipadj_adjoint_dev on torch's current stream)."""
    import torch

    class EnsembleAdjoint(torch.autograd.Function):
        """out = EnsembleAdjoint.apply(u0 [N][n], p [np] or [N][np], engine).  The forward solution lives in the engine's single
        handle: a second forward on the same engine before the backward of the first invalidates it, which backward detects
        (forward-generation counter) instead of differentiating the wrong solution.  Both directions end with
        engine.synchronize(): the device-pointer calls of the C ABI are asynchronous and only hipadj_synchronize reads the device
        status word (non-finite sensitivities, Tsit5 step-capacity overflow = the reference's unstable / MaxIters retcodes)."""

        @staticmethod
        def forward(ctx, u0, p, engine):
            if not (u0.is_cuda and p.is_cuda and u0.dtype == torch.float64 and p.dtype == torch.float64):
                raise ValueError("u0 and p must be float64 tensors on the engine's GPU")
            if u0.device.index != engine.device or p.device.index != engine.device:
                raise ValueError(f"u0 / p live on cuda:{u0.device.index} / cuda:{p.device.index}, the engine on cuda:{engine.device}")
            if tuple(u0.shape) != (engine.N, engine.n):
                raise ValueError(f"u0 must be [{engine.N}][{engine.n}], got {tuple(u0.shape)}")
            want_p = (engine.np,) if engine.p_shared else (engine.N, engine.np)
            if tuple(p.shape) != want_p:
                raise ValueError(f"p must have shape {want_p}, got {tuple(p.shape)}")
            engine.use_torch_stream()
            out = torch.empty((engine.N, engine.M, engine.n), dtype=torch.float64, device=u0.device)
            engine.forward_dev(u0.contiguous(), p.contiguous(), out)
            engine.synchronize()
            ctx.engine = engine
            ctx.generation = engine.forward_generation
            return out

        @staticmethod
        def backward(ctx, grad_out):
            eng = ctx.engine
            if eng.forward_generation != ctx.generation:
                raise RuntimeError("the engine ran another forward solve since this output was produced: its handle holds ONE forward "
                                   "solution (use one Engine per live forward, or call backward before the next forward)")
            if tuple(grad_out.shape) != (eng.N, eng.M, eng.n):
                raise ValueError(f"cotangent must be [{eng.N}][{eng.M}][{eng.n}], got {tuple(grad_out.shape)}")
            eng.use_torch_stream()
            du0 = torch.empty((eng.N, eng.n), dtype=torch.float64, device=grad_out.device)
            dp = torch.empty((eng.np,) if eng.p_shared else (eng.N, eng.np), dtype=torch.float64, device=grad_out.device)
            eng.adjoint_dev(grad_out.contiguous(), du0, dp)
            eng.synchronize()
            return du0, dp, None

    return EnsembleAdjoint
