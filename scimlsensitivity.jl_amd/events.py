"""DiscreteCallback at preset times with a state affect, composed on the host from per-piece device solves.

The reference differentiates hybrid systems by tracking the callbacks of the forward solve and replaying them as reverse callbacks
(src/callback_tracking.jl:232-470): at an event time the adjoint becomes  lam <- (da/du)' lam  (Jacobian of the affect at the LEFT state) and the
parameter gradient receives  (da/dp)' lam.  Between two events nothing differs from an ordinary problem — so an event problem is a chain of ordinary
handles: piece j integrates [e_j, e_{j+1}], the affect kernel (hipadj_affect_apply) maps its end state to the start state of piece j + 1; in reverse,
du0 of piece j + 1 goes through hipadj_affect_vjp and enters piece j as one more cotangent at its end time.  Every sensealg, stepper and model
family of the pieces is available unchanged; the hot kernels know nothing about events.

    f = sa.DeviceFunction("lv_dose", 2, 4, body).set_affect("un[0] += 2.0;")
    sol = sa.solve(ensprob, sa.Tsit5(), saveat=0.5, sensealg=sa.BacksolveAdjoint(), callback=sa.PresetTimeCallback([5.0]))
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), dgdu_discrete=delta)

Semantics (save_positions = (false, false), the form of test/Callbacks1/discrete_callbacks.jl:263-268): a save time that coincides with an event time
holds the RIGHT limit — OrdinaryDiffEq applies the discrete callbacks of a step before its regular saveat save [upstream-recall]; event times outside
(t0, t1) are ignored.  An affect may also edit the parameters (`pn`, test/Callbacks1/discrete_callbacks.jl:303-312): the pieces after such an event
run with per-trajectory parameters and the reverse callbacks carry the gradient with respect to the later parameters back through d(pn)/d(u, p).
Not covered: ContinuousCallback (root finding and the implicit event-time corrections, :367-431), save_positions with a `true`."""
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .problems import ODEProblem, EnsembleProblem, LsqShift


@dataclass
class EventSolution:
    """What solve(..., callback=PresetTimeCallback(...)) returns: the pieces (each owning its device-resident forward solution), the saved states
    assembled over all save times, and the left states at the events (needed by the reverse callbacks)."""
    pieces: list                 # EnsembleSolution per piece, in time order
    piece_cols: list             # per piece: indices into `t` of its own save times
    edges: list                  # [t0, e_1, ..., e_k, t_end]
    u_left: list                 # state at the end of piece j (before the affect), j < last
    p_piece: list                # parameters of piece j, [N][np] (they change where an affect edits pn)
    mass_matrix: object          # the model's mass matrix AT SOLVE TIME (or None): the reverse chain converts with it, whatever the registry holds later
    u: np.ndarray                # [N][M][n] = sol(ts), right limits at event times
    t: np.ndarray
    prob: object
    alg: object
    model_id: int
    device: int
    extra: dict = field(default_factory=dict)

    def close(self):
        for s in self.pieces:
            s.engine.close()


def _model_id(prob):
    mid = _lib.MODEL[prob.f]
    if mid < _lib.MODEL_USER_BASE:
        raise ValueError("callback: affects are attached to runtime-registered models (DeviceFunction(...).set_affect(body))")
    return mid


def solve_with_events(solve, ts_of, ensprob, alg, callback, *, saveat=None, dt=None, device=0, dgdu_discrete=None, checkpoints=None, save_idxs=None,
                      save_start=True, save_end=True, save_everystep=False, **kw):
    if isinstance(ensprob, ODEProblem):
        ensprob = EnsembleProblem(ensprob, ensprob.u0[None, :])
    prob = ensprob.prob
    if checkpoints is not None or save_idxs is not None or save_everystep:
        raise ValueError("callback: `checkpoints`, `save_idxs` and `save_everystep` are not combined with event problems (the pieces use their defaults)")
    mid = _model_id(prob)
    mm0 = _lib.MASS.get(mid)
    if mm0 is not None and abs(np.linalg.det(mm0)) < 1e-13 * max(1.0, float(np.abs(mm0).max())) ** mm0.shape[0]:
        raise ValueError("callback: events are not combined with a singular mass matrix (semi-explicit DAE): the chaining of the pieces divides by M'")
    t0, t1 = prob.tspan
    ts = ts_of(prob.tspan, saveat, 0.0 if dt is None else dt, False, save_start, save_end)
    if len(ts) == 0:
        raise ValueError("callback: the event problem needs at least one save time (the loss lives there)")
    ev = sorted({float(e) for e in callback.times if t0 < e < t1 and e <= ts[-1]})    # events after the last loss time cannot influence it
    edges = [t0] + ev + [t1]
    N = ensprob.u0.shape[0]
    pieces, cols, u_left, p_piece = [], [], [], []
    npar = ensprob.p.shape[-1]
    # every piece runs with per-trajectory parameters: the reverse callbacks need the parameter gradient per trajectory, and an affect may edit pn
    u0, pcur = ensprob.u0, np.ascontiguousarray(np.broadcast_to(ensprob.p, (N, npar)))
    out = None
    for j in range(len(edges) - 1):
        a, b = edges[j], edges[j + 1]
        last = j == len(edges) - 2
        own = [i for i, s in enumerate(ts) if (a <= s < b) or (last and s == b)]
        sv = [ts[i] for i in own] + ([] if last else [b])          # the piece's end state feeds the affect
        pj = EnsembleProblem(ODEProblem(prob.f, u0[0], (a, b), pcur[0], prob.dims), u0, pcur)
        p_piece.append(pcur)
        # the first piece inherits the start-point rule of the ordinary path: no_start = !save_start && t0 in ts suppresses the loss jump at t0
        # (src/concrete_solve.jl:962, src/adjoint_common.jl:761); later pieces start at an event time, where nothing is suppressed
        ns = bool(j == 0 and not save_start and len(ts) and ts[0] == t0)
        sol = solve(pj, alg, dt=dt, saveat=sv, device=device, dgdu_discrete=None, no_start=ns, **kw)
        pieces.append(sol); cols.append(own)
        if out is None:
            out = np.zeros((N, len(ts), sol.u.shape[2]))
        for q, i in enumerate(own):
            out[:, i, :] = sol.u[:, q, :]
        if not last:
            ul = np.ascontiguousarray(sol.u[:, -1, :])
            u_left.append(ul)
            u0, pcur = _lib.affect_apply(mid, ul, pcur, b, npar, device=device)
    mm = _lib.MASS.get(mid)
    return EventSolution(pieces=pieces, piece_cols=cols, edges=edges, u_left=u_left, p_piece=p_piece, mass_matrix=None if mm is None else np.array(mm), u=out, t=np.asarray(ts), prob=ensprob, alg=alg, model_id=mid, device=device,
                         extra=dict(dgdu_discrete=dgdu_discrete, callback=callback))


def adjoint_sensitivities_events(adjoint_sensitivities, sol, alg, *, t=None, dgdu_discrete=None, dgdp_discrete=None, sensealg=None, checkpoints=None, **kw):
    """(du0, dp) of an event problem: the pieces' reverse passes from the last to the first, chained by the reverse callbacks.  `sensealg` must be the one
    of the forward solve (each piece checks it, like the ordinary path); `checkpoints` is refused: the pieces use their own defaults."""
    if checkpoints is not None:
        raise ValueError("callback: `checkpoints` is not combined with event problems (the pieces use their defaults)")
    if sensealg is not None:
        kw = dict(kw, sensealg=sensealg)
    if dgdp_discrete is not None:
        raise ValueError("callback: dgdp_discrete is not supported for hybrid systems (the reference errors likewise, src/callback_tracking.jl:283-284)")
    if t is not None and not np.array_equal(np.asarray(t, dtype=np.float64), sol.t):
        raise ValueError("t must equal the save times of the forward solve")
    dg = dgdu_discrete if dgdu_discrete is not None else sol.extra.get("dgdu_discrete")
    N, M, n = sol.u.shape
    if isinstance(dg, LsqShift):
        delta = sol.u - dg.shift                                    # dgdu_discrete(out, u, p, t, i) = u - shift, at the saved (right-limit) states
    elif dg is None:
        raise ValueError("dgdu_discrete required (cotangents [N][M][n] or LsqShift)")
    else:
        delta = np.asarray(dg, dtype=np.float64).reshape(N, M, n)
    shared = sol.prob.p.ndim == 1
    npar = sol.prob.p.shape[-1]
    gp = np.zeros((N, npar))          # gradient with respect to the CURRENT piece's parameters, per trajectory, of everything later in time
    lam_in = None
    du0 = None
    for j in range(len(sol.pieces) - 1, -1, -1):
        own = sol.piece_cols[j]
        cot = [delta[:, i, :] for i in own]
        if j < len(sol.pieces) - 1:
            cot.append(lam_in)                                       # the reverse callback's output enters at the piece's end time
        dj = np.ascontiguousarray(np.stack(cot, axis=1))
        du0, dpj = adjoint_sensitivities(sol.pieces[j], alg, t=sol.pieces[j].t, dgdu_discrete=dj, **kw)
        gp = gp + np.asarray(dpj).reshape(N, npar)
        if j > 0:                                                    # reverse callback of the event between piece j - 1 and piece j
            # with a mass matrix a piece returns the reference's lam(t0) = M^{-T} dL/du (src/sensitivity_interface.jl:500); the callback acts on
            # dL/du itself, and the cotangent handed to the lower piece is a dL/du as well: convert (rows: lam' M)
            M = sol.mass_matrix                                      # snapshot taken at solve time
            lam_true = du0 if M is None else du0 @ M
            lam_in, gp = _lib.affect_vjp(sol.model_id, sol.u_left[j - 1], sol.p_piece[j - 1], sol.edges[j], lam_true, gp, device=sol.device)
    return du0, (gp.sum(axis=0) if shared else gp)
