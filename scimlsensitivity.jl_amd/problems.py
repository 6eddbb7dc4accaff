"""Problem / solver / loss descriptors of the host API (names follow SciMLBase / OrdinaryDiffEq)."""
from dataclasses import dataclass, field
import numpy as np

from . import _lib


@dataclass(frozen=True)
class RK4:
    """Fixed-step classic Runge-Kutta 4 (the `alg` handed to solve / adjoint_sensitivities)."""


@dataclass(frozen=True)
class Tsit5:
    """Adaptive Tsitouras 5(4) with PI step control and its own 4th-order interpolant — the stepper of the
    reference's own tests (test/Core3/adjoint.jl:31-43).  Per-trajectory step sequences on the device."""


class DeviceFunction:
    """ODEFunction(f!; vjp, vjp_p) for the device (src/derivative_wrappers.jl:284-359, test/Core3/user_vjp.jl:77):
    the three function BODIES as HIP C++ text over `du`/`out`, `u`, `p`, `lam`, `t` (all double), compiled at solve time
    with hiprtc for gfx950.  Usable wherever a registered model name is: ODEProblem(DeviceFunction(...), u0, tspan, p)."""

    def __init__(self, name, n, np, f, vjp, vjp_p, check=False):
        self.name, self.n, self.np = name, int(n), int(np)
        self.id = _lib.register_model(name, n, np, f, vjp, vjp_p, check=check)

    def set_cost(self, dgdu, dgdp):
        """Attach a continuous cost through its gradients — the `dgdu_continuous` / `dgdp_continuous` keywords of
        adjoint_sensitivities (src/sensitivity_interface.jl:373-526) as HIP C++ bodies writing `out` from `u`, `p`, `t`.
        Select it with solve(..., g=ModelCost())."""
        _lib.set_model_cost(self.id, dgdu, dgdp)
        return self

    def __repr__(self):
        return f"DeviceFunction({self.name!r}, n={self.n}, np={self.np}, id={self.id})"


@dataclass
class ODEProblem:
    """ODEProblem(f, u0, tspan, p) with `f` a name from the device model registry (include/hipadj.h)."""
    f: str
    u0: np.ndarray
    tspan: tuple
    p: np.ndarray
    dims: tuple = (0, 0, 0, 0)

    def __post_init__(self):
        if isinstance(self.f, DeviceFunction):
            self.f = self.f.name
        if self.f not in _lib.MODEL:
            raise ValueError(f"unknown model {self.f!r}; registered: {sorted(_lib.MODEL)}")
        self.u0 = np.ascontiguousarray(self.u0, dtype=np.float64)
        self.p = np.ascontiguousarray(self.p, dtype=np.float64)
        self.tspan = (float(self.tspan[0]), float(self.tspan[1]))


@dataclass
class EnsembleProblem:
    """EnsembleProblem(prob; prob_func): N independent trajectories that differ in u0 (and optionally p)
    (test/Core4/ensembles.jl:13-31).  u0: [N][n]; p: [np] shared or [N][np]."""
    prob: ODEProblem
    u0: np.ndarray
    p: np.ndarray = None

    def __post_init__(self):
        self.u0 = np.ascontiguousarray(self.u0, dtype=np.float64)
        if self.u0.ndim != 2:
            raise ValueError("EnsembleProblem u0 must be [N][n]")
        self.p = self.prob.p if self.p is None else np.ascontiguousarray(self.p, dtype=np.float64)


@dataclass(frozen=True)
class LsqShift:
    """dgdu_discrete(out, u, p, t, i) = u - shift  (test/Core3/adjoint.jl:49-51: `out .= -2.0 .+ u`),
    evaluated inside the reverse kernel."""
    shift: float = 0.0


@dataclass(frozen=True)
class HalfSquaredSum:
    """Continuous cost g(u, p, t) = (sum(u))^2 / 2 with dgdu_continuous = sum(u) in every component
    (the `g`/`dg` pair of test/Core3/adjoint.jl:913-919); evaluated inside the reverse kernels
    (accumulate_cost!, src/derivative_wrappers.jl:1411-1442)."""


@dataclass(frozen=True)
class FirstStateSquaredPlusFirstParam:
    """Continuous cost g(u, p, t) = u[1]^2 + p[1] with dgdu_continuous = [2 u1, 0, ...] and dgdp_continuous = [1, 0, ...]
    (test/Core7/mixed_costs.jl:46-57): the registered cost with a parameter term."""


@dataclass(frozen=True)
class ModelCost:
    """The continuous cost attached to a DeviceFunction with set_cost(dgdu, dgdp)."""


@dataclass
class EnsembleSolution:
    """What the reverse pass needs from the forward solve: the handle owning the device-resident interpolant
    tiles / checkpoints, plus the primal output at the save times."""
    engine: object
    u: np.ndarray                # [N][M][n] = sol(ts)
    t: np.ndarray                # [M]
    prob: object
    alg: object
    dt: float
    dense: bool = True
    retcode: str = "Success"
    extra: dict = field(default_factory=dict)
