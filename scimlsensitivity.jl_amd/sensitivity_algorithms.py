"""Sensitivity-algorithm configuration types — host-side mirror of the four continuous-adjoint structs of
SciMLSensitivity.jl (src/sensitivity_algorithms.jl:254-272 BacksolveAdjoint, :378-396 InterpolatingAdjoint,
:486-503 QuadratureAdjoint, :591-607 GaussAdjoint) with the same field names and defaults.

`autojacvec` (the VJPChoice seam, :1426-1602) accepts only `DeviceVJP()` / None here: the models of the device
registry carry hand-derived (df/du)^T lam and (df/dp)^T lam, i.e. the reference's user-VJP path
(src/derivative_wrappers.jl:284-359).  Any other VJP choice raises, as the reference does for unsupported
combinations (src/sensitivity_interface.jl:409-420 falls back; we do not — there is no second backend)."""
from dataclasses import dataclass


class AbstractSensitivityAlgorithm:
    pass


class AbstractAdjointSensitivityAlgorithm(AbstractSensitivityAlgorithm):
    name = "?"


@dataclass(frozen=True)
class DeviceVJP:
    """Hand-derived, device-inlined VJP of a registered model (f.vjp / f.vjp_p of the reference)."""


def _check_vjp(autojacvec):
    if autojacvec is not None and not isinstance(autojacvec, DeviceVJP):
        raise ValueError(
            f"autojacvec={autojacvec!r} is not available: device models carry hand-derived VJPs (DeviceVJP()).")


@dataclass(frozen=True)
class InterpolatingAdjoint(AbstractAdjointSensitivityAlgorithm):
    autojacvec: object = None
    checkpointing: bool = False
    noisemixing: bool = False
    name = "interpolating"

    def __post_init__(self):
        _check_vjp(self.autojacvec)


@dataclass(frozen=True)
class BacksolveAdjoint(AbstractAdjointSensitivityAlgorithm):
    autojacvec: object = None
    checkpointing: bool = True      # src/sensitivity_algorithms.jl:260-265
    noisemixing: bool = False
    name = "backsolve"

    def __post_init__(self):
        _check_vjp(self.autojacvec)


@dataclass(frozen=True)
class QuadratureAdjoint(AbstractAdjointSensitivityAlgorithm):
    autojacvec: object = None
    abstol: float = 1e-6            # src/sensitivity_algorithms.jl:493-497
    reltol: float = 1e-3
    name = "quadrature"

    def __post_init__(self):
        _check_vjp(self.autojacvec)


@dataclass(frozen=True)
class GaussAdjoint(AbstractAdjointSensitivityAlgorithm):
    autojacvec: object = None
    checkpointing: bool = False
    name = "gauss"

    def __post_init__(self):
        _check_vjp(self.autojacvec)


@dataclass(frozen=True)
class GaussKronrodAdjoint(AbstractAdjointSensitivityAlgorithm):
    """GaussKronrodAdjoint(; autojacvec, checkpointing = false) (src/sensitivity_algorithms.jl:612-711): GaussAdjoint with
    Gauss-Kronrod quadrature per step "to achieve error control"."""
    autojacvec: object = None
    checkpointing: bool = False
    name = "gausskronrod"

    def __post_init__(self):
        _check_vjp(self.autojacvec)


def ischeckpointing(sensealg, sol=None):
    """src/sensitivity_algorithms.jl:1665-1677: Backsolve uses its flag; Interpolating/Gauss checkpoint when
    asked or when the forward solution is not dense."""
    if isinstance(sensealg, BacksolveAdjoint):
        return sensealg.checkpointing
    if isinstance(sensealg, (InterpolatingAdjoint, GaussAdjoint, GaussKronrodAdjoint)):
        return sensealg.checkpointing or (sol is not None and not getattr(sol, "dense", True))
    return False
