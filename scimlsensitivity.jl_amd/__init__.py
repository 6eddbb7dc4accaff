"""scimlsensitivity.jl_amd — MI355X-native batched continuous-adjoint engine behind the SciMLSensitivity.jl
plugin surface (AbstractSensitivityAlgorithm / adjoint_sensitivities / _concrete_solve_adjoint).

Import as `scimlsensitivity_jl_amd` (shim at the repo root).  The numerical path is libhipadj.so (HIP, gfx950)
behind the C ABI in include/hipadj.h; importing this package does not require a GPU, creating an Engine does."""
from ._lib import HipadjError, model_sizes, load as load_library, LIB_PATH
from .sensitivity_algorithms import (AbstractSensitivityAlgorithm, AbstractAdjointSensitivityAlgorithm, DeviceVJP,
                                     InterpolatingAdjoint, BacksolveAdjoint, QuadratureAdjoint, GaussAdjoint, GaussKronrodAdjoint,
                                     ischeckpointing)
from .problems import (RK4, ETDRK4, Tsit5, Rosenbrock23, DeviceFunction, WideDeviceFunction, ODEProblem, EnsembleProblem, EnsembleSolution, LsqShift, LsqData, ModelLoss, HalfSquaredSum, PresetTimeCallback, ContinuousCallback,
                       FirstStateSquaredPlusFirstParam, ModelCost)
from .engine import Engine
from .interface import solve, adjoint_sensitivities, concrete_solve_adjoint, make_autograd_function
from .events import EventSolution
from .distributed import shard_range, allreduce_dp, gather_du0, comm_unique_id, init_native_allreduce
from . import build as _build

build_extension = _build.build

__all__ = [
    "HipadjError", "model_sizes", "load_library", "LIB_PATH", "AbstractSensitivityAlgorithm",
    "AbstractAdjointSensitivityAlgorithm", "DeviceVJP", "InterpolatingAdjoint", "BacksolveAdjoint",
    "QuadratureAdjoint", "GaussAdjoint", "GaussKronrodAdjoint", "ischeckpointing", "RK4", "ETDRK4", "Tsit5", "Rosenbrock23", "DeviceFunction", "WideDeviceFunction", "ODEProblem", "EnsembleProblem",
    "EnsembleSolution", "ContinuousCallback", "LsqShift", "LsqData", "ModelLoss", "HalfSquaredSum", "FirstStateSquaredPlusFirstParam", "ModelCost", "Engine", "solve", "adjoint_sensitivities", "concrete_solve_adjoint",
    "make_autograd_function", "shard_range", "allreduce_dp", "gather_du0", "comm_unique_id", "init_native_allreduce", "build_extension",
]
