"""Build libhipadj.so (gfx950) in-tree with hipcc.  `python -m scimlsensitivity_jl_amd.build` or build.build().

The library consists of one translation unit for the C ABI (csrc/hipadj_api.hip) and one per kernel family / compiled-in
model / stepper (csrc/hipadj_tu_lane.hip and csrc/hipadj_tu_family.hip, selected with -D): the device code of the ~400
kernel instantiations dominates the build, so the units are compiled in parallel and linked once.
"""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "hipadj_api.hip")
# A/B builds: HIPADJ_BUILD_LIB=<path of another .so> HIPADJ_BUILD_EXTRA="-D..." (objects go to build_<stem>/, the shipped library is untouched; load the
# variant with HIPADJ_LIBRARY=<path>)
LIB = os.environ.get("HIPADJ_BUILD_LIB") or os.path.join(HERE, "libhipadj.so")
OBJ = os.path.join(HERE, "build") if "HIPADJ_BUILD_LIB" not in os.environ else os.path.join(HERE, "build_ab", os.path.splitext(os.path.basename(LIB))[0])   # build_ab/: git- and gpurun-ignored

DEPS = sorted(glob.glob(os.path.join(CSRC, "*.h*"))) + [os.path.join(CSRC, "hipadj.map")] + [os.path.join(os.path.dirname(HERE), "include", "hipadj.h"), os.path.abspath(__file__)]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("HIPADJ_BUILD_EXTRA", "").split()

LANE_MODELS = ["ModelLV", "ModelLVT", "ModelLorenz", "ModelLinDiag", "ModelFallMass"]   # csrc/hipadj_models.hpp
FIELD_GRIDS = [8, 16, 32]
MLP_HIDDEN = [32, 64, 128]


def units():
    """(object name, source, extra flags), most expensive first so that the pool drains evenly."""
    u = []
    for m in LANE_MODELS:
        u.append((f"lane_{m}_rk4.o", "hipadj_tu_lane.hip", [f"-DHIPADJ_TU_MODEL={m}", "-DHIPADJ_TU_PART=0"]))
    for m in LANE_MODELS:
        u.append((f"lane_{m}_tsit5.o", "hipadj_tu_lane.hip", [f"-DHIPADJ_TU_MODEL={m}", "-DHIPADJ_TU_PART=1"]))
    for h in sorted(MLP_HIDDEN, reverse=True):
        u.append((f"mlp_{h}.o", "hipadj_tu_family.hip", [f"-DHIPADJ_TU_MLP={h}"]))
    for g in sorted(FIELD_GRIDS, reverse=True):
        u.append((f"field_{g}.o", "hipadj_tu_family.hip", [f"-DHIPADJ_TU_FIELD={g}"]))
    u.append(("api.o", "hipadj_api.hip", []))
    return u


API_ONLY = {"hipadj_user.hpp", "hipadj_dual.hpp", "hipadj_comm.hpp", "hipadj_api.hip"}   # included by hipadj_api.hip alone


def _stale(obj, src):
    """An object is rebuilt when its source, a header it can see, hipadj.h or this script is newer."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    api = os.path.basename(src) == "hipadj_api.hip"
    for d in DEPS:
        b = os.path.basename(d)
        if b.endswith(".hip") and d != src:
            continue
        if not api and b in API_ONLY:
            continue
        if os.path.exists(d) and os.path.getmtime(d) > t:
            return True
    return False


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, jobs=None):
    """Compile every HIP source of the package for gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the ROCm root this build's compiler lives in: runtime-registered models are compiled by the SAME toolkit's hiprtc (csrc/hipadj_user.hpp rtc_api)
    rocm_root = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    rocm_def = [f'-DHIPADJ_ROCM_PATH="{rocm_root}"']
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or int(os.environ.get("HIPADJ_BUILD_JOBS", "0")) or max(1, min(os.cpu_count() or 1, 16))

    def compile_unit(u):
        obj, src, extra = u
        if not force and not _stale(os.path.join(OBJ, obj), os.path.join(CSRC, src)):
            return os.path.join(OBJ, obj)
        cmd = [hipcc] + FLAGS + rocm_def + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return os.path.join(OBJ, obj)

    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_unit, units()))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'hipadj.map')}", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
