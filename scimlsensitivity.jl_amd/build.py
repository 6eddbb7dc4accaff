"""Build libhipadj.so (gfx950) in-tree with hipcc.  `python -m scimlsensitivity_jl_amd.build` or build.build()."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "hipadj_api.hip")
LIB = os.path.join(HERE, "libhipadj.so")
import glob

DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h*"))) + [os.path.join(os.path.dirname(HERE), "include", "hipadj.h")]   # every header and the .hip
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile every HIP source of the package for gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
