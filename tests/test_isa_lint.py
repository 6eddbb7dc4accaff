"""CPU check of the shipped gfx950 code for the spill-placement miscompile described in tests/tools/isa_lint.py: the
built-in library, and the runtime-compiled kernels of the configuration that exposed it (hiprtc needs no device)."""
import ctypes as C
import glob
import os
import sys

import pytest

import emu as E
import user_models as UM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import isa_lint  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")


def test_builtin_library_has_no_spill_copies_ahead_of_exec_restores(sa):
    sa.load_library()
    assert isa_lint.lint(os.path.join(ROOT, "scimlsensitivity.jl_amd", "libhipadj.so")) == []


@pytest.mark.parametrize("n,alg", [(3, "backsolve"), (4, "backsolve"), (4, "interpolating")])
def test_runtime_tsit5_kernels_have_no_spill_copies_ahead_of_exec_restores(tmp_path, monkeypatch, n, alg):
    from scimlsensitivity_jl_amd import _lib
    m = UM.ring(n)
    name = f"ring{n}_lint_{alg}"
    _lib.register_model(name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    cfg = E.make_config(name, alg, 53, 0.0, 0.5, 0.0, [], loss_kind=1, stepper=1, abstol=1e-9, reltol=1e-9, checkpointing=False)
    L = _lib.load()
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs
    for o in objs:
        assert isa_lint.lint(o) == []


def test_runtime_compile_retries_at_O1_when_the_check_flags_the_code_object():
    """Product-side guard (user_isa_check in csrc/hipadj_user.hpp).  The flagged placement comes from the hiprtc a torch wheel bundles (ROCm 7.0),
    which compiled the runtime models before the library bound the build toolkit's compiler; the 7.2 toolkit does not produce it
    for the probe.  So the probe runs in a child process that asks for the in-process hiprtc (HIPADJ_HIPRTC=libhiprtc.so)."""
    import subprocess, sys
    import torch
    if not os.path.exists(os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")):
        pytest.skip("no bundled hiprtc to bind")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-p", "no:cacheprovider", "-k", "inner_retry_probe", "-rs"],
                       env=dict(os.environ, HIPADJ_HIPRTC="libhiprtc.so", HIPADJ_LINT_INNER="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.skipif(os.environ.get("HIPADJ_LINT_INNER") != "1", reason="runs inside test_runtime_compile_retries_at_O1_when_the_check_flags_the_code_object")
def test_inner_retry_probe(tmp_path, monkeypatch):
    """The batched stage sum forced onto a 13-wide state (-DHIPADJ_TS5_WIDE=64) reproduces the flagged placement at -O3; the library must notice,
    rebuild at -O1 and hand out a clean code object."""
    from scimlsensitivity_jl_amd import _lib
    m = UM.ring(4)
    _lib.register_model("ring4_lint_retry", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    monkeypatch.setenv("HIPADJ_RTC_FLAGS", "-DHIPADJ_TS5_WIDE=64 -DHIPADJ_TS5_PADDED=1")   # the batched zero-padded stage sum of rounds 1-2: the code the old compiler mis-places
    cfg = E.make_config("ring4_lint_retry", "backsolve", 53, 0.0, 0.5, 0.0, [], loss_kind=1, stepper=1, abstol=1e-9, reltol=1e-9, checkpointing=False)
    L = _lib.load()
    rc = L.hipadj_model_check_config(C.byref(cfg))
    objs = sorted(glob.glob(str(tmp_path / "*.hsaco")))
    if rc != _lib.OK:
        # round 5: with the discrete-loss plumbing in the callback the old compiler mis-places the copies at -O1 as well — the guard's OTHER designed outcome: both builds
        # flagged, the configuration is refused by name instead of running lanes on stale values (user_compile, attempt 1)
        assert rc == _lib.ERR_UNSUPPORTED and b"register-spill copies ahead of an exec restore" in L.hipadj_last_error(None) and b"at -O3 and at -O1" in L.hipadj_last_error(None)
        assert len(objs) == 2 and isa_lint.lint(objs[0]) != [] and isa_lint.lint(objs[1]) != []
        return
    if len(objs) == 1:
        pytest.skip("this compiler build does not produce the flagged placement for the probe configuration")
    assert len(objs) == 2 and isa_lint.lint(objs[0]) != [] and isa_lint.lint(objs[1]) == []


@pytest.mark.parametrize("alg", ["interpolating", "gauss", "backsolve"])
@pytest.mark.parametrize("n", [3, 6])
def test_runtime_offgrid_kernels_compile_without_a_device(tmp_path, monkeypatch, n, alg):
    """Loss times off the step grid for a runtime-registered model: hipadj_model_check_config compiles k_forward, k_interp_offgrid,
    k_out_offgrid and k_finish with hiprtc (no device needed); the code objects pass the spill-placement lint."""
    from scimlsensitivity_jl_amd import _lib
    m = UM.ring(n)
    name = f"ring{n}_offgrid_{alg}"
    _lib.register_model(name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    cfg = E.make_config(name, alg, 53, 0.0, 0.5, 0.01, [0.1234, 0.3, 0.5], loss_kind=0, checkpointing=(alg == "backsolve"))
    L = _lib.load()
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs
    txt = "".join(isa_lint.disassemble(o) for o in objs)
    assert {"interpolating": "k_interp_offgrid", "gauss": "k_gauss_offgrid", "backsolve": "k_backsolve_offgrid"}[alg] in txt and "k_out_offgrid" in txt
    for o in objs:
        assert isa_lint.lint(o) == []


@pytest.mark.parametrize("alg", ["interpolating", "gauss"])
def test_runtime_checkpointed_fixed_step_kernels_compile_without_a_device(tmp_path, monkeypatch, alg):
    """checkpointing=true on the fixed step for a runtime-registered model (k_interp_ckpt / k_gauss_ckpt through hiprtc)."""
    from scimlsensitivity_jl_amd import _lib
    m = UM.ring(4)
    name = f"ring4_ckpt_{alg}"
    _lib.register_model(name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    cfg = E.make_config(name, alg, 53, 0.0, 0.5, 0.01, [0.0, 0.1, 0.2, 0.3, 0.4, 0.5], loss_kind=0, checkpointing=True)
    L = _lib.load()
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs and ("k_interp_ckpt" if alg == "interpolating" else "k_gauss_ckpt") in "".join(isa_lint.disassemble(o) for o in objs)
    for o in objs:
        assert isa_lint.lint(o) == []


@pytest.mark.parametrize("regs", ["1", "0", None])
def test_runtime_tsit5_stage_rows_are_lds_columns_unless_opted_in(tmp_path, monkeypatch, regs):
    """Adaptive Tsit5 of a runtime-registered lane model: by default (and with HIPADJ_TS5_REGS_USER=0) the stage rows are the lane-private LDS columns of
    rounds 1-3 (8 x NZ x 64 doubles); HIPADJ_TS5_REGS_USER=1 opts into the register form the compiled-in models use (no LDS) — an experiment: it passes the
    spill-placement lint and is 20-35 % faster, but one such kernel returned wrong gradients on the GPU (hipadj_user.hpp), so it is never the default."""
    import re
    import subprocess
    from scimlsensitivity_jl_amd import _lib
    m = UM.LV
    name = f"lv_ts5_rows_{regs}"
    _lib.register_model(name, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    if regs is None:
        monkeypatch.delenv("HIPADJ_TS5_REGS_USER", raising=False)
    else:
        monkeypatch.setenv("HIPADJ_TS5_REGS_USER", regs)
    cfg = E.make_config(name, "interpolating", 64, 0.0, 2.0, 0.0, [0.5, 1.0, 2.0], loss_kind=0, stepper=1, abstol=1e-8, reltol=1e-8, checkpointing=False)
    L = _lib.load()
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs
    lds = None
    for o in objs:
        assert isa_lint.lint(o) == []
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", o], capture_output=True, text=True).stdout
        for mm in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)[\s\S]*?\.name:\s+(\S+)", notes):
            if "k_adjoint_tsit5" in mm.group(2):
                lds = int(mm.group(1))
    assert lds is not None and (lds == 0 if regs == "1" else lds == 8 * 6 * 64 * 8)


@pytest.mark.parametrize("name,alg,ck", [("rober", "interpolating", False), ("rober", "quadrature", False), ("ring4", "gauss", False), ("ring6", "gausskronrod", False),
                                         ("rober", "interpolating", True), ("ring4", "gausskronrod", True)])
def test_runtime_rosenbrock23_kernels_compile_without_a_device_and_are_clean(tmp_path, monkeypatch, name, alg, ck):
    """HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE for runtime-registered lane models: k_forward_tsit5<U, 1> and k_adjoint_tsit5<U, ALG, 0, false, 1> (the W solves unrolled over the
    model's n) through hiprtc, and the spill-placement check on what it produced."""
    from scimlsensitivity_jl_amd import _lib
    m = UM.ROBER if name == "rober" else UM.ring(int(name[4:]))
    mname = f"{name}_ros23_lint_{alg}{'_ck' if ck else ''}"
    _lib.register_model(mname, m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))
    cfg = E.make_config(mname, alg, 53, 0.0, 0.5, 0.0, [0.25, 0.5], loss_kind=1, stepper=3, abstol=1e-8, reltol=1e-8, checkpointing=ck)
    L = _lib.load()
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    objs = glob.glob(str(tmp_path / "*.hsaco"))
    assert objs
    for o in objs:
        assert isa_lint.lint(o) == []
