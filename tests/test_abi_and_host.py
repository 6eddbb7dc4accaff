"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every symbol declared in
include/hipadj.h, refuses to run without a device (no CPU fallback), the planner validates configurations the way
the reference errors on misuse, and the Python mirror keeps the reference's names/defaults."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import emu as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "hipadj.h")).read()
    return sorted(set(re.findall(r"\b(hipadj_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(sa):
    from scimlsensitivity_jl_amd import _lib
    L = sa.load_library()
    names = declared_functions()
    assert set(names) == set(_lib.DECLARED_SYMBOLS)
    for nm in names:
        assert hasattr(L, nm), nm
    assert L.hipadj_version() == 110
    assert L.hipadj_status_string(-2).decode().startswith("no usable HIP device")


def test_runtime_models_are_compiled_by_the_build_toolkits_hiprtc(sa):
    """One compiler for all device code: libhipadj.so binds the hiprtc of the ROCm toolkit it was built with, also in a process whose torch wheel
    carries an older hiprtc / comgr pair (then in its own link-map namespace).  The older bundled compiler miscompiled a wide runtime model
    (tests/test_gpu_mass_matrix.py::test_wide_ring_dense_mass_matrix_gauss_regression); HIPADJ_HIPRTC overrides the choice."""
    import os, subprocess, sys
    from scimlsensitivity_jl_amd import _lib
    desc = _lib.runtime_compiler()
    root = os.path.realpath(os.environ.get("ROCM_PATH", "/opt/rocm"))
    assert desc.startswith(root + "/lib/libhiprtc.so"), desc
    with open(os.path.join(root, ".info", "version")) as fh:
        major_minor = ".".join(fh.read().strip().split(".")[:2])
    assert f"; HIP {major_minor}." in desc, desc
    import torch
    bundled = os.path.exists(os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so"))
    if bundled and "HIPADJ_NO_TORCH" not in os.environ:
        assert "[own link-map namespace]" in desc, desc
    # a torch-free host process (what the Julia glue is) takes the same library by a plain dlopen; the explicit override is honoured
    code = "import sys; sys.path.insert(0, %r); from scimlsensitivity_jl_amd import _lib; print(_lib.runtime_compiler())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPADJ_NO_TORCH="1"), capture_output=True, text=True, timeout=300).stdout
    assert out.startswith(root + "/lib/libhiprtc.so") and "namespace" not in out, out
    if bundled:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPADJ_HIPRTC="libhiprtc.so"), capture_output=True, text=True, timeout=300).stdout
        assert out.startswith("libhiprtc.so; HIP "), out


def test_runtime_compiler_survives_setenv_in_the_host_process():
    """The toolkit's hiprtc lives in its own link-map namespace next to torch's — with its own libc, whose `environ` is a copy.  Setting variables in
    the host process after the first compile (os.environ[...] = ... reallocates the array) left that copy dangling: the next getenv inside hiprtc
    crashed (seen on the GPU box as a segfault in hiprtcCreateProgram).  rtc_api().enter() re-points it before every call."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import scimlsensitivity_jl_amd as sa, user_models as UM\n"
        "from scimlsensitivity_jl_amd import _lib\n"
        "print(_lib.runtime_compiler(), flush=True)\n"
        "for i in range(3000): os.environ['HIPADJ_DUMMY_%%d' %% i] = 'y' * 200\n"
        "m = UM.ring(4); f = sa.DeviceFunction('ring4_env', 4, 5, m['f'], m['vjp'], m['vjp_p'])\n"
        "_lib.check_model(f.id)\n"
        "for i in range(3000, 6000): os.environ['HIPADJ_DUMMY_%%d' %% i] = 'z' * 100\n"
        "f.set_mass_matrix([[2.0, 0.1, 0, 0], [0, 1.5, 0, 0.2], [0, 0, 1.0, 0], [0.3, 0, 0, 1.2]]); _lib.check_model(f.id)\n"
        "print('compiled twice', flush=True)\n") % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "compiled twice" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-600:])


def test_runtime_model_kernel_choices(sa, tmp_path, monkeypatch):
    """What user_kernel_names / the planner pick for runtime models, read off the name expressions of the dumped translation units (no device needed):
    a polynomial right-hand side with n <= 3 takes the deep unrolled prefetch (k_interp<U, 8 | 6, ...>), model text that calls library math the rolled
    sweep with one knot in flight (k_interp<U, 1, ...>); a 5- to 8-state model gets the segmented kernels (SEG = true: segment state <= 160 doubles);
    dual-number models bundle their columns only up to three states."""
    import glob
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    import emu as E
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))

    def names(tag, m, alg="interpolating", auto=False):
        mid = _lib.register_model(tag, m["n"], m["np"], m["f"], None if auto else m["vjp"], None if auto else m["vjp_p"])
        cfg = E.make_config(tag, alg, 10000, 0.0, 10.0, 0.01, 0.1 * np.arange(1, 101), loss_kind=0, p_shared=False)
        cfg.model = mid
        L = _lib.load()
        assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
        tu = open(sorted(glob.glob(str(tmp_path / (tag + "_*.hip"))))[-1]).read()
        return tu
    tu = names("choice_lv", UM.LV)
    assert "k_interp_fused<hipadj::UserModel, 6, 0, true, false>" in tu          # round 3: the sweep kernel finishes the pass (one launch)
    tu = names("choice_ring3", UM.ring(3))
    assert "k_interp_fused<hipadj::UserModel, 1, 0, true, false>" in tu and "HAS_COLS = true" in tu
    tu = names("choice_ring8", UM.ring(8), alg="gauss")
    # a segment map of more than 64 entries (8-state ring: 153): the three-launch sequence stays (fused_eligible; the tail would spill heavily)
    assert "k_gauss<hipadj::UserModel, 1, 0, false, true>" in tu and "k_compose_finish<hipadj::UserModel>" in tu and "k_gauss_fused" not in tu
    tu = names("choice_ring4", UM.ring(4), alg="gauss")
    assert "k_gauss_fused<hipadj::UserModel, 1, 0, false, true>" in tu and "k_compose_finish<hipadj::UserModel>" not in tu
    assert "HAS_COLS = true" in names("choice_auto3", UM.ring(3), auto=True)
    assert "HAS_COLS = false" in names("choice_auto4", UM.ring(4), auto=True)


def test_struct_layouts_match_header(sa, tmp_path):
    """ctypes mirrors vs the C compiler's view of include/hipadj.h (sizeof / offsetof)."""
    import subprocess
    from scimlsensitivity_jl_amd import _lib
    src = tmp_path / "layout.c"
    fields_c = [f for f, _ in _lib.HipadjConfig._fields_]
    fields_s = [f for f, _ in _lib.HipadjStats._fields_]
    body = "".join('printf("c %s %%zu\\n", offsetof(hipadj_config, %s));' % (f, f) for f in fields_c)
    body += "".join('printf("s %s %%zu\\n", offsetof(hipadj_stats, %s));' % (f, f) for f in fields_s)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hipadj.h"\nint main(void){'
                   'printf("C %zu\\nS %zu\\n", sizeof(hipadj_config), sizeof(hipadj_stats));' + body + 'return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    got = {}
    for line in out:
        parts = line.split()
        if len(parts) == 2:
            got[parts[0]] = int(parts[1])
        elif len(parts) == 3:
            got[(parts[0], parts[1])] = int(parts[2])
    assert got["C"] == C.sizeof(_lib.HipadjConfig) and got["S"] == C.sizeof(_lib.HipadjStats)
    for f in fields_c:
        assert got[("c", f)] == getattr(_lib.HipadjConfig, f).offset, f
    for f in fields_s:
        assert got[("s", f)] == getattr(_lib.HipadjStats, f).offset, f


def test_model_sizes_through_abi(sa):
    assert sa.model_sizes("lorenz") == (3, 3)
    assert sa.model_sizes("lvt") == (2, 4)
    assert sa.model_sizes("mlp", (2, 128, 4096, 0)) == (2 * 4096, 128 * 2 + 128 + 128 * 128 + 128 + 2 * 128 + 2)
    assert sa.model_sizes("bruss", (32, 0, 0, 0)) == (2048, 3)
    with pytest.raises(sa.HipadjError):
        sa.model_sizes("mlp", (0, 0, 0, 0))


def test_no_cpu_fallback_without_device(sa):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sa.HipadjError) as ei:
        sa.Engine("lorenz", "interpolating", 8, 0.0, 1.0, 0.01, save_times=[1.0])
    assert ei.value.status == -2


@pytest.mark.parametrize("kw,msg", [
    (dict(dt=3e-9), "between 1 and 1e8 steps"),
    (dict(save=[0.5, 0.5]), "strictly ascending"),
    (dict(save=[1.5]), "inside [t0, t1]"),
    (dict(ntraj=0), "ntraj"),
    (dict(dt=-0.1), "dt > 0"),
])
def test_planner_rejects_misuse(kw, msg):
    cfg = E.make_config("lorenz", "interpolating", kw.get("ntraj", 4), 0.0, 1.0, kw.get("dt", 0.01), kw.get("save", [0.5, 1.0]))
    nseg, nck, nq = C.c_int(), C.c_int(), C.c_int()
    b = (C.c_int * 64)()
    rc = E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq))
    assert rc == -1 and msg in E.lib().emu_last_error().decode()


def test_planner_span_not_a_multiple_of_dt_takes_a_short_last_step():
    """dt = 0.03 on (0, 1): 34 forward steps, the last one 0.01 long; such spans always run the off-grid sweeps (the reverse steps start from T
    with the full dt); the PDE / MLP families refuse them."""
    nseg, nck, nq = C.c_int(), C.c_int(), C.c_int()
    b = (C.c_int * 64)()
    cfg = E.make_config("lorenz", "interpolating", 4, 0.0, 1.0, 0.03, [0.51, 0.99], time_segments=3)     # 0.51 = 17 * 0.03: on the knots, still off-grid
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 3 and b[3] >= 34
    cfg = E.make_config("lorenz", "gausskronrod", 4, 0.0, 1.0, 0.03, [0.51], time_segments=3)    # round 5: GaussKronrod runs the reverse step list too, sequential in time
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 1
    cfg = E.make_config("lorenz", "interpolating", 4, 0.0, 1.0, 0.03, [0.51], checkpointing=True, time_segments=3)   # ... and so do the checkpointed sweeps: t0, 0.51, T
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 1 and nck.value == 3


def test_planner_offgrid_loss_times_build_the_reverse_step_list():
    """Loss times off the step grid: accepted for InterpolatingAdjoint / GaussAdjoint (the reverse step list, itself cut into time segments:
    the bounds are positions in that list — 101 steps here: 49 + 1 + 50 + the stop at 0.505 — not knot indices), one segment for
    Backsolve, GaussKronrod and the checkpointed sweeps; Quadrature: one interval pair."""
    nseg, nck, nq = C.c_int(), C.c_int(), C.c_int()
    b = (C.c_int * 64)()
    cfg = E.make_config("lorenz", "interpolating", 4, 0.0, 1.0, 0.01, [0.505, 1.0], time_segments=4)
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 4
    assert b[0] == 0 and b[4] == 101 and all(b[i] < b[i + 1] for i in range(4)) and (b[4] - b[3]) > (b[1] - b[0])     # the 1-column top segment is the longest
    cfg = E.make_config("lorenz", "backsolve", 4, 0.0, 1.0, 0.01, [0.505, 1.0], time_segments=4, checkpointing=True)
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 1
    cfg = E.make_config("lorenz", "quadrature", 4, 0.0, 1.0, 0.01, [0.505, 1.0])
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nq.value == 2     # [0.505, 1.0] and the start correction [0, 0.505]
    cfg = E.make_config("lorenz", "gausskronrod", 4, 0.0, 1.0, 0.01, [0.505, 1.0], time_segments=4)       # round 5: the (7,15) rule per reverse step, sequential in time
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 1
    cfg = E.make_config("lorenz", "gauss", 4, 0.0, 1.0, 0.01, [0.505, 1.0], time_segments=4, checkpointing=True)   # round 5: checkpointed, the checkpoints t0, 0.505, T are stops
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == 0 and nseg.value == 1 and nck.value == 3
    cfg = E.make_config("lorenz", "interpolating", 4, 0.0, 1.0, 0.01, [0.505, 1.0], checkpointing=True, checkpoints=[0.3, 0.3 + 1e-16 * 3])
    assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 64, C.byref(nck), C.byref(nq)) == -1 and "closer than the time resolution" in E.lib().emu_last_error().decode()


def test_planner_segments_checkpoints_and_quadrature_intervals(monkeypatch):
    monkeypatch.setenv("HIPADJ_FUSED_GROUP", "0")      # the plain one-launch pass first (the rule every model gets); the grouped form of the Lorenz stage-operator sweeps below
    def plan(alg, N, S_dt, save, **k):
        cfg = E.make_config("lorenz", alg, N, 0.0, 10.0, S_dt, save, **k)
        nseg, nck, nq = C.c_int(), C.c_int(), C.c_int()
        b = (C.c_int * 128)()
        assert E.lib().emu_plan(C.byref(cfg), C.byref(nseg), b, 128, C.byref(nck), C.byref(nq)) == 0
        return nseg.value, list(b[: nseg.value + 1]), nck.value, nq.value
    ts = np.linspace(0, 10, 101)
    nseg, bounds, _, _ = plan("interpolating", 10000, 0.01, ts, time_segments=0)
    assert nseg == 13 and bounds[0] == 0 and bounds[-1] == 1000 and all(np.diff(bounds) > 0)
    assert bounds[-1] - bounds[-2] > bounds[1] - bounds[0]          # the 1-column top segment is the longest
    assert 157 * nseg <= 2048                                       # one residency round on 1024 SIMDs x 2 waves
    assert plan("interpolating", 10 ** 6, 0.01, ts, time_segments=0)[0] == 1    # big ensembles stay sequential in time
    # strong-scaling shards: two waves per SIMD where the 16-step minimum allows it (2500 trajectories: 40 waves x 51 = 2040), otherwise exactly ONE
    # residency round (1250: 20 waves x 51 = 1020 <= 1024 — not the 62 segments of the step limit, which would leave 216 SIMDs with two waves), and
    # the step limit itself when even that does not fill the chip (640: 10 waves x 62)
    assert plan("interpolating", 5000, 0.01, ts, time_segments=0)[0] == 25
    assert plan("interpolating", 2500, 0.01, ts, time_segments=0)[0] == 51
    assert plan("interpolating", 1250, 0.01, ts, time_segments=0)[0] == 51
    assert plan("interpolating", 640, 0.01, ts, time_segments=0)[0] == 62
    assert plan("interpolating", 100, 0.01, ts, time_segments=5)[0] == 5
    # round 6: the grouped form (G segments per workgroup, first composition level in LDS; hipadj_plan.hpp plan_group_choice, measured: profiles/r6_shard_group_ab.jsonl) —
    # 8-wave workgroups own a CU (<= 256 of them), 4-wave ones pair up (<= 512), the smallest shards keep one wave per SIMD; beyond 170 blocks the plain rule stands
    monkeypatch.delenv("HIPADJ_FUSED_GROUP")
    assert [plan("interpolating", N, 0.01, ts, time_segments=0)[0] for N in (640, 1250, 2500, 5000, 10000, 20000, 10 ** 6)] == [60, 48, 48, 24, 12, 6, 1]
    assert plan("interpolating", 1250, 0.01, ts, time_segments=7)[0] == 7                 # an explicit segment count is kept
    assert plan("gauss", 1250, 0.01, ts, time_segments=0)[0] == 51                        # other sweeps: the plain rule
    monkeypatch.setenv("HIPADJ_FUSED_GROUP", "0")
    assert plan("backsolve", 64, 0.01, ts, checkpointing=True)[2] == 101        # default checkpoints = saved points
    assert plan("backsolve", 64, 0.01, ts[1:-1], checkpointing=True)[2] == 101  # endpoints are always stored
    assert plan("backsolve", 64, 0.01, ts, checkpointing=True, ckpt_stride=250)[2] == 5
    assert plan("backsolve", 64, 0.01, ts, checkpointing=False)[2] == 0
    assert plan("quadrature", 64, 0.01, ts)[3] == 100
    assert plan("quadrature", 64, 0.01, ts[1:-1])[3] == 100                     # + start and end corrections
    assert plan("quadrature", 64, 0.01, [])[3] == 1


def test_sensealg_mirror_defaults_and_errors(sa):
    assert sa.BacksolveAdjoint().checkpointing is True               # src/sensitivity_algorithms.jl:260-265
    assert sa.InterpolatingAdjoint().checkpointing is False
    q = sa.QuadratureAdjoint()
    assert (q.abstol, q.reltol) == (1e-6, 1e-3)                       # :493-497
    assert sa.ischeckpointing(sa.BacksolveAdjoint()) and not sa.ischeckpointing(sa.InterpolatingAdjoint())
    assert sa.ischeckpointing(sa.GaussAdjoint(checkpointing=True))
    with pytest.raises(ValueError):
        sa.InterpolatingAdjoint(autojacvec="ZygoteVJP")
    assert isinstance(sa.GaussAdjoint(autojacvec=sa.DeviceVJP()), sa.AbstractAdjointSensitivityAlgorithm)
    with pytest.raises(ValueError):
        sa.ODEProblem("not_a_model", [1.0], (0, 1), [1.0])
    with pytest.raises(ValueError):
        sa.solve(sa.ODEProblem("lorenz", [1.0, 0, 0], (0, 1), [10.0, 28.0, 8 / 3]), "Tsit5", dt=0.01)


def test_shard_ranges_cover_the_ensemble(sa):
    for n, w in ((10000, 8), (10, 3), (7, 8), (1, 1)):
        r = [sa.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        sizes = [hi - lo for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1


# ---- runtime model ingestion (hipadj_model_register / hipadj_model_check): hiprtc compiles for gfx950 without a device
def test_runtime_model_registration_compiles_for_gfx950_and_reports_errors():
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    m = UM.ring(5)
    mid = _lib.register_model("ring5_cpu_check", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"], check=True)
    assert mid >= _lib.MODEL_USER_BASE and _lib.model_sizes("ring5_cpu_check") == (5, 6)
    with pytest.raises(_lib.HipadjError, match="undeclared identifier"):
        _lib.register_model("broken_rhs", 2, 2, "du[0] = nope; du[1] = 0.0;", "out[0] = 0.0; out[1] = 0.0;", "out[0] = 0.0; out[1] = 0.0;", check=True)
    for n, npar in ((0, 1), (9, 1), (2, 0), (2, 33)):
        with pytest.raises(_lib.HipadjError):
            _lib.register_model("bad_dims", n, npar, "", "", "")
    # automatic VJPs (autojacvec = true): only f is given, compiled for double and for forward-mode dual numbers
    _lib.register_model("mm_auto_cpu_check", 2, 3, "real s = u[0] / (p[2] + u[0]); du[0] = -p[0]*s + p[1]*u[1]; du[1] = p[0]*s - p[1]*u[1]*exp(-0.1*t);", check=True)
    with pytest.raises(_lib.HipadjError):                      # one VJP without the other
        _lib.register_model("half_vjp", 2, 2, "du[0] = u[0]; du[1] = u[1];", "out[0] = lam[0]; out[1] = lam[1];", None)


def test_runtime_model_mass_matrix_registration():
    """ODEFunction(f; mass_matrix = M) (test/Core3/adjoint.jl:1315-1325): a constant non-singular M is folded into the generated model (hand VJPs
    and dual-number VJPs both compile for gfx950); a singular one is refused with the reason unless it has the semi-explicit form (next test)."""
    import user_models as UM
    import scimlsensitivity_jl_amd as sa
    from scimlsensitivity_jl_amd import _lib
    m = UM.AFFINE3
    f = sa.DeviceFunction("affine3_mm_cpu", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"], mass_matrix=UM.AFFINE3_MM, check=True)
    assert np.array_equal(f.mass_matrix, np.array(UM.AFFINE3_MM))
    sa.DeviceFunction("affine3_mm_auto_cpu", m["n"], m["np"], m["f"], mass_matrix=UM.AFFINE3_MM, check=True)
    with pytest.raises(_lib.HipadjError, match="singular") as e:      # singular and NOT of the semi-explicit form [Md 0; 0 0]
        f.set_mass_matrix(np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))
    assert e.value.status == -6
    with pytest.raises(_lib.HipadjError, match="singular"):             # a zero row whose column is not zero
        f.set_mass_matrix(np.array([[1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]]))
    with pytest.raises(ValueError):
        f.set_mass_matrix(np.eye(2))
    with pytest.raises(_lib.HipadjError):
        f.set_mass_matrix(np.full((3, 3), np.nan))
    with pytest.raises(_lib.HipadjError):
        _lib.set_model_mass_matrix(_lib.MODEL["lorenz"], 3, np.eye(3))      # compiled-in models carry none
    f.set_mass_matrix(None)
    assert f.mass_matrix is None
    _lib.check_model(f.id)


def test_runtime_model_singular_mass_matrix_is_a_dae_for_the_stiff_stepper_only(tmp_path, monkeypatch):
    """mass_matrix = diag(1, 1, 0) on `rober` (test/Core3/adjoint.jl:1434-1454): accepted as a semi-explicit DAE; hipadj_model_check_config compiles the Rosenbrock23 kernels of
    the generated model (DAE / mass / isalg members, hiprtc, no device) for every sensealg it is offered with and refuses the other steppers by name."""
    import ctypes as C
    import emu as E
    import user_models as UM
    import scimlsensitivity_jl_amd as sa
    from scimlsensitivity_jl_amd import _lib
    m = UM.ROBERDAE
    f = sa.DeviceFunction("roberdae_cpu", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"], mass_matrix=UM.ROBERDAE_MM)
    L = _lib.load()
    for alg in ("interpolating", "gauss", "gausskronrod", "quadrature"):
        cfg = E.make_config("roberdae_cpu", alg, 8, 0.0, 100.0, 0.0, [50.0, 100.0], loss_kind=0, stepper=3, abstol=1e-8, reltol=1e-6)
        assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)
    for stepper, kw in ((0, dict(dt=0.01)), (1, dict(dt=0.0))):
        cfg = E.make_config("roberdae_cpu", "interpolating", 8, 0.0, 1.0, kw["dt"], [1.0], loss_kind=0, stepper=stepper)
        assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.ERR_UNSUPPORTED and b"singular" in L.hipadj_last_error(None)
    cfg = E.make_config("roberdae_cpu", "backsolve", 8, 0.0, 1.0, 0.0, [1.0], loss_kind=0, stepper=3, checkpointing=True)
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.ERR_UNSUPPORTED
    f.set_mass_matrix(None)            # back to an ODE model: every stepper again
    cfg = E.make_config("roberdae_cpu", "interpolating", 8, 0.0, 1.0, 0.01, [1.0], loss_kind=0, stepper=0)
    assert L.hipadj_model_check_config(C.byref(cfg)) == _lib.OK, L.hipadj_last_error(None)


def test_runtime_model_affect_registration_and_loud_failure_without_a_device():
    """DiscreteCallback affects (hipadj_model_set_affect): the affect and its dual-number VJPs compile for gfx950 with the model; the host-level
    composition (events.py) calls hipadj_affect_apply / _vjp, which fail loudly without a device — no CPU fallback."""
    import os
    import user_models as UM
    import scimlsensitivity_jl_amd as sa
    from scimlsensitivity_jl_amd import _lib
    m = UM.LV
    f = sa.DeviceFunction("lv_affect_cpu", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"]).set_affect("for (int i = 0; i < N; ++i) un[i] += p[1] / 8.0 * sin(u[i]);")
    _lib.check_model(f.id)
    bad = sa.DeviceFunction("lv_affect_bad_cpu", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"]).set_affect("un[0] += nope;")
    with pytest.raises(_lib.HipadjError, match="undeclared identifier"):
        _lib.check_model(bad.id)
    with pytest.raises(_lib.HipadjError):
        _lib.set_model_affect(_lib.MODEL["lorenz"], "un[0] += 1.0;")          # compiled-in models carry none
    assert sa.PresetTimeCallback([3, 1.5]).times == (3.0, 1.5)
    if not os.path.exists("/dev/kfd"):
        with pytest.raises(_lib.HipadjError) as e:
            _lib.affect_apply(f.id, np.ones((4, 2)), np.ones(4), 1.0, 4)
        assert e.value.status == -2
        with pytest.raises(_lib.HipadjError) as e:
            _lib.affect_vjp(f.id, np.ones((4, 2)), np.ones(4), 1.0, np.ones((4, 2)), np.zeros((4, 4)))
        assert e.value.status == -2


def test_runtime_model_plans_like_a_lane_model():
    """A registered model goes through the same planner: segmentation only while (1+n)(n+np) columns fit the registers."""
    import emu as E
    import user_models as UM
    from scimlsensitivity_jl_amd import _lib
    import ctypes as C
    for n, expect_seg in ((4, True), (6, False)):
        m = UM.ring(n)
        _lib.register_model(f"ring{n}_plan", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
        cfg = E.make_config(f"ring{n}_plan", "interpolating", 64, 0.0, 10.0, 0.01, [10.0], loss_kind=1, time_segments=0)
        L = _lib.load()
        h = C.c_void_p()
        rc = L.hipadj_create(C.byref(cfg), C.byref(h))
        # no GPU in the CPU suite: planning succeeded iff the failure is NO_DEVICE (not INVALID_ARG / UNSUPPORTED)
        assert rc in (_lib.OK, _lib.ERR_NO_DEVICE), L.hipadj_last_error(None)
        if rc == _lib.OK:
            L.hipadj_destroy(h)


def test_save_times_follow_the_reference_saving_rules(sa):
    """src/concrete_solve.jl:713-770: scalar saveat = the range t0:saveat:T with T appended when the span is not a multiple
    (fix_endpoints); a list is sorted; an empty saveat = every step, trimmed by save_start / save_end."""
    from scimlsensitivity_jl_amd.interface import _save_times
    ts = _save_times((0.0, 10.0), 0.1, 0.01)
    assert len(ts) == 101 and ts[0] == 0.0 and ts[-1] == 10.0 and np.allclose(np.diff(ts), 0.1)
    ts = _save_times((0.0, 0.3), 0.1, 0.01)                      # 0.3 / 0.1 = 2.9999999999999996 in floating point
    assert len(ts) == 4 and ts[-1] == 0.3
    ts = _save_times((0.0, 1.0), 0.35, 0.01)                     # not a multiple: 0, .35, .7 and the end point
    assert np.allclose(ts, [0.0, 0.35, 0.7, 1.0])
    ts = _save_times((1.0, 2.0), [1.75, 1.25, 2.0], 0.01)
    assert np.array_equal(ts, [1.25, 1.75, 2.0])
    assert len(_save_times((0.0, 1.0), None, 0.01)) == 0         # no discrete loss
    ts = _save_times((0.0, 1.0), None, 0.25, save_everystep=True)
    assert np.array_equal(ts, [0.0, 0.25, 0.5, 0.75, 1.0])
    assert np.array_equal(_save_times((0.0, 1.0), None, 0.25, True, False, True), [0.25, 0.5, 0.75, 1.0])
    assert np.array_equal(_save_times((0.0, 1.0), None, 0.25, True, False, False), [0.25, 0.5, 0.75])
    with pytest.raises(ValueError):
        _save_times((0.0, 1.0), None, 0.0, save_everystep=True)


def test_comm_entry_points_validate_and_bind_rccl_lazily(sa):
    """hipadj_comm_* (include/hipadj.h): NULL handles / ids are rejected; the unique id comes from RCCL bound with dlopen
    (no link-time dependency: `ldd libhipadj.so` does not list librccl)."""
    import subprocess
    L = sa.load_library()
    assert L.hipadj_comm_unique_id(None) == -1 and L.hipadj_comm_init_rank(None, None, 1, 0) == -1
    assert L.hipadj_comm_attach(None, None) == -1 and L.hipadj_comm_destroy(None) == -1
    assert L.hipadj_comm_count(None, None) == -1 and L.hipadj_comm_selfcheck(None) == -1
    a, b = sa.comm_unique_id(), sa.comm_unique_id()
    assert len(a) == 128 and a != b
    assert L.hipadj_status_string(-8) == b"RCCL error"
    needed = subprocess.check_output(["readelf", "-d", sa.LIB_PATH]).decode()
    assert "rccl" not in needed and "hiprtc" not in needed


def test_hand_declared_rccl_abi_matches_the_toolkits_header():
    """csrc/hipadj_comm.hpp binds RCCL with dlopen and declares the few enumerators / signatures it uses by hand; tests/c/rccl_abi_check.cpp
    static_asserts every one of them against <rccl/rccl.h> (compile only).  The N > 1 all-reduce has never run on this pool's 1-GPU boxes:
    this is what keeps a silently different datatype / operator number or id size from reaching the driver's 8-GPU run."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "-fsyntax-only", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "scimlsensitivity.jl_amd", "csrc"),
                        os.path.join(ROOT, "tests", "c", "rccl_abi_check.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _build_host_demo(sa, tmp_path):
    import subprocess
    exe = str(tmp_path / "host_demo")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host_demo.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_c_host_compiles_against_the_header_and_fails_loudly_without_a_device(sa, tmp_path):
    """include/hipadj.h is valid strict C99, every entry point the example uses links from libhipadj.so, and a plain C host gets
    HIPADJ_ERR_NO_DEVICE with a message on a machine without a GPU (no CPU fallback)."""
    import subprocess
    sa.load_library()
    exe = _build_host_demo(sa, tmp_path)
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr and "version 110" in r.stdout


def test_c_stiff_dae_example_compiles_and_fails_loudly_without_a_device(sa, tmp_path):
    """examples/stiff_dae_demo.c (the reference's singular-mass-matrix problem through the C ABI: model as text, singular semi-explicit mass matrix, Rosenbrock23): strict C99 against
    include/hipadj.h; registration and the mass matrix succeed without a device, hipadj_create reports HIPADJ_ERR_NO_DEVICE."""
    import subprocess
    sa.load_library()
    exe = str(tmp_path / "stiff_dae_demo")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "stiff_dae_demo.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir, "-lm"])
    if os.path.exists("/dev/kfd"):
        pytest.skip("a device is present: the GPU suite runs the example (tests/test_gpu_stiff.py)")
    r = subprocess.run([exe, "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "hipadj_create" in r.stderr and "no usable HIP device" in r.stderr


def test_c_bouncing_ball_example_compiles_and_fails_loudly_without_a_device(sa, tmp_path):
    """examples/bouncing_ball_demo.c (the reference's ContinuousCallback problem through the C ABI: f, condition and affect as text): strict C99 against include/hipadj.h;
    registration and hipadj_model_set_continuous_callback succeed without a device, hipadj_create reports HIPADJ_ERR_NO_DEVICE."""
    import subprocess
    sa.load_library()
    exe = str(tmp_path / "bouncing_ball_demo")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "bouncing_ball_demo.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir, "-lm"])
    if os.path.exists("/dev/kfd"):
        pytest.skip("a device is present: the GPU suite runs the example (tests/test_gpu_continuous_callbacks.py)")
    r = subprocess.run([exe, "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "hipadj_create" in r.stderr and "no usable HIP device" in r.stderr


def test_solve_attaches_a_continuous_callback_to_the_model(sa, monkeypatch):
    """solve(...; callback = ContinuousCallback(condition, affect)): the host mirror hands the two bodies to hipadj_model_set_continuous_callback ONCE per (model, callback) and
    refuses compiled-in models (no text to compile the bodies next to)"""
    from scimlsensitivity_jl_amd import _lib, interface
    calls = []
    monkeypatch.setattr(_lib, "set_model_continuous_callback", lambda *a: calls.append(a))
    monkeypatch.setattr(interface, "Engine", _StubEngine)
    f = sa.DeviceFunction("cc_host_mirror", 2, 2, "du[0] = u[1]; du[1] = -p[0];")
    cb = sa.ContinuousCallback("c = u[0];", "un[1] = -p[1] * u[1];", 32)
    u0 = np.array([[5.0, 0.0]]); p = np.array([9.8, 0.8])
    for _ in range(2):
        sa.solve(sa.EnsembleProblem(sa.ODEProblem(f, u0[0], (0.0, 2.5), p), u0), sa.Tsit5(), saveat=[0.5, 2.5], sensealg=sa.InterpolatingAdjoint(), abstol=1e-8, reltol=1e-8, callback=cb)
    assert calls == [(f.id, "c = u[0];", "un[1] = -p[1] * u[1];", 32, 1)]          # (..., max_events, ncond)
    with pytest.raises(ValueError, match="runtime lane model"):
        sa.solve(sa.EnsembleProblem(sa.ODEProblem("fallmass", u0[0], (0.0, 2.5), p), u0), sa.Tsit5(), saveat=[0.5, 2.5], sensealg=sa.InterpolatingAdjoint(), abstol=1e-8, reltol=1e-8, callback=cb)


class _StubEngine:
    """Stands in for the device handle (which needs a GPU) so that the HOST logic of interface.py — argument mapping, saving rules,
    cotangent packing, dgdp_discrete, error paths — runs in the CPU suite.  It records the configuration it was created with and
    returns recognisable arrays; no arithmetic of the path happens here (the parity tests proper are the `-m gpu` suite)."""
    created = []

    def __init__(self, model, alg, ntraj, t0, t1, dt, save_times=(), **kw):
        from types import SimpleNamespace
        self.model, self.alg, self.N, self.kw = model, alg, int(ntraj), dict(kw)
        self.t0, self.t1, self.dt = t0, t1, dt
        self.save = np.asarray(save_times, dtype=np.float64)
        self.M, self.n, self.np = len(self.save), 3, 3
        self.p_shared = bool(kw.get("p_shared", True))
        self.cfg = SimpleNamespace(loss_kind=kw.get("loss_kind", 0), loss_shift=kw.get("loss_shift", 0.0))
        self.adjoint_args = []
        _StubEngine.created.append(self)

    def forward(self, u0, p, want_out=True):
        return np.arange(self.N * self.M * self.n, dtype=np.float64).reshape(self.N, self.M, self.n) if (want_out and self.M) else None

    def adjoint(self, dLdu=None):
        self.adjoint_args.append(None if dLdu is None else np.array(dLdu))
        return np.ones((self.N, self.n)), (np.full(self.np, 2.0) if self.p_shared else np.full((self.N, self.np), 2.0))

    def close(self):
        pass


def test_host_mirror_logic_with_a_stub_engine(sa, monkeypatch):
    from scimlsensitivity_jl_amd import interface, _lib
    monkeypatch.setattr(interface, "Engine", _StubEngine)
    _StubEngine.created.clear()
    u0 = np.zeros((4, 3)); p = np.array([10.0, 28.0, 8 / 3])
    prob = sa.EnsembleProblem(sa.ODEProblem("lorenz", u0[0], (0.0, 1.0), p), u0)
    # saving rules -> engine configuration (src/concrete_solve.jl:713-770, 962)
    sol = sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.25, sensealg=sa.BacksolveAdjoint(), save_start=False)
    e = _StubEngine.created[-1]
    assert e.alg == "backsolve" and np.allclose(e.save, [0, 0.25, 0.5, 0.75, 1.0]) and e.kw["no_start"] is True and e.kw["checkpointing"] is True
    assert e.kw["stepper"] == 0 and e.kw["loss_kind"] == _lib.LOSS_COTANGENT and sol.u.shape == (4, 5, 3)
    sol = sa.solve(prob, sa.Tsit5(), saveat=[0.9, 0.3], sensealg=sa.QuadratureAdjoint(abstol=1e-9, reltol=1e-8), abstol=1e-7, reltol=1e-5, dgdu_discrete=sa.LsqShift(2.0))
    e = _StubEngine.created[-1]
    assert e.kw["stepper"] == 1 and np.allclose(e.save, [0.3, 0.9]) and e.kw["no_start"] is False and e.dt == 0.0
    assert (e.kw["quad_abstol"], e.kw["quad_reltol"], e.kw["abstol"], e.kw["reltol"]) == (1e-9, 1e-8, 1e-7, 1e-5)
    assert e.kw["loss_kind"] == _lib.LOSS_LSQ_SHIFT and e.kw["loss_shift"] == 2.0
    du0, dp = sa.adjoint_sensitivities(sol, sa.Tsit5(), t=sol.t, dgdu_discrete=sa.LsqShift(2.0))
    assert e.adjoint_args == [None] and du0.shape == (4, 3) and dp.shape == (3,)
    with pytest.raises(ValueError, match="specialised"):
        sa.adjoint_sensitivities(sol, sa.Tsit5(), dgdu_discrete=sa.LsqShift(3.0))
    with pytest.raises(ValueError, match="save times"):
        sa.adjoint_sensitivities(sol, sa.Tsit5(), t=[0.3, 0.8], dgdu_discrete=sa.LsqShift(2.0))
    with pytest.raises(ValueError, match="re-run solve"):
        sa.adjoint_sensitivities(sol, sa.Tsit5(), sensealg=sa.GaussAdjoint(), dgdu_discrete=sa.LsqShift(2.0))
    # a sensealg with other options, or another checkpoint list, at adjoint time is an error, not silently the solve-time value
    with pytest.raises(ValueError, match="differs from the one the forward solve"):
        sa.adjoint_sensitivities(sol, sa.Tsit5(), sensealg=sa.QuadratureAdjoint(abstol=1e-12, reltol=1e-12), dgdu_discrete=sa.LsqShift(2.0))
    sa.adjoint_sensitivities(sol, sa.Tsit5(), sensealg=sa.QuadratureAdjoint(abstol=1e-9, reltol=1e-8), dgdu_discrete=sa.LsqShift(2.0))
    solc = sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.25, sensealg=sa.BacksolveAdjoint(), checkpoints=[0.0, 0.5, 1.0])
    sa.adjoint_sensitivities(solc, sa.RK4(), checkpoints=[0.0, 0.5, 1.0], dgdu_discrete=np.zeros((4, 5, 3)))
    with pytest.raises(ValueError, match="checkpoints differ"):
        sa.adjoint_sensitivities(solc, sa.RK4(), checkpoints=[0.0, 0.25, 0.5, 0.75, 1.0], dgdu_discrete=np.zeros((4, 5, 3)))
    with pytest.raises(ValueError, match="checkpoints differ"):
        sa.adjoint_sensitivities(sol, sa.Tsit5(), checkpoints=[0.3, 0.9], dgdu_discrete=sa.LsqShift(2.0))
    # equally spaced custom checkpoints -> ckpt_stride; anything else is refused
    sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.25, sensealg=sa.BacksolveAdjoint(), checkpoints=[0.0, 0.2, 0.4, 0.6, 0.8, 1.0])
    assert _StubEngine.created[-1].kw["ckpt_stride"] == 20
    # any other ascending list goes through as the explicit checkpoint list of ABI 102
    sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.25, sensealg=sa.BacksolveAdjoint(), checkpoints=[0.0, 0.2, 0.5, 1.0])
    assert "ckpt_stride" not in _StubEngine.created[-1].kw and np.allclose(_StubEngine.created[-1].kw["checkpoints"], [0.0, 0.2, 0.5, 1.0])
    sa.solve(prob, sa.Tsit5(), saveat=[0.9, 0.3], sensealg=sa.InterpolatingAdjoint(checkpointing=True), checkpoints=[0.77, 0.1234])
    assert np.allclose(_StubEngine.created[-1].kw["checkpoints"], [0.1234, 0.77])
    with pytest.raises(ValueError, match="distinct"):
        sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.25, sensealg=sa.BacksolveAdjoint(), checkpoints=[0.0, 0.5, 0.5, 1.0])
    # cotangent forms of the pullback (src/concrete_solve.jl:776-869): dense array, vector of per-time arrays with NoTangent entries, only_end
    out, pull = sa.concrete_solve_adjoint(prob.prob, sa.RK4(), sa.InterpolatingAdjoint(), u0, p, dt=0.01, saveat=0.5)
    e = _StubEngine.created[-1]
    assert out.shape == (4, 3, 3) and pull.only_end is False
    dense = np.arange(36.0).reshape(4, 3, 3)
    pull(dense); pull([dense[:, 0], None, dense[:, 2]]); pull(dense.ravel())
    assert np.array_equal(e.adjoint_args[0], dense) and np.array_equal(e.adjoint_args[2], dense)
    assert np.array_equal(e.adjoint_args[1][:, 0], dense[:, 0]) and not e.adjoint_args[1][:, 1].any() and np.array_equal(e.adjoint_args[1][:, 2], dense[:, 2])
    with pytest.raises(ValueError, match="one entry per save time"):
        pull([dense[:, 0]])
    out, pull = sa.concrete_solve_adjoint(prob.prob, sa.RK4(), sa.InterpolatingAdjoint(), u0, p, dt=0.01, saveat=[1.0])     # a single save time == T
    e = _StubEngine.created[-1]
    assert pull.only_end is True and out.shape == (4, 1, 3)
    pull(np.ones((4, 3))); pull([2 * np.ones((4, 3))]); pull(np.ones((4, 1, 3)))                                               # sol[end], [sol[end]], Array(sol)
    assert [a.shape for a in e.adjoint_args] == [(4, 1, 3)] * 3 and e.adjoint_args[1][0, 0, 0] == 2.0
    # save_idxs: cotangents of the saved components are scattered into the full state, zeros elsewhere (:790-824)
    sol = sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.5, sensealg=sa.InterpolatingAdjoint(), save_idxs=[2, 0])
    e = _StubEngine.created[-1]
    assert sol.u.shape == (4, 3, 2) and np.array_equal(sol.u[..., 0], np.arange(36.0).reshape(4, 3, 3)[..., 2])
    delta = np.arange(24.0).reshape(4, 3, 2)
    sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=delta)
    full = e.adjoint_args[-1]
    assert full.shape == (4, 3, 3) and np.array_equal(full[..., 2], delta[..., 0]) and np.array_equal(full[..., 0], delta[..., 1]) and np.all(full[..., 1] == 0)
    with pytest.raises(ValueError, match="out of range"):
        sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.5, save_idxs=[3])
    # dgdp_discrete: sum over the save times (and the ensemble for shared p) added to dp once (src/adjoint_common.jl:775-779)
    sol = sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.5, sensealg=sa.InterpolatingAdjoint())
    dl = np.ones((4, 3, 3))
    _, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=np.zeros((4, 3, 3)), dgdp_discrete=dl)
    assert np.array_equal(dp, np.full(3, 2.0 + 12.0))
    _, dp = sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=np.zeros((4, 3, 3)), dgdp_discrete=lambda u, p_, t, i: np.full((4, 3), float(i)))
    assert np.array_equal(dp, np.full(3, 2.0 + 4 * (0 + 1 + 2)))
    # the (out, pullback) contract of _concrete_solve_adjoint
    out, pullback = sa.concrete_solve_adjoint(prob.prob, sa.RK4(), sa.GaussAdjoint(), u0, p, dt=0.01, saveat=0.5)
    du0, dp = pullback(np.ones(out.size))
    assert out.shape == (4, 3, 3) and du0.shape == (4, 3) and _StubEngine.created[-1].adjoint_args[-1].shape == (4, 3, 3)
    # misuse
    with pytest.raises(ValueError, match="needs dt"):
        sa.solve(prob, sa.RK4(), saveat=0.5)
    with pytest.raises(TypeError):
        sa.solve(prob, sa.RK4(), dt=0.01, saveat=0.5, sensealg="interpolating")
    with pytest.raises(TypeError, match="unsupported keyword"):
        sa.adjoint_sensitivities(sol, sa.RK4(), dgdu_discrete=np.zeros((4, 3, 3)), callback=object())


def test_bench_cpu_baseline_leg_runs_on_a_small_sample():
    """bench.py's cpu_baseline leg (the oracle = CPU restatement, timed beside the GPU path on rank 0 at N = 1) on a tiny sample: keys,
    units and the parity inputs it hands back; bench.py's synthetic inputs are deterministic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    u0, p = bench.inputs(300)
    u0b, _ = bench.inputs(300)
    assert np.array_equal(u0, u0b) and u0.shape == (300, 3) and np.allclose(p, [10.0, 28.0, 8 / 3])
    ts = np.linspace(0.0, bench.T_FINAL, 101)
    cb = bench.cpu_baseline(u0, p, ts, budget_s=0.05)
    assert cb["kind"] == "port" and cb["unit"] == "trajectories/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["single_thread_value"] > 0
    assert "sample" in cb and f"{cb['cores']} of {cb['host_threads']} host threads" in cb["cores_note"]
    assert cb["repeats"] >= 5 and cb["spread_min_max"][0] <= cb["value"] <= cb["spread_min_max"][1] and str(cb["cores"]) + ":" in cb["thread_probe_traj_per_s"]
    assert np.allclose(bench.save_times(), ts) and bench.oracle_problem().M == 101


def test_an_untrusted_runtime_compiler_gets_the_conservative_limits(tmp_path):
    """when the toolkit's hiprtc cannot be bound by path (a torch wheel without a ROCm installation) the soname resolves to whatever the process
    carries — the bundled ROCm 7.0 pair miscompiled wide runtime models.  rtc_trusted() then limits segment lanes to 64 doubles of state
    again (an 8-state ring runs the one-column kernels: SEG = false, k_finish_map) and says so once on stderr.  HIPADJ_RTC_TRUST=0 stands in for that situation."""
    import subprocess
    import sys
    code = r'''
import sys, glob, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import scimlsensitivity_jl_amd as sa, user_models as UM, emu as E
from scimlsensitivity_jl_amd import _lib
m = UM.ring(8)
mid = _lib.register_model("trust_ring8", m["n"], m["np"], m["f"], m["vjp"], m["vjp_p"])
cfg = E.make_config("trust_ring8", "interpolating", 10000, 0.0, 10.0, 0.01, 0.1 * np.arange(1, 101), loss_kind=0, p_shared=False); cfg.model = mid
L = _lib.load()
assert L.hipadj_model_check_config(C.byref(cfg)) == 0, L.hipadj_last_error(None)
tu = open(sorted(glob.glob(sys.argv[1] + "/trust_ring8_*.hip"))[-1]).read()
print("SEG_TRUE" if "k_interp<hipadj::UserModel, 1, 0, true>" in tu else "SEG_FALSE" if "k_interp_fused<hipadj::UserModel, 1, 0, false, false>" in tu else "UNKNOWN")
''' % (ROOT, os.path.join(ROOT, "tests"))
    out = {}
    for trust in ("1", "0"):
        d = tmp_path / trust; d.mkdir()
        env = dict(os.environ, HIPADJ_RTC_TRUST=trust, HIPADJ_RTC_DUMP=str(d))
        r = subprocess.run([sys.executable, "-c", code, str(d)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[trust] = (r.stdout.strip().splitlines()[-1], r.stderr)
    assert out["1"][0] == "SEG_TRUE" and out["0"][0] == "SEG_FALSE"
