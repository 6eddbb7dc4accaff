"""The Julia side of the boundary (julia/) cannot run here — no Julia in the image — so its ABI side is checked from C and by text:

  * tests/c/julia_seam.c replays the call sequence of HIPAdj.Handle / forward! / adjoint! with the configuration written byte by byte
    at the offsets of HIPAdj.CONFIG_OFFSETS (static-asserted against include/hipadj.h): CPU — compiles, links, loud NO_DEVICE;
    GPU — its numbers against the oracle (rtol 1e-6);
  * the field list and offset table in julia/HIPAdj/src/HIPAdj.jl agree with the header."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(sa, tmp_path):
    exe = str(tmp_path / "julia_seam")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "julia_seam.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir])
    return exe


def _lcg_stream(seed, count):
    out = np.empty(count); s = seed; mask = (1 << 64) - 1
    for i in range(count):
        s = (s * 6364136223846793005 + 1442695040888963407) & mask
        out[i] = ((s >> 11) & ((1 << 53) - 1)) / float(1 << 53) - 0.5
    return out


def test_julia_struct_text_matches_header():
    jl = open(os.path.join(ROOT, "julia", "HIPAdj", "src", "HIPAdj.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "hipadj.h")).read()
    body = jl[jl.index("struct HipadjConfig"):]
    body = body[:body.index("\nend")]
    jl_fields = re.findall(r"^\s+(\w+)::", body, re.M)
    cstruct = hdr[hdr.index("typedef struct {\n    uint32_t struct_size;      /* = sizeof(hipadj_config)"):hdr.index("} hipadj_config;")]
    cstruct = re.sub(r"/\*.*?\*/", "", cstruct, flags=re.S)
    c_fields = []
    for decl in cstruct.split(";"):
        decl = decl.strip()
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        for nm in names.split(","):
            c_fields.append(re.sub(r"[\*\s]|\[\d+\]", "", nm))
    assert jl_fields == c_fields
    offs = [int(x) for x in re.search(r"const CONFIG_OFFSETS = \(([\d,\s]+)\)", jl).group(1).replace("\n", " ").split(",")]
    assert len(offs) == len(c_fields)
    c_src = open(os.path.join(ROOT, "tests", "c", "julia_seam.c")).read()
    c_offs = {m.group(1): int(m.group(2)) for m in re.finditer(r"O_(\w+) = (\d+)", c_src)}
    assert [c_offs[f] for f in c_fields] == offs                       # the C replay asserts THESE against offsetof() at compile time
    assert int(re.search(r"const CONFIG_SIZE = (\d+)", jl).group(1)) == int(re.search(r"CONFIG_SIZE = (\d+) \}", c_src).group(1))
    assert "v == 110" in jl and "#define HIPADJ_VERSION 110" in hdr


def test_julia_call_sequence_from_c_fails_loudly_without_a_device(tmp_path):
    import scimlsensitivity_jl_amd as sa
    sa.build_extension()
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the GPU variant of this test runs the sequence for real")
    exe = _build(sa, tmp_path)
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 1 and "hipadj status -2" in r.stderr and "no usable HIP device" in r.stderr and "version 110" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("alg,oalg", [(0, "INTERPOLATING"), (1, "BACKSOLVE"), (2, "GAUSS"), (3, "QUADRATURE")])
def test_julia_call_sequence_from_c_matches_oracle(tmp_path, alg, oalg):
    import oracle as O
    import scimlsensitivity_jl_amd as sa
    sa.build_extension()
    exe = _build(sa, tmp_path)
    N, n, M = 96, 3, 11
    r = subprocess.run([exe, str(N), str(alg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    vals = {l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in r.stdout.strip().split("\n")}
    st = _lcg_stream(20240926, 3 * N + n * M * N)
    u0 = st[:3 * N].reshape(N, 3) * 0.1; u0[:, 0] += 1.0
    delta = st[3 * N:].reshape(N, M, n)
    ts = 0.1 * np.arange(M); ts[-1] = 1.0
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    ref = O.Problem("LORENZ", alg=oalg, stepper="RK4", t0=0.0, t1=1.0, dt=0.01, save_times=ts, loss="COTANGENT", checkpointing=(alg == 1))
    rdu0, rdp, rout, _ = ref.adjoint_ensemble(u0, p, delta)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    assert rel(vals["dp"], rdp) < 1e-6 and rel(vals["du0_first"], rdu0[0]) < 1e-6 and rel(vals["du0_last"], rdu0[-1]) < 1e-6
    assert rel(vals["out_last"], rout[-1, -1]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("alg", [0, 1, 2, 3])
def test_julia_call_sequence_over_eight_virtual_shards(tmp_path, alg):
    """ONE handle over several devices through the HOST-pointer calls the Julia binding makes (`Handle(...; devices = ...)`, hipadj_config.device_ids).
    A 1-GPU box runs the eight shards as virtual shards on device 0: out and du0 are sliced, so they must equal the single-device run; dp is the sum of the
    shards' partials in shard order — compared at 1e-12 (the summation order differs from the single handle's block tree)."""
    import scimlsensitivity_jl_amd as sa
    sa.build_extension()
    exe = _build(sa, tmp_path)
    N = 203        # not a multiple of 8: ragged ranges
    runs = []
    for G in (0, 8):
        r = subprocess.run([exe, str(N), str(alg), str(G)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        runs.append({l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in r.stdout.strip().split("\n")})
    one, eight = runs
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    assert np.array_equal(one["out_last"], eight["out_last"])
    # du0 of a trajectory does not depend on its neighbours — up to the planner's choice of time segments, which follows the shard's size: roundoff, not bits
    assert rel(eight["du0_first"], one["du0_first"]) < 1e-12 and rel(eight["du0_last"], one["du0_last"]) < 1e-12
    assert rel(eight["dp"], one["dp"]) < 1e-12


def _build_model_calls(sa, tmp_path):
    exe = str(tmp_path / "julia_model_calls")
    libdir = os.path.dirname(sa.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "julia_model_calls.c"),
                           "-o", exe, "-L" + libdir, "-lhipadj", "-Wl,-rpath," + libdir])
    return exe


def test_julia_model_calls_from_c(tmp_path):
    """register_model / set_mass_matrix! / set_affect! / affect_apply / affect_vjp as HIPAdj.jl calls them.  Without a device: registration, the
    singular-matrix refusal and the gfx950 compile work, the first device call fails loudly (status -2); with one: the affect numbers."""
    import scimlsensitivity_jl_amd as sa
    sa.build_extension()
    exe = _build_model_calls(sa, tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert "singular -6" in r.stdout and "semi_explicit 0" in r.stdout and "version 110" in r.stdout
    if not os.path.exists("/dev/kfd"):
        assert r.returncode == 1 and "hipadj status -2" in r.stderr and "no usable HIP device" in r.stderr
        return
    assert r.returncode == 0, r.stderr
    vals = {l.split()[0]: np.array([float(x) for x in l.split()[1:]]) for l in r.stdout.strip().split("\n") if l.split()[0] in ("out", "pout", "lam_out", "gp_out")}
    # un[0] += 2 p[3]; pn[1] = 1.1 p[1]   with p = [1.5, 1.0, 3.0, 0.5]
    assert np.allclose(vals["out"], [2.0, 2.0, 6.0, 6.0]) and np.allclose(vals["pout"], [1.5, 1.1, 3.0, 0.5])
    # lam_out = lam (dun/du = I, dpn/du = 0); gp_out = (dun/dp)' lam + (dpn/dp)' gp: row 0: gp = [0.1, 0.2, 0.3, 0.4], lam = [1, -1]
    assert np.allclose(vals["lam_out"], [1.0, 3.0]) and np.allclose(vals["gp_out"], [0.1, 1.1 * 0.2, 0.3, 0.4 + 2.0 * 1.0])
