"""Wide runtime models (hipadj_wmodel_register, csrc/hipadj_wide.hpp) without a GPU: argument validation, the emitters of the host mirror,
compilation of every kernel of the family for gfx950 with hiprtc (no device needed), and the planner's answers."""
import ctypes as C

import numpy as np
import pytest


def test_registration_validates_its_arguments(sa):
    L = sa.load_library()
    mid = C.c_int32()
    f, v = b"HIPADJ_W_FOR(i, N) du[i] = -u[i];", b"HIPADJ_W_FOR(i, N) dlam[i] = -lam[i];"
    bad = [(0, 4, 0, 0, 0, 0), (5000, 4, 0, 0, 0, 0), (16, 0, 0, 0, 0, 0), (16, 4, 100, 0, 0, 0), (16, 4, 2048, 0, 0, 0), (16, 4, 0, -1, 0, 0),
           (16, 4, 0, 0, 17, 0), (16, 4, 0, 0, 3, 2), (4096, 4, 64, 0, 0, 0), (4096, 4, 0, 8000, 0, 0)]
    for n, npar, T, nw, nacc, a0 in bad:
        assert L.hipadj_wmodel_register(b"bad", n, npar, T, nw, nacc, a0, f, v, C.byref(mid)) == -1, (n, npar, T, nw, nacc, a0)
    assert L.hipadj_wmodel_register(b"bad", 16, 4, 0, 0, 0, 0, None, v, C.byref(mid)) == -1
    assert L.hipadj_wmodel_register(b"decay16", 16, 4, 0, 0, 0, 0, f, v, C.byref(mid)) == 0 and mid.value >= 1000
    n, npar = C.c_int32(), C.c_int32()
    assert L.hipadj_model_sizes(mid.value, (C.c_int32 * 4)(0, 0, 0, 0), C.byref(n), C.byref(npar)) == 0 and (n.value, npar.value) == (16, 4)
    # the small-model entry point points at the wide one
    assert L.hipadj_model_register(b"big", 9, 3, f, None, None, C.byref(mid)) == -1 and b"hipadj_wmodel_register" in L.hipadj_last_error(None)


@pytest.mark.parametrize("make", ["chain", "chain3", "linear", "index"])
def test_emitted_models_compile_for_gfx950(sa, make):
    fun = dict(chain=lambda: sa.WideDeviceFunction.dense_chain("t_chain", (2, 50, 2), input_power=3),
               chain3=lambda: sa.WideDeviceFunction.dense_chain("t_chain3", (3, 16, 24, 3)),
               linear=lambda: sa.WideDeviceFunction.dense_linear("t_lin", 100),          # np = 10 000: the gradient accumulator moves to HBM
               index=lambda: sa.WideDeviceFunction.index_affine("t_idx", 30, 50))[make]()
    from scimlsensitivity_jl_amd import _lib
    _lib.check_model(fun.id)      # forward + Interpolating / Gauss / Backsolve / Quadrature kernels; raises with the compiler log on an error


def test_a_body_that_does_not_compile_is_reported_with_the_log(sa):
    from scimlsensitivity_jl_amd import _lib
    fun = sa.WideDeviceFunction("t_broken", 16, 2, "HIPADJ_W_FOR(i, N) du[i] = undefined_symbol;", "HIPADJ_W_FOR(i, N) dlam[i] = 0.0;")
    with pytest.raises(sa.HipadjError, match="undefined_symbol"):
        _lib.check_model(fun.id)


def test_planner_answers_for_wide_models(sa):
    from scimlsensitivity_jl_amd import _lib
    L = sa.load_library()
    fun = sa.WideDeviceFunction.dense_linear("t_plan", 12)
    ts = np.linspace(0.0, 1.0, 6)

    def check(**kw):
        c = _lib.HipadjConfig()
        c.struct_size = C.sizeof(_lib.HipadjConfig)
        c.model, c.alg, c.stepper, c.ntraj, c.t0, c.t1, c.dt = fun.id, 0, 0, 3, 0.0, 1.0, 0.02
        c.nsave, c.save_times = len(ts), ts.ctypes.data_as(C.POINTER(C.c_double))
        c.p_shared, c.abstol, c.reltol = 1, 1e-6, 1e-3
        for k, v in kw.items():
            setattr(c, k, v)
        return L.hipadj_model_check_config(C.byref(c)), L.hipadj_last_error(None).decode()

    for alg in (0, 1, 2, 3):
        assert check(alg=alg, checkpointing=int(alg == 1))[0] == 0
    # adaptive Tsit5: all four sensealgs (their kernels compile here, no device needed)
    off_grid = np.array([0.0, 0.137, 0.61, 1.0])
    for alg in (0, 1, 2, 3):
        assert check(stepper=1, alg=alg, dt=0.0, nsave=4, save_times=off_grid.ctypes.data_as(C.POINTER(C.c_double)), checkpointing=int(alg == 1))[0] == 0
    assert check(stepper=1, alg=4, dt=0.0)[0] == 0                      # GaussKronrodAdjoint: adaptive (7,15) rule per step, both steppers
    for alg in (0, 2, 4):                                              # checkpointing = true on the adaptive solution (round 5): the interval re-solve inside the sweep compiles
        assert check(stepper=1, alg=alg, dt=0.0, checkpointing=1)[0] == 0, alg
    rc, msg = check(stepper=1, alg=3, dt=0.0, checkpointing=1); assert rc == -1 and "no checkpointing" in msg
    rc, msg = check(stepper=1, alg=2, abstol=0.0); assert rc == -1 and "abstol" in msg
    assert check(alg=4)[0] == 0
    # the built-in continuous costs: WideWithCost<UserW, kind> kernels compile (a sample here; every sensealg x stepper x kind runs on the GPU, test_gpu_wide.py)
    assert check(alg=0, cont_cost=1)[0] == 0 and check(alg=3, cont_cost=2)[0] == 0
    assert check(alg=1, cont_cost=2, stepper=1, dt=0.0, checkpointing=1)[0] == 0 and check(alg=2, cont_cost=1, stepper=1, dt=0.0)[0] == 0
    rc, msg = check(cont_cost=3); assert rc == -1 and "has no cost" in msg       # HIPADJ_CCOST_MODEL without hipadj_wmodel_set_cost
    fun.set_cost(body="HIPADJ_W_FOR(i, N) dlam[i] += u[i]; if (WP) { HIPADJ_W_FOR(j, NP) gp[j] += 0.0; }")
    for alg in (0, 1, 2, 3, 4):
        assert check(alg=alg, cont_cost=3, checkpointing=int(alg == 1))[0] == 0, alg          # WideWithCost<UserW, 3> kernels compile for gfx950
    assert check(alg=0, cont_cost=3, stepper=1, dt=0.0)[0] == 0 and check(alg=1, cont_cost=3, stepper=1, dt=0.0, checkpointing=1)[0] == 0
    fun.set_cost(body=None)
    # checkpointing = true on the fixed step (round 4, k_wide_adjoint_ck): Interpolating / Gauss / GaussKronrod re-solve checkpoint intervals; Quadrature keeps the dense solution
    for alg in (0, 2, 4):
        assert check(alg=alg, checkpointing=1)[0] == 0 and check(alg=alg, checkpointing=1, ckpt_stride=7)[0] == 0
    rc, msg = check(alg=3, checkpointing=1); assert rc == -6 and "QuadratureAdjoint keeps the dense" in msg
    # loss times off the step grid: Interpolating / Gauss without checkpointing (round 4, k_wide_adjoint_og), Backsolve with its default checkpoints or none and Quadrature
    # (round 5, k_wide_backsolve_og / k_wide_quad_adj_og + the GK pass over the reverse step list); the others name what is offered
    off = np.array([0.0, 0.333, 1.0])
    for alg, kw in ((0, {}), (2, {}), (4, {}), (1, dict(checkpointing=1)), (1, {}), (3, {}), (1, dict(checkpointing=1, ckpt_stride=5))):
        assert check(alg=alg, nsave=3, save_times=off.ctypes.data_as(C.POINTER(C.c_double)), **kw)[0] == 0, (alg, kw)
    for alg, kw in ((4, dict(checkpointing=1)), (0, dict(checkpointing=1)), (2, dict(checkpointing=1))):     # the checkpointed sweeps over the reverse step list: lane models
        rc, msg = check(alg=alg, nsave=3, save_times=off.ctypes.data_as(C.POINTER(C.c_double)), **kw); assert rc == -6 and "step grid" in msg, (alg, msg)


def test_dense_chain_emitter_matches_its_golden_text(sa):
    """tests/golden/dense_chain_bodies.json holds the SPMD bodies the Python emitter writes for two chains; julia/test/runtests.jl compares
    `HIPAdj.dense_chain_bodies` (the Julia emitter, never executed in this image) with the same file — one text, two host languages."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dense_chain_bodies.json")) as f:
        gold = json.load(f)
    for key, g in gold.items():
        m = sa.WideDeviceFunction.dense_chain("regold_" + key, g["widths"], input_power=g["input_power"])
        assert m.np == g["np"] and m.source["f"] == g["f"] and m.source["vjp"] == g["vjp"], key


def test_kernel_choice_of_the_wide_family(sa, tmp_path, monkeypatch):
    """Which kernels a wide handle instantiates, read off the name expressions of the dumped translation units (no device needed): the forward solve never
    sees a cost; the reverse kernels of a handle with a built-in cost are instantiated for WideWithCost<UserW, kind>; stepper and sensealg pick the sweep."""
    import glob
    from scimlsensitivity_jl_amd import _lib
    L = sa.load_library()
    fun = sa.WideDeviceFunction.dense_linear("t_choice", 8)
    ts = np.linspace(0.0, 1.0, 6)
    monkeypatch.setenv("HIPADJ_RTC_DUMP", str(tmp_path))

    def names(**kw):
        c = _lib.HipadjConfig()
        c.struct_size = C.sizeof(_lib.HipadjConfig)
        c.model, c.alg, c.stepper, c.ntraj, c.t0, c.t1, c.dt = fun.id, 0, 0, 3, 0.0, 1.0, 0.02
        c.nsave, c.save_times = len(ts), ts.ctypes.data_as(C.POINTER(C.c_double))
        c.p_shared, c.abstol, c.reltol = 1, 1e-6, 1e-3
        for k, v in kw.items():
            setattr(c, k, v)
        before = set(glob.glob(str(tmp_path / "*.hip")))
        assert L.hipadj_model_check_config(C.byref(c)) == 0, L.hipadj_last_error(None)
        new = sorted(set(glob.glob(str(tmp_path / "*.hip"))) - before)
        assert new, "served from the code cache: every configuration here must be a new one"
        return open(new[-1]).read().split("hipadj_name_expressions")[-1]

    e = names(alg=0)
    assert "k_wide_forward<hipadj::UserW>" in e and "k_wide_adjoint<hipadj::UserW, 0>" in e
    e = names(alg=4, cont_cost=1)
    assert "k_wide_forward<hipadj::UserW>" in e and "k_wide_adjoint<hipadj::WideWithCost<hipadj::UserW, 1>, 4>" in e
    e = names(alg=3, stepper=1, dt=0.0)
    assert "k_wide_forward_ts5<hipadj::UserW>" in e and "k_wide_adjoint_ts5<hipadj::UserW, 3, false>" in e and "k_wide_quad_gk<hipadj::UserW, 32, true>" in e
    e = names(alg=1, stepper=1, dt=0.0, checkpointing=1, cont_cost=2)
    assert "k_wide_forward_ts5<hipadj::UserW>" in e and "k_wide_backsolve_ts5<hipadj::WideWithCost<hipadj::UserW, 2>>" in e
    e = names(alg=2, stepper=1, dt=0.0, checkpointing=1)                 # round 5: the interval re-solve inside the adaptive sweep
    assert "k_wide_adjoint_ts5<hipadj::UserW, 2, true>" in e
    off = np.array([0.0, 0.333, 1.0])                                   # round 5: loss times off the step grid, the two sensealgs that were missing
    e = names(alg=3, nsave=3, save_times=off.ctypes.data_as(C.POINTER(C.c_double)))
    assert "k_wide_quad_adj_og<hipadj::UserW>" in e and "k_wide_out_offgrid<hipadj::UserW>" in e and "k_wide_quad_gk<hipadj::UserW, 32, false, true>" in e
    e = names(alg=1, nsave=3, save_times=off.ctypes.data_as(C.POINTER(C.c_double)), checkpointing=1)
    assert "k_wide_backsolve_og<hipadj::UserW>" in e
    e = names(alg=3, cont_cost=2)
    assert "k_wide_quad_adj<hipadj::WideWithCost<hipadj::UserW, 2>>" in e and "k_wide_quad_gk<hipadj::WideWithCost<hipadj::UserW, 2>, 32, false>" in e


def test_wide_models_refuse_mass_matrix_and_lane_style_cost_or_affect_text(sa):
    """set_mass_matrix on a wide model used to overflow the n <= 8 stack arrays of the lane family; the wide kernels
    consult neither a mass matrix nor cost / affect text, so all three are refused at the C ABI with HIPADJ_ERR_UNSUPPORTED."""
    fun = sa.WideDeviceFunction.dense_linear("t_refuse40", 40)
    with pytest.raises(sa.HipadjError) as e:
        fun.set_mass_matrix(np.eye(40) * 2.0)
    assert e.value.status == -6 and "wide" in str(e.value)
    with pytest.raises(ValueError):                 # the lane family's cost forms do not apply: a wide model takes ONE SPMD body (hipadj_wmodel_set_cost)
        fun.set_cost(g="g = u[0] * u[0];")
    with pytest.raises(ValueError):
        fun.set_cost(dgdu="out[0] = u[0];", dgdp="out[0] = 0.0;")
    from scimlsensitivity_jl_amd import _lib
    L = sa.load_library()
    assert L.hipadj_model_set_cost(fun.id, b"out[0] = u[0];", b"out[0] = 0.0;") == -6 and L.hipadj_model_set_cost_function(fun.id, b"g = u[0];") == -6
    assert L.hipadj_wmodel_set_cost(0, b"") == -1                      # not a wide model id
    fun.set_cost(body="HIPADJ_W_FOR(i, N) dlam[i] += u[i];")        # g = |u|^2 / 2: accepted, compiled with the sweeps of a handle that selects it
    fun.set_cost(body=None)
    with pytest.raises(sa.HipadjError) as e:        # the lane family's affect setter (dual-number reverse callback) refuses a wide model and names the right entry point
        _lib.set_model_affect(fun.id, "un[0] += 1.0;")
    assert e.value.status == -6 and "hipadj_wmodel_set_affect" in str(e.value)
    fun.set_affect("un[0] += 1.0;", "")             # round 5: a wide model takes the affect together with its reverse callback as text (hipadj_wmodel_set_affect)
    assert L.hipadj_wmodel_set_affect(0, b"un[0] += 1.0;", b"") == -1          # not a wide model id
    fun.set_affect(None)
    fun.set_mass_matrix(None)       # removing what was never there stays a no-op


def test_a_reregistered_name_drops_the_traced_mass_matrix_of_its_predecessor(sa):
    """engine.py maps du0 = M^{-T} nu(t0) for traced models registered with a mass matrix, looked up by model NAME: the entry must not outlive a re-registration
    of that name without a matrix (the next handle would map du0 with a matrix the device no longer integrates with)."""
    from scimlsensitivity_jl_amd import problems
    ring = lambda u, p, t, ops: p[0:6] * (ops.roll(u, -1) - u)
    sa.WideDeviceFunction.from_callable("mm_stale", ring, 6, 6, mass_matrix=np.eye(6) * 2.0)
    assert "mm_stale" in problems.WIDE_MASS_MATRICES
    sa.WideDeviceFunction.from_callable("mm_stale", ring, 6, 6)
    assert "mm_stale" not in problems.WIDE_MASS_MATRICES


def test_dense_chain_declaration_is_validated_by_the_library(sa):
    """hipadj_wmodel_declare_dense_chain (ABI 109): the structure behind a wide model is declared to the LIBRARY, which selects the kernel family in hipadj_create
    (csrc/hipadj_route.hpp) — here, without a device, the declaration's own checks: widths must reproduce the registered n / np, only wide models take one, NULL withdraws it."""
    from scimlsensitivity_jl_amd import _lib
    fun = sa.WideDeviceFunction.dense_chain("decl_chain_host", (2, 32, 32, 2))        # registers AND declares
    assert fun.np == 2 * 32 + 32 + 32 * 32 + 32 + 2 * 32 + 2
    _lib.declare_dense_chain(fun.id, (2, 32, 32, 2))
    with pytest.raises(sa.HipadjError, match="do not reproduce"):
        _lib.declare_dense_chain(fun.id, (2, 16, 32, 2))
    with pytest.raises(sa.HipadjError, match="do not reproduce"):
        _lib.declare_dense_chain(fun.id, (3, 32, 32, 3))
    _lib.declare_dense_chain(fun.id, None)                                            # withdrawn
    lane = sa.DeviceFunction("decl_lane_host", 2, 1, "du[0] = p[0] * u[1]; du[1] = -u[0];", "out[0] = -lam[1]; out[1] = p[0] * lam[0];", "out[0] = lam[0] * u[1];")
    with pytest.raises(sa.HipadjError, match="hipadj_wmodel_register"):
        _lib.declare_dense_chain(lane.id, (2, 4, 2))
    L = _lib.load()
    import ctypes as C
    w = (C.c_int32 * 3)(2, 50, 2)
    assert L.hipadj_wmodel_declare_dense_chain(fun.id, w, 3, 7, 1) == -1          # an activation other than HIPADJ_ACT_TANH
