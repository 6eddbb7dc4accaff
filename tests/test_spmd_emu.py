"""The SPMD bodies of wide models under T COOPERATING host threads (tests/spmd_emu.py): what the one-thread harness cannot see — barriers between producer and consumer phases,
ownership of every output / gradient entry, collectives reached by every thread — for the three emitters, the traced models and a deliberately broken body."""
import numpy as np
import pytest

import oracle as O
import spmd_emu as SE


def _check(m, omodel, dims, n, npar, threads, reverse, seed=3):
    rng = np.random.default_rng(seed)
    u, p, lam, t = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, npar), rng.standard_normal(n), 0.37
    du = m.f(u, p, t, threads=threads, reverse=reverse)
    dlam, gp = m.vjp(lam, u, p, t, w=0.7, threads=threads, reverse=reverse)
    rdu = O.model_f(omodel, u, p, t, dims)
    rdl, rgp = O.model_vjp(omodel, lam, u, p, t, dims)
    assert np.max(np.abs(du - rdu)) <= 1e-13 * max(1.0, np.max(np.abs(rdu)))
    assert np.max(np.abs(dlam - rdl)) <= 1e-13 * max(1.0, np.max(np.abs(rdl)))
    assert np.max(np.abs(gp - 0.7 * rgp)) <= 1e-13 * max(1.0, np.max(np.abs(rgp)))
    dl2, gp2 = m.vjp(lam, u, p, t, w=0.7, wp=False, threads=threads, reverse=reverse)
    assert np.max(np.abs(dl2 - rdl)) <= 1e-13 * max(1.0, np.max(np.abs(rdl))) and not gp2.any()


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("threads", [64, 128])
@pytest.mark.parametrize("which", ["chain_2_50_2", "chain_3_8_3", "chain_3_16_24_3", "linear_12", "idxaff_6x10"])
def test_emitter_bodies_under_cooperating_threads(sa, which, threads, reverse):
    if which.startswith("chain"):
        w = tuple(int(x) for x in which.split("_")[1:])
        fun = sa.WideDeviceFunction.dense_chain("se_" + which, w, input_power=3)
        if len(w) == 3:
            omodel, dims = "MLP1", (w[0], w[1], 0, 0)
        else:
            pytest.skip("the oracle's chain model has one hidden layer; the deep chain is compared with its own one-thread run below")
    elif which == "linear_12":
        fun, omodel, dims = sa.WideDeviceFunction.dense_linear("se_lin12", 12), "DENSELIN", (12, 0, 0, 0)
    else:
        fun, omodel, dims = sa.WideDeviceFunction.index_affine("se_idx", 6, 10), "IDXAFF", (6, 10, 0, 0)
    src = fun.source
    m = SE.SpmdModel(src["f"], src["vjp"], fun.n, fun.np, lds_doubles=src.get("lds_doubles", 4096), nacc=src.get("nacc", 0), acc_first=src.get("acc_first", 0))
    _check(m, omodel, dims, fun.n, fun.np, threads, reverse)


@pytest.mark.parametrize("reverse", [False, True])
def test_deep_chain_equals_its_one_thread_run(sa, reverse):
    fun = sa.WideDeviceFunction.dense_chain("se_deep", (3, 16, 24, 3), input_power=1)
    src = fun.source
    m = SE.SpmdModel(src["f"], src["vjp"], fun.n, fun.np, lds_doubles=4096)
    rng = np.random.default_rng(5)
    u, p, lam = rng.uniform(0.2, 1.2, 3), rng.uniform(-0.8, 0.9, fun.np), rng.standard_normal(3)
    one = (m.f(u, p, 0.1, threads=1), ) + m.vjp(lam, u, p, 0.1, w=0.7, threads=1)
    many = (m.f(u, p, 0.1, threads=64, reverse=reverse), ) + m.vjp(lam, u, p, 0.1, w=0.7, threads=64, reverse=reverse)
    for a, b in zip(one, many):
        assert np.max(np.abs(a - b)) <= 1e-13 * max(1.0, np.max(np.abs(a)))


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("case", ["ring64", "denselin12", "mlp1_2_50", "rxn40"])
def test_traced_bodies_under_cooperating_threads(case, reverse):
    from scimlsensitivity_jl_amd import wtrace
    from test_wtrace import CASES, host_model
    fn, n, npar, omodel, dims = CASES[case]
    fb, vb, nw, nacc, a0 = wtrace.bodies(fn, n, npar)
    m = SE.SpmdModel(fb, vb, n, npar, lds_doubles=nw, nacc=nacc, acc_first=a0)
    f1, vjp1 = host_model(fn, n, npar)                 # the one-thread harness
    rng = np.random.default_rng(3)
    u, p, lam, t = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, npar), rng.standard_normal(n), 0.37
    du = m.f(u, p, t, threads=64, reverse=reverse)
    dlam, gp = m.vjp(lam, u, p, t, w=0.7, threads=64, reverse=reverse)
    rdl, rgp = vjp1(lam, u, p, t, w=0.7)
    assert np.max(np.abs(du - f1(u, p, t))) <= 1e-13 * max(1.0, np.max(np.abs(du)))
    assert np.max(np.abs(dlam - rdl)) <= 1e-13 * max(1.0, np.max(np.abs(rdl))) and np.max(np.abs(gp - rgp)) <= 1e-13 * max(1.0, np.max(np.abs(rgp)))


def test_a_missing_barrier_and_a_divergent_collective_are_caught():
    """The emulation's reason to exist, on two deliberately broken bodies: (i) a consumer phase that reads `ws` without the wg_sync() after the producer phase gives the right
    numbers under one thread and wrong ones under cooperating threads in at least one order; (ii) a wg_sum reached by some threads only is reported (the device would hang)."""
    good = "HIPADJ_W_FOR(i, N) ws[i] = 2.0 * u[i];\nwg_sync();\nHIPADJ_W_FOR(i, N) du[i] = ws[(i + 1) % N] + ws[(i + N - 1) % N];"
    bad = good.replace("wg_sync();\n", "")
    vj = "HIPADJ_W_FOR(i, N) dlam[i] = 0.0;"
    u, p = np.arange(1.0, 9.0), np.zeros(1)
    want = 2.0 * (np.roll(u, -1) + np.roll(u, 1))
    mg, mb = SE.SpmdModel(good, vj, 8, 1, lds_doubles=8), SE.SpmdModel(bad, vj, 8, 1, lds_doubles=8)
    for rev in (False, True):
        assert np.array_equal(mg.f(u, p, 0.0, threads=8, reverse=rev), want)
    assert np.array_equal(mb.f(u, p, 0.0, threads=1), want)                                       # one thread: the broken body looks fine
    assert any(not np.array_equal(mb.f(u, p, 0.0, threads=8, reverse=rev), want) for rev in (False, True))
    div = "double s = 0.0; if (tid < 4) s = wg_sum(u[tid]); HIPADJ_W_FOR(i, N) du[i] = s;"
    md = SE.SpmdModel(div, vj, 8, 1, lds_doubles=8)
    with pytest.raises(RuntimeError, match="same collective"):
        md.f(u, p, 0.0, threads=8)
