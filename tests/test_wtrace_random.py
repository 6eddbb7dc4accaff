"""Randomised check of the array tracer (scimlsensitivity.jl_amd/wtrace.py: the automatic joint VJP of wide models, the counterpart of the reference's AD-generated
vecjacobian!, src/derivative_wrappers.jl:649-1145): random right-hand sides composed from every traced operation — elementwise chains with scalar and array operands,
gathers, rolls, sums, parameter and constant matrix products — are traced, both SPMD bodies are emitted, compiled for the host and run under T cooperating threads in both
schedules (tests/spmd_emu.py); f must equal the SAME Python function evaluated eagerly with numpy, the joint VJP must equal central differences of lam . f.  No GPU."""
import numpy as np
import pytest

import spmd_emu as SE


class NpOps:
    """The eager numpy twin of wtrace.Ops: the same callable evaluates under either."""
    sin, cos, exp, tanh, sqrt, log, sinh, cosh, atan = np.sin, np.cos, np.exp, np.tanh, np.sqrt, np.log, np.sinh, np.cosh, np.arctan
    const = staticmethod(lambda v: np.asarray(v, dtype=np.float64))
    gather = staticmethod(lambda x, idx: x[np.asarray(idx)])
    roll = staticmethod(lambda x, s: np.roll(x, int(s)))
    sum = staticmethod(lambda x: np.sum(x))
    matvec_const = staticmethod(lambda A, x: np.asarray(A) @ x)
    matvec = staticmethod(lambda ps, m, x: ps.reshape((int(m), len(x)), order="F") @ x)


def random_rhs(seed, n):
    """(fn, np): a random right-hand side of n states.  Parameter layout: three array slices of n entries, two scalars, one n x n matrix."""
    rng = np.random.default_rng(seed)
    npar = 3 * n + 2 + n * n
    perm = rng.permutation(n); dup = rng.integers(0, n, n)            # a permutation gather and one with repeated / missing sources
    cvec = rng.uniform(0.5, 1.5, n); A = rng.standard_normal((n, n)) / np.sqrt(n)
    shifts = [int(s) for s in rng.integers(-3, 4, 4)]

    def build(depth):
        kind = rng.integers(0, 6 if depth == 0 else 14)
        if depth == 0 or kind < 6:
            k = int(kind) % 6
            if k == 0: return lambda u, p, t, ops: u
            if k == 1: s = shifts[rng.integers(0, 4)]; return lambda u, p, t, ops: ops.roll(u, s)
            if k == 2: idx = perm if rng.integers(0, 2) else dup; return lambda u, p, t, ops: ops.gather(u, idx)
            if k == 3: a = int(rng.integers(0, 3)) * n; return lambda u, p, t, ops: p[a:a + n]
            if k == 4: return lambda u, p, t, ops: ops.const(cvec)
            return lambda u, p, t, ops: u * u
        x = build(depth - 1)
        if kind == 6: f = ("sin", "cos", "tanh", "atan")[rng.integers(0, 4)]; return lambda u, p, t, ops: getattr(ops, f)(x(u, p, t, ops))
        if kind == 7: y = build(depth - 1); return lambda u, p, t, ops: x(u, p, t, ops) + y(u, p, t, ops)
        if kind == 8: y = build(depth - 1); return lambda u, p, t, ops: x(u, p, t, ops) * y(u, p, t, ops)
        if kind == 9: y = build(depth - 1); return lambda u, p, t, ops: x(u, p, t, ops) / (1.0 + y(u, p, t, ops) ** 2)
        if kind == 10:
            which = rng.integers(0, 4); k = 3 * n + int(rng.integers(0, 2)); y = build(depth - 1)
            if which == 0: return lambda u, p, t, ops: p[k] * x(u, p, t, ops)
            if which == 1: return lambda u, p, t, ops: (0.3 + t) * x(u, p, t, ops)
            if which == 2: return lambda u, p, t, ops: ops.sum(y(u, p, t, ops)) * x(u, p, t, ops) - 0.25
            return lambda u, p, t, ops: x(u, p, t, ops) - 1.7 * p[k]
        if kind == 11: return lambda u, p, t, ops: ops.exp(-(x(u, p, t, ops) ** 2))
        if kind == 12: return lambda u, p, t, ops: ops.matvec(p[3 * n + 2:3 * n + 2 + n * n], n, x(u, p, t, ops))
        return lambda u, p, t, ops: ops.matvec_const(A, x(u, p, t, ops))

    a, b = build(3), build(2)
    return (lambda u, p, t, ops: a(u, p, t, ops) + 0.5 * b(u, p, t, ops) - 0.1 * u), npar


@pytest.mark.parametrize("n", [7, 70])
@pytest.mark.parametrize("seed", range(12))
def test_random_traced_models_against_numpy_and_finite_differences(seed, n):
    from scimlsensitivity_jl_amd import wtrace
    fn, npar = random_rhs(1000 * n + seed, n)
    fb, vb, nw, nacc, a0 = wtrace.bodies(fn, n, npar)
    m = SE.SpmdModel(fb, vb, n, npar, lds_doubles=nw, nacc=nacc, acc_first=a0)
    rng = np.random.default_rng(seed)
    u, p, lam, t = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, npar), rng.standard_normal(n), 0.37
    ev = lambda uu, pp: np.asarray(fn(uu, pp, t, NpOps), dtype=np.float64) * np.ones(n)
    want = ev(u, p)
    scale = max(1.0, np.max(np.abs(want)))
    res = {}
    for threads, rev in ((1, False), (64, False), (64, True)):
        du = m.f(u, p, t, threads=threads, reverse=rev)
        assert np.max(np.abs(du - want)) <= 1e-12 * scale, (seed, threads, rev)
        res[(threads, rev)] = m.vjp(lam, u, p, t, w=0.7, threads=threads, reverse=rev)
    dlam, gp = res[(64, False)]
    for key in ((1, False), (64, True)):                      # the schedules agree (sums in another order: roundoff)
        assert np.max(np.abs(res[key][0] - dlam)) <= 1e-11 * max(1.0, np.max(np.abs(dlam))) and np.max(np.abs(res[key][1] - gp)) <= 1e-11 * max(1.0, np.max(np.abs(gp)))
    eps = 1e-6
    for x, g, sc in ((u, dlam, 1.0), (p, gp, 0.7)):
        for k in rng.choice(len(x), size=min(len(x), 10), replace=False):
            xp, xm = x.copy(), x.copy(); xp[k] += eps; xm[k] -= eps
            fd = (lam @ (ev(xp, p) if x is u else ev(u, xp)) - lam @ (ev(xm, p) if x is u else ev(u, xm))) / (2 * eps)
            assert abs(g[k] - sc * fd) <= 2e-6 * max(1.0, abs(fd), np.max(np.abs(g))), (seed, "u" if x is u else "p", int(k), g[k], sc * fd)


@pytest.mark.parametrize("seed", range(10))
def test_random_traced_costs_against_finite_differences(seed):
    """A continuous cost g(u, p, t) traced to the SPMD body hipadj_wmodel_set_cost takes (wtrace.cost_body: dg/du added into dlam, w dg/dp into the gradient row —
    accumulate_cost!, src/derivative_wrappers.jl:1411-1442): random scalar expressions, against central differences of the same function evaluated with numpy."""
    from scimlsensitivity_jl_amd import wtrace
    n = 9
    fn, npar = random_rhs(7000 + seed, n)
    rng = np.random.default_rng(seed)
    wts = rng.uniform(0.5, 1.5, n)
    k1, k2 = 3 * n, 3 * n + 1

    def g(u, p, t, ops):           # a scalar: weighted sum of squares of a random array expression + scalar parameters
        r = fn(u, p, t, ops)
        return ops.sum(ops.const(wts) * r * r) * p[k1] + 0.5 * p[k2] * p[k2] + ops.sum(u) * t
    body, nw, nacc, a0 = wtrace.cost_body(g, n, npar)
    m = SE.SpmdModel("", "", n, npar, lds_doubles=nw, nacc=nacc, acc_first=a0, cost_body=body)
    u, p, t = rng.uniform(0.2, 1.2, n), rng.uniform(-0.8, 0.9, npar), 0.37
    ev = lambda uu, pp: float(g(uu, pp, t, NpOps))
    base = rng.standard_normal(n)                                   # the body ADDS to dlam
    res = [m.cost(u, p, t, w=0.7, threads=th, reverse=rev, dlam0=base) for th, rev in ((1, False), (64, False), (64, True))]
    for a in res[1:]:
        assert np.max(np.abs(a[0] - res[0][0])) <= 1e-11 * max(1.0, np.max(np.abs(res[0][0]))) and np.max(np.abs(a[1] - res[0][1])) <= 1e-11 * max(1.0, np.max(np.abs(res[0][1])))
    dlam, gp = res[1]
    eps = 1e-6
    for x, got, sc in ((u, dlam - base, 1.0), (p, gp, 0.7)):
        for k in rng.choice(len(x), size=min(len(x), 12), replace=False):
            xp, xm = x.copy(), x.copy(); xp[k] += eps; xm[k] -= eps
            fd = ((ev(xp, p) if x is u else ev(u, xp)) - (ev(xm, p) if x is u else ev(u, xm))) / (2 * eps)
            assert abs(got[k] - sc * fd) <= 2e-6 * max(1.0, abs(fd), np.max(np.abs(got))), (seed, "u" if x is u else "p", int(k), got[k], sc * fd)
    d2, g2 = m.cost(u, p, t, w=0.7, wp=False, threads=64, dlam0=base)     # WP = false: the state part only
    assert np.max(np.abs(d2 - dlam)) <= 1e-12 * max(1.0, np.max(np.abs(dlam))) and not g2.any()
