#!/usr/bin/env python
"""Golden gradients for the ContinuousCallback cases, from CLOSED FORMS differentiated with forward-mode dual numbers (no ODE solver, no oracle): the role ForwardDiff through
the solve plays in /root/reference/test/Callbacks2/continuous_callbacks.jl:129-139, 340-341.

  ball        `fiip` of :10-14, du = [u2, -p1], u0 = [5, 0], tspan (0, 2.5), p = [9.8, 0.8], saveat 0.5 (:5-7, 22-24), condition u1, affect u2 <- -p2 u2 (:212-217),
              save_positions = (false, false), G = sum(sol) (:184): one bounce at t* = sqrt(2 u1(0) / p1)
  ball_long   the same over (0, 5): three bounces
  ball_mse    condition u1, affect u1 += 3, u2 <- u2^2, G = sum((1 - u)^2) / 2 (:239-250)
  relax       du = p1 - u, u0 = [0], tspan (0, 10), p = [100, 50], condition u - 3/4 p1, affect u += p2, G = u(10) (:314-338; the reference's comment holds the answer,
              [0.9999546000702386, 0.00018159971904994378], :342)
  *_saved     the same with save_positions = (true, true) (the constructor's default; :200-217, 239-250): the loss also takes the state just before and just after each affect
  walls       VectorContinuousCallback (test/Callbacks2/vector_continuous_callbacks.jl:10-25, 79-96): du = [u2, -p1, u4, 0], out = [u1, (u3 - 10) u3], component 1 reflects u2,
              component 2 reflects u4; MSE loss; u0 = [50, 0, 0, 2.01], tspan (0, 10), p = [9.8, 0.9]
  clock       the same model with out = [sin t, cos t] and the affect u <- [0.5, 1, 0, 0] (:100-116): conditions that depend on time only, an affect with a zero Jacobian
  moving      NOT from the reference: the ball on a floor that rises with 0.3 t, condition u1 - 0.3 t, affect u2 <- -p2 (u2 - 0.3) + 0.3 + 0.1 t — condition and affect
              depend on t explicitly

Every state between two events is a polynomial (or an exponential) in t with coefficients that depend on (u0, p) and on the earlier event times; the event times are roots in
closed form.  Writes tests/golden/continuous_callbacks.json.  Needs numpy only."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class D:
    """value + gradient with respect to the K seeds"""
    def __init__(self, v, g):
        self.v = float(v); self.g = np.asarray(g, dtype=np.float64)

    @staticmethod
    def lift(x, K):
        return x if isinstance(x, D) else D(x, np.zeros(K))

    def _o(self, o):
        return D.lift(o, len(self.g))

    def __add__(self, o): o = self._o(o); return D(self.v + o.v, self.g + o.g)
    __radd__ = __add__
    def __neg__(self): return D(-self.v, -self.g)
    def __sub__(self, o): o = self._o(o); return D(self.v - o.v, self.g - o.g)
    def __rsub__(self, o): return self._o(o) - self
    def __mul__(self, o): o = self._o(o); return D(self.v * o.v, self.g * o.v + self.v * o.g)
    __rmul__ = __mul__
    def __truediv__(self, o): o = self._o(o); return D(self.v / o.v, (self.g * o.v - self.v * o.g) / (o.v * o.v))
    def __rtruediv__(self, o): return self._o(o) / self


def dsqrt(x): r = math.sqrt(x.v); return D(r, x.g / (2 * r))
def dexp(x): e = math.exp(x.v); return D(e, e * x.g)
def dlog(x): return D(math.log(x.v), x.g / x.v)


def seeds(vals):
    K = len(vals)
    return [D(v, np.eye(K)[i]) for i, v in enumerate(vals)]


def ballistic(kind, u0, p, T, ts, loss, save_positions=False, terminate=False):
    """x'' = -g between events; kind 1: floor at 0, v <- -e v; kind 2: x += 3, v <- v^2; kind 4: floor 0.3 t, v <- -e (v - 0.3) + 0.3 + 0.1 t"""
    x, v, g, e = seeds([u0[0], u0[1], p[0], p[1]])
    tb = D(0.0, np.zeros(4))                    # start time of the current piece
    G = D(0.0, np.zeros(4)); out = []; events = []; ev_states = []
    k = 0
    while True:
        # next root of x + v s - g s^2 / 2 - floor(tb + s) = 0, s > 0
        w = v - 0.3 if kind == 4 else v         # relative velocity
        x0 = x - 0.3 * tb if kind == 4 else x   # height above the floor
        disc = w * w + 2.0 * g * x0
        s = (w + dsqrt(disc)) / g if disc.v >= 0 else None
        if s is not None and s.v <= 1e-12:      # standing on the floor right after an event: the other root
            s = 2.0 * w / g
        te = tb + s if s is not None and s.v > 1e-12 else None
        while k < len(ts) and (te is None or ts[k] < te.v or te.v >= T):
            if ts[k] > T: break
            d = ts[k] - tb
            xs = x + v * d - 0.5 * g * d * d; vs = v - g * d
            out.append([xs.v, vs.v])
            G = G + loss(xs, vs)
            k += 1
        if te is None or te.v >= T or k >= len(ts):
            break
        xm = x + v * s - 0.5 * g * s * s; vm = v - g * s
        events.append(te.v)
        if kind == 1: x, v = xm, -e * vm
        elif kind == 2: x, v = xm + 3.0, vm * vm
        else: x, v = xm, -e * (vm - 0.3) + 0.3 + 0.1 * te
        if save_positions:                        # save_positions = (true, true): the solution also holds the state just before and just after the affect (:202, 207, 215, 222)
            G = G + loss(xm, vm) + loss(x, v)
            ev_states.append([[xm.v, vm.v], [x.v, v.v]])
        tb = te
        if terminate:                             # terminate!(integrator): the solution ends here (:226-236); the save times after it carry no loss
            if not save_positions:
                ev_states.append([[xm.v, vm.v], [x.v, v.v]])
            break
    return dict(u0=list(u0), p=list(p), tspan=[0.0, T], ts=list(ts), kind=kind, u_at_ts=out, event_times=events, G=G.v, du0=G.g[:2].tolist(), dp=G.g[2:].tolist(),
                **(dict(event_states=ev_states) if save_positions or terminate else {}))


def ball2d(kind, u0, p, T, ts, save_positions=False):
    """du = [u2, -p1, u4, 0] (test/Callbacks2/vector_continuous_callbacks.jl:10-16), MSE loss sum((1 - u)^2) / 2 (:79).  kind 5: out = [u1, (u3 - 10) u3], component 1 reflects
    u2, component 2 reflects u4, both with restitution p2 (:80-96); kind 6: out = [sin t, cos t], either resets u to [0.5, 1, 0, 0] (:100-116)"""
    x, v, y, w, g, e = seeds([u0[0], u0[1], u0[2], u0[3], p[0], p[1]])
    K = 6
    tb = D(0.0, np.zeros(K)); G = D(0.0, np.zeros(K)); out = []; events = []; ev_states = []; ev_idx = []
    loss = lambda a, b, c, d: 0.5 * ((1.0 - a) * (1.0 - a) + (1.0 - b) * (1.0 - b) + (1.0 - c) * (1.0 - c) + (1.0 - d) * (1.0 - d))
    k = 0
    while True:
        cands = []
        if kind == 5:
            disc = v * v + 2.0 * g * x
            if disc.v >= 0:
                sx = (v + dsqrt(disc)) / g
                if sx.v <= 1e-12: sx = 2.0 * v / g
                if sx.v > 1e-12: cands.append((sx, 0))
            if w.v > 0: sy = (10.0 - y) / w
            elif w.v < 0: sy = (0.0 - y) / w
            else: sy = None
            if sy is not None and sy.v <= 1e-12 and w.v != 0:      # standing on a wall right after its event: the other wall
                sy = (10.0 - y) / w if w.v > 0 else (0.0 - y) / w
            if sy is not None and sy.v > 1e-12: cands.append((sy, 1))
        else:
            m = math.floor(tb.v / (math.pi / 2) + 1e-9) + 1
            cands.append((D(m * math.pi / 2, np.zeros(K)) - tb, 0 if m % 2 == 0 else 1))      # sin t = 0 at even multiples of pi / 2, cos t = 0 at odd ones
        s, idx = min(cands, key=lambda c: c[0].v) if cands else (None, None)
        te = tb + s if s is not None else None
        while k < len(ts) and (te is None or ts[k] < te.v or te.v >= T):
            if ts[k] > T: break
            d = ts[k] - tb
            st = (x + v * d - 0.5 * g * d * d, v - g * d, y + w * d, w)
            out.append([q.v if isinstance(q, D) else q for q in st])
            G = G + loss(*st)
            k += 1
        if te is None or te.v >= T or k >= len(ts):
            break
        left = (x + v * s - 0.5 * g * s * s, v - g * s, y + w * s, w)
        events.append(te.v); ev_idx.append(idx)
        if kind == 5:
            x, v, y, w = (left[0], -e * left[1], left[2], left[3]) if idx == 0 else (left[0], left[1], left[2], -e * left[3])
        else:
            x, v, y, w = [D(c, np.zeros(K)) for c in (0.5, 1.0, 0.0, 0.0)]
        if save_positions:
            G = G + loss(*left) + loss(x, v, y, w)
            ev_states.append([[q.v for q in left], [x.v, v.v, y.v, w.v]])
        tb = te
    return dict(u0=list(u0), p=list(p), tspan=[0.0, T], ts=list(ts), kind=kind, u_at_ts=out, event_times=events, event_components=ev_idx, G=G.v, du0=G.g[:4].tolist(), dp=G.g[4:].tolist(),
                **(dict(event_states=ev_states) if save_positions else {}))


def relax():
    u0v, pv, T = [0.0], [100.0, 50.0], 10.0
    u0, a, m = seeds([u0v[0], pv[0], pv[1]])
    # u = a + (u0 - a) e^{-t};  u = 3/4 a  at  e^{-t*} = (a / 4) / (a - u0)
    ts_ = -1.0 * dlog((0.25 * a) / (a - u0))
    up = 0.75 * a + m
    uT = a + (up - a) * dexp(-1.0 * (T - ts_))
    return dict(u0=u0v, p=pv, tspan=[0.0, T], ts=[T], kind=3, event_times=[ts_.v], G=uT.v, du0=uT.g[:1].tolist(), dp=uT.g[1:].tolist())


if __name__ == "__main__":
    ts = np.arange(0.0, 2.5 + 1e-12, 0.5).tolist()
    ssum = lambda x, v: x + v
    mse = lambda x, v: 0.5 * ((1.0 - x) * (1.0 - x) + (1.0 - v) * (1.0 - v))
    out = dict(
        ball=ballistic(1, [5.0, 0.0], [9.8, 0.8], 2.5, ts, ssum),
        ball_long=ballistic(1, [5.0, 0.0], [9.8, 0.8], 5.0, np.arange(0.0, 5.0 + 1e-12, 0.5).tolist(), ssum),
        ball_mse=ballistic(2, [5.0, 0.0], [9.8, 0.8], 2.5, ts, mse),
        relax=relax(),
        moving=ballistic(4, [5.0, 0.0], [9.8, 0.8], 4.0, np.arange(0.0, 4.0 + 1e-12, 0.5).tolist(), ssum),
        # save_positions = (true, true), the constructor's default and the setting of most of the reference's testsets: g also sums the saved event states
        ball_saved=ballistic(1, [5.0, 0.0], [9.8, 0.8], 2.5, ts, ssum, save_positions=True),                 # "= callback with parameter dependence and save", :212-217
        ball_long_saved=ballistic(1, [5.0, 0.0], [9.8, 0.8], 5.0, np.arange(0.0, 5.0 + 1e-12, 0.5).tolist(), ssum, save_positions=True),
        ball_mse_saved=ballistic(2, [5.0, 0.0], [9.8, 0.8], 2.5, ts, mse, save_positions=True),            # "callback with non-linear affect", MSE loss, :239-250
        moving_saved=ballistic(4, [5.0, 0.0], [9.8, 0.8], 4.0, np.arange(0.0, 4.0 + 1e-12, 0.5).tolist(), ssum, save_positions=True),
        # terminate!: "= callback with terminate" (:226-236) — the loss takes the save times before the bounce; with save_positions also the two states at it (the second one
        # is the solution's last point)
        ball_terminate=ballistic(1, [5.0, 0.0], [9.8, 0.8], 2.5, ts, ssum, terminate=True),
        ball_terminate_saved=ballistic(1, [5.0, 0.0], [9.8, 0.8], 2.5, ts, ssum, save_positions=True, terminate=True),
        # VectorContinuousCallback, test/Callbacks2/vector_continuous_callbacks.jl: u0 = [50, 0, 0, 2.01], tspan (0, 10), p = [9.8, 0.9] (:23-25), saveat 0.5
        walls=ball2d(5, [50.0, 0.0, 0.0, 2.01], [9.8, 0.9], 10.0, np.arange(0.0, 10.0 + 1e-12, 0.5).tolist()),
        walls_saved=ball2d(5, [50.0, 0.0, 0.0, 2.01], [9.8, 0.9], 10.0, np.arange(0.0, 10.0 + 1e-12, 0.5).tolist(), save_positions=True),
        clock=ball2d(6, [50.0, 0.0, 0.0, 2.01], [9.8, 0.9], 10.0, np.arange(0.0, 10.0 + 1e-12, 0.5).tolist()),
        clock_saved=ball2d(6, [50.0, 0.0, 0.0, 2.01], [9.8, 0.9], 10.0, np.arange(0.0, 10.0 + 1e-12, 0.5).tolist(), save_positions=True),
        source="tests/golden/make_continuous_callbacks.py: closed forms differentiated with dual numbers")
    json.dump(out, open(os.path.join(HERE, "continuous_callbacks.json"), "w"), indent=1)
    print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != "u_at_ts"} if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
