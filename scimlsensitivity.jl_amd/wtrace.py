"""Automatic joint VJP for WIDE runtime models (workgroup-per-trajectory family, csrc/hipadj_wide.hpp): trace a right-hand side written with ARRAY operations,
differentiate the recorded graph in reverse mode and emit the two SPMD bodies `hipadj_wmodel_register` takes —

    f   (du, u, p, t, ws, tid)                       du = f(u, p, t)
    vjp<WP>(dlam, gp, acc, w, lam, u, p, t, ws, tid) dlam = (df/du)' lam; if WP: gradient += w (df/dp)' lam

i.e. what the reference obtains from its AD backends for `vecjacobian!(dlam, y, lam, p, t, S; dgrad)` (src/derivative_wrappers.jl:256-267, 649-764, 982-1145) and
what a wide model so far needed hand-written (problems.py: dense_chain / dense_linear / index_affine).  trace.py does the same for the lane family (n <= 8) with
scalar graphs; beyond that the code must stay SPMD — a loop over components with the same instructions for every thread — so the traced values here are whole
ARRAYS and every operation knows how its adjoint is distributed:

    elementwise (+ - * / neg, ** numeric, sin cos exp log sqrt tanh ...)   same index on both sides; a scalar operand (p[k], t, a sum) is broadcast and its
                                                                            adjoint is a sum over the workgroup (`acc` for parameters when their indices are contiguous)
    u / p[a:b] / constant arrays                                            leaves; p[a:b][i] is OWNED by one thread: its gradient entry is written without a reduction
    gather(x, idx), roll(x, s)                                              static index maps; the adjoint is the gather through the inverse map (no atomics, fixed order)
    sum(x)                                                                  workgroup sum (wg_sum); adjoint = broadcast
    matvec(p[a:b] as m x k column-major, x)                                 dense parameter matrix: adjoints W' g and the outer product into the gradient
    matvec_const(A, x)                                                      constant matrix (e.g. the inverse of a mass matrix): adjoint A' g

    def ring(u, p, t, ops):                     # du_i = p_i (u_{i+1} - u_i) + p_n sin(u_{i-1})
        n = u.length
        return p[0:n] * (ops.roll(u, -1) - u) + p[n] * ops.sin(ops.roll(u, 1))
    fun = sa.WideDeviceFunction.from_callable("ring64", ring, n=64, np=65)

Arrays that a gather, a matvec or a second stage needs as a whole are materialised in the model's LDS scratch `ws`; chains of elementwise operations are fused
into the loop that consumes them and recomputed in the reverse pass (no tape).  Nothing of `f` is ever evaluated on the host for a solve."""
import math

import numpy as np

_UNARY = {
    # name: (C expression of the value from a, C expression of d/da from (a, z))
    "sin": ("sin({a})", "cos({a})"), "cos": ("cos({a})", "(-sin({a}))"), "exp": ("exp({a})", "{z}"), "log": ("log({a})", "(1.0 / {a})"),
    "sqrt": ("sqrt({a})", "(0.5 / {z})"), "tanh": ("tanh({a})", "(1.0 - {z} * {z})"), "neg": ("(-{a})", "(-1.0)"),
    "sinh": ("sinh({a})", "cosh({a})"), "cosh": ("cosh({a})", "sinh({a})"), "atan": ("atan({a})", "(1.0 / (1.0 + {a} * {a}))"),
}


def _c(x):
    s = repr(float(x))
    if s in ("inf", "-inf", "nan"):
        raise ValueError("non-finite constant in a traced model")
    return s if ("." in s or "e" in s) else s + ".0"


class Arr:
    """One traced value: an array of `length` doubles (length 0 = a scalar, broadcast in elementwise operations)."""
    _count = [0]

    def __init__(self, kind, length, args=(), **data):
        self.kind, self.length, self.args, self.data = kind, int(length), tuple(args), data
        Arr._count[0] += 1
        self.id = Arr._count[0]

    # ---- elementwise arithmetic
    def _bin(self, o, op, swap=False):
        o = o if isinstance(o, Arr) else Arr("cscalar", 0, value=float(o))
        a, b = (o, self) if swap else (self, o)
        if a.length and b.length and a.length != b.length:
            raise ValueError(f"elementwise {op} of arrays of different lengths ({a.length}, {b.length})")
        return Arr("ew", max(a.length, b.length), (a, b), op=op)

    def __add__(self, o): return self._bin(o, "+")
    def __radd__(self, o): return self._bin(o, "+", True)
    def __sub__(self, o): return self._bin(o, "-")
    def __rsub__(self, o): return self._bin(o, "-", True)
    def __mul__(self, o): return self._bin(o, "*")
    def __rmul__(self, o): return self._bin(o, "*", True)
    def __truediv__(self, o): return self._bin(o, "/")
    def __rtruediv__(self, o): return self._bin(o, "/", True)
    def __neg__(self): return Arr("ew", self.length, (self,), op="neg")
    def __pos__(self): return self

    def __pow__(self, e):
        if isinstance(e, Arr):
            raise TypeError("only numeric exponents are traced (x ** 3)")
        return Arr("ew", self.length, (self,), op="pow", e=float(e))

    def __bool__(self):
        raise TypeError("a traced array has no truth value: branches on the state are not traceable")

    def __getitem__(self, k):
        if self.kind != "p":
            raise TypeError("only the parameter vector is sliced (p[a:b], p[k]); use ops.gather for the state")
        if isinstance(k, slice):
            a, b, st = k.indices(self.length)
            if st != 1 or b <= a:
                raise ValueError("parameter slices are contiguous: p[a:b]")
            return Arr("pslice", b - a, off=a)
        k = int(k)
        if not 0 <= k < self.length:
            raise IndexError(k)
        return Arr("pscalar", 0, k=k)


class Ops:
    """The function namespace handed to the traced callable."""

    def __getattr__(self, name):
        if name in _UNARY:
            return lambda x: Arr("ew", x.length, (x,), op=name)
        raise AttributeError(name)

    @staticmethod
    def const(values):
        """A constant coefficient array (e.g. grid coordinates, stoichiometric weights)."""
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        return Arr("const", len(v), values=v)

    @staticmethod
    def gather(x, idx):
        """y[i] = x[idx[i]] with a static integer index array."""
        idx = np.ascontiguousarray(idx, dtype=np.int64).ravel()
        if x.length == 0 or idx.min() < 0 or idx.max() >= x.length:
            raise ValueError("gather: index out of range")
        return Arr("gather", len(idx), (x,), idx=idx)

    @staticmethod
    def roll(x, shift):
        """numpy.roll: y[i] = x[(i - shift) mod L]."""
        L = x.length
        return Ops.gather(x, (np.arange(L) - int(shift)) % L)

    @staticmethod
    def sum(x):
        return Arr("sum", 0, (x,))

    @staticmethod
    def matvec_const(A, x):
        """A @ x with a CONSTANT matrix A (m x k, numpy): e.g. the inverse of a mass matrix."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        if A.ndim != 2 or x.length != A.shape[1]:
            raise ValueError("matvec_const: a constant m x k matrix and an array of k entries")
        return Arr("cmatvec", A.shape[0], (x,), A=A)

    @staticmethod
    def matvec(pslice, m, x):
        """reshape(p[a:b], (m, k)) @ x with the matrix in column-major order (Lux / Julia `reshape`): W[i, j] = p[a + i + j m]."""
        if pslice.kind != "pslice" or x.length == 0 or pslice.length != int(m) * x.length:
            raise ValueError("matvec: a parameter slice of m * len(x) entries and an array")
        return Arr("matvec", int(m), (x,), off=pslice.data["off"], m=int(m), k=x.length)


_LEAF = ("u", "pslice", "const")
_SCALAR_LEAF = ("pscalar", "t", "cscalar")


def _topo(root):
    order, seen = [], set()
    stack = [(root, False)]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n); continue
        if n.id in seen:
            continue
        seen.add(n.id)
        stack.append((n, True))
        for a in n.args:
            stack.append((a, False))
    return order


class _Gen:
    def __init__(self, root, n, npar, scalar_root=False):
        """scalar_root: `root` is a continuous cost g(u, p, t) (a scalar); vjp_body() then emits the cost body of hipadj_wmodel_set_cost — g_u ADDED into dlam, w g_p into gp / acc"""
        if not scalar_root and root.length != n:
            raise ValueError(f"the traced right-hand side has {root.length} components, the model {n}")
        if scalar_root and root.length != 0:
            raise ValueError("a traced cost must be a scalar (ops.sum(...), or arithmetic on scalars)")
        self.root, self.n, self.np, self.scalar_root = root, n, npar, scalar_root
        self.order = _topo(root)
        self.users = {x.id: [] for x in self.order}
        for x in self.order:
            for a in x.args:
                self.users[a.id].append(x)
        # ---- which values live where
        self.mat = {}                 # node id -> ws offset of the materialised forward value (arrays), or None for du itself
        self.tables, self.table_of = [], {}
        self.ws = 0
        for x in self.order:          # sources of gathers / matvecs must be addressable
            if x.kind in ("gather", "matvec", "cmatvec"):
                s = x.args[0]
                if s.kind not in _LEAF and s.id not in self.mat:
                    self.mat[s.id] = self._alloc(s.length)
        for x in self.order:
            if x.kind in ("matvec", "cmatvec") and x is not root:
                self.mat[x.id] = self._alloc(x.length)
        if not scalar_root:
            self.mat[root.id] = None
        # scalar-valued nodes: sums and elementwise operations on scalars only
        self.scalars = [x for x in self.order if x.length == 0 and x.kind in ("sum", "ew")]
        # reverse pass: adjoint buffers of the materialised arrays (not of du: that is lam), scatter temporaries of the gathers
        self.adj = {i: self._alloc(self._node(i).length) for i in self.mat if i != root.id or scalar_root}
        self.tmp = {x.id: self._alloc(x.length) for x in self.order if x.kind == "gather" and x.args[0].kind != "const"}
        sc = sorted({x.data["k"] for x in self.order if x.kind == "pscalar"})
        self.acc_first, self.nacc = (sc[0], len(sc)) if sc and sc[-1] - sc[0] + 1 == len(sc) and len(sc) <= 16 and not scalar_root else (0, 0)   # (a cost body never uses `acc`: the range belongs to the model)
        self.pscalars = sc

    def _node(self, i):
        return next(x for x in self.order if x.id == i)

    def _alloc(self, L):
        o = self.ws
        self.ws += L
        return o

    # ---- constant tables (function-scope static const arrays)
    def _table(self, ctype, values, tag):
        key = (ctype, tag, values.tobytes())
        if key not in self.table_of:
            name = f"{tag}{len(self.tables)}"
            body = ", ".join(str(int(v)) for v in values) if ctype == "int" else ", ".join(_c(v) for v in values)
            self.tables.append(f"static const {ctype} {name}[{len(values)}] = {{{body}}};")
            self.table_of[key] = name
        return self.table_of[key]

    def _tables_text(self):
        return "\n".join(self.tables)

    # ---- addressing
    def _base(self, x):
        """C expression of an array that can be indexed by any thread: leaves and materialised values."""
        if x.kind == "u": return "u"
        if x.kind == "pslice": return f"(p + {x.data['off']})"
        if x.kind == "const": return self._table("double", x.data["values"], "ctab")
        if x.id in self.mat:
            return "du" if self.mat[x.id] is None else f"(ws + {self.mat[x.id]})"
        raise AssertionError("not addressable")

    def _scalar_name(self, x):
        if x.kind == "pscalar": return f"p[{x.data['k']}]"
        if x.kind == "t": return "t"
        if x.kind == "cscalar": return _c(x.data["value"])
        return f"s{x.id}"

    def _idx(self, g, i):
        idx, L = g.data["idx"], g.args[0].length
        sh = int(idx[0])
        if len(idx) == L and np.array_equal(idx, (np.arange(L) + sh) % L):          # a rotation: no table
            return f"(({i}) + {sh} < {L} ? ({i}) + {sh} : ({i}) + {sh} - {L})" if sh else f"({i})"
        return f"{self._table('int', idx, 'itab')}[{i}]"

    # ---- forward expression of an array-valued node at index i: SSA locals for the fused elementwise chain
    def _value(self, x, i, lines, cache):
        if x.length == 0:
            return self._scalar_name(x)
        if x.id in cache:
            return cache[x.id]
        if x.kind in _LEAF or (x.id in self.mat and not cache.get("__root__") == x.id):
            r = f"{self._base(x)}[{i}]"
        elif x.kind == "gather":
            r = f"{self._base(x.args[0])}[{self._idx(x, i)}]"
        elif x.kind == "ew":
            a = [self._value(y, i, lines, cache) for y in x.args]
            op = x.data["op"]
            if op in ("+", "-", "*", "/"):
                e = f"{a[0]} {op} {a[1]}"
            elif op == "pow":
                ex = x.data["e"]
                e = " * ".join([a[0]] * int(ex)) if ex == int(ex) and 1 <= ex <= 4 else f"pow({a[0]}, {_c(ex)})"
            else:
                e = _UNARY[op][0].format(a=a[0])
            r = f"v{x.id}"
            lines.append(f"const double {r} = {e};")
        else:
            raise AssertionError(x.kind)
        cache[x.id] = r
        return r

    def _partials(self, x, a, z):
        op = x.data["op"]
        if op == "+": return ["1.0", "1.0"]
        if op == "-": return ["1.0", "(-1.0)"]
        if op == "*": return [a[1], a[0]]
        if op == "/": return [f"(1.0 / {a[1]})", f"(-{z} / {a[1]})"]
        if op == "pow":
            ex = x.data["e"]
            if ex == int(ex) and 1 <= ex <= 4:
                return [f"({_c(ex)} * " + " * ".join([a[0]] * (int(ex) - 1)) + ")"] if ex > 1 else ["1.0"]
            return [f"({_c(ex)} * pow({a[0]}, {_c(ex - 1.0)}))"]
        return [_UNARY[op][1].format(a=a[0], z=z)]

    # ---- units of the forward pass, in dependency order
    def _units(self):
        u = []
        for x in self.order:
            if x.length == 0 and x.kind in ("sum", "ew"):
                u.append(x)
            elif x.id in self.mat:
                u.append(x)
        return u

    def _forward_unit(self, x, out):
        if x.length == 0 and x.kind == "ew":          # scalar arithmetic: every thread computes it
            lines, cache = [], {}
            a = [self._scalar_name(y) for y in x.args]
            op = x.data["op"]
            e = f"{a[0]} {op} {a[1]}" if op in ("+", "-", "*", "/") else (f"pow({a[0]}, {_c(x.data['e'])})" if op == "pow" else _UNARY[op][0].format(a=a[0]))
            out.append(f"const double s{x.id} = {e};")
            return
        if x.kind == "sum":
            lines, cache = [], {}
            v = self._value(x.args[0], "i", lines, cache)
            out.append(f"double s{x.id}; {{ double part = 0.0; HIPADJ_W_FOR(i, {x.args[0].length}) {{ {' '.join(lines)} part += {v}; }} s{x.id} = wg_sum(part); }}")
            return
        dst = self._base(x)
        if x.kind == "cmatvec":
            A = x.data["A"]; m, k = A.shape
            tab = self._table("double", A.ravel(), "mtab")
            src = self._base(x.args[0])
            out.append(f"HIPADJ_W_FOR(i, {m}) {{ double s = 0.0; for (int j = 0; j < {k}; ++j) s += {tab}[i * {k} + j] * {src}[j]; {dst}[i] = s; }}")
        elif x.kind == "matvec":
            m, k, off = x.data["m"], x.data["k"], x.data["off"]
            src = self._base(x.args[0])
            out.append(f"HIPADJ_W_FOR(i, {m}) {{ double s = 0.0; for (int j = 0; j < {k}; ++j) s += p[{off} + i + j * {m}] * {src}[j]; {dst}[i] = s; }}")
        else:
            lines, cache = [], {"__root__": x.id}
            v = self._value(x, "i", lines, cache)
            out.append(f"HIPADJ_W_FOR(i, {x.length}) {{ {' '.join(lines)} {dst}[i] = {v}; }}")
        out.append("wg_sync();")

    def forward_body(self):
        out = []
        for x in self._units():
            self._forward_unit(x, out)
        if out and out[-1] == "wg_sync();":
            out.pop()                                  # the kernels close a model body with their own barrier
        return self._tables_text() + ("\n" if self.tables else "") + "\n".join(out)

    # ---- reverse pass
    def _target(self, s):
        """where the adjoint of an addressable array accumulates: (C base expression, needs WP, scale by w)"""
        if s.kind == "u": return "dlam", False
        if s.kind == "pslice": return f"(gp + {s.data['off']})", True
        if s.kind == "const": return None, False
        return f"(ws + {self.adj[s.id]})", False

    def _reverse_loop(self, x, seed, gathers):
        """Body of one loop over the indices of the unit whose value is array node x: recompute the fused elementwise chain at index i, then push `seed` (the
        adjoint of x at index i) back to what the chain reads — leaves, materialised values, scalars, gathers (through their scatter temporaries)."""
        lines, cache = [], {"__root__": x.id}
        self._value(x, "i", lines, cache)
        tree, seen = [], set()                         # the elementwise nodes evaluated in this loop, operands before users

        def walk(y):
            if y.id in seen or y.length == 0:
                return
            seen.add(y.id)
            if y.kind == "ew" and (y is x or y.id not in self.mat):
                for z in y.args:
                    walk(z)
                tree.append(y)
        walk(x)
        body = list(lines) + [f"double a{y.id} = 0.0;" for y in tree if y is not x]

        def push(arg, c):
            if arg.length == 0:
                if arg.kind == "pscalar":
                    body.append(f"if (WP) gs{arg.data['k']} += {c};")
                elif arg.kind in ("sum", "ew"):
                    body.append(f"b{arg.id} += {c};")
            elif arg.id in self.mat and arg is not x:                   # a materialised value read from its buffer at this index
                body.append(f"ws[{self.adj[arg.id]} + i] += {c};")
            elif arg.kind == "ew":
                body.append(f"a{arg.id} += {c};")
            elif arg.kind == "gather":
                if arg.id in self.tmp:
                    body.append(f"ws[{self.tmp[arg.id]} + i] {'+=' if arg.id in gathers else '='} {c};")
                    gathers[arg.id] = arg
            else:
                base, wp = self._target(arg)
                if base:
                    body.append(f"if (WP) {base}[i] += w * ({c});" if wp else f"{base}[i] += {c};")
        if x.kind != "ew":                             # the unit's value is a bare gather (a materialised gather of a gather): its adjoint goes straight on
            push(x, seed)
        for y in reversed(tree):
            a = [cache[z.id] if z.length else self._scalar_name(z) for z in y.args]
            ybar = seed if y is x else f"a{y.id}"
            for arg, dz in zip(y.args, self._partials(y, a, cache[y.id])):
                push(arg, ybar if dz == "1.0" else f"{ybar} * {dz}")
        return body

    def vjp_body(self):
        root = self.root
        units = self._units()
        out = []
        fw = []
        for x in units:                                # forward values of everything but du itself
            if x is root:
                continue
            self._forward_unit(x, fw)
        out += fw
        zero = []
        if not self.scalar_root:
            zero.append(f"HIPADJ_W_FOR(i, {self.n}) dlam[i] = 0.0;")
        for i, off in self.adj.items():
            zero.append(f"HIPADJ_W_FOR(i, {self._node(i).length}) ws[{off} + i] = 0.0;")
        out += zero
        out.append("wg_sync();")
        for k in self.pscalars:
            out.append(f"double gs{k} = 0.0, gu{k} = 0.0;")      # gs: per-thread partial (summed over the workgroup), gu: uniform contribution (every thread holds the same)
        for x in self.scalars:
            out.append(f"double b{x.id} = 0.0, bu{x.id} = 0.0;")
        if self.scalar_root:
            out.append(f"bu{root.id} = 1.0;")        # d g / d g
        for x in reversed(units):
            if x.length == 0 and x.kind == "ew":       # scalar arithmetic: adjoint = workgroup sum of the partials + the uniform part, handed on as uniform
                out.append(f"const double bt{x.id} = wg_sum(b{x.id}) + bu{x.id};")
                a = [self._scalar_name(y) for y in x.args]
                for arg, dz in zip(x.args, self._partials(x, a, f"s{x.id}")):
                    c = f"bt{x.id} * {dz}"
                    if arg.kind == "pscalar": out.append(f"gu{arg.data['k']} += {c};")
                    elif arg.kind in ("sum", "ew"): out.append(f"bu{arg.id} += {c};")
                continue
            gathers = {}
            if x.kind == "sum":
                out.append(f"const double bt{x.id} = wg_sum(b{x.id}) + bu{x.id};")
                src = x.args[0]
                body = self._reverse_loop(src, f"bt{x.id}", gathers)
                out.append(f"HIPADJ_W_FOR(i, {src.length}) {{ {' '.join(body)} }}")
            elif x.kind == "cmatvec":
                A = x.data["A"]; m, k = A.shape
                tab = self._table("double", A.ravel(), "mtab")
                zb = "lam" if x is root else f"(ws + {self.adj[x.id]})"
                base, wp = self._target(x.args[0])
                if base:
                    line = f"HIPADJ_W_FOR(j, {k}) {{ double s = 0.0; for (int i = 0; i < {m}; ++i) s += {tab}[i * {k} + j] * {zb}[i]; "
                    out.append(line + (f"if (WP) {base}[j] += w * s; }}" if wp else f"{base}[j] += s; }}"))
            elif x.kind == "matvec":
                m, k, off = x.data["m"], x.data["k"], x.data["off"]
                zb = "lam" if x is root else f"(ws + {self.adj[x.id]})"
                src = x.args[0]
                base, wp = self._target(src)
                sb = self._base(src)
                if base:
                    line = f"HIPADJ_W_FOR(j, {k}) {{ double s = 0.0; for (int i = 0; i < {m}; ++i) s += p[{off} + i + j * {m}] * {zb}[i]; "
                    line += (f"if (WP) {base}[j] += w * s; }}" if wp else f"{base}[j] += s; }}")
                    out.append(line)
                if m >= k:
                    out.append(f"if (WP) {{ for (int j = 0; j < {k}; ++j) {{ const double xj = w * {sb}[j]; HIPADJ_W_FOR(i, {m}) gp[{off} + i + j * {m}] += {zb}[i] * xj; }} }}")
                else:
                    out.append(f"if (WP) {{ for (int i = 0; i < {m}; ++i) {{ const double zi = w * {zb}[i]; HIPADJ_W_FOR(j, {k}) gp[{off} + i + j * {m}] += zi * {sb}[j]; }} }}")
            else:
                seed = "lam[i]" if x is root else f"ws[{self.adj[x.id]} + i]"
                body = self._reverse_loop(x, seed, gathers)
                out.append(f"HIPADJ_W_FOR(i, {x.length}) {{ {' '.join(body)} }}")
            out.append("wg_sync();")
            for g in gathers.values():                 # the scatter of a gather's adjoint, as a gather through the inverse index map: fixed order, no atomics
                src = g.args[0]
                base, wp = self._target(src)
                if not base:
                    continue
                idx, L = g.data["idx"], src.length
                tmp = f"(ws + {self.tmp[g.id]})"
                sh = int(idx[0])
                if len(idx) == L and np.array_equal(idx, (np.arange(L) + sh) % L):
                    inv = f"(j - {sh} >= 0 ? j - {sh} : j - {sh} + {L})" if sh else "j"
                    acc = f"{tmp}[{inv}]"
                    out.append(f"HIPADJ_W_FOR(j, {L}) {{ " + (f"if (WP) {base}[j] += w * {acc};" if wp else f"{base}[j] += {acc};") + " }")
                else:
                    order = np.argsort(idx, kind="stable")
                    ptr = np.searchsorted(idx[order], np.arange(L + 1))
                    pt, iv = self._table("int", ptr, "ptab"), self._table("int", order, "vtab")
                    out.append(f"HIPADJ_W_FOR(j, {L}) {{ double s = 0.0; for (int q = {pt}[j]; q < {pt}[j + 1]; ++q) s += {tmp}[{iv}[q]]; "
                               + (f"if (WP) {base}[j] += w * s;" if wp else f"{base}[j] += s;") + " }")
                out.append("wg_sync();")
        # parameters used as scalars: the workgroup sum of the per-thread partials (through `acc` when their indices are contiguous) + the uniform part
        fin = []
        for k in self.pscalars:
            if self.nacc:
                fin.append(f"acc[{k - self.acc_first}] += w * gs{k}; if (tid == 0) acc[{k - self.acc_first}] += w * gu{k};")
            else:
                fin.append(f"{{ const double s = wg_sum(gs{k}) + gu{k}; if (tid == 0) gp[{k}] += w * s; }}")
        if fin:
            out.append("if (WP) { " + " ".join(fin) + " }")
        while out and out[-1] == "wg_sync();":
            out.pop()
        return self._tables_text() + ("\n" if self.tables else "") + "\n".join(out)


def cost_body(fn, n, npar):
    """Trace a continuous cost g = fn(u, p, t, ops) (a scalar expression) and return (body, lds_doubles, nacc, acc_first) for hipadj_wmodel_set_cost."""
    Arr._count[0] = 0
    u, p, t = Arr("u", n), Arr("p", npar), Arr("t", 0)
    g = fn(u, p, t, Ops())
    if not isinstance(g, Arr) or g.length != 0:
        raise TypeError("the traced cost must return a scalar expression (ops.sum(...))")
    if g.kind not in ("sum", "ew"):
        g = g * 1.0
    G = _Gen(g, n, npar, scalar_root=True)
    body = G.vjp_body()
    return body, G.ws, G.nacc, G.acc_first


def bodies(fn, n, npar):
    """Trace fn(u, p, t, ops) -> du (an array expression of length n) and return (f_body, vjp_body, lds_doubles, nacc, acc_first)."""
    Arr._count[0] = 0
    u, p, t = Arr("u", n), Arr("p", npar), Arr("t", 0)
    du = fn(u, p, t, Ops())
    if not isinstance(du, Arr):
        raise TypeError("the traced function must return the array expression of du")
    if du.kind not in ("ew", "matvec", "cmatvec"):
        du = du * 1.0                                # a bare leaf / gather as the right-hand side: give it an elementwise root
    g = _Gen(du, n, npar)
    fb = g.forward_body()
    vb = g.vjp_body()
    return fb, vb, g.ws, g.nacc, g.acc_first
