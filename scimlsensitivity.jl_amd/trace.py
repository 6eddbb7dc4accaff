"""A path from a host-language right-hand side to the device: trace a Python callable `f(du, u, p, t)` once with symbolic
operands, differentiate the recorded expression graph in reverse mode, and emit the three C bodies `hipadj_model_register` takes —
the reference's `ODEFunction(f!; vjp, vjp_p)` seam (src/derivative_wrappers.jl:284-359, test/Core3/user_vjp.jl:77-134) with C text
instead of closures.  This is the Python-mirror counterpart of julia/HIPAdj/ext/HIPAdjSymbolicsExt.jl (Symbolics -> C), and it IS
executed by the test suite.

    def lorenz(du, u, p, t):
        du[0] = p[0] * (u[1] - u[0]); du[1] = u[0] * (p[1] - u[2]) - u[1]; du[2] = u[0] * u[1] - p[2] * u[2]
    fun = sa.DeviceFunction.from_callable("lorenz_py", lorenz, n=3, np=3)        # f, (df/du)^T lam and (df/dp)^T lam as C text

Supported inside `f`: + - * / unary minus, ** with a numeric exponent, and the functions of this module (sin cos tan exp log sqrt tanh
sinh cosh atan fabs); Python control flow is evaluated at trace time (loops unroll, branches on symbolic values are an error).
No host evaluation of `f` ever takes part in a solve: the emitted text is compiled for gfx950 and runs on the device."""
import math


class Node:
    """One value of the traced computation.  op: 'const' | 'var' | '+' '-' '*' '/' 'neg' 'pow' | a unary function name."""
    __slots__ = ("op", "args", "val", "name")

    def __init__(self, op, args=(), val=None, name=None):
        self.op, self.args, self.val, self.name = op, tuple(args), val, name

    # ---- operator overloads: build the graph, fold constants, drop neutral elements
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return sub(self, o)
    def __rsub__(self, o): return sub(o, self)
    def __mul__(self, o): return mul(self, o)
    def __rmul__(self, o): return mul(o, self)
    def __truediv__(self, o): return div(self, o)
    def __rtruediv__(self, o): return div(o, self)
    def __neg__(self): return neg(self)
    def __pos__(self): return self

    def __pow__(self, e):
        if isinstance(e, Node):
            if e.op != "const":
                raise TypeError("only numeric exponents are traced (x ** 2.5); write exp(y * log(x)) for symbolic ones")
            e = e.val
        return powc(self, float(e))

    def __bool__(self):
        raise TypeError("a traced value has no truth value: branches on the state are not traceable (the device code is straight-line)")

    def _no_compare(self, o):
        raise TypeError("comparisons of traced values are not traceable (the device code is straight-line)")

    __lt__ = __le__ = __gt__ = __ge__ = __eq__ = __ne__ = _no_compare     # == / != too: an identity comparison would silently bake one branch into the device code
    __hash__ = object.__hash__                                              # the graph maps key nodes by identity

    def __abs__(self): return fabs(self)


def _n(x):
    return x if isinstance(x, Node) else Node("const", val=float(x))


def _isc(x, v=None):
    return x.op == "const" and (v is None or x.val == v)


def add(a, b):
    a, b = _n(a), _n(b)
    if _isc(a) and _isc(b): return _n(a.val + b.val)
    if _isc(a, 0.0): return b
    if _isc(b, 0.0): return a
    return Node("+", (a, b))


def sub(a, b):
    a, b = _n(a), _n(b)
    if _isc(a) and _isc(b): return _n(a.val - b.val)
    if _isc(b, 0.0): return a
    if _isc(a, 0.0): return neg(b)
    return Node("-", (a, b))


def mul(a, b):
    a, b = _n(a), _n(b)
    if _isc(a) and _isc(b): return _n(a.val * b.val)
    if _isc(a, 0.0) or _isc(b, 0.0): return _n(0.0)
    if _isc(a, 1.0): return b
    if _isc(b, 1.0): return a
    if _isc(a, -1.0): return neg(b)
    if _isc(b, -1.0): return neg(a)
    return Node("*", (a, b))


def div(a, b):
    a, b = _n(a), _n(b)
    if _isc(a) and _isc(b): return _n(a.val / b.val)
    if _isc(a, 0.0): return _n(0.0)
    if _isc(b, 1.0): return a
    return Node("/", (a, b))


def neg(a):
    a = _n(a)
    if _isc(a): return _n(-a.val)
    if a.op == "neg": return a.args[0]
    return Node("neg", (a,))


def powc(a, e):
    a = _n(a)
    if _isc(a): return _n(a.val ** e)
    if e == 0.0: return _n(1.0)
    if e == 1.0: return a
    if e == 2.0: return Node("*", (a, a))
    return Node("pow", (a,), val=e)


_UNARY = {  # name -> (python function, derivative as a graph of the argument x and the value y)
    "sin": (math.sin, lambda x, y: cos(x)), "cos": (math.cos, lambda x, y: neg(sin(x))), "tan": (math.tan, lambda x, y: add(1.0, mul(y, y))),
    "exp": (math.exp, lambda x, y: y), "log": (math.log, lambda x, y: div(1.0, x)), "sqrt": (math.sqrt, lambda x, y: div(0.5, y)),
    "tanh": (math.tanh, lambda x, y: sub(1.0, mul(y, y))), "sinh": (math.sinh, lambda x, y: cosh(x)), "cosh": (math.cosh, lambda x, y: sinh(x)),
    "atan": (math.atan, lambda x, y: div(1.0, add(1.0, mul(x, x)))), "fabs": (math.fabs, lambda x, y: div(x, y)),
}


def _unary(name):
    fn = _UNARY[name][0]

    def f(x):
        if not isinstance(x, Node):
            return fn(x)
        return _n(fn(x.val)) if _isc(x) else Node(name, (x,))
    f.__name__ = name
    return f


sin, cos, tan, exp, log, sqrt, tanh, sinh, cosh, atan, fabs = (_unary(k) for k in ("sin", "cos", "tan", "exp", "log", "sqrt", "tanh", "sinh", "cosh", "atan", "fabs"))


class _Out(list):
    """du of f(du, u, p, t): item assignment only; unassigned components are zero"""

    def __init__(self, n):
        super().__init__(_n(0.0) for _ in range(n))

    def __setitem__(self, i, v):
        if isinstance(i, slice):
            raise TypeError("assign the components of du one by one")
        list.__setitem__(self, i, _n(v))


def trace(f, n, np_):
    """Run f(du, u, p, t) once on symbolic operands; returns the n output graphs."""
    u = [Node("var", name=f"u[{i}]") for i in range(n)]
    p = [Node("var", name=f"p[{i}]") for i in range(np_)]
    t = Node("var", name="t")
    du = _Out(n)
    r = f(du, u, p, t)
    if r is not None and r is not du:
        raise TypeError("f must be in-place: f(du, u, p, t) writes du and returns None")
    return list(du), u, p, t


def topo(roots):
    seen, order = set(), []

    def visit(x):
        stack = [(x, False)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node); continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for a in node.args:
                if id(a) not in seen:
                    stack.append((a, False))
    for r in roots:
        visit(r)
    return order


def vjp_graphs(outs, wrt, seeds):
    """Reverse-mode differentiation of the traced graph: returns [sum_i seeds[i] * d outs[i] / d w  for w in wrt] as graphs."""
    order = topo(outs)
    adj = {}
    for o, s in zip(outs, seeds):
        adj[id(o)] = add(adj.get(id(o), 0.0), s)
    for node in reversed(order):
        g = adj.get(id(node))
        if g is None or node.op in ("const", "var"):
            continue
        a = node.args
        if node.op == "+":
            contrib = ((a[0], g), (a[1], g))
        elif node.op == "-":
            contrib = ((a[0], g), (a[1], neg(g)))
        elif node.op == "*":
            contrib = ((a[0], mul(g, a[1])), (a[1], mul(g, a[0])))
        elif node.op == "/":
            contrib = ((a[0], div(g, a[1])), (a[1], neg(mul(g, div(node, a[1])))))
        elif node.op == "neg":
            contrib = ((a[0], neg(g)),)
        elif node.op == "pow":
            contrib = ((a[0], mul(g, mul(node.val, powc(a[0], node.val - 1.0)))),)
        else:
            contrib = ((a[0], mul(g, _UNARY[node.op][1](a[0], node))),)
        for x, c in contrib:
            adj[id(x)] = add(adj.get(id(x), 0.0), c)
    return [_n(adj.get(id(w), 0.0)) for w in wrt]


def emit(outs, lhs, real="double"):
    """C text assigning lhs[i] = outs[i], with one temporary per shared subexpression."""
    order = topo(outs)
    uses = {}
    for node in order:
        for a in node.args:
            uses[id(a)] = uses.get(id(a), 0) + 1
    names, lines = {}, []

    def lit(v):
        return repr(float(v)) if math.isfinite(v) else ("INFINITY" if v > 0 else "-INFINITY" if v < 0 else "NAN")

    def ref(x):
        return names[id(x)]
    for node in order:
        a = node.args
        if node.op == "const":
            names[id(node)] = lit(node.val) if node.val >= 0 else f"({lit(node.val)})"; continue
        if node.op == "var":
            names[id(node)] = node.name; continue
        if node.op in "+-*/" and len(node.op) == 1:
            ex = f"({ref(a[0])} {node.op} {ref(a[1])})"
        elif node.op == "neg":
            ex = f"(-{ref(a[0])})"
        elif node.op == "pow":
            ex = f"pow({ref(a[0])}, {lit(node.val)})"
        else:
            ex = f"{node.op}({ref(a[0])})"
        if uses.get(id(node), 0) > 1:
            nm = f"w{len(lines)}"
            lines.append(f"const {real} {nm} = {ex};")
            names[id(node)] = nm
        else:
            names[id(node)] = ex
    for i, o in enumerate(outs):
        lines.append(f"{lhs}[{i}] = {ref(o)};")
    return " ".join(lines)


def evaluate(outs, env):
    """Numeric value of traced graphs for env = {'u[0]': ..., 'p[1]': ..., 't': ..., 'lam[0]': ...} — test helper (host, float)."""
    vals = {}
    for node in topo(outs):
        a = [vals[id(x)] for x in node.args]
        if node.op == "const": v = node.val
        elif node.op == "var": v = env[node.name]
        elif node.op == "+": v = a[0] + a[1]
        elif node.op == "-": v = a[0] - a[1]
        elif node.op == "*": v = a[0] * a[1]
        elif node.op == "/": v = a[0] / a[1]
        elif node.op == "neg": v = -a[0]
        elif node.op == "pow": v = a[0] ** node.val
        else: v = _UNARY[node.op][0](a[0])
        vals[id(node)] = v
    return [vals[id(o)] for o in outs]


def bodies(f, n, np_, auto_vjp=False, bundle=False):
    """(f_body, vjp_u_body, vjp_p_body) for hipadj_model_register; auto_vjp = True: only f (locals `real`), the device differentiates
    it with forward-mode dual numbers (the reference's autojacvec = true)."""
    outs, u, p, t = trace(f, n, np_)
    if auto_vjp:
        return emit(outs, "du", real="real"), None, None
    lam = [Node("var", name=f"lam[{i}]") for i in range(n)]
    # bundle = True: VJP temporaries are `auto`, so that the device can compile these bodies for lam = Cols<G> (a bundle of segment columns through one
    # pass of the body, csrc/hipadj_models.hpp) as well as for lam = double; with `double` temporaries (the default until the bundled kernels of traced
    # models have had their GPU parity run) a lam term held in a temporary forces the per-column form, which is what every GPU test of round 2 exercised
    real = "auto" if bundle else "double"
    return emit(outs, "du"), emit(vjp_graphs(outs, u, lam), "out", real=real), emit(vjp_graphs(outs, p, lam), "out", real=real)


def discrete_loss_bodies(l, n, np_):
    """A discrete loss written in the host language, `l(u, p, t, i, d) -> scalar` (u, p, d indexable; t the loss time; i its 0-based index as a real number; d the data
    column of this trajectory and time), traced once: returns (dgdu_body, dgdp_body, l_body) for hipadj_model_set_discrete_loss[_function] - the gradients by reverse-mode
    differentiation of the recorded graph, the loss itself for the device-side value (hipadj_loss_value).  The reference evaluates the user's dgdu_discrete / dgdp_discrete
    closures on the host per loss time (src/adjoint_common.jl:771-779); here they become device text once."""
    u = [Node("var", name=f"u[{k}]") for k in range(n)]
    p = [Node("var", name=f"p[{k}]") for k in range(np_)]
    d = [Node("var", name=f"d[{k}]") for k in range(n)]
    t = Node("var", name="t")
    i = Node("var", name="(double)i")
    out = _n(l(u, p, t, i, d))
    gu = vjp_graphs([out], u, [_n(1.0)])
    gp = vjp_graphs([out], p, [_n(1.0)])
    lb = "real l_[1]; " + emit([out], "l_", real="real") + " l = l_[0];"      # emit assigns an array; the body's result variable is the scalar `l`
    return emit(gu, "out"), emit(gp, "out"), lb
