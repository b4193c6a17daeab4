"""C++-syntax scalar expressions of ``dolfin.Expression`` strings, evaluated over arrays of points.

DOLFIN JIT-compiles ``Expression("300 + 60*x[1]", degree=1)`` as C++ (SolverBase.py:310-314, 364, 387;
examples/test_electrostatics.py:73-78).  Here the string is parsed by a small recursive-descent parser for the C
expression grammar it can contain - no Python ``eval``: a case file cannot reach anything but arithmetic - and
evaluated with numpy over all points at once.  C semantics are kept where they differ from Python's:

* integer literals stay integers, ``1/2`` is 0 and ``7/2*x[0]`` is ``3*x[0]`` (truncating division, ``%`` likewise);
* ``a ? b : c``, ``&&``, ``||``, ``!`` and the comparisons yield 0 / 1;
* ``^`` is C's XOR, which C++ rejects on doubles: it raises here instead of silently meaning a power.

Names: ``x[0..2]``, the user parameters of the Expression (numbers), the <cmath> functions and pi / DOLFIN_PI / DOLFIN_EPS.
"""
from __future__ import annotations

import math
import re

import numpy as np

DOLFIN_EPS = 3.0e-16

FUNCTIONS = {
    "sin": np.sin, "cos": np.cos, "tan": np.tan, "exp": np.exp, "log": np.log, "log10": np.log10, "sqrt": np.sqrt,
    "pow": np.power, "fabs": np.abs, "abs": np.abs, "atan": np.arctan, "atan2": np.arctan2,
    "asin": np.arcsin, "acos": np.arccos, "sinh": np.sinh, "cosh": np.cosh, "tanh": np.tanh,
    "floor": np.floor, "ceil": np.ceil, "fmin": np.minimum, "fmax": np.maximum, "min": np.minimum, "max": np.maximum,
    "erf": np.vectorize(math.erf, otypes=[float]),
}
CONSTANTS = {"pi": math.pi, "DOLFIN_PI": math.pi, "M_PI": math.pi, "DOLFIN_EPS": DOLFIN_EPS}

_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)|(\d+)|([A-Za-z_]\w*)|"
                    r"(&&|\|\||<=|>=|==|!=|[-+*/%<>!?:(),\[\]^]))")


class CExprError(ValueError):
    pass


def _tokens(src):
    out, pos = [], 0
    src = src.rstrip()
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise CExprError("unexpected character %r at position %d" % (src[pos:pos + 1], pos))
        if m.group(1) is not None:
            out.append(("f", float(m.group(1))))
        elif m.group(2) is not None:
            out.append(("i", int(m.group(2))))
        elif m.group(3) is not None:
            out.append(("n", m.group(3)))
        else:
            out.append(("o", m.group(4)))
        pos = m.end()
    out.append(("end", None))
    return out


class _Parser:
    """expr := ternary;  precedence (low to high): ?:  ||  &&  == !=  < > <= >=  + -  * / %  unary  primary."""

    def __init__(self, src):
        self.t = _tokens(src)
        self.i = 0

    def peek(self, op=None):
        k, v = self.t[self.i]
        return (k == "o" and v == op) if op is not None else (k, v)

    def take(self, op=None):
        k, v = self.t[self.i]
        if op is not None and not (k == "o" and v == op):
            raise CExprError("expected %r, found %r" % (op, v if k != "end" else "end of expression"))
        self.i += 1
        return k, v

    def parse(self):
        node = self.ternary()
        if self.t[self.i][0] != "end":
            if self.t[self.i] == ("o", "^"):
                raise CExprError("'^' is XOR in C++ (not a power): write pow(a, b)")
            raise CExprError("unexpected %r" % (self.t[self.i][1],))
        return node

    def ternary(self):
        c = self.binary(0)
        if self.peek("?"):
            self.take()
            a = self.ternary()
            self.take(":")
            b = self.ternary()
            return ("?", c, a, b)
        return c

    LEVELS = (("||",), ("&&",), ("==", "!="), ("<", ">", "<=", ">="), ("+", "-"), ("*", "/", "%"))

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        node = self.binary(level + 1)
        while self.t[self.i][0] == "o" and self.t[self.i][1] in self.LEVELS[level]:
            op = self.take()[1]
            node = (op, node, self.binary(level + 1))
        return node

    def unary(self):
        if self.peek("-") or self.peek("+") or self.peek("!"):
            op = self.take()[1]
            return ("u" + op, self.unary())
        return self.primary()

    def primary(self):
        k, v = self.take()
        if k == "f":
            return ("num", v, False)
        if k == "i":
            return ("num", v, True)
        if k == "o" and v == "(":
            node = self.ternary()
            self.take(")")
            return node
        if k == "n":
            if self.peek("["):
                self.take()
                ik, iv = self.take()
                self.take("]")
                if v != "x" or ik != "i" or iv > 2:
                    raise CExprError("only x[0], x[1], x[2] can be indexed")
                return ("x", iv)
            if self.peek("("):
                self.take()
                args = []
                if not self.peek(")"):
                    args.append(self.ternary())
                    while self.peek(","):
                        self.take()
                        args.append(self.ternary())
                self.take(")")
                if v not in FUNCTIONS:
                    raise CExprError("unknown function %s()" % v)
                return ("call", v, args)
            return ("name", v)
        if k == "o" and v == "^":
            raise CExprError("'^' is XOR in C++ (not a power): write pow(a, b)")
        raise CExprError("unexpected %r" % (v if k != "end" else "end of expression",))


def _trunc_div(a, b):
    q = np.floor_divide(np.abs(a), np.abs(b))
    return np.where((a < 0) != (b < 0), -q, q)


def _eval(node, x, names):
    """-> (value array or scalar, is_integer)."""
    kind = node[0]
    if kind == "num":
        return node[1], node[2]
    if kind == "x":
        return x[node[1]], False
    if kind == "name":
        name = node[1]
        if name in names:
            return float(names[name]), False          # DOLFIN declares every user parameter as a double member
        if name in CONSTANTS:
            return CONSTANTS[name], False
        raise CExprError("unknown name '%s' (parameters: %s)" % (name, ", ".join(sorted(names)) or "none"))
    if kind == "call":
        args = [np.asarray(_eval(a, x, names)[0], dtype=np.float64) for a in node[2]]
        try:
            return FUNCTIONS[node[1]](*args), False
        except TypeError as e:
            raise CExprError("%s(): %s" % (node[1], e))
    if kind == "?":
        c, _ = _eval(node[1], x, names)
        a, ia = _eval(node[2], x, names)
        b, ib = _eval(node[3], x, names)
        return np.where(np.asarray(c) != 0, a, b), ia and ib
    if kind == "u-":
        v, iv = _eval(node[1], x, names)
        return -v if not isinstance(v, np.ndarray) else np.negative(v), iv
    if kind == "u+":
        return _eval(node[1], x, names)
    if kind == "u!":
        v, _ = _eval(node[1], x, names)
        return (np.asarray(v) == 0).astype(np.int64), True
    a, ia = _eval(node[1], x, names)
    b, ib = _eval(node[2], x, names)
    both = ia and ib
    if kind == "+":
        return a + b, both
    if kind == "-":
        return a - b, both
    if kind == "*":
        return a * b, both
    if kind == "/":
        if both:
            if np.any(np.asarray(b) == 0):
                raise CExprError("integer division by zero")
            return _trunc_div(np.asarray(a), np.asarray(b)), True
        return np.asarray(a, dtype=np.float64) / b, False
    if kind == "%":
        if not both:
            raise CExprError("'%' needs integer operands in C++ (use fmod)")
        return np.asarray(a) - _trunc_div(np.asarray(a), np.asarray(b)) * np.asarray(b), True
    cmp_ops = {"<": np.less, ">": np.greater, "<=": np.less_equal, ">=": np.greater_equal, "==": np.equal, "!=": np.not_equal}
    if kind in cmp_ops:
        return cmp_ops[kind](a, b).astype(np.int64), True
    if kind == "&&":
        return ((np.asarray(a) != 0) & (np.asarray(b) != 0)).astype(np.int64), True
    if kind == "||":
        return ((np.asarray(a) != 0) | (np.asarray(b) != 0)).astype(np.int64), True
    raise CExprError("internal: unknown node %r" % (kind,))


class CExpr:
    """One parsed scalar expression; ``CExpr(src)(pts[n,3], params) -> float64[n]``."""

    def __init__(self, src):
        if not isinstance(src, str):
            src = repr(float(src))
        self.src = src
        self.tree = _Parser(src).parse()

    def __call__(self, pts, params=None):
        pts = np.asarray(pts, dtype=np.float64)
        names = {}
        for k, v in (params or {}).items():
            if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
                names[k] = v
            elif hasattr(v, "__float__"):
                names[k] = float(v)
        v, _ = _eval(self.tree, (pts[:, 0], pts[:, 1], pts[:, 2]), names)
        return np.broadcast_to(np.asarray(v, dtype=np.float64), (pts.shape[0],)).copy()
