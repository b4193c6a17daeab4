"""Recording stub of the `dolfin` namespace — TEST TOOLING, used only by
tests/golden/make_reference_form_goldens.py inside the build container.

FEniCS cannot be installed here, but the reference's *form-building* code is plain Python on
top of `from dolfin import *`.  This stub lets that code run unchanged: UFL-like objects build a
printable expression tree, DirichletBC / LinearVariationalSolver / PETScKrylovSolver record
their arguments instead of computing.  What comes out pins the Python-side semantics of the
reference (which Dirichlet sets, which integrals, which coefficients, which signs — quirks
included) for a given settings dict; it says nothing about DOLFIN's arithmetic.
Neither this stub nor the reference travels to the GPU box: only the JSON goldens do.
"""
import numbers

RECORD = {"solves": []}
DOLFIN_EPS = 3e-16


def near(a, b, eps=DOLFIN_EPS):
    return abs(a - b) < eps


def _wrap(x):
    if isinstance(x, Expr):
        return x
    if isinstance(x, numbers.Number):
        return Expr("num", x)
    if isinstance(x, (tuple, list)):
        return Expr("vec", *[_wrap(v) for v in x])
    raise TypeError("cannot use %r in a form" % (x,))


class Expr(object):
    def __init__(self, op, *args):
        self.op, self.args = op, args

    # arithmetic ------------------------------------------------------------
    def __add__(self, o):
        if isinstance(o, Form):
            return o.__radd__(self)
        return Expr("add", self, _wrap(o))

    def __radd__(self, o):
        if isinstance(o, numbers.Number) and o == 0:
            return self
        return Expr("add", _wrap(o), self)

    def __sub__(self, o):
        return Expr("sub", self, _wrap(o))

    def __rsub__(self, o):
        return Expr("sub", _wrap(o), self)

    def __neg__(self):
        return Expr("neg", self)

    def __mul__(self, o):
        if isinstance(o, Measure):
            return Form([(1, Integral(self, o))])
        if isinstance(o, Form):
            return o.__rmul__(self)
        return Expr("mul", self, _wrap(o))

    def __rmul__(self, o):
        return Expr("mul", _wrap(o), self)

    def __truediv__(self, o):
        return Expr("div", self, _wrap(o))

    __div__ = __truediv__

    def __rtruediv__(self, o):
        return Expr("div", _wrap(o), self)

    def __pow__(self, o):
        return Expr("pow", self, _wrap(o))

    def __getitem__(self, i):
        return Expr("index", self, _wrap(i))

    def __len__(self):
        return getattr(self, "_len", 3)

    def __call__(self, *a):
        return Expr("eval", self, *[_wrap(v) for v in a])

    @property
    def ufl_shape(self):
        return getattr(self, "_shape", ())

    @property
    def T(self):
        return Expr("transpose", self)

    def s(self):
        if self.op == "num":
            v = self.args[0]
            return repr(float(v)) if not float(v).is_integer() or abs(v) > 1e15 else "%d" % v
        if self.op == "symbol":
            return self.args[0]
        return "%s(%s)" % (self.op, ", ".join(a.s() if isinstance(a, Expr) else repr(a) for a in self.args))

    __repr__ = s
    __str__ = s


def _fn(name):
    def f(*a):
        return Expr(name, *[_wrap(x) for x in a])
    f.__name__ = name
    return f


inner = _fn("inner")
dot = _fn("dot")
grad = _fn("grad")
div = _fn("div")
sym = _fn("sym")
tr = _fn("tr")
sqrt = _fn("sqrt")
avg = _fn("avg")
jump = _fn("jump")
nabla_grad = _fn("nabla_grad")
outer = _fn("outer")
exp = _fn("exp")
ln = _fn("ln")


def pow(a, b):  # noqa: A001  (the reference calls pow(T, 4) on UFL objects)
    if isinstance(a, Expr) or isinstance(b, Expr):
        return Expr("pow", _wrap(a), _wrap(b))
    import builtins
    return builtins.pow(a, b)


def Identity(n):
    return Expr("Identity", _wrap(n))


def as_matrix(m):
    return Expr("as_matrix", *[_wrap(list(r)) for r in m])


def as_vector(v):
    return _wrap(list(v))


def Circumradius(mesh):
    return Expr("symbol", "Circumradius")


def FacetNormal(mesh):
    return Expr("symbol", "n")


class Constant(Expr):
    def __init__(self, value, **kw):
        Expr.__init__(self, "Constant", _wrap(value))
        self.value = value
        if isinstance(value, (tuple, list)):
            self._len = len(value)
            self._shape = (len(value),)

    def values(self):
        import numpy as np
        return np.atleast_1d(np.asarray(self.value, dtype=float))


class Expression(Expr):
    def __init__(self, code=None, degree=None, cppcode=None, **params):
        Expr.__init__(self, "Expression", Expr("symbol", repr(code)))
        self.code, self.degree, self.params = code, degree, params
        if isinstance(code, (tuple, list)):
            self._len = len(code)
            self._shape = (len(code),)


class _Geometry(object):
    def __init__(self, d):
        self._d = d

    def dim(self):
        return self._d


class Mesh(object):
    def __init__(self, filename=None, dim=3):
        self.filename = filename
        self._dim = dim

    def geometry(self):
        return _Geometry(self._dim)

    def topology(self):
        return _Geometry(self._dim)

    def mpi_comm(self):
        return None

    def hmin(self):
        return 1.0

    def ufl_cell(self):
        return "tetrahedron" if self._dim == 3 else "triangle"


def BoxMesh(*a):
    return Mesh("BoxMesh%r" % (tuple(repr(x) for x in a),))


def UnitCubeMesh(*a):
    return Mesh("UnitCubeMesh%r" % (a,))


def UnitSquareMesh(*a):
    return Mesh("UnitSquareMesh%r" % (a,), dim=2)


def RectangleMesh(*a):
    return Mesh("RectangleMesh%r" % (tuple(repr(x) for x in a),), dim=2)


class Point(object):
    def __init__(self, *xyz):
        self.xyz = xyz

    def __repr__(self):
        return "Point%r" % (self.xyz,)


class MeshFunction(object):
    def __init__(self, value_type, mesh, dim_or_file, value=None):
        self.mesh, self.arg = mesh, dim_or_file
        self.marks = []

    def set_all(self, v):
        pass

    def __repr__(self):
        return "MeshFunction(%r)" % (self.arg,)


class SubDomain(object):
    def mark(self, mf, marker_id):
        mf.marks.append((type(self).__name__, marker_id))


class AutoSubDomain(SubDomain):
    def __init__(self, fn):
        self.fn = fn


class _Element(object):
    def __init__(self, degree):
        self._d = degree

    def degree(self):
        return self._d


class FiniteElement(object):
    def __init__(self, family, cell, degree):
        self.family, self.cell, self.degree_, self.kind = family, cell, degree, "scalar"

    def __mul__(self, other):
        return MixedElement([self, other])

    def label(self):
        return "%s%d" % (self.family, self.degree_)


class VectorElement(FiniteElement):
    def __init__(self, family, cell, degree):
        FiniteElement.__init__(self, family, cell, degree)
        self.kind = "vector"

    def label(self):
        return "Vector%s%d" % (self.family, self.degree_)


class MixedElement(object):
    def __init__(self, elements):
        self.elements, self.kind = list(elements), "mixed"

    def label(self):
        return "Mixed(%s)" % " x ".join(e.label() for e in self.elements)


class FunctionSpace(object):
    def __init__(self, mesh, family, degree=None, constrained_domain=None, _kind="scalar", _label="V"):
        if isinstance(family, (FiniteElement, MixedElement)):          # FunctionSpace(mesh, element)
            element = family
            self._mesh, self.family, self._degree = mesh, element.label(), getattr(element, "degree_", None)
            self.kind, self.label, self.element = element.kind, "W" if element.kind == "mixed" else _label, element
            self._ufl_element = _Element(self._degree)
            return
        self._mesh, self.family, self._degree, self.kind, self.label = mesh, family, degree, _kind, _label
        self._ufl_element = _Element(degree)

    def mesh(self):
        return self._mesh

    def sub(self, i):
        if self.kind == "mixed":
            e = self.element.elements[i]
            V = FunctionSpace(self._mesh, e, _label="%s.sub(%d)" % (self.label, i))
            return V
        return FunctionSpace(self._mesh, self.family, self._degree, _kind="component", _label="%s.sub(%d)" % (self.label, i))

    def dofmap(self):
        return self

    def set(self, vec, value):
        pass

    def set_x(self, vec, value, component):
        pass

    def __repr__(self):
        return "%s[%s%s]" % (self.label, self.family, self._degree)


def VectorFunctionSpace(mesh, family, degree, constrained_domain=None):
    return FunctionSpace(mesh, family, degree, _kind="vector", _label="V")


class _Vec(object):
    def __setitem__(self, k, v):
        pass

    def apply(self, mode):
        pass

    def copy(self):
        return _Vec()


class Function(Expr):
    _count = 0

    def __init__(self, V, name=None):
        if isinstance(V, Function):
            name, V = "copy(%s)" % V.name_, V.V
        Function._count += 1
        self.name_ = name or "w%d" % Function._count
        Expr.__init__(self, "symbol", self.name_)
        self.V = V
        if getattr(V, "kind", "") == "vector":
            d = _vdim(V)
            self._len, self._shape = d, (d,)

    def assign(self, other):
        pass

    def vector(self):
        return _Vec()

    def function_space(self):
        return self.V

    def rename(self, a, b):
        pass


def TrialFunction(V):
    f = Expr("symbol", "u_trial")
    if getattr(V, "kind", "") == "mixed":
        f._elements = V.element.elements
    if getattr(V, "kind", "") == "vector":
        f._len, f._shape = _vdim(V), (_vdim(V),)
    return f


def TestFunction(V):
    f = Expr("symbol", "v_test")
    if getattr(V, "kind", "") == "mixed":
        f._elements = V.element.elements
    if getattr(V, "kind", "") == "vector":
        f._len, f._shape = _vdim(V), (_vdim(V),)
    return f


def _vdim(V):
    """components of a VectorFunctionSpace = the geometric dimension of its mesh"""
    m = getattr(V, "mesh_", None) or (V.mesh() if hasattr(V, "mesh") and callable(V.mesh) else None)
    return m.geometry().dim() if m is not None else 3


def split(f):
    """Components of a function of a mixed space: symbols named after the function."""
    V = getattr(f, "V", None)
    name = f.args[0] if f.op == "symbol" else f.s()
    elements = V.element.elements if V is not None and getattr(V, "kind", "") == "mixed" else getattr(f, "_elements", [])
    out = []
    for i, e in enumerate(elements):
        c = Expr("symbol", "%s[%d]" % (name, i))
        if e.kind == "vector":
            c._len, c._shape = 3, (3,)
        out.append(c)
    return tuple(out)


def TensorFunctionSpace(mesh, family, degree):
    return FunctionSpace(mesh, family, degree, _kind="tensor", _label="T")


def interpolate(expr, V):
    f = Function(V, "interpolate(%s)" % expr.s())
    return f


def project(expr, V):
    return Function(V, "project(%s)" % expr.s())


class Measure(object):
    def __init__(self, kind, subdomain_data=None, subdomain_id=None, domain=None):
        self.kind, self.data, self.id = kind, subdomain_data, subdomain_id

    def __call__(self, subdomain_id=None, domain=None, **kw):
        return Measure(self.kind, self.data, subdomain_id)

    def __rmul__(self, o):
        return Form([(1, Integral(_wrap(o), self))])

    def s(self):
        return self.kind if self.id is None else "%s(%s)" % (self.kind, self.id)


dx = Measure("dx")
ds = Measure("ds")
dS = Measure("dS")


class Integral(object):
    def __init__(self, integrand, measure):
        self.integrand, self.measure = integrand, measure


class Form(object):
    def __init__(self, terms):
        self.terms = list(terms)      # [(sign/scalar expr, Integral)]

    def __add__(self, o):
        if isinstance(o, numbers.Number) and o == 0:
            return self
        return Form(self.terms + o.terms)

    def __radd__(self, o):
        if isinstance(o, numbers.Number) and o == 0:
            return self
        if isinstance(o, Form):
            return Form(o.terms + self.terms)
        raise TypeError("cannot add %r to a Form" % (o,))

    def __sub__(self, o):
        return Form(self.terms + [(-c, i) for c, i in o.terms])

    def __neg__(self):
        return Form([(-c, i) for c, i in self.terms])

    def __mul__(self, o):
        return Form([(c, Integral(Expr("mul", i.integrand, _wrap(o)), i.measure)) for c, i in self.terms])

    def __rmul__(self, o):
        return Form([(c, Integral(Expr("mul", _wrap(o), i.integrand), i.measure)) for c, i in self.terms])

    def describe(self):
        return [{"sign": c, "integrand": i.integrand.s(), "measure": i.measure.s()} for c, i in self.terms]


def lhs(F):
    return ("lhs", F)


def rhs(F):
    return ("rhs", F)


def system(F):
    return ("lhs", F), ("rhs", F)


def action(F, u):
    return Form([(c, Integral(Expr("action", i.integrand, u), i.measure)) for c, i in F.terms])


def derivative(F, u, du):
    return ("derivative", F, u.s(), du.s())


def assemble(form, **kw):
    RECORD.setdefault("assembled", []).append(form.describe() if isinstance(form, Form) else repr(form))
    return 1.0   # e.g. the boundary area used by the 'force' BC


class DirichletBC(object):
    def __init__(self, V, value, markers, marker_id, method=None):
        self.V, self.value, self.markers, self.marker_id = V, value, markers, marker_id

    def describe(self):
        return {"space": self.V.label, "value": self.value.s() if isinstance(self.value, Expr) else repr(self.value),
                "marker": self.marker_id}


class _Params(dict):
    def __getitem__(self, k):
        if k not in self:
            dict.__setitem__(self, k, _Params())
        return dict.__getitem__(self, k)


parameters = _Params()


class LinearVariationalProblem(object):
    def __init__(self, a, L, u, bcs):
        self.a, self.L, self.u, self.bcs = a, L, u, bcs


class LinearVariationalSolver(object):
    def __init__(self, problem):
        self.problem = problem
        self.parameters = {"linear_solver": "default", "preconditioner": "default", "symmetric": False,
                           "print_rhs": False, "print_matrix": False, "lu_solver": {}, "krylov_solver": {}}

    def solve(self):
        F = self.problem.a[1]
        RECORD["solves"].append({"kind": "LinearVariationalSolver", "terms": F.describe(),
                                 "bcs": [b.describe() for b in self.problem.bcs],
                                 "parameters": {k: v for k, v in self.parameters.items() if not isinstance(v, dict)}})


def assemble_system(a, L, bcs, **kw):
    RECORD["solves"].append({"kind": "assemble_system+PETScKrylovSolver(cg, petsc_amg)", "terms": a[1].describe(),
                             "bcs": [b.describe() for b in bcs]})
    return _Mat(), _Vec()


class _Mat(object):
    def set_near_nullspace(self, ns):
        pass


def as_backend_type(A):
    return A


class PETScPreconditioner(object):
    def __init__(self, name):
        self.name = name


class PETScOptions(object):
    opts = {}

    @staticmethod
    def set(k, v=None):
        PETScOptions.opts[k] = v


class PETScKrylovSolver(object):
    def __init__(self, method, pc):
        self.method, self.pc = method, pc
        self.parameters = {}

    def set_operator(self, A):
        pass

    def solve(self, x, b):
        RECORD["solves"][-1]["krylov"] = {"method": self.method, "pc": self.pc.name, "petsc_options": dict(PETScOptions.opts)}


class VectorSpaceBasis(object):
    def __init__(self, vecs):
        self.n = len(vecs)
        RECORD["nullspace_vectors"] = self.n

    def orthonormalize(self):
        pass


class NonlinearVariationalProblem(object):
    def __init__(self, F, u, bcs, J):
        self.F, self.u, self.bcs, self.J = F, u, bcs, J


class NonlinearVariationalSolver(object):
    def __init__(self, problem):
        self.problem = problem
        self.parameters = {"newton_solver": {}}

    def solve(self):
        RECORD["solves"].append({"kind": "NonlinearVariationalSolver", "terms": self.problem.F.describe(),
                                 "bcs": [b.describe() for b in self.problem.bcs]})


class Timer(object):
    def __init__(self, name):
        pass

    def start(self):
        pass

    def stop(self):
        pass

    def elapsed(self):
        return (0.0,)


class MPI(object):
    comm_world = None

    @staticmethod
    def size(c):
        return 1

    @staticmethod
    def rank(c):
        return 0


def mpi_comm_world():
    raise RuntimeError("2018+ API")   # the reference falls back to MPI.comm_world (SolverBase.py:109-114)


def set_log_active(flag):
    pass


def set_log_level(level):
    pass


ERROR = 40


def plot(*a, **k):
    pass


def interactive():
    pass


def File(name, *a):
    class _F(object):
        def __lshift__(self, o):
            pass
    return _F()


class PointSource(object):
    pass


def dolfin_version():
    return "2019.1.0"


__version__ = "2019.1.0"
