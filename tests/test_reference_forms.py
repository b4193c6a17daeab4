"""Form-recognition parity against the REFERENCE's own Python code.

tests/golden/reference_forms.json was produced by running /root/reference/FenicsSolver's solver
classes, unchanged, against a recording stub of dolfin (tests/golden/make_reference_form_goldens.py):
for a given settings dict it holds the Dirichlet conditions and the integrals the reference hands to
DOLFIN.  Here the same settings go through fenicssolver_amd's generate_form and both sides are
reduced to the same canonical object — the residual F as a polynomial over basis monomials
(u_trial, v_test, grad(.), w_prev, ...) per integration measure — and compared coefficient by
coefficient.  Signs, theta weights, the capacity-scaled Neumann term (Appendix B-Q8) and the
reversed elasticity loads (B-Q3) are all covered by this."""
import collections
import copy
import json
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_forms.json")))


# ------------------------------------------------------------------ canonical polynomial algebra
class Poly(dict):
    """{monomial (sorted tuple of factor names): coefficient}"""

    @staticmethod
    def const(c):
        return Poly({(): float(c)})

    @staticmethod
    def sym(name):
        return Poly({(name,): 1.0})

    def __add__(self, o):
        r = Poly(self)
        for k, v in _p(o).items():
            r[k] = r.get(k, 0.0) + v
        return r

    def __neg__(self):
        return Poly({k: -v for k, v in self.items()})

    def __sub__(self, o):
        return self + (-_p(o))

    def __mul__(self, o):
        r = Poly()
        for k1, v1 in self.items():
            for k2, v2 in _p(o).items():
                k = tuple(sorted(k1 + k2))
                r[k] = r.get(k, 0.0) + v1 * v2
        return r

    def scale(self, c):
        return Poly({k: v * c for k, v in self.items()})

    def wrap(self, fn):
        """apply a linear operator to the single field factor of every monomial (grad, div, sym...)"""
        r = Poly()
        for k, v in self.items():
            fields = [f for f in k if not f.startswith("c:")]
            assert len(fields) == 1, (fn, k)
            nk = tuple(sorted([f for f in k if f.startswith("c:")] + ["%s %s" % (fn, fields[0])]))
            r[nk] = r.get(nk, 0.0) + v
        return r


def _p(x):
    return x if isinstance(x, Poly) else Poly.const(x)


def _env():
    def Constant(v):
        if isinstance(v, Poly):
            return v
        return Poly.const(v)

    def vec(*a):
        vals = [list(x.values())[0] if x else 0.0 for x in map(_p, a)]
        return Poly.sym("c:vec(%s)" % ",".join("%g" % v for v in vals))

    def Expression(code):
        if isinstance(code, str):
            try:
                return Poly.const(float(code))
            except ValueError:
                return Poly.sym("c:expr(%s)" % code)
        return Poly.sym("c:expr(%s)" % ",".join(code))

    env = dict(mul=lambda a, b: _p(a) * _p(b), add=lambda a, b: _p(a) + _p(b), sub=lambda a, b: _p(a) - _p(b),
               neg=lambda a: -_p(a), inner=lambda a, b: _p(a) * _p(b), dot=lambda a, b: _p(a) * _p(b),
               div=lambda *a: (_p(a[0]).wrap("div") if len(a) == 1 else _p(a[0]).scale(1.0 / list(_p(a[1]).values())[0])),
               grad=lambda a: _p(a).wrap("grad"), sym=lambda a: _p(a).wrap("sym"), Identity=lambda n: Poly.sym("c:I"),
               Constant=Constant, vec=vec, Expression=Expression, u_trial=Poly.sym("u_trial"),
               v_test=Poly.sym("v_test"), w_prev=Poly.sym("w_prev"), n=Poly.sym("c:n"))
    return env


def golden_poly(solve):
    """{(measure, monomial): coefficient} of a recorded reference form."""
    out = {}
    for t in solve["terms"]:
        src = re.sub(r"\bw\d+\b", "w_prev", t["integrand"])
        src = re.sub(r"copy\(w_prev\)", "w_prev", src)
        poly = eval(src, {"__builtins__": {}}, _env())
        for mono, c in _p(poly).items():
            key = (t["measure"], mono)
            out[key] = out.get(key, 0.0) + t["sign"] * c
    return {k: v for k, v in out.items() if v != 0.0}


def assert_same_poly(got, ref):
    assert set(got) == set(ref), (sorted(set(got) ^ set(ref)), got, ref)
    for k in ref:
        assert math.isclose(got[k], ref[k], rel_tol=1e-12), (k, got[k], ref[k])


# ------------------------------------------------------------------ fenicssolver_amd forms -> the same polynomial
def scalar_form_poly(F):
    d = F.describe()
    out = collections.defaultdict(float)
    kind, k = d["conductivity"][0], d["conductivity"][1]
    assert kind == "const"
    theta = d["theta"] if d["transient"] else 1.0
    out[("dx", ("grad u_trial", "grad v_test"))] += theta * k
    if d["transient"]:
        c = d["capacity"][1]
        out[("dx", ("u_trial", "v_test"))] += c / d["dt"]
        out[("dx", ("v_test", "w_prev"))] += -c / d["dt"]
        out[("dx", ("grad v_test", "grad w_prev"))] += (1.0 - theta) * k
    for (i, g, _origin) in d["facet_loads"]:
        out[("ds(%d)" % i, ("v_test",))] += -g
    for (i, h, ta) in d["robin"]:           # - h (Ta - T) q ds
        out[("ds(%d)" % i, ("v_test",))] += -h * ta
        out[("ds(%d)" % i, ("u_trial", "v_test"))] += h
    for s in d["sources"]:
        assert s[0] == "const"
        out[("dx", ("v_test",))] += -s[1]
    if d["advection"] is not None:
        v, cap = d["advection"]
        out[("dx", tuple(sorted(["c:vec(%s)" % ",".join("%g" % x for x in v), "grad u_trial", "v_test"])))] += cap
    return {k: v for k, v in out.items() if v != 0.0}


def elasticity_form_poly(F, mu2_name="sym grad u_trial", body_symbol="c:expr(10*rho,0,0.0)"):
    d = F.describe()
    out = collections.defaultdict(float)
    out[("dx", tuple(sorted([mu2_name, "grad v_test"])))] += 2.0 * d["mu"]
    out[("dx", tuple(sorted(["c:I", "div u_trial", "grad v_test"])))] += d["lambda"]
    sgn = -d["load_sign"]            # F = a(u,v) + sum(loads) in the reference  <=>  load_sign = -1
    if d["body_force"] is not None:
        out[("dx", tuple(sorted([body_symbol, "v_test"])))] += sgn * 1.0
    for (i, g, origin) in d["tractions"]:
        out[("ds(%d)" % i, tuple(sorted(["c:vec(%s)" % ",".join("%g" % x for x in g), "v_test"])))] += sgn * 1.0
    if d["thermal"] is not None:
        coef, T, T_ref = d["thermal"]
        out[("dx", tuple(sorted(["c:I", "grad v_test"])))] += -coef * (T - T_ref)
    return {k: v for k, v in out.items() if v != 0.0}


def bc_list(bcs):
    out = []
    for b in bcs:
        V = b.function_space
        space = "V" if V.component() is None else "V.sub(%d)" % V.component()
        vals = np.asarray(b.value.values() if hasattr(b.value, "values") else b.value, dtype=float).ravel()
        out.append((space, tuple(vals.tolist()), b.marker_id))
    return out


def golden_bcs(solve):
    out = []
    for b in solve["bcs"]:
        nums = tuple(float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", b["value"].replace("vec", "")))
        out.append((b["space"], nums, b["marker"]))
    return out


# ------------------------------------------------------------------ the cases (same settings as the golden script)
QUIET = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}


def _form_of(solver):
    solver.init_solver()
    solver.current_step = 0
    return solver.generate_form(0, None, None, solver.w_current, solver.w_prev)


def _heat_settings(transient=False, **extra):
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    mesh = UnitCubeMesh(4, 4, 4)
    Q = FunctionSpace(mesh, "CG", 1)
    bcs = collections.OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'heatFlux', 'value': Constant(36.0)}}}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}}}
    bcs["left"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 3, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'symmetry', 'value': None}}}
    st = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
          'boundary_conditions': bcs, 'body_source': 5.0, 'initial_values': {'temperature': 300},
          'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.1},
          'solver_settings': {'transient_settings': {'transient': transient, 'starting_time': 0, 'time_step': 0.1,
                                                     'ending_time': 0.1},
                              'reference_values': {'temperature': 300}, 'solver_parameters': {}},
          'report_settings': dict(QUIET), 'scalar_name': 'temperature'}
    st.update(extra)
    return st


def test_config1_json_terms(data_dir):
    from fenicssolver_amd.main import load_settings
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    s = load_settings(os.path.join(data_dir, "TestHeatTransfer.json"))
    s["mesh"] = os.path.join(data_dir, "mesh.xml")
    s["report_settings"] = dict(QUIET)
    F, bcs = _form_of(ScalarTransportSolver(s))
    g = GOLD["config1_json"]["solves"][0]
    assert g["kind"] == "LinearVariationalSolver"
    assert_same_poly(scalar_form_poly(F), golden_poly(g))
    assert bc_list(bcs) == golden_bcs(g)


@pytest.mark.parametrize("case,kw", [("heat_flux_htc_source", {}), ("heat_transient", {"transient": True}),
                                     ("heat_convection", {"convective_velocity": "const"})])
def test_heat_cases_terms(case, kw):
    from fenicssolver_amd.fem import Constant
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    kw = dict(kw)
    if kw.get("convective_velocity") == "const":
        kw["convective_velocity"] = Constant((0.005, -0.005, 0.0))
    solver = ScalarTransportSolver(_heat_settings(**kw))
    solver.material['conductivity'] = 0.6
    F, bcs = _form_of(solver)
    g = GOLD[case]["solves"][0]
    assert_same_poly(scalar_form_poly(F), golden_poly(g))
    assert bc_list(bcs) == golden_bcs(g) == []


def test_dirichlet_neumann_robin_terms():
    from fenicssolver_amd.fem import Constant
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    st = _heat_settings()
    st['body_source'] = None
    b = st['boundary_conditions']
    b["hot"]['values']['temperature'] = {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}
    b["cold"]['values']['temperature'] = {'variable': 'temperature', 'type': 'fixedGradient', 'value': Constant(2.0)}
    b["left"]['values']['temperature'] = {'variable': 'temperature', 'type': 'Robin', 'value': Constant(310),
                                          'gradient': Constant(1.5)}
    F, bcs = _form_of(ScalarTransportSolver(st))
    g = GOLD["heat_dirichlet_neumann_robin"]["solves"][0]
    assert_same_poly(scalar_form_poly(F), golden_poly(g))     # capacity-scaled Neumann term (Appendix B-Q8)
    assert bc_list(bcs) == golden_bcs(g)                       # same Dirichlet sets, same order


def _elasticity_solver(bcs, **extra):
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 8, 2, 2)
    st = copy.deepcopy(SB.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = None
    st.update(extra)
    return LinearElasticitySolver(st)


def _sides():
    from fenicssolver_amd.fem import SubDomain, near

    class Left(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0)

    class Right(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 10)
    return Left(), Right()


def test_elasticity_terms():
    from fenicssolver_amd.fem import Constant, Expression
    left, right = _sides()
    bf = Expression(("10*rho", "0", "0.0"), rho=7800, omega=100, degree=2)
    # stress + body force + thermal stress, per-component clamp
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': left, 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
    bcs["tensile"] = {'boundary': right, 'boundary_id': 2, 'type': 'stress', 'value': Constant((1e8, 0, 0))}
    F, dbc = _form_of(_elasticity_solver(bcs, body_source=bf, temperature_distribution=Expression("343", degree=1)))
    g = GOLD["elasticity_stress_body_thermal"]["solves"][0]
    assert g["kind"].startswith("assemble_system") and g["krylov"]["method"] == "cg" and g["krylov"]["pc"] == "petsc_amg"
    assert_same_poly(elasticity_form_poly(F), golden_poly(g))     # loads ADDED to F (B-Q3), thermal conventional
    assert bc_list(dbc) == golden_bcs(g)
    assert F.body_force == (78000.0, 0.0, 0.0)
    assert GOLD["elasticity_stress_body_thermal"]["nullspace_vectors"] == 6
    # prescribed displacement
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': left, 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}
    bcs["displ"] = {'boundary': right, 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant((0, 0, 1e-3))}
    F, dbc = _form_of(_elasticity_solver(bcs, body_source=bf))
    g = GOLD["elasticity_displacement"]["solves"][0]
    assert_same_poly(elasticity_form_poly(F), golden_poly(g))
    assert bc_list(dbc) == golden_bcs(g)


def test_elasticity_2d_goes_through_solve_linear_problem():
    """A plane-strain case: the reference solves 2-D problems with solve_linear_problem (LinearVariationalSolver), not
    solve_amg (LinearElasticitySolver.py:247-253); same integrals, per-component Dirichlet sets, Identity(2)."""
    from fenicssolver_amd.fem import RectangleMesh, Point, VectorFunctionSpace, Constant, SubDomain, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver

    class Left(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0)

    class Right(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 4)

    class Top(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[1], 1)
    mesh = RectangleMesh(Point(0, 0), Point(4, 1), 8, 2)
    st = copy.deepcopy(SB.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), Constant(0))}
    bcs["roller"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'displacement', 'value': (Constant(1e-3), None)}
    bcs["load"] = {'boundary': Top(), 'boundary_id': 3, 'type': 'stress', 'value': Constant((0, -5e6))}
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = None
    st['body_source'] = Constant((0, -76440.0))
    solver = LinearElasticitySolver(st)
    assert solver.dimension == 2 and solver.function_space.ufl_element().value_size() == 2
    F, dbc = _form_of(solver)
    g = GOLD["elasticity_2d"]["solves"][0]
    assert g["kind"] == "LinearVariationalSolver" and "Identity(2)" in g["terms"][0]["integrand"]
    assert_same_poly(elasticity_form_poly(F, body_symbol="c:vec(0,-76440)"), golden_poly(g))
    assert bc_list(dbc) == golden_bcs(g)
    # the class takes the same branch: solve_form -> solve_linear_problem for dimension 2
    called = []
    solver.solve_linear_problem = lambda F_, u_, b_: called.append("linear") or u_
    solver.solve_amg = lambda F_, u_, b_: called.append("amg") or u_
    solver.solve_form(F, solver.w_current, dbc)
    assert called == ["linear"]


def test_reference_elasticity_example_on_vector_p2():
    """examples/test_linear_elasticity.py as its __main__ runs it: BoxMesh 40x10x10, VectorFunctionSpace(mesh, 'Lagrange', 2),
    left face clamped in x only, right face displaced by (0, 0, 1e-3), body force, thermal stress at 343 K."""
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, Constant, Expression
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    left, right = _sides()
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 40, 10, 10)
    st = copy.deepcopy(SB.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    V = VectorFunctionSpace(mesh, "Lagrange", 2)
    st['function_space'] = V
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': left, 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
    bcs["displ"] = {'boundary': right, 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant((0, 0, 1 * 1e-3))}
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = Expression("343", degree=2)
    st['body_source'] = Expression(("10*rho", "0", "0.0"), omega=100, rho=7800, degree=2)
    solver = LinearElasticitySolver(st)
    gold = GOLD["elasticity_example_p2"]
    assert gold["function_space"]["degree"] == V.degree() == 2 and solver.function_space.ufl_element().value_size() == 3
    F, dbc = _form_of(solver)
    g = gold["solves"][0]
    assert_same_poly(elasticity_form_poly(F), golden_poly(g))
    assert bc_list(dbc) == golden_bcs(g)
    # topological Dirichlet sets on P2: vertices AND edge midpoints of the marked faces (Appendix D-3)
    X = V.node_coordinates()
    nv = mesh.num_vertices()
    fixed, displ = dbc
    assert len(fixed.dofs) == 21 * 21 and np.all(fixed.dofs % 3 == 0) and np.allclose(X[fixed.dofs // 3, 0], 0.0)
    assert np.count_nonzero(fixed.dofs // 3 >= nv) == 21 * 21 - 11 * 11
    assert len(displ.dofs) == 3 * 21 * 21 and np.allclose(X[displ.dofs // 3, 0], 10.0)
    assert np.array_equal(displ.values.reshape(-1, 3), np.tile([0.0, 0.0, 1e-3], (21 * 21, 1)))


# ------------------------------------------------------------------ Taylor-Hood Navier-Stokes
def _ns_expected_terms(desc, state="W0", prev="WPREV", temperature_law=None):
    """The integrals the reference builds for a NavierStokesForm description (CoupledNavierStokesSolver.py:315-381),
    in the recording stub's notation with the state / previous-step functions replaced by placeholders.
    temperature_law: the ('pT', p_ref, c_p, T_ref, c_T) tuple of CoupledNavierStokesSolver.temperature_law() - nu(p, T) is attached to
    the device space, not to the form description."""
    eps = lambda f: "mul(0.5, add(grad(%s), transpose(grad(%s))))" % (f, f)   # noqa: E731
    num = lambda x: repr(float(x)) if not float(x).is_integer() else "%d" % x  # noqa: E731
    law = desc.get("viscosity_law")
    nu_expr = None if not law else "mul(%s, pow(div(%s[1], %s), %s))" % (num(desc["nu"]), state, num(law[0]), num(law[1]))
    if temperature_law is not None:
        assert not law and temperature_law[0] == 'pT'
        _, pref, cp, tref, ct = temperature_law
        nu_expr = "mul(mul(%s, add(1, mul(div(%s[1], %s), %s))), sub(1, mul(div(%s[2], %s), %s)))" % (
            num(desc["nu"]), state, num(pref), num(cp), state, num(tref), num(ct))
        law = temperature_law
    visc = "mul(%s, inner(%s, %s))" % (num(desc["nu"] * 2.0) if not law else "mul(%s, 2)" % nu_expr, eps("u_trial[0]"), eps("v_test[0]"))
    t = [(+1, visc),
         (-1, "mul(div(u_trial[1], %s), div(v_test[0]))" % num(desc["rho"])),
         (+1, "mul(div(u_trial[0]), div(v_test[1], %s))" % num(desc["rho"]))]
    if desc["body_force"] is not None:
        t.append((-1, "inner(Constant(vec(%s)), v_test[0])" % ", ".join(num(x) for x in desc["body_force"])))
    t.append((+1, "inner(dot(grad(u_trial[0]), %s[0]), v_test[0])" % state))
    if desc["inv_dt"]:
        t.append((+1, "mul(%s, inner(sub(u_trial[0], %s[0]), v_test[0]))" % (num(desc["inv_dt"]), prev)))
    if desc.get("g2"):
        mode, kappa1 = desc["g2"]
        h = "mul(2, Circumradius)"
        stream = "inner(dot(%s[0], grad(u_trial[0])), dot(%s[0], grad(v_test[0])))" % (state, state)
        if mode == 1:          # Re <= 1: delta1 = kappa1 h h
            t.append((-1, "mul(mul(mul(%s, %s), %s), %s)" % (num(kappa1), h, h, stream)))
        else:                  # steady, convection dominated: delta1 = kappa1/2 h / sqrt(a.a)
            t.append((-1, "mul(div(mul(%s, %s), sqrt(dot(%s[0], %s[0]))), %s)" % (num(kappa1 / 2.0), h, state, state, stream)))
    t = [(sg, body, "dx") for sg, body in t]
    for marker, value in desc.get("pressure_boundaries", []):
        if value is not None:
            t.append((+1, "inner(mul(%s, n), v_test[0])" % value, "ds(%d)" % marker))
        t.append((+1, "mul(%s, inner(mul(add(grad(u_trial[0]), transpose(grad(u_trial[0]))), n), v_test[0]))"
                  % (num(-desc["nu"]) if not law else "neg(%s)" % nu_expr), "ds(%d)" % marker))
    return t


@pytest.mark.parametrize("case,transient,body", [("navier_stokes_steady", False, None),
                                                 ("navier_stokes_transient_gravity", True, (0, 0, -9.8)),
                                                 ("navier_stokes_pressure_boundaries", False, None),
                                                 ("navier_stokes_g2_steady", False, None),
                                                 ("navier_stokes_g2_low_re", False, None)])
def test_navier_stokes_terms(case, transient, body):
    """Same settings -> the same integrals (signs, the 2*nu, the 1/rho on both pressure terms, gravity without rho,
    backward Euler) and the same Dirichlet conditions on W.sub(0) as the reference hands to NonlinearVariationalSolver."""
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    mesh = UnitCubeMesh(2, 2, 2)
    bcs = collections.OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[2] - 1) < 1e-12), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    pressure = case.endswith("pressure_boundaries")
    if pressure:
        bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[0] - 1) < 1e-12), 'boundary_id': 3,
                         'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(5.0)}]}
        bcs["far"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[0]) < 1e-12), 'boundary_id': 4,
                      'values': [{'variable': "pressure", 'type': 'farfield', 'value': Constant(0.0)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG',
              'boundary_conditions': bcs, 'body_source': Constant(body) if body else None,
              'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 2.0, 'kinematic_viscosity': 0.01}})
    s['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01,
                                                  'ending_time': 0.01}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['report_settings'] = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    if "g2" in case:           # the G2 streamline term (CoupledNavierStokesSolver.py:334-363), subtracted as the reference does
        s['advection_settings'] = {'stabilization_method': 'G2', 'Re': 0.5 if case.endswith("low_re") else 100, 'kappa1': 4, 'kappa2': 2}
    solver = CoupledNavierStokesSolver(s)
    solver.init_solver()
    F, dbcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    desc = F.describe()
    assert desc["g2"] == ([1 if case.endswith("low_re") else 2, 4.0] if "g2" in case else None)
    assert desc["newton"] is True              # using_nonlinear_solver: action(F, w) + derivative (:241-243)

    gold = GOLD[case]["solves"][0]
    assert gold["kind"] == "NonlinearVariationalSolver"
    # strip the action(..., w_current) wrapper and name the state / previous-step functions
    terms = []
    state = None
    for t in gold["terms"]:
        m = re.match(r"^action\((.*), (interpolate\(Expression\(.*?\)\)\))\)$", t["integrand"])
        assert m
        state = state or m.group(2)
        assert m.group(2) == state
        body_ = m.group(1).replace(state, "W0")
        body_ = re.sub(r"\bw\d+\b", "WPREV", body_)
        terms.append((t["sign"], body_, t["measure"]))
    assert sorted(terms) == sorted(_ns_expected_terms(desc))
    # Dirichlet conditions: velocity sub space (and the pressure sub space for a pressure outlet), same order and values
    expect_bcs = [("W.sub(0)", 1), ("W.sub(0)", 2)] + ([("W.sub(1)", 3)] if pressure else [])
    assert [(b["space"], b["marker"]) for b in gold["bcs"]] == expect_bcs
    assert [b.marker_id for b in dbcs] == [m_ for _, m_ in expect_bcs]
    if pressure:
        assert np.all(dbcs[2].dofs % 4 == 3) and np.all(dbcs[2].values == 5.0)
    assert np.all(dbcs[0].values == 0.0)
    v2 = dbcs[1].values.reshape(-1, 3)
    assert np.all(v2[:, 0] == 1.0) and np.all(v2[:, 1:] == 0.0)
    assert np.all(dbcs[1].dofs % 4 != 3)       # velocity components only


@pytest.mark.parametrize("case,transient,body", [("navier_stokes_2d_steady", False, None),
                                                 ("navier_stokes_2d_transient_gravity", True, (0, -9.8))])
def test_navier_stokes_2d_terms(case, transient, body):
    """The 2-D set-up of the reference's own CFD example (examples/test_cfd_solver.py:83-170: UnitSquareMesh channel, no-slip side
    walls, velocity inlet, pressure outlet): the reference - imported unchanged on the recording stub - and this package build
    the same integrals and the same Dirichlet conditions; the class is dimension-free upstream (:84-102)."""
    from fenicssolver_amd.fem import UnitSquareMesh, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    mesh = UnitSquareMesh(4, 6)
    bcs = collections.OrderedDict()
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[1], 1.0)), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(2.0)}]}
    bcs["static"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and (near(x[0], 0.0) or near(x[0], 1.0))), 'boundary_id': 1,
                     'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0))}]}
    bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[1], 0.0)), 'boundary_id': 2,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 1))}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG',
              'boundary_conditions': bcs, 'body_source': Constant(body) if body else None,
              'initial_values': {'velocity': (0, 0.2), 'pressure': 0},
              'material': {'density': 1.5, 'kinematic_viscosity': 0.1}})
    s['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01, 'ending_time': 0.01}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1), 'pressure': 0}
    s['report_settings'] = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    solver.init_solver()
    assert solver.dimension == 2 and solver.function_space.velocity_dim() == 2
    F, dbcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    desc = F.describe()
    gold = GOLD[case]["solves"][0]
    assert gold["kind"] == "NonlinearVariationalSolver"
    terms, state = [], None
    for t in gold["terms"]:
        m = re.match(r"^action\((.*), (interpolate\(Expression\(.*?\)\)\))\)$", t["integrand"])
        assert m
        state = state or m.group(2)
        body_ = re.sub(r"\bw\d+\b", "WPREV", m.group(1).replace(state, "W0"))
        terms.append((t["sign"], body_, t["measure"]))
    assert sorted(terms) == sorted(_ns_expected_terms(desc))
    assert [(b["space"], b["marker"]) for b in gold["bcs"]] == [("W.sub(1)", 3), ("W.sub(0)", 1), ("W.sub(0)", 2)]
    assert [b.marker_id for b in dbcs] == [3, 1, 2]
    # Dirichlet sets in the block-4 layout: pressure slots of the outlet vertices; both velocity components (never the dummy third
    # slot) of the wall / inlet nodes, vertices and edge mid-points
    X = solver.function_space.node_coordinates()
    assert np.all(dbcs[0].dofs % 4 == 3) and np.all(dbcs[0].values == 2.0) and np.allclose(X[dbcs[0].dofs // 4, 1], 1.0)
    assert set(np.unique(dbcs[1].dofs % 4)) == {0, 1} and np.all(dbcs[1].values == 0.0)
    v2 = dbcs[2].values.reshape(-1, 2)
    assert np.all(v2[:, 0] == 0.0) and np.all(v2[:, 1] == 1.0) and np.allclose(X[dbcs[2].dofs[::2] // 4, 1], 0.0)
    assert len(dbcs[2].dofs) == 2 * (5 + 4)          # 5 vertices + 4 edge mid-points on the inlet
    # the initial field the reference interpolates (Expression(('0', '0.2', '0'))): velocity (0, 0.2), pressure 0
    a = solver.w_current.vector().get_local().reshape(-1, 4)
    assert np.all(a[:, 0] == 0.0) and np.all(a[:, 1] == 0.2) and np.all(a[:, 2:] == 0.0)


def test_navier_stokes_non_newtonian_terms():
    """material['Newtonian'] = False (CoupledNavierStokesSolver.viscosity :194-213): the reference multiplies nu by
    pow(p / reference pressure, 0.1) with the pressure of the CURRENT iterate, in the cell term (:306) and in the boundary
    term of a pressure outlet (:401).  The form description of this package says the same."""
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    mesh = UnitCubeMesh(2, 2, 2)
    bcs = collections.OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[2] - 1) < 1e-12), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[0] - 1) < 1e-12), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(1.0e5)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG',
              'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 1.0e5},
              'material': {'density': 2.0, 'kinematic_viscosity': 0.01, 'Newtonian': False}})
    s['solver_settings']['transient_settings'] = {'transient': False, 'starting_time': 0.0, 'time_step': 0.01, 'ending_time': 0.01}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 1.0e5}
    s['report_settings'] = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    solver.init_solver()
    F, dbcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    desc = F.describe()
    assert desc["viscosity_law"] == [1.0e5, 0.1] and desc["nu"] == 0.01
    gold = GOLD["navier_stokes_non_newtonian"]["solves"][0]
    terms, state = [], None
    for t in gold["terms"]:
        m = re.match(r"^action\((.*), (interpolate\(Expression\(.*?\)\)\))\)$", t["integrand"])
        assert m
        state = state or m.group(2)
        assert m.group(2) == state
        terms.append((t["sign"], m.group(1).replace(state, "W0"), t["measure"]))
    assert sorted(terms) == sorted(_ns_expected_terms(desc))
    assert [(b["space"], b["marker"]) for b in gold["bcs"]] == [("W.sub(0)", 1), ("W.sub(0)", 2), ("W.sub(1)", 3)]
    assert [b.marker_id for b in dbcs] == [1, 2, 3]


def test_navier_stokes_coupled_temperature_terms():
    """solving_temperature (CoupledNavierStokesSolver.py:236-239, 247-286): the reference adds, on W.sub(2), the conduction
    term, the convection by the CURRENT velocity iterate times the capacity rho*cp, and the interior-penalty term with
    alpha = 0.1 (times the capacity), with the temperature Dirichlet sets; no viscous heating, no source.  The temperature
    form of this package says the same, and the flow terms are those of the uncoupled case."""
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    gold = GOLD["navier_stokes_coupled_temperature"]["solves"][0]
    state = None
    terms = []
    for t in gold["terms"]:
        m = re.match(r"^action\((.*), (interpolate\(Expression\(.*?\)\)\))\)$", t["integrand"])
        assert m
        state = state or m.group(2)
        terms.append((t["sign"], m.group(1).replace(state, "W0"), t["measure"]))
    thermal = sorted(x for x in terms if "[2]" in x[1])
    assert thermal == sorted([
        (+1, "inner(mul(0.1, grad(u_trial[2])), grad(v_test[2]))", "dx"),
        (+1, "mul(mul(inner(W0[0], grad(u_trial[2])), v_test[2]), 6)", "dx"),
        (+1, "mul(mul(mul(Constant(0.1), pow(avg(mul(2, Circumradius)), 2)), inner(jump(grad(u_trial[2]), n), jump(grad(v_test[2]), n))), 6)", "dS")])
    assert [(b["space"], b["marker"], b["value"]) for b in gold["bcs"] if b["space"] == "W.sub(2)"] == \
        [("W.sub(2)", 1, "Constant(350)"), ("W.sub(2)", 2, "Constant(300)")]

    mesh = UnitCubeMesh(2, 2, 2)
    bcs = collections.OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))},
                               {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(350)}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[2] - 1) < 1e-12), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))},
                             {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(300)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG', 'solving_temperature': True,
              'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0, 'temperature': 320},
              'material': {'density': 2.0, 'kinematic_viscosity': 0.01, 'specific_heat_capacity': 3.0, 'thermal_conductivity': 0.1}})
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0, 'temperature': 300}
    s['report_settings'] = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    solver.init_solver()
    F, dbcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    flow = [x for x in terms if "[2]" not in x[1]]
    assert sorted(flow) == sorted(_ns_expected_terms(F.describe()))
    FT, tbcs = solver.generate_thermal_form(0, None, None, solver.w_current, solver.w_prev)
    d = FT.describe()
    assert d["conductivity"] == ("const", 0.1) or d["conductivity"] == 0.1 or tuple(d["conductivity"])[-1] == 0.1
    assert d["ip_coefficient"] == pytest.approx(0.1 * 6.0) and FT.advection[1] == pytest.approx(6.0)
    assert np.ndim(FT.advection[0]) == 3                     # one velocity per cell and test function: the P2 iterate, exactly
    assert [b.marker_id for b in tbcs] == [1, 2] and np.all(tbcs[0].values == 350.0) and np.all(tbcs[1].values == 300.0)
    assert not FT.transient and FT.source is None if hasattr(FT, "source") else True


def test_navier_stokes_non_newtonian_with_temperature_terms():
    """material['Newtonian'] = False together with solving_temperature (CoupledNavierStokesSolver.viscosity :199-203): the
    reference multiplies nu by (1 + (p / p_ref) 0.1) (1 - (T / T_ref) 0.2) with the pressure and the temperature of the CURRENT
    iterate, in the cell term (:306) and in the boundary term of a pressure outlet (:401).  Held to the golden recorded from the
    imported reference: temperature_law() of this package (what fs_space_set_viscosity_law hands to the kernels), the flow terms
    around it, the thermal terms, every Dirichlet set - and the oracle's viscosity_at(), which the -m gpu tests hold the device
    law to, evaluated against the golden's own expression."""
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    from oracle import ns_oracle
    gold = GOLD["navier_stokes_non_newtonian_temperature"]["solves"][0]
    assert gold["kind"] == "NonlinearVariationalSolver"
    terms, state = [], None
    for t in gold["terms"]:
        m = re.match(r"^action\((.*), (interpolate\(Expression\(.*?\)\)\))\)$", t["integrand"])
        assert m
        state = state or m.group(2)
        assert m.group(2) == state
        terms.append((t["sign"], m.group(1).replace(state, "W0"), t["measure"]))

    mesh = UnitCubeMesh(2, 2, 2)
    bcs = collections.OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))},
                               {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(350)}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[2] - 1) < 1e-12), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))},
                             {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(300)}]}
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and abs(x[0] - 1) < 1e-12), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(1.0e5)},
                                {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(330)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG', 'solving_temperature': True,
              'boundary_conditions': bcs, 'body_source': None,
              'initial_values': {'velocity': (0, 0, 0), 'pressure': 1.0e5, 'temperature': 320},
              'material': {'density': 2.0, 'kinematic_viscosity': 0.01, 'specific_heat_capacity': 3.0, 'thermal_conductivity': 0.1,
                           'Newtonian': False}})
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 1.0e5, 'temperature': 300}
    s['report_settings'] = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    solver.init_solver()
    law = solver.temperature_law()
    assert law == ('pT', 1.0e5, 0.1, 300.0, 0.2) and solver.viscosity_law() is None
    # (the form is described without touching a device: the law travels with the space, _attach_temperature_law)
    F = solver._cell_form(0, solver.w_current, solver.w_prev)
    dbcs, F.pressure_boundaries = solver.update_boundary_conditions(0, None, None, None)
    desc = F.describe()
    assert desc["viscosity_law"] is None and desc["nu"] == 0.01
    flow = [x for x in terms if "[2]" not in x[1].replace("W0[2]", "")]
    assert sorted(flow) == sorted(_ns_expected_terms(desc, temperature_law=law))
    assert sum("W0[2]" in x[1] for x in flow) == 2                       # the temperature enters the cell term and the outlet term
    thermal = sorted(x for x in terms if x not in flow)
    assert thermal == sorted([
        (+1, "inner(mul(0.1, grad(u_trial[2])), grad(v_test[2]))", "dx"),
        (+1, "mul(mul(inner(W0[0], grad(u_trial[2])), v_test[2]), 6)", "dx"),
        (+1, "mul(mul(mul(Constant(0.1), pow(avg(mul(2, Circumradius)), 2)), inner(jump(grad(u_trial[2]), n), jump(grad(v_test[2]), n))), 6)", "dS")])
    assert [(b["space"], b["marker"]) for b in gold["bcs"]] == [("W.sub(0)", 1), ("W.sub(0)", 2), ("W.sub(1)", 3),
                                                                 ("W.sub(2)", 1), ("W.sub(2)", 2), ("W.sub(2)", 3)]
    assert [b.marker_id for b in dbcs] == [1, 2, 3] and np.all(dbcs[2].dofs % 4 == 3) and np.all(dbcs[2].values == 1.0e5)
    FT, tbcs = solver.generate_thermal_form(0, None, None, solver.w_current, solver.w_prev)
    assert [b.marker_id for b in tbcs] == [1, 2, 3] and [float(b.values[0]) for b in tbcs] == [350.0, 300.0, 330.0]
    assert FT.describe()["ip_coefficient"] == pytest.approx(0.1 * 6.0) and FT.advection[1] == pytest.approx(6.0)

    # the golden's viscosity, evaluated as the stub wrote it down, against the oracle's law (to which the device law is held)
    visc = next(x[1] for x in flow if x[2] == "dx" and "W0[2]" in x[1])
    nu_txt = visc[len("mul(mul("):visc.index(", 2), inner(")]
    ops = {"mul": lambda a, b: a * b, "div": lambda a, b: a / b, "add": lambda a, b: a + b, "sub": lambda a, b: a - b}
    rng = np.random.default_rng(5)
    for p_, T_ in zip(rng.uniform(0.5e5, 2e5, 8), rng.uniform(250.0, 400.0, 8)):
        val = eval(nu_txt.replace("W0[1]", repr(float(p_))).replace("W0[2]", repr(float(T_))), {"__builtins__": {}}, ops)
        assert val == pytest.approx(float(ns_oracle.viscosity_at(0.01, law, p_, T_)), rel=1e-15)
        assert val == pytest.approx(0.01 * (1 + 0.1 * p_ / 1e5) * (1 - 0.2 * T_ / 300.0), rel=1e-14)


def test_reference_g2_transient_branch_is_broken_upstream():
    """F_static reads an undefined time_iter_ in the transient, convection-dominated G2 branch (:354-355): the reference
    raises NameError there.  The GPU path uses the step's dt in that formula (the evident intent) - see fs_ns_form.g2_mode."""
    assert GOLD["navier_stokes_g2_transient"]["reference_raises"].startswith("NameError")


# ------------------------------------------------------------------ SUPG ("SPUG")
@pytest.mark.parametrize("transient", [False, True])
def test_supg_is_the_plain_form_with_the_test_function_replaced(transient):
    """What the reference builds for advection_settings = {'stabilization_method': 'SPUG', 'Pe': 10} is, integral by
    integral, the unstabilised form with every test function q replaced by
        q + tau (v . grad q),   tau = 0.5 h (4/(Pe h) + 2 |v|)^-1,   h = 2 Circumradius
    (ScalarTransportSolver.py:259-270, SPUG_method == 2) - volume, source AND boundary integrals.  That substitution is
    what forms.ScalarForm.supg_pe stands for (kernels: fs_assemble.hip supg_tau / k_facet_supg; oracle: supg_weights)."""
    from fenicssolver_amd.fem import Constant
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    vel = "Constant(vec(0.005, -0.005, 0))"
    h = "mul(2, Circumradius)"
    tau = "mul(mul(0.5, %s), pow(add(div(4, mul(10, %s)), mul(2, sqrt(dot(%s, %s)))), -1))" % (h, h, vel, vel)
    tq = "add(v_test, mul(%s, inner(%s, grad(v_test))))" % (tau, vel)
    g = GOLD["heat_convection_supg_transient" if transient else "heat_convection_supg"]["solves"][0]
    stripped = []
    for t in g["terms"]:
        assert tq in t["integrand"], t["integrand"]              # every integral carries the modified test function
        stripped.append({"sign": t["sign"], "integrand": t["integrand"].replace(tq, "v_test"), "measure": t["measure"]})
        assert "Circumradius" not in stripped[-1]["integrand"]   # ... and nothing else of the stabilisation
    # our side: the same settings give the plain convection form + supg_pe = Pe
    kw = {"convective_velocity": Constant((0.005, -0.005, 0.0)), "advection_settings": {'stabilization_method': 'SPUG', 'Pe': 10.0}}
    if transient:
        kw["transient"] = True
    solver = ScalarTransportSolver(_heat_settings(**kw))
    solver.material['conductivity'] = 0.6
    F, bcs = _form_of(solver)
    assert F.supg_pe == 10.0 and F.describe()["supg_pe"] == 10.0
    assert_same_poly(scalar_form_poly(F), golden_poly({"terms": stripped}))
    assert bc_list(bcs) == golden_bcs(g) == []


# ------------------------------------------------------------------ interior penalty ("IP")
def test_ip_is_the_convection_form_plus_one_interior_facet_integral():
    """advection_settings = {'stabilization_method': 'IP', 'alpha': 0.1}: the reference adds exactly one integral,
        alpha avg(h)^2 inner(jump(grad T, n), jump(grad q, n)) capacity dS,   h = 2 Circumradius
    (ScalarTransportSolver.py:312-315), to the unstabilised convection form.  forms.ScalarForm.ip_coefficient =
    alpha * capacity stands for it (kernel: fs_assemble.hip k_interior_penalty; oracle: assemble_interior_penalty)."""
    from fenicssolver_amd.fem import Constant
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    g = GOLD["heat_convection_ip"]["solves"][0]
    facet = [t for t in g["terms"] if t["measure"] == "dS"]
    rest = [t for t in g["terms"] if t["measure"] != "dS"]
    assert len(facet) == 1 and facet[0]["sign"] == 1
    assert facet[0]["integrand"] == ("mul(mul(mul(Constant(0.1), pow(avg(mul(2, Circumradius)), 2)), "
                                     "inner(jump(grad(u_trial), n), jump(grad(v_test), n))), 4200000)")
    plain = GOLD["heat_convection"]["solves"][0]
    key = lambda ts: [(t["sign"], t["integrand"], t["measure"]) for t in ts]      # noqa: E731
    assert key(rest) == key(plain["terms"])
    kw = {"convective_velocity": Constant((0.005, -0.005, 0.0)), "advection_settings": {'stabilization_method': 'IP', 'alpha': 0.1}}
    solver = ScalarTransportSolver(_heat_settings(**kw))
    solver.material['conductivity'] = 0.6
    F, bcs = _form_of(solver)
    assert F.ip_coefficient == pytest.approx(0.1 * 4200000.0) and F.supg_pe == 0.0
    assert_same_poly(scalar_form_poly(F), golden_poly({"terms": rest}))
