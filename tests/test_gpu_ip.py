"""Interior-penalty stabilisation ('IP', ScalarTransportSolver.py:312-315): the interior-facet term on the device against
the oracle's per-facet restatement, its defining properties, and the solver class against an oracle solve."""
from collections import OrderedDict

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def _mesh(n=3, p1=(1.0, 0.8, 1.3)):
    co, ce = fo.box_mesh((0, 0, 0), p1, n, n + 1, n)
    rng = np.random.default_rng(3)
    interior = np.all((co > 1e-12) & (co < np.array(p1) - 1e-12), axis=1)
    co = co + interior[:, None] * rng.uniform(-0.04, 0.04, co.shape)      # no two cells alike
    return co, ce


def test_interior_penalty_matrix_matches_oracle_and_kills_linear_fields(gpu):
    from fenicssolver_amd.fem import Mesh
    co, ce = _mesh()
    m = Mesh(coords=co, cells=ce)
    ce = m.cells()          # ordered vertex ids (mesh.order())
    fcells, pairs = m.interior_facet_cells()
    _, _, cnt = fo.facet_numbering(ce)
    assert len(fcells) == int((cnt == 2).sum())
    dm = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(dm, 1, 1, coupled_pairs=pairs)
    plain = gpu.DeviceSpace(dm, 1, 1)
    assert V.nnz > plain.nnz                       # the vertices opposite a facet couple although they share no cell
    A = gpu.DeviceMatrix(V)
    A.zero()
    A.add_interior_penalty(fcells, 0.37)
    ref = fo.assemble_interior_penalty(co, ce, 0.37)
    got = _csr(A)
    assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
    assert abs(got - got.T).max() <= 1e-13 * abs(ref).max()
    lin = 3.0 + co @ np.array([0.7, -1.1, 0.4])    # the gradient of a linear field does not jump
    assert np.abs(got @ lin).max() <= 1e-11 * abs(ref).max() * np.abs(lin).max()
    rng = np.random.default_rng(0)
    for _ in range(3):                             # positive semi-definite
        v = rng.standard_normal(len(co))
        assert v @ (got @ v) >= -1e-12 * abs(ref).max()
    # the ordinary operators assemble unchanged on the wider pattern
    A.assemble(stiffness=2.0, mass=0.5)
    K = fo.assemble_p1_scalar(co, ce, 2.0, 0.5)
    assert abs(_csr(A) - K).max() <= 1e-12 * abs(K).max()
    with pytest.raises(gpu.BackendError):          # without the facet couplings in the pattern
        gpu.DeviceMatrix(plain).add_interior_penalty(fcells, 1.0)


def test_ip_stabilised_advection_through_the_solver_class(gpu):
    """Advection-dominated transport across the box (cell Peclet number ~ 40): the solver class with
    advection_settings = {'stabilization_method': 'IP', 'alpha': 0.1} solves (K + C + IP) T = b; compared with the
    oracle's direct solve of the same operators.  The penalty damps the wiggles of the plain Galerkin solution."""
    from fenicssolver_amd.fem import BoxMesh, Point, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    n = 6
    m = BoxMesh(Point(0, 0, 0), Point(1, 1, 1), n, n, n)
    Q = FunctionSpace(m, "CG", 1)
    bcs = OrderedDict()
    bcs["in"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    bcs["out"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 1.0)), 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}

    def make(method):
        s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
             'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
             'material': {'density': 1.0, 'specific_heat_capacity': 1.0, 'thermal_conductivity': 0.002},
             'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                                 'reference_values': {'temperature': 300},
                                 'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
             'convective_velocity': Constant((1.0, 0.0, 0.0)),
             'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0},
             'scalar_name': 'temperature'}
        if method:
            s['advection_settings'] = {'stabilization_method': method, 'alpha': 0.1}
        return ScalarTransportSolver(s)

    T_ip = make('IP').solve().vector().array()
    co, ce = m.coordinates(), m.cells()
    K = fo.assemble_p1_scalar(co, ce, 0.002)
    Cm = fo.assemble_matrix(len(co), ce, fo.p1_advection_local(co, ce, (1.0, 0.0, 0.0), 1.0))
    P = fo.assemble_interior_penalty(co, ce, 0.1 * 1.0)
    lo, hi = np.nonzero(co[:, 0] == 0.0)[0], np.nonzero(co[:, 0] == 1.0)[0]
    dofs = np.concatenate([lo, hi])
    vals = np.concatenate([np.full(len(lo), 300.0), np.full(len(hi), 360.0)])
    A, b = fo.apply_dirichlet((K + Cm + P).tocsr(), np.zeros(len(co)), dofs, vals, False)
    ref = fo.solve_direct(A, b)
    assert np.abs(T_ip - ref).max() <= 1e-7 * 360.0
    A0, b0 = fo.apply_dirichlet((K + Cm).tocsr(), np.zeros(len(co)), dofs, vals, False)
    galerkin = fo.solve_direct(A0, b0)
    overshoot = lambda T: max(T.max() - 360.0, 300.0 - T.min())           # noqa: E731
    assert overshoot(galerkin) > 1.0 and overshoot(T_ip) < 0.5 * overshoot(galerkin)


def test_interior_penalty_on_triangles(gpu):
    """The same term over the interior EDGES of a 2-D mesh (advection_settings 'IP' on a triangular mesh): kernel against the
    oracle, linear fields in the kernel, and the solver class against the oracle's direct solve."""
    from fenicssolver_amd.fem import Mesh, RectangleMesh, Point, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    rng = np.random.default_rng(3)
    co, ce = fo.rectangle_mesh((0, 0), (1.0, 0.8), 6, 5)
    interior = np.all((co > 1e-12) & (co < np.array([1.0, 0.8]) - 1e-12), axis=1)
    co = co + interior[:, None] * rng.uniform(-0.03, 0.03, co.shape)
    m = Mesh(coords=co, cells=ce)
    ce = m.cells()
    fcells, pairs = m.interior_facet_cells()
    _, _, cnt = fo.tri_edge_numbering(ce)
    assert len(fcells) == int((cnt == 2).sum())
    dm = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(dm, 1, 1, coupled_pairs=pairs)
    A = gpu.DeviceMatrix(V)
    A.zero()
    A.add_interior_penalty(fcells, 0.37)
    ref = fo.assemble_tri_interior_penalty(co, ce, 0.37)
    got = _csr(A)
    assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
    lin = 3.0 + co @ np.array([0.7, -1.1])
    assert np.abs(got @ lin).max() <= 1e-11 * abs(ref).max() * np.abs(lin).max()
    v = rng.standard_normal(len(co))
    assert v @ (got @ v) >= -1e-12 * abs(ref).max()

    mesh = RectangleMesh(Point(0, 0), Point(1, 1), 10, 10)
    Q = FunctionSpace(mesh, "CG", 1)
    bcs = OrderedDict()
    bcs["in"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    bcs["out"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 1.0)), 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
         'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 1.0, 'specific_heat_capacity': 1.0, 'thermal_conductivity': 0.002},
         'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                             'reference_values': {'temperature': 300},
                             'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
         'convective_velocity': Constant((1.0, 0.0)), 'advection_settings': {'stabilization_method': 'IP', 'alpha': 0.1},
         'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0},
         'scalar_name': 'temperature'}
    T = ScalarTransportSolver(s).solve().vector().array()
    c2, t2 = mesh.coordinates(), mesh.cells()
    K = fo.assemble_generic(len(c2), t2, fo.tri_stiffness_local(c2, t2, 0.002) + fo.tri_advection_local(c2, t2, (1.0, 0.0), 1.0))
    P = fo.assemble_tri_interior_penalty(c2, t2, 0.1)
    lo, hi = np.nonzero(c2[:, 0] == 0.0)[0], np.nonzero(c2[:, 0] == 1.0)[0]
    dofs, vals = np.concatenate([lo, hi]), np.concatenate([np.full(len(lo), 300.0), np.full(len(hi), 360.0)])
    ref = fo.solve_direct(*fo.apply_dirichlet((K + P).tocsr(), np.zeros(len(c2)), dofs, vals, False))
    assert np.abs(T - ref).max() <= 1e-7 * 360.0
    gal = fo.solve_direct(*fo.apply_dirichlet(K.tocsr(), np.zeros(len(c2)), dofs, vals, False))
    overshoot = lambda X: max(X.max() - 360.0, 300.0 - X.min())           # noqa: E731
    assert overshoot(gal) > 1.0 and overshoot(T) < 0.6 * overshoot(gal)


@pytest.mark.parametrize("tdim", [3, 2])
def test_interior_penalty_on_p2_spaces(gpu, tdim):
    """The IP term on CG2 spaces (the reference's form is degree-agnostic, ScalarTransportSolver.py:312-315): jump(grad phi, n) is
    linear along a facet, the kernel integrates the quadratic integrand exactly (edge mid-points of the triangle / 2-point Gauss
    on the edge); against the oracle's per-facet restatement with a HIGHER rule <= 1e-12, globally quadratic fields in the
    kernel, positive semi-definite; then through the solver class (P2 advection with 'IP') against the oracle's direct solve."""
    from fenicssolver_amd.fem import Mesh, BoxMesh, RectangleMesh, Point, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    rng = np.random.default_rng(5)
    if tdim == 3:
        co, ce = _mesh(2)
    else:
        co, ce = fo.rectangle_mesh((0, 0), (1.0, 0.8), 5, 4)
        interior = np.all((co > 1e-12) & (co < np.array([1.0, 0.8]) - 1e-12), axis=1)
        co = co + interior[:, None] * rng.uniform(-0.03, 0.03, co.shape)
    m = Mesh(coords=co, cells=ce)
    ce = m.cells()
    Q = FunctionSpace(m, "CG", 2)
    fcells = m.interior_facet_cells()[0]
    V = Q.device(facet_coupling=True)
    plain = gpu.DeviceSpace(m.device(), 1, 2)
    assert V.nnz > plain.nnz
    A = gpu.DeviceMatrix(V)
    A.zero()
    A.add_interior_penalty(fcells, 0.37)
    ref = fo.assemble_p2_interior_penalty(co, ce, 0.37)
    got = _csr(A)
    assert got.shape == ref.shape
    assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
    X = Q.node_coordinates()
    quad = 1.0 + X[:, 0] ** 2 - 0.5 * X[:, 0] * X[:, 1] + 2.0 * X[:, 1] + (X[:, 2] ** 2 if tdim == 3 else 0.0)
    assert np.abs(got @ quad).max() <= 1e-10 * abs(ref).max() * np.abs(quad).max()        # no gradient jump of a global quadratic
    v = rng.standard_normal(got.shape[0])
    assert v @ (got @ v) >= -1e-12 * abs(ref).max()
    with pytest.raises(gpu.BackendError):
        gpu.DeviceMatrix(plain).add_interior_penalty(fcells, 1.0)

    mesh = BoxMesh(Point(0, 0, 0), Point(1, 1, 1), 3, 3, 3) if tdim == 3 else RectangleMesh(Point(0, 0), Point(1, 1), 6, 6)
    Q2 = FunctionSpace(mesh, "CG", 2)
    vel = (1.0, 0.0, 0.0)[:tdim]
    bcs = OrderedDict()
    bcs["in"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    bcs["out"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 1.0)), 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q2, 'periodic_boundary': None,
         'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 1.0, 'specific_heat_capacity': 1.0, 'thermal_conductivity': 0.01},
         'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                             'reference_values': {'temperature': 300}, 'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
         'convective_velocity': Constant(vel), 'advection_settings': {'stabilization_method': 'IP', 'alpha': 0.1},
         'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    T = ScalarTransportSolver(s).solve().vector().array()
    c2, t2 = mesh.coordinates(), mesh.cells()
    if tdim == 3:
        cd, _ = fo.p2_cell_dofs(len(c2), t2)
        K = fo.assemble_generic(len(T), cd, fo.p2_stiffness_local(c2, t2, 0.01) + fo.p2_advection_local(c2, t2, vel, 1.0))
    else:
        cd, _ = fo.tri_p2_cell_dofs(len(c2), t2)
        K = fo.assemble_generic(len(T), cd, fo.tri_p2_stiffness_local(c2, t2, 0.01) + fo.tri_p2_advection_local(c2, t2, vel, 1.0))
    P = fo.assemble_p2_interior_penalty(c2, t2, 0.1)
    Xn = Q2.node_coordinates()
    lo, hi = np.nonzero(Xn[:, 0] == 0.0)[0], np.nonzero(Xn[:, 0] == 1.0)[0]
    dofs, vals = np.concatenate([lo, hi]), np.concatenate([np.full(len(lo), 300.0), np.full(len(hi), 360.0)])
    ref_T = fo.solve_direct(*fo.apply_dirichlet((K + P).tocsr(), np.zeros(len(T)), dofs, vals, False))
    assert np.abs(T - ref_T).max() <= 1e-7 * 360.0
