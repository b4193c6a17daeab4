"""The solver API's domain-decomposed path (fenicssolver_amd/parallel.py + SolverBase.assemble_system
localisation), checked on one GPU two ways:
  * parallel.FORCE_DECOMPOSED_PATH = True sends the whole case through the partition/localise/gather code with
    one part: results must equal the plain single-GPU path;
  * three emulated ranks (no communicator): every rank's owned rows of (A, b), mapped back to
    global numbering, must equal the single-GPU system - coefficients, facet terms, Crank-Nicolson
    old-step terms and the symmetric Dirichlet elimination included.
The RCCL exchange itself is covered by test_gpu_comm.py and the gloo tests of the partition."""
import copy
import os
from collections import OrderedDict

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

QUIET = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}


def _heat_p2_case(n=4):
    """P2 heat: Dirichlet + flux + HTC (the CG2 facet mass matrix, ghost vertex nodes included) + per-subdomain
    conductivity and source."""
    solver = _heat_case(n, degree=2)
    return solver


def _periodic_subdomain(axis, length):
    from fenicssolver_amd.fem import SubDomain, near

    class Periodic(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[axis], 0.0) and on_boundary

        def map(self, x, y):
            for i in range(len(x)):
                y[i] = x[i] - (length if i == axis else 0.0)
    return Periodic()


def _heat_case(n=5, transient=False, degree=1, supg=False, distributed=False, ip=False, periodic=None, source_field=False):
    """periodic: axis of a periodic_boundary (reference SolverBase.py:260-275) - 1: across the slabs a decomposition cuts (slave
    and master on one rank), 2: ALONG the decomposition axis (the first and the last rank become neighbours; the hot / HTC faces
    then move to x)."""
    from fenicssolver_amd.fem import BoxMesh, Point, FunctionSpace, AutoSubDomain, Constant, MeshFunction, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    m = BoxMesh(Point(0, 0, 0), Point(1, 1, 2), n, n, 2 * n, distributed=distributed)
    pb = None if periodic is None else _periodic_subdomain(periodic, (1.0, 1.0, 2.0)[periodic])
    Q = FunctionSpace(m, "CG", degree) if pb is None else FunctionSpace(m, "CG", degree, constrained_domain=pb)
    bcs = OrderedDict()
    if periodic == 2:
        bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 1.0)), 'boundary_id': 1, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
        bcs["flux"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'heatFlux', 'value': Constant(36.0)}}}
        bcs["htc"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 3, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}}}
    else:
        bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 2.0)), 'boundary_id': 1, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
        bcs["flux"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0.0)), 'boundary_id': 2, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'heatFlux', 'value': Constant(36.0)}}}
        bcs["htc"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0.0)), 'boundary_id': 3, 'values': {
            'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': pb,
         'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 10.0, 'specific_heat_capacity': 2.0, 'thermal_conductivity': 0.6},
         'solver_settings': {'transient_settings': {'transient': transient, 'starting_time': 0, 'time_step': 0.1,
                                                    'ending_time': 0.3},
                             'reference_values': {'temperature': 300},
                             'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
         'report_settings': dict(QUIET), 'scalar_name': 'temperature'}
    if supg:        # advection with the 'SPUG' test function in the volume, source and boundary (flux, HTC) integrals
        s['convective_velocity'] = Constant((0.02, -0.01, 0.03))
        s['advection_settings'] = {'stabilization_method': 'SPUG', 'Pe': 10.0}
    if ip:          # interior-penalty stabilisation: a dS integral - under decomposition the part takes a second cell layer
        s['convective_velocity'] = Constant((0.02, -0.01, 0.03))
        s['advection_settings'] = {'stabilization_method': 'IP', 'alpha': 0.1}
    solver = ScalarTransportSolver(s)
    cen = m.coordinates()[m.cells().astype(np.int64)].mean(axis=1)
    sub = MeshFunction("size_t", m, 3)
    sub.array()[:] = np.where(cen[:, 2] < 0.9, 1, 2)
    solver.subdomains = sub
    solver.material['conductivity'] = {'lower': {'subdomain_id': 1, 'value': 0.6},
                                       'upper': {'subdomain_id': 2, 'value': 6.0}}
    solver.body_source = {'heater': {'subdomain_id': 2, 'value': 50.0}}
    if source_field:      # a body source given as an Expression: its nodal interpolant, also under the SUPG test function (round 5)
        from fenicssolver_amd.fem import Expression
        solver.body_source = Expression("50.0 + 30.0*x[0] - 20.0*x[1]*x[2]", degree=degree)
    return solver


def _elastic_p2_case():
    solver = _elastic_case(degree=2)
    return solver


def _elastic_case(distributed=False, degree=1, fine=1, pressure_field=False, wide=False, periodic=None):
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    bcs = OrderedDict()
    if distributed:     # slabs are cut along z: the beam lies along z
        # (wide: mesh lines of 71 nodes, so that every 64-row slice of the operator holds at most one line end - the DIA form the
        # row dictionary needs)
        mesh = BoxMesh(Point(0, 0, 0), Point(3, 1, 10), 70, 8, 24, distributed=True) if wide else \
            BoxMesh(Point(0, 0, 0), Point(1, 1, 10), 2 * fine, 2 * fine, 12 * fine, distributed=True)
        bcs["fixed"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0)), 'boundary_id': 1, 'type': 'Dirichlet',
                        'value': Constant((0, 0, 0))}
        bcs["tensile"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 10)), 'boundary_id': 2, 'type': 'stress',
                          'value': Constant((0, 1e6, 1e8))}
    else:
        mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 12 * fine, 2 * fine, 2 * fine)
        bcs["fixed"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0)), 'boundary_id': 1, 'type': 'Dirichlet',
                        'value': Constant((0, 0, 0))}
        bcs["tensile"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 10)), 'boundary_id': 2, 'type': 'stress',
                          'value': Constant((1e8, 0, 0))}
    if pressure_field:      # a pressure that varies over the face (hydrostatic-like): nodal loads worked out on the host
        ax = 2 if distributed else 0
        bcs["side"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1)), 'boundary_id': 3, 'type': 'pressure',
                       'value': Expression("1e6*(1+0.3*x[%d])" % ax, degree=1)}
    s = copy.deepcopy(SB.default_case_settings)
    s['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                     'thermal_expansion_coefficient': 2e-6}
    pb = None if periodic is None else _periodic_subdomain(periodic, 1.0)     # (the beam's cross-section is the unit square)
    s['function_space'] = VectorFunctionSpace(mesh, "Lagrange", degree, constrained_domain=pb)
    s['periodic_boundary'] = pb
    s['boundary_conditions'] = bcs
    s['solver_settings']['reference_values'] = {'temperature': 293}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-12}
    s['report_settings'] = dict(QUIET)
    s['body_source'] = Expression(("10*rho", "0", "0.0"), rho=7800, degree=2)
    s['temperature_distribution'] = Expression("300+40*x[2]" if distributed else "300+40*x[0]", degree=1)
    return LinearElasticitySolver(s)


def _cavity_case(n=4, transient=True, distributed=False, thermal=False):
    """Lid-driven cavity, Taylor-Hood, two backward-Euler steps with Newton (configs[4] in small)."""
    from fenicssolver_amd.fem import UnitCubeMesh, BoxMesh, Point, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    mesh = BoxMesh(Point(0, 0, 0), Point(1, 1, 1.5), n, n, n + 2, distributed=distributed)
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 1.5)), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': 0.01}})
    s['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01,
                                                  'ending_time': 0.02 - 1e-9}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-10}
    s['report_settings'] = dict(QUIET)
    if thermal:
        # solving_temperature with the non-Newtonian law nu (1 + 0.1 p/p_ref)(1 - 0.2 T/T_ref): the temperature (its own P1 space with
        # the interior-penalty term, i.e. a two-layer part under decomposition) feeds back into the momentum equation
        s['solving_temperature'] = True
        s['material'] = {'density': 1.0, 'kinematic_viscosity': 0.05, 'Newtonian': False, 'specific_heat_capacity': 3.0,
                         'thermal_conductivity': 0.1}
        s['boundary_conditions']['walls']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(420.0)})
        s['boundary_conditions']['lid']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300.0)})
        s['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 0, 'temperature': 320.0}
        s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 10.0, 'temperature': 300.0}
    return CoupledNavierStokesSolver(s)


def _channel_case(distributed=False):
    """Pressure-driven channel along z (the partition axis): pressure Dirichlet + the reference's pressure-boundary
    integrals on the inlet / outlet facets, which lie in the first and the last rank's parts."""
    from fenicssolver_amd.fem import BoxMesh, Point, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    nu = 0.3
    mesh = BoxMesh(Point(0, 0, 0), Point(1, 1, 2), 3, 3, 6, distributed=distributed)
    prof = Expression(("0", "0", "x[0]*(1-x[0])"), degree=2)
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and (near(x[0], 0) or near(x[0], 1) or near(x[1], 0) or near(x[1], 1))),
                    'boundary_id': 1, 'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': prof}]}
    bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 0)), 'boundary_id': 2,
                    'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(2 * nu * 2)}]}
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 2)), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(0.0)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': nu}})
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-10}
    s['report_settings'] = dict(QUIET)
    return CoupledNavierStokesSolver(s)


def _radiation_case(degree=1):
    """examples/test_heat_transfer.py test_radiation(): radiation to the ambient on every exterior facet and a
    temperature-dependent conductivity, Newton through solve_nonlinear_problem."""
    from fenicssolver_amd.fem import BoxMesh, Point, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    m = BoxMesh(Point(0, 0, 0), Point(1, 1, 2), 4, 4, 8) if degree == 1 else BoxMesh(Point(0, 0, 0), Point(1, 1, 2), 3, 3, 6)
    Q = FunctionSpace(m, "CG", degree)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 2.0)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0.0)), 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
         'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.6},
         'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                             'reference_values': {'temperature': 300},
                             'solver_parameters': {'krylov_relative_tolerance': 1e-13}},
         'radiation_settings': {'ambient_temperature': 280.0, 'emissivity': 0.9},
         'report_settings': dict(QUIET), 'scalar_name': 'temperature'}
    solver = ScalarTransportSolver(s)
    solver.material['conductivity'] = lambda T: 0.6 * (1.0 + 0.002 * (T - 300.0))
    solver.material['emissivity'] = 0.9
    return solver


# cases that do not go through _device_solve (Newton loops, the saddle-point path): no captured (A, b) test
NS_CASES = {"cavity": _cavity_case, "channel": _channel_case, "radiation": _radiation_case, "radiation_p2": lambda: _radiation_case(2),
            "cavity_thermal": lambda: _cavity_case(transient=False, thermal=True)}

# BoxMesh(distributed=True): every rank builds only its z-slab on the host (one rank: the same mesh as the replicated one)
DIST_CASES = {"heat_dist": lambda: _heat_case(distributed=True), "heat_cn_dist": lambda: _heat_case(transient=True, distributed=True),
              "elasticity_dist": lambda: _elastic_case(distributed=True), "elasticity_fine_dist": lambda: _elastic_case(distributed=True, fine=3),
              # CG2 spaces on the distributed box: node plan from local cells only, host <-> device through a local permutation
              "heat_p2_dist": lambda: _heat_case(4, degree=2, distributed=True),
              "heat_p2_cn_dist": lambda: _heat_case(4, transient=True, degree=2, distributed=True),
              "elasticity_p2_dist": lambda: _elastic_case(distributed=True, degree=2),
              "elasticity_pfield_dist": lambda: _elastic_case(distributed=True, pressure_field=True),
              "elasticity_wide_dist": lambda: _elastic_case(distributed=True, wide=True),
              # Taylor-Hood on the distributed box (round 4): Newton loop, pressure hierarchy and projections on this rank's slab only
              "cavity_dist": lambda: _cavity_case(distributed=True), "channel_dist": lambda: _channel_case(distributed=True)}

def _file_mesh_case(degree=1):
    """data/TestHeatTransfer.json on data/mesh.xml (a Gmsh mesh in file order): decomposed by RCB-free coordinate slabs, every part
    numbered in the locality order of fs_mesh_locality_order when FS_RENUMBER=1."""
    from fenicssolver_amd.main import load_settings
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    s = load_settings(os.path.join(data, "TestHeatTransfer.json"))
    s["mesh"] = os.path.join(data, "mesh.xml")
    s["report_settings"] = dict(QUIET)
    s["fe_degree"] = degree
    return ScalarTransportSolver(s)


CASES = {"heat": lambda: _heat_case(), "heat_cn": lambda: _heat_case(transient=True), "elasticity": _elastic_case,
         # 37 x 7 x 7 nodes: a hierarchy of several levels (the 12 x 2 x 2 beam is a single level)
         "elasticity_fine": lambda: _elastic_case(fine=3), "elasticity_pfield": lambda: _elastic_case(pressure_field=True),
         "elasticity_p2_pfield": lambda: _elastic_case(degree=2, pressure_field=True),
         "heat_file": _file_mesh_case, "heat_file_p2": lambda: _file_mesh_case(2),
         "heat_supg": lambda: _heat_case(supg=True), "heat_supg_field": lambda: _heat_case(supg=True, source_field=True),
         "heat_p2_supg_field": lambda: _heat_case(supg=True, source_field=True, degree=2), "heat_ip": lambda: _heat_case(ip=True), "heat_ip_cn": lambda: _heat_case(ip=True, transient=True),
         "heat_p2": _heat_p2_case, "elasticity_p2": _elastic_p2_case,
         # periodic_boundary under decomposition (round 4): slaves owned by their masters' rank, masters as extra ghosts
         "heat_periodic_y": lambda: _heat_case(periodic=1), "heat_periodic_z": lambda: _heat_case(periodic=2),
         "heat_periodic_z_cn": lambda: _heat_case(periodic=2, transient=True),
         "elasticity_periodic": lambda: _elastic_case(periodic=1),
         # ... and of CG2 spaces (round 5): the cells around the masters in the part, edges owned through order ids
         "heat_p2_periodic_y": lambda: _heat_case(4, degree=2, periodic=1), "heat_p2_periodic_z": lambda: _heat_case(4, degree=2, periodic=2),
         "elasticity_p2_periodic": lambda: _elastic_case(degree=2, periodic=1)}


@pytest.mark.parametrize("case", sorted(CASES) + sorted(NS_CASES))
def test_forced_single_part_equals_plain_path(gpu, monkeypatch, case):
    from fenicssolver_amd import parallel
    make = CASES.get(case) or NS_CASES[case]
    plain = make().solve().vector().array()
    monkeypatch.setattr(parallel, "FORCE_DECOMPOSED_PATH", True)
    assert parallel.active()
    solver = make()
    forced = solver.solve().vector().array()
    assert solver.function_space.localizer() is not None
    tol = 1e-9 if case in CASES else 1e-7          # Newton + FGMRES to 1e-10 on the update
    assert np.abs(forced - plain).max() <= tol * np.abs(plain).max()


class _Captured(Exception):
    pass


def _capture_system(monkeypatch, make, rank=None, world=1):
    """Run make().solve() up to the first linear solve and return (A as global-numbered csr rows,
    b, global row ids, ndof) of the rank."""
    from fenicssolver_amd import parallel, backend, SolverBase as SB
    got = {}

    def fake_solve(self, A, b, u, label, method="cg", **kwargs):
        rp, ci, va, shape = A.to_csr()
        got.update(A=sp.csr_matrix((va, ci, rp), shape=shape), b=b.get(), loc=u.function_space().localizer(),
                   ncomp=u.function_space()._ncomp)
        raise _Captured()

    with monkeypatch.context() as mp:
        mp.setattr(SB.SolverBase, "_device_solve", fake_solve)
        if rank is not None:
            mp.setattr(parallel, "world", lambda: (rank, world, 0))
            mp.setattr(parallel, "ensure_comm", lambda: (rank, world))
            mp.setattr(backend.DeviceSpace, "set_halo", lambda self, *a, **k: None)
        with pytest.raises(_Captured):
            make().solve()
    return got


@pytest.mark.parametrize("case", sorted(CASES))
def test_three_emulated_ranks_reproduce_the_global_system(gpu, monkeypatch, case):
    ref = _capture_system(monkeypatch, CASES[case])
    A, b = ref["A"].tocsr(), ref["b"]
    ndof = A.shape[0]
    covered = np.zeros(ndof, dtype=int)
    for r in range(3):
        got = _capture_system(monkeypatch, CASES[case], rank=r, world=3)
        loc, nc = got["loc"], got["ncomp"]
        l2g_dof = (loc.l2g[:, None].astype(np.int64) * nc + np.arange(nc)).ravel()     # node-level (P2: vertices + edges)
        rows = l2g_dof[:loc.n_owned * nc]
        covered[rows] += 1
        Al = got["A"].tocoo()
        Ag = sp.csr_matrix((Al.data, (rows[Al.row], l2g_dof[Al.col])), shape=(ndof, ndof))
        diff = abs(Ag[rows] - A[rows])
        scale = abs(A).max()
        assert (diff.max() if diff.nnz else 0.0) <= 1e-12 * scale, (case, r)
        assert np.abs(got["b"] - b[rows]).max() <= 1e-11 * max(np.abs(b).max(), 1e-300), (case, r)
    assert np.all(covered == 1)
