"""GPU parity tests of the quadratic (P2) scalar path against the oracle: edge numbering and
sparsity bit-exact, 10x10 element matrices (4-point quadrature, exact mass), loads, Dirichlet on
vertex + edge nodes, convergence order 3, and the solver API with fe_degree = 2."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def _p2(gpu, co, ce, mesh=None):
    mesh = mesh or gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1, degree=2)
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    return mesh, V, cd, edges


def test_p2_edges_pattern_and_matrices(gpu, data_dir):
    for co, ce in (fo.unit_cube_mesh(3), fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))):
        mesh, V, cd, edges = _p2(gpu, co, ce)
        n = len(co) + len(edges)
        assert V.n_owned == n
        assert np.array_equal(V.edges(), edges)                    # bit-exact edge-node numbering (both rules)
        lex = fo.edge_numbering(ce)[0]
        grouped = len(np.unique(lex[:, 1].astype(np.int64) - lex[:, 0])) <= 16
        assert grouped == (len(co) == 64)                          # the Kuhn cube is grouped, the Gmsh mesh is not
        A = gpu.DeviceMatrix(V)
        rng = np.random.default_rng(0)
        kc = rng.uniform(0.5, 2.0, len(ce))
        for kw, Ke in ((dict(stiffness=3.0), fo.p2_stiffness_local(co, ce, 3.0)),
                       (dict(stiffness=("cell", kc)), fo.p2_stiffness_local(co, ce, kc)),
                       (dict(stiffness=1.5, mass=2.0), fo.p2_stiffness_local(co, ce, 1.5) + fo.p2_mass_local(co, ce, 2.0)),
                       (dict(mass=("cell", kc)), fo.p2_mass_local(co, ce, kc))):
            A.assemble(**kw)
            M = _csr(A)
            R = fo.assemble_generic(n, cd, Ke)
            assert np.array_equal(M.indptr, R.indptr) and np.array_equal(M.indices, R.indices)
            assert np.abs(M.data - R.data).max() <= 1e-12 * np.abs(R.data).max(), kw
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    assert len(fo.edge_numbering(ce)[0]) == 6123                   # config-1 mesh has 6 123 edges


def test_p2_loads_and_solve_quadratic_patch(gpu):
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 1.5, 0.5), 3, 4, 2)
    mesh, V, cd, edges = _p2(gpu, co, ce)
    X = fo.p2_dof_coordinates(co, edges)
    n = len(X)
    b = gpu.DeviceVector(n)
    gpu.assemble_vector(V, b, source=-1.0)
    ref = fo.assemble_generic_vector(n, cd, fo.p2_source_local(co, ce, -1.0))
    assert np.abs(b.get() - ref).max() <= 1e-13 * np.abs(ref).max()
    fn = np.random.default_rng(1).uniform(0, 1, n)
    gpu.assemble_vector(V, b, source=("nodal", fn))
    Me = fo.p2_mass_local(co, ce, 1.0)
    ref2 = fo.assemble_generic_vector(n, cd, np.einsum("cab,cb->ca", Me, fn[cd.astype(np.int64)]))
    assert np.abs(b.get() - ref2).max() <= 1e-12 * np.abs(ref2).max()
    # a quadratic field is reproduced exactly:  -lap u = -1 with u below
    u = 1 + X[:, 0] ** 2 - 0.5 * X[:, 1] ** 2 + X[:, 0] * X[:, 2] + 2 * X[:, 1]
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    gpu.assemble_vector(V, b, source=-1.0)
    hi = np.array([1.0, 1.5, 0.5])
    onb = np.nonzero(((X == 0) | (X == hi)).any(axis=1))[0]
    A.apply_dirichlet(b, onb, u[onb], symmetric=True)
    x = gpu.DeviceVector(n)
    st = gpu.krylov_solve(A, b, x, rtol=1e-13, max_iter=5000)
    assert st["converged"] == 1
    assert np.abs(x.get() - u).max() <= 1e-10
    # facet load on z = top: edge nodes get g*area/3, vertices nothing
    facets, _, cnt = fo.facet_numbering(ce)
    tri = facets[(cnt == 1) & np.all(co[facets][:, :, 2] == 0.5, axis=1)]
    b.fill(0.0)
    gpu.assemble_facet_vector(V, b, tri, 7.0)
    got = b.get()
    assert np.all(got[:len(co)] == 0.0) and abs(got.sum() - 7.0 * 1.0 * 1.5) < 1e-12


def test_p2_third_order_convergence(gpu):
    errs = []
    for n in (4, 8, 16):
        mesh = gpu.DeviceMesh.box(n, n, n)
        V = gpu.DeviceSpace(mesh, 1, degree=2)
        xyz, cells, _ = mesh.get()
        edges = V.edges().astype(np.int64)
        X = np.concatenate([xyz, 0.5 * (xyz[edges[:, 0]] + xyz[edges[:, 1]])])
        u = np.sin(np.pi * X[:, 0]) * np.sin(np.pi * X[:, 1]) * np.sin(np.pi * X[:, 2])
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=1.0)
        b = gpu.DeviceVector(V.n_owned)
        gpu.assemble_vector(V, b, source=("nodal", 3 * np.pi ** 2 * u))
        onb = np.nonzero(((X == 0.0) | (X == 1.0)).any(axis=1))[0]
        A.apply_dirichlet(b, onb, 0.0, symmetric=True)
        x = gpu.DeviceVector(V.n_owned)
        st = gpu.krylov_solve(A, b, x, rtol=1e-12, max_iter=20000)
        assert st["converged"] == 1
        errs.append(np.sqrt(np.mean((x.get() - u) ** 2)))
    r1, r2 = np.log2(errs[0] / errs[1]), np.log2(errs[1] / errs[2])
    assert r1 > 2.6 and r2 > 2.7, (errs, r1, r2)      # O(h^3) (Appendix C6)


def test_p2_solver_api_config1_and_box(gpu, data_dir):
    from fenicssolver_amd.main import load_settings
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    s = load_settings(os.path.join(data_dir, "TestHeatTransfer.json"))
    s["fe_degree"] = 2
    s["report_settings"] = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    s["solver_settings"]["solver_parameters"]["krylov_relative_tolerance"] = 1e-12
    solver = ScalarTransportSolver(s)
    T = solver.solve()
    X = solver.function_space.node_coordinates()
    assert T.vector().size() == 1069 + 6123
    assert np.abs(T.vector().array() - (350.0 - 2.5 * X[:, 2])).max() <= 1e-8
    # body source + flux on a box, against the oracle's LU
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    m = UnitCubeMesh(3, 3, 3)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(360)}
    bcs["in"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'type': 'heatFlux', 'value': Constant(12.0)}
    st = {'solver_name': 'x', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs,
          'body_source': 5.0, 'initial_values': {'temperature': 300},
          'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.6},
          'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 1},
                              'reference_values': {'temperature': 300},
                              'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
          'report_settings': dict(s["report_settings"]), 'scalar_name': 'temperature'}
    sol = ScalarTransportSolver(st)
    Tb = sol.solve().vector().array()
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    n = len(co) + len(edges)
    A = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.6))
    b = fo.assemble_generic_vector(n, cd, fo.p2_source_local(co, ce, 5.0))
    facets, _, cnt = fo.facet_numbering(ce)
    fm = sol.boundary_facets.array()
    tri = facets[fm == 2].astype(np.int64)
    area = fo.facet_areas(co, tri)
    nv = len(co)
    ekey = edges[:, 0].astype(np.int64) * nv + edges[:, 1]
    sorter = np.argsort(ekey)          # the structured cube groups its edge nodes by v1 - v0
    for (i, j) in ((0, 1), (0, 2), (1, 2)):
        eid = sorter[np.searchsorted(ekey[sorter], tri[:, i] * nv + tri[:, j])]
        np.add.at(b, nv + eid, 12.0 * area / 3.0)
    dofs = fo.p2_facet_dofs(nv, edges, facets, fm, 1)
    Ab, bb = fo.apply_dirichlet(A, b, dofs, 360.0, True)
    ref = fo.solve_direct(Ab, bb)
    assert np.abs(Tb - ref).max() <= 1e-8 * np.abs(ref).max()


def test_p2_htc_facet_matrix_and_solver_class(gpu):
    """Robin / HTC on a CG2 space: int h T q ds with the exact P2 facet mass matrix against the oracle's quadrature, and
    the solver class (fe_degree 2, HTC on one face, Dirichlet on the opposite one) against an oracle solve."""
    from fenicssolver_amd.fem import BoxMesh, Point, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 1.2), 3, 2, 3)
    mesh, V, cd, edges = _p2(gpu, co, ce)
    n = len(co) + len(edges)
    facets, _, cnt = fo.facet_numbering(ce)
    markers = fo.mark_facets(co, ce, lambda x, on_b: on_b and np.isclose(x[2], 0.0), 3)
    sel = np.nonzero(markers == 3)[0]
    hvals = np.linspace(50.0, 150.0, len(sel))
    A = gpu.DeviceMatrix(V)
    A.zero()
    A.add_facet_mass(facets[sel], hvals)
    ref = fo.assemble_p2_facet_mass(co, edges, facets, markers, 3, hvals)
    got = _csr(A)
    assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
    # through the solver class
    m = BoxMesh(Point(0, 0, 0), Point(1.0, 0.8, 1.2), 3, 2, 3)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 1.2)), 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
    bcs["htc"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0.0)), 'boundary_id': 3, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
         'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 10.0, 'specific_heat_capacity': 2.0, 'thermal_conductivity': 0.6},
         'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                             'reference_values': {'temperature': 300}, 'solver_parameters': {'krylov_relative_tolerance': 1e-12}},
         'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0},
         'scalar_name': 'temperature'}
    T = ScalarTransportSolver(s).solve().vector().array()
    K = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.6))
    R = fo.assemble_p2_facet_mass(co, edges, facets, markers, 3, 100.0)
    b = R @ np.full(n, 300.0)                                   # int h Ta q ds = R * (Ta as a P2 function)
    mk1 = fo.mark_facets(co, ce, lambda x, on_b: on_b and np.isclose(x[2], 1.2), 1)
    dofs = fo.p2_facet_dofs(len(co), edges, facets, mk1, 1)
    Ab, bb = fo.apply_dirichlet((K + R).tocsr(), b, dofs, np.full(len(dofs), 360.0), True)
    ref_T = fo.solve_direct(Ab, bb)
    assert np.abs(T - ref_T).max() <= 1e-8 * 360.0
    # physical check: 1-D conduction with a Robin end, T(z) = 300 + q (1/h + z/k), q = (360 - 300)/(1/h + L/k)
    X = fo.p2_dof_coordinates(co, edges)
    q = 60.0 / (1.0 / 100.0 + 1.2 / 0.6)
    assert np.abs(T - (300.0 + q * (1.0 / 100.0 + X[:, 2] / 0.6))).max() <= 1e-7


def test_p2_advection_kernel_and_solver_class(gpu, data_dir):
    """fe_degree 2 with a convective velocity (ScalarTransportSolver.py:305-311): C_ab = capacity * int phi_a (v . grad phi_b) dx
    on the CG2 basis - kernel (Keast's 5-point degree-3 rule) against the oracle (14-point degree-5 rule) for constant and
    per-cell velocities on an unstructured and a structured mesh, then the solver class (BiCGStab) against a direct solve."""
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    import scipy.sparse as sps
    rng = np.random.default_rng(9)
    for co, ce in (fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml")), fo.unit_cube_mesh(4)):
        mesh = gpu.DeviceMesh(co, ce)
        V = gpu.DeviceSpace(mesh, 1, degree=2)
        cd, edges = fo.p2_cell_dofs(len(co), ce)
        assert np.array_equal(V.edges().astype(np.int64), edges.astype(np.int64))
        n = len(co) + len(edges)
        A = gpu.DeviceMatrix(V)
        for vel in (np.array([0.3, -0.2, 0.5]), rng.uniform(-1, 1, (len(ce), 3))):
            A.assemble(stiffness=0.7, mass=0.1, advection=vel, advection_scale=2.5)
            ref = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.7) + fo.p2_mass_local(co, ce, 0.1)
                                      + fo.p2_advection_local(co, ce, vel, 2.5)).tocsr()
            rp, ci, va, shape = A.to_csr()
            got = sps.csr_matrix((va, ci, rp), shape=shape)
            assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
        # SUPG on CG2 (round 4): the test function q + tau (v . grad q) in every term - its gradient carries the Hessian of q
        for vel in (np.array([0.3, -0.2, 0.5]), rng.uniform(-1, 1, (len(ce), 3))):
            for kk, mm, sc in ((0.7, 0.1, 2.5), (-0.35, 11.0, 0.0)):           # the operator; the old-step matrix of Crank-Nicolson
                A.assemble(stiffness=kk, mass=mm, advection=vel, advection_scale=sc, supg_pe=2.0)
                ref = fo.assemble_generic(n, cd, fo.p2_supg_system_local(co, ce, vel, 2.0, kk, sc, mm)).tocsr()
                rp, ci, va, shape = A.to_csr()
                got = sps.csr_matrix((va, ci, rp), shape=shape)
                assert abs(got - ref).max() <= 1e-12 * abs(ref).max()
            plain = fo.assemble_generic(n, cd, fo.p2_supg_system_local(co, ce, vel, None, 0.7, 2.5, 0.1)).tocsr()
            A.assemble(stiffness=0.7, mass=0.1, advection=vel, advection_scale=2.5, supg_pe=2.0)
            rp, ci, va, shape = A.to_csr()
            assert abs(sps.csr_matrix((va, ci, rp), shape=shape) - plain).max() > 1e-3 * abs(plain).max()      # not a no-op
            b = gpu.DeviceVector(V.n_owned)
            fc = rng.uniform(1.0, 3.0, len(ce))
            gpu.assemble_vector(V, b, source=("cell", fc), supg=(vel, 2.0))
            refb = fo.assemble_generic_vector(n, cd, fo.p2_supg_source_local(co, ce, vel, 2.0, fc))
            assert np.abs(b.get() - refb).max() <= 1e-12 * np.abs(refb).max()
            # a NODAL source (a Function on the CG2 space): closed-form first moments in the kernel, degree-5 quadrature in the oracle
            fn = rng.uniform(1.0, 3.0, n)
            gpu.assemble_vector(V, b, source=("nodal", fn), supg=(vel, 2.0))
            refn = fo.assemble_generic_vector(n, cd, fo.p2_supg_source_local(co, ce, vel, 2.0, f_nodal=fn, cell_dofs=cd))
            plainn = fo.assemble_generic_vector(n, cd, fo.p2_supg_source_local(co, ce, vel, 1e-300, f_nodal=fn, cell_dofs=cd))
            assert np.abs(b.get() - refn).max() <= 1e-12 * np.abs(refn).max() and np.abs(refn - plainn).max() > 1e-3 * np.abs(refn).max()
            # ds terms: the cells behind the boundary facets (cell, opposite vertex)
            from oracle import ns_oracle as nso
            fcells = np.array(nso.boundary_facet_cells(nso.TaylorHood(co, ce), lambda x: True)).reshape(-1, 2)[::3]
            gval, hval = rng.uniform(1.0, 2.0, len(fcells)), rng.uniform(5.0, 9.0, len(fcells))
            A.assemble(stiffness=0.7)
            base = sps.csr_matrix((A.to_csr()[2], A.to_csr()[1], A.to_csr()[0]), shape=A.to_csr()[3])
            b = gpu.DeviceVector(V.n_owned)
            gpu.assemble_facet_supg(V, A, b, fcells[:, 0], fcells[:, 1], vel, 2.0, g=gval, h=hval)
            dA, db = fo.p2_supg_facet_terms(co, ce, cd.astype(np.int64), n, fcells, vel, 2.0, gval, hval)
            rp, ci, va, shape = A.to_csr()
            assert abs((sps.csr_matrix((va, ci, rp), shape=shape) - base) - dA).max() <= 1e-12 * abs(dA).max()
            assert np.abs(b.get() - db).max() <= 1e-12 * np.abs(db).max()
    m = UnitCubeMesh(3, 3, 3)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(360)}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant(300)}
    st = {'solver_name': 'x', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs,
          'body_source': 5.0, 'initial_values': {'temperature': 300}, 'convective_velocity': Constant((0.05, -0.08, 0.02)),
          'material': {'density': 10.0, 'specific_heat_capacity': 2.0, 'thermal_conductivity': 0.6},
          'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 1},
                              'reference_values': {'temperature': 300},
                              'solver_parameters': {'krylov_relative_tolerance': 1e-12, 'maximum_iterations': 20000}},
          'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    T = ScalarTransportSolver(st).solve().vector().array()
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    n = len(co) + len(edges)
    A = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.6) + fo.p2_advection_local(co, ce, (0.05, -0.08, 0.02), 20.0))
    b = fo.assemble_generic_vector(n, cd, fo.p2_source_local(co, ce, 5.0))
    X = Q.node_coordinates()
    top, bot = np.nonzero(X[:, 1] == 1.0)[0], np.nonzero(X[:, 1] == 0.0)[0]
    ref = fo.solve_direct(*fo.apply_dirichlet(A.tocsr(), b, np.concatenate([top, bot]),
                                              np.concatenate([np.full(len(top), 360.0), np.full(len(bot), 300.0)]), False))
    assert np.abs(T - ref).max() <= 1e-7 * np.abs(ref).max()
    assert np.abs(T - (300 + 60 * X[:, 1])).max() > 0.5              # the convection bends the profile


def test_p2_radiation_newton(gpu):
    """radiation_settings with fe_degree 2: the residual int m (T_h^4 - T_amb^4) q ds with the P2 iterate (degree-10 integrand),
    device Newton against a quasi-Newton iteration written with the oracle and an independent quadrature."""
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    m = UnitCubeMesh(3, 3, 2)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(360)}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant(300)}
    st = {'solver_name': 'x', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs,
          'body_source': None, 'initial_values': {'temperature': 300},
          'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.6, 'emissivity': 0.9},
          'radiation_settings': {'ambient_temperature': 280.0, 'emissivity': 0.9},
          'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 1},
                              'reference_values': {'temperature': 300},
                              'solver_parameters': {'krylov_relative_tolerance': 1e-13, 'maximum_iterations': 20000}},
          'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    solver = ScalarTransportSolver(st)
    T = solver.solve().vector().array()
    assert solver.nonlinear and 2 <= solver.newton_iterations <= 30
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    nv, n = len(co), len(co) + len(edges)
    X = Q.node_coordinates()
    K = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.6)).tocsr()
    facets, _, cnt = fo.facet_numbering(ce)
    ext = facets[cnt == 1].astype(np.int64)
    emap = {(int(a), int(b)): nv + k for k, (a, b) in enumerate(edges.astype(np.int64))}
    tab = np.array([list(t) + [emap[tuple(sorted((int(t[i]), int(t[j]))))] for i, j in ((0, 1), (0, 2), (1, 2))] for t in ext])
    area = fo.facet_areas(co, ext)
    g8, w8 = np.polynomial.legendre.leggauss(8)
    g8, w8 = 0.5 * (g8 + 1), 0.5 * w8
    pts, wq = [], []
    for u, wu in zip(g8, w8):
        for v, wv in zip(g8, w8):
            pts.append((1 - u, u * (1 - v), u * v)); wq.append(2 * wu * wv * u)
    pts, wq = np.array(pts), np.array(wq)
    phi = np.stack([pts[:, i] * (2 * pts[:, i] - 1) for i in range(3)] + [4 * pts[:, a] * pts[:, b] for a, b in ((0, 1), (0, 2), (1, 2))], axis=1)
    mrad, Ta = 0.9 * 5.670367e-8, 280.0
    top, bot = np.nonzero(X[:, 1] == 1.0)[0], np.nonzero(X[:, 1] == 0.0)[0]
    dofs = np.concatenate([top, bot])
    vals = np.concatenate([np.full(len(top), 360.0), np.full(len(bot), 300.0)])
    fm = (cnt == 1).astype(np.int32)
    J = (K + fo.assemble_p2_facet_mass(co, edges, facets, fm, 1, 4.0 * mrad * 330.0 ** 3)).tocsr()       # frozen linearisation
    Tn = np.full(n, 300.0)
    Tn[dofs] = vals
    for it in range(200):
        Tq = Tn[tab] @ phi.T
        loads = area[:, None] * ((mrad * (Ta ** 4 - Tq ** 4) * wq[None, :]) @ phi)
        b = np.zeros(n)
        np.add.at(b, tab.ravel(), loads.ravel())
        r = K @ Tn - b
        r[dofs] = 0.0
        if np.linalg.norm(r) < 1e-10:
            break
        Tn = Tn + fo.solve_direct(*fo.apply_dirichlet(J, -r, dofs, 0.0, True))
    assert it < 199
    assert np.abs(T - Tn).max() <= 1e-6
    assert np.abs(T - (300 + 60 * X[:, 1])).max() > 1e-3


def test_p2_temperature_dependent_conductivity(gpu):
    """material['conductivity'] = lambda T: ... with fe_degree 2 (the nonlinear variant of examples/test_heat_transfer.py:50-54):
    k(T_h) evaluated at the quadrature points of the stiffness integrand (FS_COEF_CELL_QP); Newton (Picard Jacobian) on the
    device against the same fixed-point iteration written with the oracle."""
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    from oracle import ns_oracle as nso
    m = UnitCubeMesh(3, 3, 2)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(360)}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant(300)}
    st = {'solver_name': 'x', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs,
          'body_source': 40.0, 'initial_values': {'temperature': 300},
          'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.6},
          'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 1},
                              'reference_values': {'temperature': 300},
                              'solver_parameters': {'krylov_relative_tolerance': 1e-13, 'maximum_iterations': 20000}},
          'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    kfun = lambda T: 0.6 * (1.0 + 0.01 * (T - 300.0))                      # noqa: E731  (linear in T: the integrand is a quartic)
    solver = ScalarTransportSolver(st)
    solver.material['conductivity'] = kfun
    T = solver.solve().vector().array()
    assert solver.nonlinear and 2 <= solver.newton_iterations <= 40
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    n = len(co) + len(edges)
    X = Q.node_coordinates()
    b = fo.assemble_generic_vector(n, cd, fo.p2_source_local(co, ce, 40.0))
    top, bot = np.nonzero(X[:, 1] == 1.0)[0], np.nonzero(X[:, 1] == 0.0)[0]
    dofs = np.concatenate([top, bot])
    vals = np.concatenate([np.full(len(top), 360.0), np.full(len(bot), 300.0)])
    Tn = np.full(n, 300.0)
    Tn[dofs] = vals
    for it in range(100):
        Tc = Tn[cd.astype(np.int64)]
        K = fo.assemble_generic(n, cd, fo.p2_stiffness_local_qp(co, ce, lambda lam: kfun(Tc @ nso.p2_shape(lam)[0]))).tocsr()
        r = K @ Tn - b
        r[dofs] = 0.0
        if np.linalg.norm(r) < 1e-9:
            break
        Tn = Tn + fo.solve_direct(*fo.apply_dirichlet(K, -r, dofs, 0.0, True))
    assert it < 99
    assert np.abs(T - Tn).max() <= 1e-6
    lin = fo.solve_direct(*fo.apply_dirichlet(fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.6)).tocsr(), b, dofs, vals, True))
    assert np.abs(T - lin).max() > 0.1                                     # the nonlinearity matters


@pytest.mark.parametrize("transient,source", [(False, "constant"), (True, "constant"), (False, "field")])
def test_p2_supg_stabilised_convection_matches_oracle(gpu, transient, source):
    """advection_settings 'SPUG' with fe_degree 2 (ScalarTransportSolver.py:259-270 is degree-agnostic; round 4): every test function
    is q + tau (v . grad q) - diffusion (with the Hessian of q), advection, capacity, body source, HTC and flux boundary terms.
    source = field (round 5): an Expression body source, i.e. its CG2 interpolant times Tq (ScalarTransportSolver.py:213-226, 259-276)."""
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    from oracle import ns_oracle as nso
    vel, pe, rho_cp, k = (0.8, -0.5, 0.3), 5.0, 2.0 * 3.0, 0.6
    m = UnitCubeMesh(3, 3, 2)
    Q = FunctionSpace(m, "CG", 2)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(360)}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0.0)), 'boundary_id': 2, 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}
    bcs["side"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 1.0)), 'boundary_id': 3, 'type': 'heatFlux', 'value': Constant(36.0)}
    st = {'solver_name': 'x', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs,
          'body_source': 7.0 if source == "constant" else Expression("7.0 + 30.0*x[0] - 20.0*x[1]*x[2]", degree=2),
          'initial_values': {'temperature': 300}, 'convective_velocity': Constant(vel),
          'advection_settings': {'stabilization_method': 'SPUG', 'Pe': pe},
          'material': {'density': 2.0, 'specific_heat_capacity': 3.0, 'thermal_conductivity': k},
          'solver_settings': {'transient_settings': {'transient': transient, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 0.3},
                              'reference_values': {'temperature': 300},
                              'solver_parameters': {'krylov_relative_tolerance': 1e-13, 'maximum_iterations': 20000}},
          'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    solver = ScalarTransportSolver(st)
    T = solver.solve().vector().array()
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    cdl = cd.astype(np.int64)
    n = len(co) + len(edges)
    facets, cell_facets, cnt = fo.facet_numbering(ce)
    fm = fo.mark_facets(co, ce, lambda x, ob: abs(x[1] - 1.0) < 3e-16, 1)
    fm = fo.mark_facets(co, ce, lambda x, ob: abs(x[1]) < 3e-16, 2, fm)
    fm = fo.mark_facets(co, ce, lambda x, ob: abs(x[0] - 1.0) < 3e-16, 3, fm)
    assert np.array_equal(fm, solver.boundary_facets.array())
    allb = np.array(nso.boundary_facet_cells(nso.TaylorHood(co, ce), lambda x: True)).reshape(-1, 2)

    def marked_cells(mid):
        return np.array([fc for fc in allb if fm[cell_facets[fc[0], fc[1]]] == mid]).reshape(-1, 2)

    def facet_load(mid, g):         # int g q ds = (facet mass matrix) x (g as a P2 function)
        return fo.assemble_p2_facet_mass(co, edges, facets, fm, mid, 1.0) @ np.full(n, g)
    A_op = fo.assemble_generic(n, cd, fo.p2_supg_system_local(co, ce, vel, pe, k * (0.5 if transient else 1.0), rho_cp,
                                                              rho_cp / 0.1 if transient else 0.0)).tocsr()
    R = fo.assemble_p2_facet_mass(co, edges, facets, fm, 2, 100.0)
    dA2, db2 = fo.p2_supg_facet_terms(co, ce, cdl, n, marked_cells(2), vel, pe, g=100.0 * 300.0, h=100.0)
    _, db3 = fo.p2_supg_facet_terms(co, ce, cdl, n, marked_cells(3), vel, pe, g=36.0)
    X = Q.node_coordinates()
    if source == "constant":
        src = fo.assemble_generic_vector(n, cd, fo.p2_supg_source_local(co, ce, vel, pe, 7.0))
    else:
        fn = 7.0 + 30.0 * X[:, 0] - 20.0 * X[:, 1] * X[:, 2]
        src = fo.assemble_generic_vector(n, cd, fo.p2_supg_source_local(co, ce, vel, pe, f_nodal=fn, cell_dofs=cd))
    load = src + facet_load(3, 36.0) + facet_load(2, 100.0 * 300.0) + db2 + db3
    top = np.nonzero(X[:, 1] == 1.0)[0]
    A = (A_op + R + dA2).tocsr()
    if not transient:
        ref = fo.solve_direct(*fo.apply_dirichlet(A, load, top, 360.0, False))
    else:
        B = fo.assemble_generic(n, cd, fo.p2_supg_system_local(co, ce, vel, pe, -0.5 * k, 0.0, rho_cp / 0.1)).tocsr()
        ref = np.full(n, 300.0)
        t = 0.0
        while t < 0.3:
            ref = fo.solve_direct(*fo.apply_dirichlet(A, B @ ref + load, top, 360.0, False))
            t += 0.1
    assert np.abs(T - ref).max() <= 1e-8 * np.abs(ref).max()
    st2 = dict(st)
    st2.pop('advection_settings')
    st2['function_space'] = FunctionSpace(UnitCubeMesh(3, 3, 2), "CG", 2)
    T2 = ScalarTransportSolver(st2).solve().vector().array()
    assert np.abs(T - T2).max() > 1e-3                # the stabilisation is not a no-op at this Peclet number


def test_cg2_box_operator_solved_in_lattice_order_gives_the_same_solve(gpu):
    """Option lattice_order (fs_lattice.hip, round 5): a scalar CG2 operator on a uniform box is permuted, inside the Krylov solve,
    into the x-fastest order of the half grid (vertices and edge mid-points interleaved as they sit in space, one dummy row per mesh
    line), solved there through the row-dictionary product with rounds of twelve runs, and permuted back - the API numbering is
    untouched.  Same Krylov iteration count, same solution to rounding, also from a nonzero guess and for a non-symmetric operator
    (BiCGStab); a box whose lines are shorter than a slice keeps the space's own numbering.  (Measured slower than the space's
    numbering at 10 M rows - the option is off by default; this pins its arithmetic.)
    In lattice order the product of the two-launch iteration is the TILE product (k_lattice_spmv: x through LDS windows, a wave per
    line parity, class lists broadcast): option lattice_check has every solve compare it with the work-item product on a vector of
    pseudo-random numbers, every row, bit for bit (a difference is an error of the solve); cg_fused = 0 keeps the two-launch iteration
    at this size, so that the iterations themselves run on the tile product and its fused dots."""
    import bench
    n = 32
    prob = bench.P2Problem(n, (0, n + 1), 2, 0, 1)
    prob.A.assemble(stiffness=20.0)
    prob.b.fill(0.0)
    prob.A.apply_dirichlet(prob.b, prob.dofs, prob.vals, symmetric=True)
    got = {}
    try:
        gpu.set_option("lattice_check", 1)
        for lat in (0, 1, 2):
            gpu.set_option("lattice_order", 1 if lat else 0)
            gpu.set_option("cg_fused", 0 if lat == 2 else -1)
            x = gpu.DeviceVector(prob.V.n_owned)
            st = gpu.krylov_solve(prob.A, prob.b, x, rtol=1e-10, max_iter=5000)
            x0 = x.get().copy()
            guess = x0 * (1.0 + 1e-3 * np.cos(np.arange(len(x0))))
            x.set(guess)
            st2 = gpu.krylov_solve(prob.A, prob.b, x, rtol=1e-10, max_iter=5000, nonzero_guess=True)
            y = gpu.DeviceVector(prob.V.n_owned)
            st3 = gpu.krylov_solve(prob.A, prob.b, y, rtol=1e-10, max_iter=5000, method="bicgstab")
            got[lat] = (st, x0, st2, x.get().copy(), st3, y.get().copy())
        small = bench.P2Problem(12, (0, 13), 2, 0, 1)          # 26 rows per line: no DIA slices in lattice order -> not used
        small.A.assemble(stiffness=20.0)
        small.b.fill(0.0)
        small.A.apply_dirichlet(small.b, small.dofs, small.vals, symmetric=True)
        st_small = gpu.krylov_solve(small.A, small.b, small.x, rtol=1e-10, max_iter=5000)
    finally:
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
        gpu.set_option("cg_fused", -1)
    (s2, x2, r2, g2, b2, y2) = got[2]           # the two-launch iteration on the tile product
    assert s2["lattice_order"] == 1 and s2["fused_iteration"] == 0 and s2["converged"] == 1
    assert abs(s2["iterations"] - got[0][0]["iterations"]) <= 1 and np.abs(x2 - got[0][1]).max() <= 1e-9 * np.abs(got[0][1]).max()
    assert r2["converged"] == 1 and np.abs(g2 - got[0][3]).max() <= 1e-8 * np.abs(got[0][1]).max()
    (s0, x0, r0, g0, b0, y0), (s1, x1, r1, g1, b1, y1) = got[0], got[1]
    assert s0["lattice_order"] == 0 and s1["lattice_order"] == 1 and r1["lattice_order"] == 1 and b1["lattice_order"] == 1
    assert s1["row_classes"] > 0 and s1["converged"] == 1 and abs(s1["iterations"] - s0["iterations"]) <= 1
    scale = np.abs(x0).max()
    assert np.abs(x1 - x0).max() <= 1e-9 * scale and np.abs(x1 - prob.exact_owned).max() <= 1e-6 * scale
    assert r1["converged"] == 1 and abs(r1["iterations"] - r0["iterations"]) <= 2 and np.abs(g1 - g0).max() <= 1e-8 * scale
    assert b1["converged"] == 1 and np.abs(y1 - y0).max() <= 1e-7 * scale
    assert st_small["lattice_order"] == 0 and st_small["converged"] == 1
    assert np.abs(small.x.get()[:small.n_owned] - small.exact_owned).max() <= 1e-6 * scale



def test_large_cg2_boxes_are_solved_in_lattice_order_by_default(gpu):
    """Automatic choice (option lattice_order = -1, the default since the tile product beat the work-item product of the space's own
    numbering - DESIGN.md section 3): from 270 000 rows on a scalar CG2 box operator is solved in the solver's lattice order, smaller
    ones in the space's numbering; the two give the same solve."""
    import bench
    n = 32                       # 65^3 = 274 625 rows (n = 31, 250 047 rows, keeps the space's numbering: measured slower there)
    prob = bench.P2Problem(n, (0, n + 1), 2, 0, 1)
    prob.A.assemble(stiffness=20.0)
    prob.b.fill(0.0)
    prob.A.apply_dirichlet(prob.b, prob.dofs, prob.vals, symmetric=True)
    try:
        gpu.set_option("lattice_check", 1)
        x1 = gpu.DeviceVector(prob.V.n_owned)
        s1 = gpu.krylov_solve(prob.A, prob.b, x1, rtol=1e-10, max_iter=5000)
        gpu.set_option("lattice_order", 0)
        x0 = gpu.DeviceVector(prob.V.n_owned)
        s0 = gpu.krylov_solve(prob.A, prob.b, x0, rtol=1e-10, max_iter=5000)
    finally:
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
    assert s1["lattice_order"] == 1 and s1["row_classes"] > 0 and s1["converged"] == 1
    assert s0["lattice_order"] == 0 and s0["converged"] == 1 and abs(s1["iterations"] - s0["iterations"]) <= 1
    scale = np.abs(x0.get()).max()
    assert np.abs(x1.get() - x0.get()).max() <= 1e-9 * scale
    small = bench.P2Problem(24, (0, 25), 2, 0, 1)      # 49^3 = 117 649 rows: the space's own numbering
    small.A.assemble(stiffness=20.0)
    small.b.fill(0.0)
    small.A.apply_dirichlet(small.b, small.dofs, small.vals, symmetric=True)
    st = gpu.krylov_solve(small.A, small.b, small.x, rtol=1e-10, max_iter=5000)
    assert st["lattice_order"] == 0 and st["converged"] == 1


def test_tile_product_on_lines_longer_than_a_tile(gpu):
    """k_lattice_spmv takes the strips of interior lines as one long line per line number once a mesh line is longer than a tile
    (138 rows at n = 68: 35 tiles a plane instead of 64): a tile that runs over the end of a line goes on at the start of the line four
    lines up, window and rows alike.  Option lattice_check compares every row of the tile product with the work-item product bit for
    bit (a difference fails the solve); the solve agrees with the one in the space's own numbering, and a second solve on the same
    matrix - its dictionary and with it the tile lists KEPT - gives the first one's iterations and solution exactly."""
    import bench
    n = 68                      # 137^3 = 2 571 353 rows, 138 to a line of the lattice order
    prob = bench.P2Problem(n, (0, n + 1), 2, 0, 1)
    prob.A.assemble(stiffness=20.0)
    prob.b.fill(0.0)
    prob.A.apply_dirichlet(prob.b, prob.dofs, prob.vals, symmetric=True)
    try:
        gpu.set_option("lattice_check", 1)
        gpu.set_option("cg_fused", 0)
        x1 = gpu.DeviceVector(prob.V.n_owned)
        s1 = gpu.krylov_solve(prob.A, prob.b, x1, rtol=1e-10, max_iter=5000)
        x2 = gpu.DeviceVector(prob.V.n_owned)
        s2 = gpu.krylov_solve(prob.A, prob.b, x2, rtol=1e-10, max_iter=5000)
        gpu.set_option("lattice_check", 0)
        x3 = gpu.DeviceVector(prob.V.n_owned)
        s3 = gpu.krylov_solve(prob.A, prob.b, x3, rtol=1e-10, max_iter=5000)
        gpu.set_option("lattice_order", 0)
        x0 = gpu.DeviceVector(prob.V.n_owned)
        s0 = gpu.krylov_solve(prob.A, prob.b, x0, rtol=1e-10, max_iter=5000)
    finally:
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
        gpu.set_option("cg_fused", -1)
    assert s1["lattice_order"] == 1 and s1["row_classes"] > 0 and s1["converged"] == 1 and s1["fused_iteration"] == 0
    assert s0["lattice_order"] == 0 and s0["converged"] == 1 and abs(s1["iterations"] - s0["iterations"]) <= 1
    scale = np.abs(x0.get()).max()
    assert np.abs(x1.get() - x0.get()).max() <= 1e-9 * scale
    assert np.abs(x1.get()[:prob.n_owned] - prob.exact_owned).max() <= 1e-6 * scale
    for s, x in ((s2, x2), (s3, x3)):
        assert s["lattice_order"] == 1 and s["classes_kept"] == 1 and s["iterations"] == s1["iterations"]
        assert np.array_equal(x.get(), x1.get())


def test_tile_product_on_a_box_that_is_not_a_cube(gpu):
    """The tile product on 70 x 26 x 18 cells (142 x 53 x 37 rows in lattice order: lines longer than a tile, a last strip of one
    line, a last plane of tiles with one plane): every row against the work-item product bit for bit (lattice_check), the solve
    against the one in the space's own numbering and against the exact linear profile along x."""
    nx, ny, nz = 70, 26, 18
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 1, degree=2)
    xyz, cells, _ = mesh.get()
    edges = V.edges().astype(np.int64)
    nv = len(xyz)
    cv = xyz[:, 0]
    ce = 0.5 * (xyz[edges[:, 0], 0] + xyz[edges[:, 1], 0])
    co = np.concatenate([cv, ce])
    lo, hi = np.flatnonzero(co == 0.0), np.flatnonzero(co == 1.0)
    dofs = np.concatenate([lo, hi]).astype(np.int32)
    vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
    A = gpu.DeviceMatrix(V)
    b = gpu.DeviceVector(V.n_owned)
    A.assemble(stiffness=20.0)
    b.fill(0.0)
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    got = {}
    try:
        gpu.set_option("lattice_check", 1)
        gpu.set_option("cg_fused", 0)
        for lat in (1, 0):
            gpu.set_option("lattice_order", lat)
            x = gpu.DeviceVector(V.n_owned)
            got[lat] = (gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000), x.get().copy())
    finally:
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
        gpu.set_option("cg_fused", -1)
    (s1, x1), (s0, x0) = got[1], got[0]
    assert s1["lattice_order"] == 1 and s1["row_classes"] > 0 and s1["converged"] == 1 and s1["fused_iteration"] == 0
    assert s0["lattice_order"] == 0 and s0["converged"] == 1 and abs(s1["iterations"] - s0["iterations"]) <= 1
    scale = np.abs(x0).max()
    assert np.abs(x1 - x0).max() <= 1e-9 * scale
    assert np.abs(x1[:V.n_owned] - (350.0 - 50.0 * co[:V.n_owned])).max() <= 1e-6 * scale


def test_two_lattice_ordered_operators_solved_in_turn(gpu):
    """The tile product's tables (class lists, per-tile class table, launch geometry) belong to the operator solved last: two CG2 box
    operators of different shape solved in turn - each solve rebuilds them, re-captures its batch of launches on them (the tables are
    part of the captured launches' identity), checks the tile product against the work-item product bit for bit - give, the second
    time round, the first round's iterations and solutions exactly."""
    import bench
    cube = bench.P2Problem(32, (0, 33), 2, 0, 1)
    cube.A.assemble(stiffness=20.0)
    cube.b.fill(0.0)
    cube.A.apply_dirichlet(cube.b, cube.dofs, cube.vals, symmetric=True)
    mesh = gpu.DeviceMesh.box(70, 26, 18, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 1, degree=2)
    xyz, _, _ = mesh.get()
    edges = V.edges().astype(np.int64)
    co = np.concatenate([xyz[:, 0], 0.5 * (xyz[edges[:, 0], 0] + xyz[edges[:, 1], 0])])
    lo, hi = np.flatnonzero(co == 0.0), np.flatnonzero(co == 1.0)
    A = gpu.DeviceMatrix(V)
    b = gpu.DeviceVector(V.n_owned)
    A.assemble(stiffness=20.0)
    b.fill(0.0)
    A.apply_dirichlet(b, np.concatenate([lo, hi]).astype(np.int32), np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), symmetric=True)
    runs = []
    try:
        gpu.set_option("lattice_check", 1)
        gpu.set_option("lattice_order", 1)
        gpu.set_option("cg_fused", 0)
        for rnd in range(2):
            for (AA, bb, nn) in ((cube.A, cube.b, cube.V.n_owned), (A, b, V.n_owned)):
                x = gpu.DeviceVector(nn)
                st = gpu.krylov_solve(AA, bb, x, rtol=1e-10, max_iter=5000)
                runs.append((st, x.get().copy()))
    finally:
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
        gpu.set_option("cg_fused", -1)
    for st, _ in runs:
        assert st["lattice_order"] == 1 and st["converged"] == 1 and st["row_classes"] > 0
    for k in (0, 1):
        assert runs[k][0]["iterations"] == runs[k + 2][0]["iterations"] and np.array_equal(runs[k][1], runs[k + 2][1])
    assert np.abs(runs[0][1][:cube.n_owned] - cube.exact_owned).max() <= 1e-6 * 350.0
    assert np.abs(runs[1][1][:V.n_owned] - (350.0 - 50.0 * co[:V.n_owned])).max() <= 1e-6 * 350.0


def test_marching_window_product_of_cg2_boxes_against_the_tile_product(gpu):
    """k_lat_march (fs_latmarch.h, round 6): in the solver's lattice order the product of a scalar CG2 box operator marches LDS
    windows through the lattice planes - loop structure from the compile-time parity stencils (fs_cg2_stencil.h), a wave per mesh line,
    the coefficients of a (line, step) one contiguous step list, the ends of the lines by the column tiles of the tile product in the
    same launch.  Option lattice_march = 0 keeps the tile product k_lattice_spmv.  Both against the work-item product bit for bit
    on every row (lattice_check: a difference fails the solve), the solves against each other and against the exact profile: a box of
    142 x 53 x 37 lattice rows (lines of two 64-pair pieces, a last patch of four lines, chunks of planes of unequal length), one of
    83 x 57 x 45 (lines of a single piece, Dirichlet values on the faces across the mesh lines' direction: the classes change along
    the march), a cube of 137 rows a side, and a box with lines of three pieces (262 rows: two waves a line)."""
    out = {}
    boxes = {"142 x 53 x 37": (70, 26, 18, 0), "83 x 57 x 45": (41, 28, 22, 2), "137^3": (68, 68, 68, 1), "262 x 17 x 17": (130, 8, 8, 1)}
    try:
        gpu.set_option("lattice_check", 1)
        gpu.set_option("lattice_order", 1)
        gpu.set_option("cg_fused", 0)
        for name, (nx, ny, nz, axis) in boxes.items():
            mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
            V = gpu.DeviceSpace(mesh, 1, degree=2)
            xyz, cells, _ = mesh.get()
            edges = V.edges().astype(np.int64)
            co = np.concatenate([xyz[:, axis], 0.5 * (xyz[edges[:, 0], axis] + xyz[edges[:, 1], axis])])
            lo, hi = np.flatnonzero(co == 0.0), np.flatnonzero(co == 1.0)
            dofs = np.concatenate([lo, hi]).astype(np.int32)
            vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
            A = gpu.DeviceMatrix(V)
            b = gpu.DeviceVector(V.n_owned)
            A.assemble(stiffness=20.0)
            b.fill(0.0)
            A.apply_dirichlet(b, dofs, vals, symmetric=True)
            for march in (1, 0):
                gpu.set_option("lattice_march", march)
                x = gpu.DeviceVector(V.n_owned)
                st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
                out[(name, march)] = (st, x.get()[:V.n_owned].copy(), 350.0 - 50.0 * co[:V.n_owned])
    finally:
        gpu.set_option("lattice_march", 1)
        gpu.set_option("lattice_order", -1)
        gpu.set_option("lattice_check", 0)
        gpu.set_option("cg_fused", -1)
    for name in boxes:
        (s1, x1, exact), (s0, x0, _) = out[(name, 1)], out[(name, 0)]
        assert s1["product_kind"] == 5 and s0["product_kind"] == 2, (name, s1["product_kind"], s0["product_kind"])
        assert s1["lattice_order"] == 1 and s0["lattice_order"] == 1 and s1["converged"] == 1 and s0["converged"] == 1, name
        assert abs(s1["iterations"] - s0["iterations"]) <= 1, name
        scale = np.abs(x0).max()
        assert np.abs(x1 - x0).max() <= 1e-9 * scale and np.abs(x1 - exact).max() <= 1e-6 * scale, name
