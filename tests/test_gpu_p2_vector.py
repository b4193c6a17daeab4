"""Vector P2 (CG2 nodes, 3 unknowns per node): the space of the reference's own elasticity example
(/root/reference/examples/test_linear_elasticity.py:105-106, VectorFunctionSpace(mesh, 'CG', 2)).  HIP kernels through
the C-ABI against the oracle's 30x30 element matrices (oracle/fem_oracle.py: p2_elasticity_local, pinned on the host by
tests/test_host_api.py against the exact sympy reference-tet matrix and patch tests)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu
E, NU = 2e11, 0.27


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def _space(gpu, co, ce):
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 3, degree=2)
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    assert np.array_equal(V.edges(), edges)
    assert V.n_owned == 3 * (len(co) + len(edges))
    return mesh, V, cd, edges


def test_p2_elasticity_matrix_equals_oracle_and_kills_rigid_body_modes(gpu, data_dir):
    for co, ce in (fo.box_mesh((0, 0, 0), (4.0, 1.0, 1.0), 4, 2, 2), fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))):
        mesh, V, cd, edges = _space(gpu, co, ce)
        A = gpu.DeviceMatrix(V)
        A.assemble(lame=fo.lame(E, NU))
        M = _csr(A)
        R, _, _ = fo.assemble_p2_elasticity(co, ce, E, NU)
        assert M.shape == R.shape
        assert abs(M - R).max() <= 1e-12 * abs(R).max()
        # a second assembly on top (add) doubles it; with a mass term the diagonal blocks gain rho * M_P2
        A.assemble(lame=fo.lame(E, NU), add=True)
        assert abs(_csr(A) - 2 * R).max() <= 1e-12 * abs(R).max()
        A.assemble(lame=fo.lame(E, NU), mass=7800.0)
        Mm = fo.assemble_generic(len(co) + len(edges), cd, fo.p2_mass_local(co, ce, 7800.0))
        Rm = R + sp.kron(Mm, sp.identity(3))
        assert abs(_csr(A) - Rm).max() <= 1e-12 * abs(Rm).max()
        # rigid-body modes evaluated at the P2 nodes are in the kernel of K (Appendix C5)
        A.assemble(lame=fo.lame(E, NU))
        X = fo.p2_dof_coordinates(co, edges)
        x = gpu.DeviceVector(V.n_local)
        y = gpu.DeviceVector(V.n_owned)
        for r in fo.rigid_body_modes(X):
            x.set(r)
            A.spmv(x, y)
            assert np.abs(y.get()).max() <= 1e-10 * abs(R).max() * np.abs(r).max()


def test_p2_vector_load_vectors_equal_oracle(gpu):
    co, ce = fo.box_mesh((0, 0, 0), (2.0, 1.0, 1.5), 3, 2, 2)
    mesh, V, cd, edges = _space(gpu, co, ce)
    n = len(co) + len(edges)
    b = gpu.DeviceVector(V.n_owned)
    f = (3.0, -2.0, 7800.0 * 9.81)
    gpu.assemble_vector(V, b, vector_value=f)
    ref = fo.assemble_p2_vector_source(co, ce, f)
    assert np.abs(b.get() - ref).max() <= 1e-13 * np.abs(ref).max()
    # thermal-stress load int c div v dx: constant c, per-cell c, P1 c through its vertex values
    vd = fo.p2_vector_cell_dofs(cd)
    gpu.assemble_vector(V, b, div_coef=2.5)
    ref = fo.assemble_generic_vector(3 * n, vd, fo.p2_div_load_local(co, ce, c_const=2.5))
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    rng = np.random.default_rng(3)
    cv = rng.uniform(1.0, 2.0, len(co))
    nodal = np.concatenate([cv, np.zeros(len(edges))])        # nodal array over the space's nodes; vertex entries are used
    gpu.assemble_vector(V, b, div_coef=("nodal", nodal))
    ref = fo.assemble_generic_vector(3 * n, vd, fo.p2_div_load_local(co, ce, c_vertex=cv))
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    # both together, added to what is there
    gpu.assemble_vector(V, b, vector_value=f, div_coef=2.5, add=True)
    ref = ref + fo.assemble_p2_vector_source(co, ce, f) + fo.assemble_generic_vector(3 * n, vd, fo.p2_div_load_local(co, ce, c_const=2.5))
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    # traction int g . v ds on the face x = x_max
    facets = fo.facet_numbering(ce)[0]
    fm = fo.mark_facets(co, ce, lambda x, on: on and abs(x[0] - 2.0) < 1e-12, 5)
    tri = facets[fm == 5]
    g = (1e3, 0.0, -4e3)
    b.fill(0.0)
    gpu.assemble_facet_vector(V, b, tri, g)
    ref = fo.assemble_p2_facet_vector_load(co, edges, facets, fm, 5, g)
    assert np.abs(b.get() - ref).max() <= 1e-13 * np.abs(ref).max()


def test_p2_cantilever_solve_equals_oracle_direct_solve(gpu):
    """The reference example's set-up in small: box 10 x 1 x 1, left face clamped, right face displaced by (0, 0, 1e-3)
    (examples/test_linear_elasticity.py:42-62, 112-129), plus a body force."""
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 8, 2, 2)
    mesh, V, cd, edges = _space(gpu, co, ce)
    X = fo.p2_dof_coordinates(co, edges)
    n = len(X)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=fo.lame(E, NU))
    b = gpu.DeviceVector(V.n_owned)
    f = (0.0, 0.0, -7800.0 * 10.0)
    gpu.assemble_vector(V, b, vector_value=f)
    facets = fo.facet_numbering(ce)[0]
    fm = fo.mark_facets(co, ce, lambda x, on: on and abs(x[0]) < 1e-12, 1)
    fm = fo.mark_facets(co, ce, lambda x, on: on and abs(x[0] - 10.0) < 1e-12, 2, markers=fm)
    left = fo.p2_facet_dofs(len(co), edges, facets, fm, 1).astype(np.int64)
    right = fo.p2_facet_dofs(len(co), edges, facets, fm, 2).astype(np.int64)
    assert np.allclose(X[left, 0], 0.0) and np.allclose(X[right, 0], 10.0)
    dofs = np.concatenate([(left[:, None] * 3 + np.arange(3)).ravel(), (right[:, None] * 3 + np.arange(3)).ravel()])
    vals = np.concatenate([np.zeros(3 * len(left)), np.tile([0.0, 0.0, 1e-3], len(right))])
    A.apply_dirichlet(b, dofs.astype(np.int32), vals, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=1e-12, max_iter=50000, norm="preconditioned")
    assert st["converged"] == 1
    R, _, _ = fo.assemble_p2_elasticity(co, ce, E, NU)
    Ab, bb = fo.apply_dirichlet(R, fo.assemble_p2_vector_source(co, ce, f), dofs, vals, True)
    ref = fo.solve_direct(Ab, bb)
    assert np.abs(x.get() - ref).max() <= 1e-7 * np.abs(ref).max()
    assert abs(x.get().reshape(n, 3)[right, 2] - 1e-3).max() <= 1e-15


# ---- through the solver API ------------------------------------------------------------------------------------------
QUIET = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}


def _example_solver(nx, ny, nz, degree=2, thermal=True, body=True, **extra):
    """examples/test_linear_elasticity.py:42-129 on an nx x ny x nz box."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, Constant, Expression, SubDomain, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver

    class Left(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0)

    class Right(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 10)
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), nx, ny, nz)
    st = copy.deepcopy(SB.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': E, 'poisson_ratio': NU, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", degree)
    bcs = OrderedDict()
    bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
    bcs["displ"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant((0, 0, 1e-3))}
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['solver_settings']['solver_parameters'] = {'relative_tolerance': 1e-12, 'maximum_iterations': 100000}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = Expression("343", degree=degree) if thermal else None
    if body:
        st['body_source'] = Expression(("10*rho", "0", "0.0"), omega=100, rho=7800, degree=2)
    st.update(extra)
    return LinearElasticitySolver(st)


def _oracle_example(co, ce, thermal=True, body=True):
    """The same problem restated on the oracle: bug-compatible load sign (Appendix B-Q3), thermal term conventional."""
    R, cd, edges = fo.assemble_p2_elasticity(co, ce, E, NU)
    n = len(co) + len(edges)
    b = np.zeros(3 * n)
    if body:
        b -= fo.assemble_p2_vector_source(co, ce, (78000.0, 0.0, 0.0))
    if thermal:
        c = E / (1 - 2 * NU) * 2e-6 * (343.0 - 293.0)
        b += fo.assemble_generic_vector(3 * n, fo.p2_vector_cell_dofs(cd), fo.p2_div_load_local(co, ce, c_const=c))
    facets = fo.facet_numbering(ce)[0]
    fm = fo.mark_facets(co, ce, lambda x, on: abs(x[0]) < 1e-12, 1)
    fm = fo.mark_facets(co, ce, lambda x, on: abs(x[0] - 10.0) < 1e-12, 2, markers=fm)
    left = fo.p2_facet_dofs(len(co), edges, facets, fm, 1).astype(np.int64)
    right = fo.p2_facet_dofs(len(co), edges, facets, fm, 2).astype(np.int64)
    dofs = np.concatenate([left * 3, (right[:, None] * 3 + np.arange(3)).ravel()])
    vals = np.concatenate([np.zeros(len(left)), np.tile([0.0, 0.0, 1e-3], len(right))])
    Ab, bb = fo.apply_dirichlet(R, b, dofs, vals, True)
    return fo.solve_direct(Ab, bb), cd, edges


def test_reference_example_p2_small_equals_oracle(gpu):
    solver = _example_solver(8, 2, 2)
    u = solver.solve()
    assert solver.function_space.degree() == 2 and u.vector().size() == solver.function_space.dim()
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 8, 2, 2)
    ref, cd, edges = _oracle_example(co, ce)
    assert np.array_equal(solver.function_space.edge_nodes(), edges)
    assert np.abs(u.vector().array() - ref).max() <= 1e-6 * np.abs(ref).max()
    # consistent L2 projection of the von Mises stress (LinearElasticitySolver.py:71-76) against the oracle's sparse LU
    vm = solver.von_Mises(u).vector().array()
    vm_ref, _ = fo.von_mises_projection(co, ce, u.vector().array().reshape(-1, 3), E, NU, degree=2, cell_dofs=cd)
    assert np.abs(vm - vm_ref).max() <= 1e-8 * np.abs(vm_ref).max()
    assert vm.shape == (len(co),) and vm.max() > 0


def test_von_mises_projection_p1_equals_oracle(gpu):
    solver = _example_solver(8, 2, 2, degree=1)
    u = solver.solve()
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 8, 2, 2)
    vm = solver.von_Mises(u).vector().array()
    vm_ref, b_ref = fo.von_mises_projection(co, ce, u.vector().array().reshape(-1, 3), E, NU, degree=1)
    assert np.abs(vm - vm_ref).max() <= 1e-9 * np.abs(vm_ref).max()
    # a consistent projection is NOT the lumped vertex average round 1 computed
    s = solver.sigma(u)
    dev = s - np.trace(s, axis1=1, axis2=2)[:, None, None] / 3.0 * np.eye(3)
    vmc = np.sqrt(1.5 * np.einsum("cij,cij->c", dev, dev))
    X = co[ce.astype(np.int64)]
    vol = np.abs(np.linalg.det(np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=2))) / 6.0
    num, den = np.zeros(len(co)), np.zeros(len(co))
    np.add.at(num, ce.ravel(), np.repeat(vmc * vol, 4))
    np.add.at(den, ce.ravel(), np.repeat(vol, 4))
    assert np.abs(vm - num / den).max() > 1e-3 * np.abs(vm).max()
    # ... but both conserve the integral of vm
    M = fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, 1.0))
    assert abs((M @ vm).sum() - (vmc * vol).sum()) <= 1e-9 * (vmc * vol).sum()


def test_reference_example_p2_full_size(gpu):
    """The example's own mesh (40 x 10 x 10 cells, 41 x 11 x 11 vertices + 30 870... edge nodes): runs, meets its Dirichlet
    data, and agrees with the oracle restated on the same mesh in the energy norm of the load-free part."""
    solver = _example_solver(40, 10, 10, thermal=True, body=True)
    u = solver.solve()
    V = solver.function_space
    X = V.node_coordinates()
    U = u.node_values()
    assert U.shape == (V.num_nodes(), 3) and V.num_nodes() == 81 * 21 * 21
    assert np.abs(U[np.isclose(X[:, 0], 10.0)] - [0.0, 0.0, 1e-3]).max() <= 1e-15
    assert np.abs(U[np.isclose(X[:, 0], 0.0), 0]).max() == 0.0
    st = solver.last_solve_stats
    assert st["converged"] == 1 and st["true_rel_residual"] <= 1e-8
    # solve_amg as in the reference (LinearElasticitySolver.py:247-250): CG + smoothed aggregation on the P2 operator, rigid-body
    # modes evaluated at the vertices AND the edge mid-points; Jacobi-CG needs about 3 000 iterations on this mesh
    assert st["amg_levels"] >= 2 and st["iterations"] <= 80
    # residual of the discrete equilibrium equations against the oracle's operator (free dofs)
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 40, 10, 10)
    R, cd, edges = fo.assemble_p2_elasticity(co, ce, E, NU)
    n = len(co) + len(edges)
    b = -fo.assemble_p2_vector_source(co, ce, (78000.0, 0.0, 0.0))
    b += fo.assemble_generic_vector(3 * n, fo.p2_vector_cell_dofs(cd), fo.p2_div_load_local(co, ce, c_const=E / (1 - 2 * NU) * 2e-6 * 50.0))
    r = R @ u.vector().array() - b
    free = np.ones(3 * n, dtype=bool)
    free[3 * np.nonzero(np.isclose(X[:, 0], 0.0))[0]] = False
    free[(3 * np.nonzero(np.isclose(X[:, 0], 10.0))[0][:, None] + np.arange(3)).ravel()] = False
    assert np.linalg.norm(r[free]) <= 1e-7 * np.linalg.norm(b)


def test_elastodynamics_inertia_term(gpu):
    """solving_dynamics = True (LinearElasticitySolver.py:216-220): from the second step on the explicit acceleration of
    SolverBase.get_acceleration enters the right-hand side as rho M a; checked against the oracle's consistent mass matrix."""
    import scipy.sparse as sps
    tr = {'transient': True, 'starting_time': 0.0, 'time_step': 1e-3, 'ending_time': 3e-3}
    solver = _example_solver(6, 2, 2, degree=1, thermal=False, body=True)
    solver.transient_settings = tr
    solver.solving_dynamics = True
    solver.init_solver()
    rng = np.random.default_rng(5)
    n = solver.function_space.dim()
    for f in (solver.w_current, solver.w_prev, solver.w_pp):
        f.vector().set_local(1e-4 * rng.standard_normal(n))
    F, bcs = solver.generate_form(1, None, None, solver.w_current, solver.w_prev)
    assert F.inertia is not None and F.inertia[0] == 7800.0
    A, b = solver.assemble_system(F, [], symmetric=True)
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 6, 2, 2)
    M = sps.kron(fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, 7800.0)), sps.identity(3)).tocsr()
    dt = 1e-3
    w0, w1, w2 = solver.w_current.vector().array(), solver.w_prev.vector().array(), solver.w_pp.vector().array()
    accel = ((w0 - w1) / dt - (w1 - w2) / dt) / (1.0 / dt)                 # the reference's own scaling (SolverBase.py:477-482)
    ref = M @ accel - fo.assemble_p1_vector_source(co, ce, (78000.0, 0.0, 0.0))
    assert np.abs(b.get() - ref).max() <= 1e-11 * np.abs(ref).max()


def test_p2_amg_device_modes_equal_the_host_near_null_space(gpu):
    """The six rigid-body modes built on the device for a CG2 space (vertices + edge mid-points) span what
    SolverBase.build_nullspace returns; both hierarchies solve the cantilever in the same number of iterations, and the
    operator annihilates the modes on the interior rows."""
    solver = _example_solver(12, 3, 3, thermal=False, body=True)
    solver.init_solver()
    F, bcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    A, b = solver.assemble_system(F, bcs, symmetric=True)
    V = solver.function_space
    ns = solver.build_nullspace(V)
    assert ns.shape == (6, V.dim())
    assert np.abs(ns @ ns.T - np.eye(6)).max() <= 1e-12
    its = []
    for near in ("rigid_body", ns):
        H = gpu.AMG(A, nullspace=near)
        x = gpu.DeviceVector(V.device().n_local)
        st = H.solve(b, x, rtol=1e-10, max_iter=200)
        assert st["converged"] == 1 and H.info()["levels"] >= 2
        its.append(st["iterations"])
        H.close()
    assert abs(its[0] - its[1]) <= 2 and its[0] <= 60
    A0, _ = solver.assemble_system(F, [], symmetric=True)           # no Dirichlet rows: K B = 0
    y = gpu.DeviceVector(V.dim())
    for k in range(6):
        A0.spmv(gpu.DeviceVector(V.dim(), ns[k]), y)
        assert np.abs(y.get()).max() <= 1e-9 * 2e11 * np.abs(ns[k]).max()


@pytest.mark.parametrize("degree", [1, 2])
def test_spatially_varying_body_force(gpu, degree):
    """A body_source Expression that varies in space (a centrifugal load): its interpolant in the displacement space times the
    consistent mass matrix - FFC's quadrature for an Expression of the element's degree - against the oracle's direct solve."""
    import scipy.sparse as sps
    from fenicssolver_amd.fem import Expression
    solver = _example_solver(8, 2, 2, degree=degree, thermal=False, body=False)
    solver.settings['body_source'] = Expression(("rho*omega*omega*x[0]", "rho*omega*omega*x[1]", "-9.8*rho"), rho=7800.0, omega=30.0, degree=degree)
    solver.body_source = solver.settings['body_source']
    u = solver.solve().vector().array()
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 8, 2, 2)
    V = solver.function_space
    X = V.node_coordinates()
    f = np.stack([7800.0 * 900.0 * X[:, 0], 7800.0 * 900.0 * X[:, 1], np.full(len(X), -9.8 * 7800.0)], axis=1)
    if degree == 1:
        R = fo.assemble_p1_elasticity(co, ce, E, NU)
        M = sps.kron(fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, 1.0)), sps.identity(3), format="csr")
        left = np.nonzero(np.isclose(co[:, 0], 0.0))[0]
        right = np.nonzero(np.isclose(co[:, 0], 10.0))[0]
    else:
        R, cd, edges = fo.assemble_p2_elasticity(co, ce, E, NU)
        M = sps.kron(fo.assemble_generic(len(X), cd, fo.p2_mass_local(co, ce, 1.0)), sps.identity(3), format="csr")
        left = np.nonzero(np.isclose(X[:, 0], 0.0))[0]
        right = np.nonzero(np.isclose(X[:, 0], 10.0))[0]
    b = -(M @ f.reshape(-1))                                         # the reference ADDS its loads to F (Appendix B-Q3)
    dofs = np.concatenate([left * 3, (right[:, None] * 3 + np.arange(3)).ravel()])
    vals = np.concatenate([np.zeros(len(left)), np.tile([0.0, 0.0, 1e-3], len(right))])
    ref = fo.solve_direct(*fo.apply_dirichlet(R, b, dofs, vals, True))
    assert np.abs(u - ref).max() <= 1e-6 * np.abs(ref).max()
    assert np.abs(ref).max() > 2e-3                                  # the load matters next to the 1 mm end displacement


@pytest.mark.parametrize("degree", [1, 2])
def test_pressure_boundary_that_varies_in_space(gpu, degree):
    """'pressure' with an Expression value (a hydrostatic load on the top face): n * p with p through its P1 interpolant on
    every facet, integrated against the P1 / P2 test functions - checked with the 6-point degree-4 rule on every facet."""
    from fenicssolver_amd.fem import Expression, SubDomain, near
    from oracle import ns_oracle as nso

    class Top(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[2], 1.0)
    solver = _example_solver(6, 2, 2, degree=degree, thermal=False, body=False)
    solver.boundary_conditions["load"] = {'boundary': Top(), 'boundary_id': 3, 'type': 'pressure', 'value': Expression("1e6*(1+0.3*x[0])", degree=1)}
    solver.settings['boundary_conditions'] = solver.boundary_conditions
    solver.generate_boundary_facets() if hasattr(solver, 'generate_boundary_facets') else None
    u = solver.solve().vector().array()
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 6, 2, 2)
    V = solver.function_space
    X = V.node_coordinates()
    n = len(X)
    facets, _, cnt = fo.facet_numbering(ce)
    top = facets[(cnt == 1) & np.all(co[facets.astype(np.int64)][:, :, 2] == 1.0, axis=1)].astype(np.int64)
    area = fo.facet_areas(co, top)
    pvert = 1e6 * (1 + 0.3 * co[:, 0])
    b = np.zeros(3 * n)
    if degree == 2:
        edges = V.edge_nodes().astype(np.int64)
        emap = {(int(a), int(c)): len(co) + k for k, (a, c) in enumerate(edges)}
    for f, tri in enumerate(top):
        nodes = list(tri) + ([emap[tuple(sorted((int(tri[i]), int(tri[j]))))] for i, j in ((0, 1), (0, 2), (1, 2))] if degree == 2 else [])
        for bary, w in zip(nso.TRI_QP, nso.TRI_QW):
            pq = pvert[tri] @ bary
            if degree == 1:
                phi = bary
            else:
                phi = np.concatenate([bary * (2 * bary - 1), [4 * bary[0] * bary[1], 4 * bary[0] * bary[2], 4 * bary[1] * bary[2]]])
            for a, node in enumerate(nodes):
                b[3 * node + 2] += -1.0 * w * area[f] * pq * phi[a] * 1.0          # outward normal (0,0,1); loads are ADDED to F (B-Q3)
    if degree == 1:
        R = fo.assemble_p1_elasticity(co, ce, E, NU)
    else:
        R, cd, _ = fo.assemble_p2_elasticity(co, ce, E, NU)
    left = np.nonzero(np.isclose(X[:, 0], 0.0))[0]
    right = np.nonzero(np.isclose(X[:, 0], 10.0))[0]
    dofs = np.concatenate([left * 3, (right[:, None] * 3 + np.arange(3)).ravel()])
    vals = np.concatenate([np.zeros(len(left)), np.tile([0.0, 0.0, 1e-3], len(right))])
    ref = fo.solve_direct(*fo.apply_dirichlet(R, b, dofs, vals, True))
    assert np.abs(u - ref).max() <= 1e-6 * np.abs(ref).max()
