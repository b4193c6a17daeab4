"""CPU tests (no GPU): the oracle against the reference's known answers, the host-side data
model and form recognition, and the C-ABI surface of libfsamd.so."""
import json
import os
import re

import numpy as np
import pytest

from oracle import fem_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- oracle vs known answers
def test_facet_numbering_reproduces_reference_markers(data_dir):
    """lexicographic facet numbering puts the 100+100 marked facets of
    data/mesh_facet_region.xml exactly on z=0 (id 1) and z=20 (id 2)."""
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    dim, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, cell_facets, cnt = fo.facet_numbering(ce)
    assert (len(co), len(ce), len(facets), dim) == (1069, 4355, 9410, 2)
    assert int((cnt == 1).sum()) == 1400
    assert np.bincount(fm).tolist() == [9210, 100, 100]
    for mid, z in ((1, 0.0), (2, 20.0)):
        sel = fm == mid
        assert np.all(co[facets[sel]][:, :, 2] == z) and np.all(cnt[sel] == 1)
    dimc, cm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_physical_region.xml"))
    assert dimc == 3 and np.all(cm == 3) and len(cm) == 4355
    # golden fixture: vertex triples of the marked facets
    gold = np.load(os.path.join(ROOT, "tests", "golden", "config1_marked_facets.npz"))
    assert np.array_equal(facets[fm == 1], gold["id1"]) and np.array_equal(facets[fm == 2], gold["id2"])
    # mesh sanity: volume 10*5*20, Euler characteristic 1
    detJ, _ = fo.p1_geometry(co, ce)
    assert np.all(detJ != 0) and abs(np.abs(detJ).sum() / 6 - 1000.0) < 1e-9
    edges, _ = fo.edge_numbering(ce)
    assert len(co) - len(edges) + len(facets) - len(ce) == 1


def test_config1_exact_solution(data_dir):
    """data/TestHeatTransfer.json: k=20, T=350 on id 1, 300 on id 2 -> T = 350 - 2.5 z (C2, C8)."""
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    A = fo.assemble_p1_scalar(co, ce, 20.0)
    assert A.nnz == 13315 and abs(A - A.T).max() < 1e-12 and abs(A.sum(axis=1)).max() < 1e-11
    d1, d2 = fo.dirichlet_dofs_p1(facets, fm, 1), fo.dirichlet_dofs_p1(facets, fm, 2)
    assert len(d1) == 66 and len(d2) == 66
    dofs = np.concatenate([d1, d2])
    vals = np.concatenate([np.full(66, 350.0), np.full(66, 300.0)])
    exact = 350.0 - 2.5 * co[:, 2]
    for sym in (False, True):
        Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), dofs, vals, symmetric=sym)
        assert np.abs(fo.solve_direct(Ab, bb) - exact).max() < 1e-9
    Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), dofs, vals, symmetric=True)
    x, it, _ = fo.pcg_jacobi(Ab, bb, rtol=1e-8)
    assert it == 93 and np.abs(x - exact).max() < 1e-4
    x2, it2, _ = fo.pcg_jacobi_single_reduction(Ab, bb, rtol=1e-8)
    assert abs(it2 - it) <= 1 and np.abs(x2 - x).max() < 1e-6
    gold = np.load(os.path.join(ROOT, "tests", "golden", "config1_solution.npy"))
    assert np.abs(fo.solve_direct(Ab, bb) - gold).max() < 1e-9


def test_reference_tet_element_matrices():
    """Appendix C3: exact P1 stiffness and mass matrices on the reference tetrahedron."""
    co = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    ce = np.array([[0, 1, 2, 3]], dtype=np.int32)
    K = fo.p1_stiffness_local(co, ce, 1.0)[0]
    assert np.allclose(K * 6, [[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]], atol=1e-15)
    M = fo.p1_mass_local(co, ce, 1.0)[0]
    assert np.allclose(M * 120, np.ones((4, 4)) + np.eye(4), atol=1e-15)


def test_kuhn_cube_structure_counts():
    """Appendix C7: nnz/row of the P1 pattern on dolfin.UnitCubeMesh(n,n,n)."""
    for n, expect in ((8, 12.48), (16, 13.63)):
        co, ce = fo.unit_cube_mesh(n)
        rp, ci = fo.csr_pattern(len(co), ce)
        assert abs(len(ci) / len(co) - expect) < 0.01 and np.diff(rp).max() == 15
        assert len(ce) == 6 * n ** 3
    co, ce = fo.unit_cube_mesh(3)
    detJ, _ = fo.p1_geometry(co, ce)
    assert abs(np.abs(detJ).sum() / 6 - 1.0) < 1e-14


def test_patch_tests_and_nullspace():
    """C4/C5: linear fields are reproduced exactly; rigid-body modes are in the elasticity kernel."""
    co, ce = fo.box_mesh((0, 0, 0), (2.0, 1.0, 1.5), 3, 2, 2)
    A = fo.assemble_p1_scalar(co, ce, 3.0)
    lin = 1.0 + 2.0 * co[:, 0] - 0.5 * co[:, 1] + 0.25 * co[:, 2]
    on_b = np.nonzero(((co == 0) | (co == np.array([2.0, 1.0, 1.5]))).any(axis=1))[0]
    Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), on_b, lin[on_b], True)
    assert np.abs(fo.solve_direct(Ab, bb) - lin).max() < 1e-12
    K = fo.assemble_p1_elasticity(co, ce, 2e11, 0.27)
    for r in fo.rigid_body_modes(co):
        assert np.abs(K @ r).max() < 1e-9 * abs(K).max() * np.abs(r).max()
    assert abs(K - K.T).max() < 1e-6 * abs(K).max()


def test_c_oracle_agrees_with_numpy_oracle():
    from oracle import c_oracle as co_
    xyz, cells = co_.box_mesh(5, 4, 3, (0, 0, 0), (1, 2, 3))
    c2, e2 = fo.box_mesh((0, 0, 0), (1, 2, 3), 5, 4, 3)
    assert np.array_equal(xyz, c2) and np.array_equal(cells, e2)
    rp, ci = co_.csr_pattern(len(xyz), cells)
    rp2, ci2 = fo.csr_pattern(len(c2), e2)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    v = co_.assemble_p1(xyz, cells, 20.0, rp, ci)
    A = fo.assemble_p1_scalar(c2, e2, 20.0)
    assert np.abs(v - A.data).max() < 1e-13 * np.abs(A.data).max()
    P = fo.heat_box_problem(16)
    x, it, _ = fo.pcg_jacobi(P["A"], P["b"])
    r = co_.heat_box_solve(16, 16, 16)
    assert r["iterations"] == it and np.abs(r["x"] - x).max() < 1e-9


def test_c_oracle_p2_and_elasticity_agree_with_numpy_oracle():
    """The C cell loops the full-size parity tests of configs[2] / configs[3] lean on (orc_assemble_p1_elasticity, orc_assemble_p2,
    orc_csr_pattern_generic) against the numpy oracle - an independent statement of the same forms (vectorised, other data layout)."""
    from oracle import c_oracle as co_
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 1.2), 5, 4, 6)
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    n = len(co) + len(edges)
    rp, ci = co_.csr_pattern_generic(n, cd)
    R = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 0.7)).tocsr()
    R.sort_indices()
    assert np.array_equal(R.indptr, rp) and np.array_equal(R.indices, ci)
    assert np.abs(R.data - co_.assemble_p2(co, ce, cd, 0.7, rp, ci)).max() <= 1e-13 * np.abs(R.data).max()
    E, nu = 2e11, 0.27
    mu, lam = fo.lame(E, nu)
    cd12 = (ce.astype(np.int64)[:, :, None] * 3 + np.arange(3)).reshape(len(ce), 12)
    rp, ci = co_.csr_pattern_generic(3 * len(co), cd12)
    R = fo.assemble_p1_elasticity(co, ce, E, nu).tocsr()
    R.sort_indices()
    assert np.array_equal(R.indptr, rp) and np.array_equal(R.indices, ci)
    assert np.abs(R.data - co_.assemble_p1_elasticity(co, ce, mu, lam, rp, ci)).max() <= 1e-13 * np.abs(R.data).max()


# ---------------------------------------------------------------- C-ABI surface
def test_library_exports_every_declared_symbol():
    from fenicssolver_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "fenicssolver_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libfsamd.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in lib.fs_version()


def test_every_tunable_is_documented_in_the_header():
    """fs_set_option names its tunables in the library's source; the header is their only documentation."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "fenicssolver_amd", "csrc", "fs_krylov.hip")).read()
    names = re.findall(r'!strcmp\(name, "([a-z0-9_]+)"\)', src)
    header = open(os.path.join(root, "include", "fenicssolver_amd.h")).read()
    assert len(names) >= 15 and not [n for n in names if '"%s"' % n not in header]


def test_product_fails_loudly_without_gpu(data_dir):
    from fenicssolver_amd import backend
    if backend.device_count() > 0:
        pytest.skip("a GPU is visible: the loud-failure path cannot be exercised")
    with pytest.raises(backend.BackendError):
        backend.init(0)
    from fenicssolver_amd.main import load_settings
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    s = load_settings(os.path.join(data_dir, "TestHeatTransfer.json"))
    s["mesh"] = os.path.join(data_dir, "mesh.xml")
    solver = ScalarTransportSolver(s)
    with pytest.raises(backend.BackendError):
        solver.solve()     # no CPU fallback anywhere on the product path


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fenicssolver_amd/ may import, link or include it."""
    pkg = os.path.join(ROOT, "fenicssolver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                text = open(path).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
                assert "liboracle" not in text, path
            elif f.endswith((".hip", ".h", ".cpp", "Makefile")):
                text = open(path).read()
                assert not re.search(r"#include\s*[\"<][^\">]*oracle", text) and "liboracle" not in text, path


# ---------------------------------------------------------------- host data model
def test_load_settings_contract(data_dir, tmp_path):
    from fenicssolver_amd.main import load_settings, main
    d = {"solver_name": "x"}
    assert load_settings(d) is d
    s = load_settings(os.path.join(data_dir, "TestHeatTransfer.json"))
    ref = json.load(open(os.path.join(data_dir, "TestHeatTransfer.json")))
    assert s["solver_name"] == "ScalarTransportSolver" and s["material"] == ref["material"]
    assert s["boundary_conditions"] == ref["boundary_conditions"]
    with pytest.raises(TypeError):
        load_settings(12345)
    with pytest.raises(NameError):
        main({"solver_name": "NoSuchSolver"})


def test_mesh_and_markers_from_xml(data_dir):
    from fenicssolver_amd.fem import Mesh, MeshFunction
    m = Mesh(os.path.join(data_dir, "mesh.xml"))
    assert (m.num_vertices(), m.num_cells(), m.num_facets(), m.num_entities(1)) == (1069, 4355, 9410, 6123)
    assert np.all(np.diff(m.cells().astype(np.int64), axis=1) > 0)      # mesh.order()
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    facets, cf, cnt = fo.facet_numbering(ce)
    assert np.array_equal(m.facets(), facets) and np.array_equal(m.cell_facets(), cf)
    assert np.array_equal(m.exterior_facets(), cnt == 1)
    mf = MeshFunction("size_t", m, os.path.join(data_dir, "mesh_facet_region.xml"))
    assert mf.dim() == 2 and np.bincount(mf.array()).tolist() == [9210, 100, 100]


def test_boxmesh_matches_oracle_and_subdomain_marking():
    from fenicssolver_amd.fem import BoxMesh, Point, MeshFunction, AutoSubDomain, SubDomain, near
    m = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 6, 2, 3)
    co, ce = fo.box_mesh((0, 0, 0), (10, 1, 1), 6, 2, 3)
    assert np.array_equal(m.coordinates(), co) and np.array_equal(m.cells(), ce)

    class Left(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0.0)

    mf = MeshFunction("size_t", m, 2)
    mf.set_all(0)
    Left().mark(mf, 1)
    AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 10.0)).mark(mf, 2)
    AutoSubDomain(lambda x: near(x[2], 1.0)).mark(mf, 3)
    ref = fo.mark_facets(co, ce, lambda x, ob: abs(x[0]) < 3e-16, 1)
    ref = fo.mark_facets(co, ce, lambda x, ob: ob and abs(x[0] - 10.0) < 3e-16, 2, ref)
    ref = fo.mark_facets(co, ce, lambda x, ob: abs(x[2] - 1.0) < 3e-16, 3, ref)
    assert np.array_equal(mf.array(), ref)
    assert (mf.array() == 1).sum() == 2 * 2 * 3 and (mf.array() == 3).sum() == 2 * 6 * 2


def test_expression_and_dirichlet():
    from fenicssolver_amd.fem import (UnitCubeMesh, FunctionSpace, VectorFunctionSpace, Expression, Constant,
                                      MeshFunction, AutoSubDomain, DirichletBC, interpolate, near, SolverError)
    m = UnitCubeMesh(3, 3, 3)
    V = FunctionSpace(m, "CG", 1)
    f = interpolate(Expression("1 + x[0]*x[0] + 2*x[1] - pow(x[2], 3) + sin(pi*x[0])", degree=1), V)
    co = m.coordinates()
    assert np.allclose(f.vector().array(), 1 + co[:, 0] ** 2 + 2 * co[:, 1] - co[:, 2] ** 3 + np.sin(np.pi * co[:, 0]))
    e = Expression(("10*rho", "0", "0.0"), rho=7800, omega=100, degree=2)   # examples/test_linear_elasticity.py:68
    assert np.allclose(e.eval_points(co[:2]), [[78000.0, 0, 0]] * 2)
    # C++ semantics, not Python's (fenicssolver_amd/cexpr.py): ternary, integer division, no eval of arbitrary code
    assert np.array_equal(Expression("x[0] > 0.5 ? 1 : 2", degree=1).eval_points(co), np.where(co[:, 0] > 0.5, 1.0, 2.0))
    assert np.array_equal(Expression("1/2*x[0] + 7/2", degree=1).eval_points(co), np.full(len(co), 3.0))
    assert np.allclose(Expression("1./2*x[0]", degree=1).eval_points(co), 0.5 * co[:, 0])
    assert np.array_equal(Expression("x[0] >= 1 && !(x[1] < 1) || x[2] == 0", degree=1).eval_points(co),
                          (((co[:, 0] >= 1) & ~(co[:, 1] < 1)) | (co[:, 2] == 0)).astype(float))
    for bad in ("x[0]^2", "().__class__.__subclasses__()", "__import__('os').system('true')", "open('x')", "x[3]", "1 +"):
        with pytest.raises(SolverError):
            Expression(bad, degree=1)
    with pytest.raises(SolverError):
        Expression("2*undefined_parameter", degree=1).eval_points(co)
    mf = MeshFunction("size_t", m, 2)
    AutoSubDomain(lambda x: near(x[0], 0.0)).mark(mf, 1)
    bc = DirichletBC(V, Constant(350), mf, 1)
    assert np.array_equal(bc.dofs, np.nonzero(co[:, 0] == 0)[0]) and np.all(bc.values == 350.0)
    W = VectorFunctionSpace(m, "Lagrange", 1)
    bcy = DirichletBC(W.sub(1), Constant(0.5), mf, 1)
    assert np.array_equal(bcy.dofs, np.nonzero(co[:, 0] == 0)[0] * 3 + 1)
    bcv = DirichletBC(W, Constant((0, 0, 1e-3)), mf, 1)
    assert bcv.dofs.size == 3 * 16 and np.allclose(bcv.values.reshape(-1, 3), [0, 0, 1e-3])
    with pytest.raises(SolverError):
        FunctionSpace(m, "CG", 3)          # P3 is not built: loud, not silent
    # vector P2 (the reference's elasticity example): vertices + edge midpoints of the marked facets, 3 dofs per node
    W2 = VectorFunctionSpace(m, "Lagrange", 2)
    X2 = W2.node_coordinates()
    on_face = np.nonzero(X2[:, 0] == 0)[0]
    bc2 = DirichletBC(W2.sub(2), Constant(1.0), mf, 1)
    assert W2.dim() == 3 * len(X2) and np.array_equal(bc2.dofs, on_face * 3 + 2) and len(on_face) == 7 * 7
    assert np.array_equal(W2.cell_nodes()[:, :4], m.cells())
    mid = 0.5 * (X2[W2.cell_nodes()[:, 2]] + X2[W2.cell_nodes()[:, 3]])
    assert np.allclose(X2[W2.cell_nodes()[:, 4]], mid)                       # UFC edge 0 = (v2, v3)
    # values given as strings / Functions (translate_value interpolates them into the space): evaluated at the BC nodes
    g = interpolate(Expression("300 + x[1]", degree=1), V)
    bcf = DirichletBC(V, g, mf, 1)
    assert np.allclose(bcf.values, 300.0 + co[bcf.dofs, 1])
    gv = interpolate(Expression(("x[0]", "2*x[1]", "3*x[2]"), degree=1), W)
    bcfv = DirichletBC(W, gv, mf, 1)
    assert np.allclose(bcfv.values.reshape(-1, 3), co[bcfv.dofs[::3] // 3] * [1, 2, 3])
    bcfc = DirichletBC(W.sub(1), gv, mf, 1)                                      # a vector Function on a sub space: its component
    assert np.allclose(bcfc.values, 2 * co[bcfc.dofs // 3, 1])


def test_scalar_form_recognition_config1(data_dir):
    """Same settings dict in -> same Dirichlet sets and operator as the reference builds
    (ScalarTransportSolver.py:228-359 for TestHeatTransfer.json)."""
    from fenicssolver_amd.main import load_settings
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    s = load_settings(os.path.join(data_dir, "TestHeatTransfer.json"))
    s["mesh"] = os.path.join(data_dir, "mesh.xml")
    solver = ScalarTransportSolver(s)
    assert solver.dimension == 3 and not solver.transient
    solver.init_solver()
    assert np.all(solver.w_current.vector().array() == 293.0)      # initial_values
    solver.current_step = 0
    F, bcs = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    d = F.describe()
    assert d["conductivity"] == ("const", 20.0) and d["capacity"] is None and not d["transient"]
    assert d["sources"] == [] and d["facet_loads"] == [] and d["robin"] == []
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    assert [b.marker_id for b in bcs] == [1, 2]
    assert np.array_equal(bcs[0].dofs, fo.dirichlet_dofs_p1(facets, fm, 1)) and np.all(bcs[0].values == 350.0)
    assert np.array_equal(bcs[1].dofs, fo.dirichlet_dofs_p1(facets, fm, 2)) and np.all(bcs[1].values == 300.0)
    assert solver.capacity() == 1000 * 500
    assert solver.set_solver_parameters()["relative_tolerance"] == 1e-12  # 1e-7 in the JSON is capped at LU-equivalent accuracy (Q2)


def test_scalar_form_recognition_heat_flux_htc_transient():
    """The BC vocabulary of examples/test_heat_transfer.py:42-71,136-161 on a 3D box."""
    from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    from fenicssolver_amd.SolverBase import SolverError
    m = UnitCubeMesh(4, 4, 4)
    Q = FunctionSpace(m, "CG", 1)
    top = AutoSubDomain(lambda x: near(x[1], 1.0))
    bottom = AutoSubDomain(lambda x: near(x[1], 0.0))
    left = AutoSubDomain(lambda x: near(x[0], 0.0))
    bcs = {"hot": {'boundary': top, 'boundary_id': 1, 'values': {
               'temperature': {'variable': 'temperature', 'type': 'heatFlux', 'value': Constant(36.0)}}},
           "cold": {'boundary': bottom, 'boundary_id': 2, 'values': {
               'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100),
                               'ambient': Constant(300)}}},
           "left": {'boundary': left, 'boundary_id': 3, 'values': {
               'temperature': {'variable': 'temperature', 'type': 'symmetry', 'value': None}}}}
    settings = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
                'boundary_conditions': bcs, 'body_source': 5.0, 'initial_values': {'temperature': 300},
                'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.1},
                'solver_settings': {'transient_settings': {'transient': True, 'starting_time': 0, 'time_step': 0.1,
                                                           'ending_time': 1},
                                    'reference_values': {'temperature': 300}, 'solver_parameters': {}},
                'scalar_name': 'temperature'}
    solver = ScalarTransportSolver(settings)
    solver.material['conductivity'] = 0.6      # users patch the material after construction (:170)
    solver.init_solver()
    solver.current_step = 0
    F, dbc = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    d = F.describe()
    assert dbc == []
    assert d["conductivity"] == ("const", 0.6) and d["capacity"] == ("const", 4200000.0)
    assert d["transient"] and d["dt"] == 0.1 and d["theta"] == 0.5
    assert d["facet_loads"] == [(1, 36.0, 'flux')] and d["robin"] == [(2, 100.0, 300.0)]
    assert d["sources"] == [("const", 5.0)]
    assert (solver.boundary_facets.array() == 1).sum() == 32
    settings['convective_velocity'] = Constant((0.005, -0.005, 0.0))
    s2 = ScalarTransportSolver(settings)
    s2.init_solver()
    s2.current_step = 0
    F2, _ = s2.generate_form(0, None, None, s2.w_current, s2.w_prev)
    assert F2.describe()["advection"] == ((0.005, -0.005, 0.0), 4200000.0) and not F2.symmetric
    settings['advection_settings'] = {'stabilization_method': 'IP', 'alpha': 0.1}
    F3, _ = ScalarTransportSolver(settings).generate_form(0, None, None, solver.w_current, solver.w_prev)
    assert F3.describe()["ip_coefficient"] == pytest.approx(0.1 * 4200000.0)      # alpha * capacity (:312-315)
    settings['advection_settings'] = {'stabilization_method': 'G2'}
    with pytest.raises(SolverError):
        ScalarTransportSolver(settings).generate_form(0, None, None, solver.w_current, solver.w_prev)


def test_elasticity_form_recognition():
    """examples/test_linear_elasticity.py:70-129 (P1 instead of P2)."""
    import copy
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, SubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 8, 2, 2)

    class Left(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0)

    class Right(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 10)

    from collections import OrderedDict
    bcs = OrderedDict()
    bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
    bcs["tensile"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'stress', 'value': Constant((1e8, 0, 0))}
    s = copy.deepcopy(SB.default_case_settings)
    s['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                     'thermal_expansion_coefficient': 2e-6}
    s['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    s['boundary_conditions'] = bcs
    s['solver_settings']['reference_values'] = {'temperature': 293}
    s['temperature_distribution'] = Expression("343", degree=1)
    s['body_source'] = Expression(("10*rho", "0", "0.0"), omega=100, rho=7800, degree=2)
    solver = LinearElasticitySolver(s)
    assert solver.settings['vector_name'] == 'displacement'
    solver.init_solver()
    solver.current_step = 0
    F, dbc = solver.generate_form(0, None, None, solver.w_current, solver.w_prev)
    d = F.describe()
    mu, lm = fo.lame(2e11, 0.27)
    assert d["mu"] == mu and d["lambda"] == lm and d["load_sign"] == -1.0      # quirk Q3 kept
    assert d["body_force"] == (78000.0, 0.0, 0.0)
    assert d["tractions"] == [(2, (1e8, 0.0, 0.0), 'stress(vector)')]
    assert d["thermal"] == (2e11 / (1 - 0.54) * 2e-6, 343.0, 293.0)
    co = mesh.coordinates()
    assert len(dbc) == 1 and np.array_equal(dbc[0].dofs, np.nonzero(co[:, 0] == 0)[0] * 3)
    ns = solver.build_nullspace(solver.function_space)
    assert ns.shape == (6, 3 * len(co)) and np.allclose(ns @ ns.T, np.eye(6), atol=1e-12)
    K = fo.assemble_p1_elasticity(co, mesh.cells(), 2e11, 0.27)
    assert np.abs(K @ ns.T).max() < 1e-9 * abs(K).max()


# ---------------------------------------------------------------- P2
def test_p2_oracle_known_answers():
    """Appendix C3: exact P2 stiffness of the reference tetrahedron (x30), exact mass (row sums), loads."""
    co = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    ce = np.array([[0, 1, 2, 3]], dtype=np.int32)
    K = fo.p2_stiffness_local(co, ce, 1.0)[0] * 30
    ref = np.array([[9, 1, 1, 1, 2, 2, 2, -6, -6, -6], [1, 3, 0, 0, 0, -1, -1, 1, 1, -4], [1, 0, 3, 0, -1, 0, -1, 1, -4, 1],
                    [1, 0, 0, 3, -1, -1, 0, -4, 1, 1], [2, 0, -1, -1, 16, 4, 4, -8, -8, -8], [2, -1, 0, -1, 4, 16, 4, -8, -8, -8],
                    [2, -1, -1, 0, 4, 4, 16, -8, -8, -8], [-6, 1, 1, -4, -8, -8, -8, 24, 4, 4], [-6, 1, -4, 1, -8, -8, -8, 4, 24, 4],
                    [-6, -4, 1, 1, -8, -8, -8, 4, 4, 24]], dtype=float)
    assert np.abs(K - ref).max() < 1e-13
    M = fo.p2_mass_reference()
    assert abs(M.sum() - 1.0) < 1e-15 and np.allclose(M, M.T)
    assert np.allclose(M.sum(axis=1), [-1 / 20.0] * 4 + [1 / 5.0] * 6)       # = int phi_a
    # quadratic patch test + structure counts (C7: P2 max 65 entries per row on Kuhn cubes)
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), 2, 2, 2)
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    X = fo.p2_dof_coordinates(co, edges)
    n = len(X)
    A = fo.assemble_generic(n, cd, fo.p2_stiffness_local(co, ce, 1.0))
    assert np.diff(A.indptr).max() == 65
    u = 1 + X[:, 0] ** 2 - 0.5 * X[:, 1] ** 2 + X[:, 0] * X[:, 2] + 2 * X[:, 1]
    b = fo.assemble_generic_vector(n, cd, fo.p2_source_local(co, ce, -1.0))
    onb = np.nonzero(((X == 0) | (X == 1)).any(axis=1))[0]
    Ab, bb = fo.apply_dirichlet(A, b, onb, u[onb], True)
    assert np.abs(fo.solve_direct(Ab, bb) - u).max() < 1e-12


def test_p2_host_space_matches_oracle_numbering(data_dir):
    from fenicssolver_amd.fem import Mesh, UnitCubeMesh, FunctionSpace, MeshFunction, AutoSubDomain, DirichletBC, Expression, near
    for m in (UnitCubeMesh(3, 3, 3), Mesh(os.path.join(data_dir, "mesh.xml"))):
        V = FunctionSpace(m, "CG", 2)
        co, ce = m.coordinates(), m.cells()
        cd, edges = fo.p2_cell_dofs(len(co), ce)
        assert np.array_equal(V.edge_nodes(), edges) and V.dim() == len(co) + len(edges)
        assert np.array_equal(V.node_coordinates(), fo.p2_dof_coordinates(co, edges))
    m = UnitCubeMesh(3, 3, 3)
    V = FunctionSpace(m, "CG", 2)
    mf = MeshFunction("size_t", m, 2)
    AutoSubDomain(lambda x: near(x[2], 0.0)).mark(mf, 1)
    bc = DirichletBC(V, Expression("1+x[0]*x[1]", degree=2), mf, 1)
    co, ce = m.coordinates(), m.cells()
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    facets, _, _ = fo.facet_numbering(ce)
    ref = fo.p2_facet_dofs(len(co), edges, facets, mf.array(), 1)
    assert np.array_equal(bc.dofs, ref) and len(ref) == 49
    Xb = V.node_coordinates()[ref]
    assert np.allclose(bc.values, 1 + Xb[:, 0] * Xb[:, 1])


def test_2d_host_data_model_matches_oracle():
    """UnitSquareMesh / RectangleMesh ('right' diagonal), edge = facet numbering, marking, DirichletBC, point evaluation."""
    from fenicssolver_amd.fem import (UnitSquareMesh, RectangleMesh, Point, MeshFunction, AutoSubDomain, near, FunctionSpace,
                                      DirichletBC, Constant, Expression, interpolate, PointSource)
    m = RectangleMesh(Point(0, 0), Point(2.0, 1.0), 6, 4)
    co, ce = fo.rectangle_mesh((0, 0), (2.0, 1.0), 6, 4)
    assert np.array_equal(m.coordinates(), co) and np.array_equal(m.cells(), ce)
    assert m.geometry().dim() == 2 and m.topology().dim() == 2
    edges, cf, cnt = fo.tri_edge_numbering(ce)
    assert np.array_equal(m.facets(), edges) and np.array_equal(m.cell_facets(), cf)
    assert np.array_equal(m.exterior_facets(), cnt == 1) and m.num_entities(1) == len(edges) and m.num_entities(2) == len(ce)
    mf = MeshFunction("size_t", m, 1)
    AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 2.0)).mark(mf, 3)
    ref = fo.mark_edges(co, ce, lambda x, ob: ob and abs(x[0] - 2.0) < 3e-16, 3)
    assert np.array_equal(mf.array(), ref) and (ref == 3).sum() == 4
    V = FunctionSpace(m, "CG", 1)
    bc = DirichletBC(V, Expression("10*x[1]", degree=1), mf, 3)
    right = np.nonzero(co[:, 0] == 2.0)[0]
    assert np.array_equal(bc.dofs, right) and np.allclose(bc.values, 10 * co[right, 1])
    f = interpolate(Expression("1+2*x[0]-x[1]", degree=1), V)
    assert abs(f(1.3, 0.45) - (1 + 2.6 - 0.45)) < 1e-13
    ps = PointSource(V, Point(1.3, 0.45), 2.0)
    assert abs(ps.weights.sum() - 2.0) < 1e-13 and len(ps.dofs) == 3
    assert UnitSquareMesh(40, 40).num_cells() == 3200


def test_host_blas_pools_are_capped_under_the_cpu_quota():
    """Importing the package caps the BLAS pools at half the CPUs the cgroup grants (an uncapped 64-thread OpenBLAS pool
    spinning under a 16-CPU quota freezes the thread that feeds the GPU: DESIGN.md, Navier-Stokes section)."""
    import os
    import fenicssolver_amd
    n = fenicssolver_amd.granted_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    chosen = any(os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"))
    if chosen and fenicssolver_amd._thread_pool_limit is None:
        return                                   # the user chose: nothing to check
    threadpoolctl = pytest.importorskip("threadpoolctl")
    import numpy  # noqa: F401
    for pool in threadpoolctl.threadpool_info():
        if pool.get("user_api") == "blas":
            assert pool["num_threads"] <= max(1, n // 2)


def _write_ascii_xdmf(path, co, ce, cell_attr=None, fmt="XML"):
    """What dolfin.XDMFFile writes with XDMFFile.Encoding_ASCII (and meshio --ascii): inline DataItems."""
    kind, g = ("Tetrahedron", "XYZ") if ce.shape[1] == 4 else ("Triangle", "XY")
    with open(path, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n<Xdmf Version="3.0" xmlns:xi="http://www.w3.org/2001/XInclude">\n'
                 '  <Domain>\n    <Grid Name="mesh" GridType="Uniform">\n')
        fh.write('      <Topology NumberOfElements="%d" TopologyType="%s" NodesPerElement="%d">\n        <DataItem Dimensions="%d %d" '
                 'NumberType="UInt" Format="%s">' % (len(ce), kind, ce.shape[1], len(ce), ce.shape[1], fmt))
        fh.write("\n".join(" ".join(str(v) for v in row) for row in ce) if fmt == "XML" else "mesh.h5:/Mesh/mesh/topology")
        fh.write('</DataItem>\n      </Topology>\n      <Geometry GeometryType="%s">\n        <DataItem Dimensions="%d %d" Format="%s">'
                 % (g, len(co), co.shape[1], fmt))
        fh.write("\n".join(" ".join(repr(float(v)) for v in row) for row in co) if fmt == "XML" else "mesh.h5:/Mesh/mesh/geometry")
        fh.write('</DataItem>\n      </Geometry>\n')
        if cell_attr is not None:
            fh.write('      <Attribute Name="subdomains" AttributeType="Scalar" Center="Cell">\n        <DataItem Dimensions="%d 1" Format="XML">'
                     % len(ce))
            fh.write(" ".join(str(int(v)) for v in cell_attr))
            fh.write('</DataItem>\n      </Attribute>\n')
        fh.write('    </Grid>\n  </Domain>\n</Xdmf>\n')


def test_xdmf_mesh_reader_and_settings_paths(tmp_path):
    """settings['mesh'] = 'case.xdmf' (SolverBase.py:246-252): ASCII XDMF is read, markers come from the SubDomains;
    HDF5-backed XDMF goes through libhdf5; broken files and a periodic_boundary without map() are refused loudly."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd import SolverBase as SB, case
    from fenicssolver_amd.fem import AutoSubDomain, Constant, near, SolverError
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 2.0, 3.0), 2, 3, 2)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(ce))
    ids = (co[ce.astype(np.int64)].mean(axis=1)[:, 2] > 1.5).astype(int) + 1
    path = str(tmp_path / "box.xdmf")
    _write_ascii_xdmf(path, co, ce[perm][:, ::-1], cell_attr=ids[perm])          # cells shuffled, vertices unsorted
    bundle = case.read_mesh_file(path)
    m = bundle.mesh
    assert np.array_equal(m.coordinates(), co) and m.num_cells() == len(ce)
    assert np.array_equal(np.sort(m.cells(), axis=0), np.sort(np.sort(ce, axis=1), axis=0))      # same cells, mesh.order()ed
    assert np.array_equal(np.sort(bundle.cell_markers.array()), np.sort(ids)) and bundle.facet_markers is None
    # through the solver: markers from the SubDomains, k per region from the XDMF cell attribute
    s = copy.deepcopy(SB.default_case_settings)
    s.update(mesh=path, scalar_name="temperature", report_settings=dict(logging_level=50, logging_file=None, plotting_freq=0, saving_freq=0))
    bcs = OrderedDict()
    bcs["bottom"] = {'boundary': AutoSubDomain(lambda x, on: on and near(x[2], 0.0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant(350)}
    bcs["top"] = {'boundary': AutoSubDomain(lambda x, on: on and near(x[2], 3.0)), 'boundary_id': 2, 'type': 'Dirichlet', 'value': 300}
    s['boundary_conditions'] = bcs
    s['material'] = {'thermal_conductivity': 20.0, 'density': 1.0, 'specific_heat_capacity': 1.0}
    solver = ScalarTransportSolver(s)
    assert solver.mesh.num_vertices() == len(co) and solver.function_space.dim() == len(co)
    assert np.count_nonzero(solver.boundary_facets.array() == 1) == 2 * 2 * 3 and np.count_nonzero(solver.boundary_facets.array() == 2) == 12
    assert sorted(set(solver.subdomains.array().tolist())) == [1, 2]
    F, dbc = solver.generate_form(0, None, None, None, None)
    assert [len(b.dofs) for b in dbc] == [12, 12]
    # 2-D triangles
    co2, ce2 = fo.rectangle_mesh((0, 0), (1, 1), 3, 2)
    p2 = str(tmp_path / "sq.xdmf")
    _write_ascii_xdmf(p2, co2, ce2)
    m2 = case.read_mesh_file(p2).mesh
    assert m2.topology().dim() == 2 and np.array_equal(m2.coordinates(), co2)
    # heavy data in a side HDF5 file (what DOLFIN and meshio write by default): read through libhdf5
    from fenicssolver_amd import hdf5io
    with hdf5io.H5File(str(tmp_path / "mesh.h5"), "w") as h5:
        h5.write("/Mesh/mesh/topology", ce[perm][:, ::-1])
        h5.write("/Mesh/mesh/geometry", co)
    p3 = str(tmp_path / "h5.xdmf")
    _write_ascii_xdmf(p3, co, ce, fmt="HDF")
    m3 = case.read_mesh_file(p3).mesh
    assert np.array_equal(m3.coordinates(), co) and np.array_equal(m3.cells(), m.cells())
    os.remove(str(tmp_path / "mesh.h5"))
    with pytest.raises(SolverError, match="mesh.h5"):
        case.read_mesh_file(p3)
    open(str(tmp_path / "m.h5"), "wb").write(b"\x89HDF\r\n\x1a\n")
    with pytest.raises(SolverError, match="HDF5"):
        case.read_mesh_file(str(tmp_path / "m.h5"))
    with pytest.raises(SolverError):
        case.read_mesh_file(str(tmp_path / "missing.xml"))
    # a periodic_boundary must be able to map slaves onto masters (SolverBase.py:260-275)
    s2 = copy.deepcopy(SB.default_case_settings)
    s2.update(mesh=path, scalar_name="temperature", boundary_conditions=bcs, material=s['material'], report_settings=s['report_settings'],
              periodic_boundary=AutoSubDomain(lambda x: near(x[0], 0.0)))
    with pytest.raises(SolverError, match="periodic"):
        ScalarTransportSolver(s2)


def test_time_grid_and_value_rules():
    from fenicssolver_amd import case
    from fenicssolver_amd.fem import SolverError
    g = case.TimeGrid({'transient': True, 'starting_time': 1.0, 'time_step': 0.25, 'ending_time': 2.0})
    assert g.step(3) == 0.25 and g.time(3) == 1.5
    g = case.TimeGrid({'transient': True, 'starting_time': 0.0, 'time_series': [0.0, 0.1, 0.3, 0.7], 'ending_time': 0.7})
    assert abs(g.step(1) - 0.2) < 1e-15 and g.time(2) == 0.3          # t[i+1] - t[i], not the reference's t[i] - t[i] (B-Q4)
    with pytest.raises(SolverError):
        g.step(3)
    assert case.boundary_variable({'type': 'a', 'values': {'temperature': {'type': 'b'}}}, 'temperature') == {'type': 'b'}
    assert case.boundary_variable({'type': 'a', 'values': [{'variable': 'velocity', 'type': 'c'}]}, 'velocity')['type'] == 'c'
    assert case.boundary_variable({'type': 'a', 'values': {'pressure': {}}}, 'temperature')['type'] == 'a'


def test_measure_names_marked_boundary_parts():
    """dolfin.Measure as the reference uses it: ds = Measure("ds", subdomain_data=boundary_facets); ds(i) (SolverBase users
    get it as the 4th argument of update_boundary_conditions)."""
    from fenicssolver_amd.fem import UnitCubeMesh, MeshFunction, AutoSubDomain, Measure, SolverError, near
    mesh = UnitCubeMesh(2, 2, 2)
    mf = MeshFunction("size_t", mesh, 2)
    mf.set_all(0)
    AutoSubDomain(lambda x: near(x[0], 0.0)).mark(mf, 4)
    ds = Measure("ds", subdomain_data=mf)
    part = ds(4)
    assert part.subdomain_id == 4 and part.integral_type() == "exterior_facet" and repr(part) == "ds(4)"
    assert len(part.facets()) == 8 and np.array_equal(part.facets(), mf.where(4))
    assert ds.subdomain_id is None and Measure("dx", domain=mesh)(1).integral_type() == "cell"
    with pytest.raises(SolverError):
        Measure("dq")
    with pytest.raises(SolverError):
        ds.facets()


def test_periodic_vertex_pairs_follow_dolfins_rule():
    """constrained_domain: masters = boundary vertices inside(); a slave is a boundary vertex whose map() lands inside;
    doubly periodic corners fold onto one master; unmatched meshes and empty selections are errors."""
    from fenicssolver_amd.fem import UnitSquareMesh, FunctionSpace, SubDomain, SolverError, near, periodic_vertex_pairs

    class PX(SubDomain):
        def inside(self, x, on_boundary):
            return near(x[0], 0.0) and on_boundary

        def map(self, x, y):
            y[0], y[1] = x[0] - 1.0, x[1]

    class PXY(SubDomain):          # the doubly periodic square of the DOLFIN demos: left and bottom edges are masters
        def inside(self, x, on_boundary):
            return bool((near(x[0], 0) or near(x[1], 0)) and not ((near(x[0], 0) and near(x[1], 1)) or (near(x[0], 1) and near(x[1], 0))) and on_boundary)

        def map(self, x, y):
            if near(x[0], 1) and near(x[1], 1):
                y[0], y[1] = x[0] - 1.0, x[1] - 1.0
            elif near(x[0], 1):
                y[0], y[1] = x[0] - 1.0, x[1]
            else:
                y[0], y[1] = x[0], x[1] - 1.0
    mesh = UnitSquareMesh(4, 3)
    co = mesh.coordinates()
    sl, ma = periodic_vertex_pairs(mesh, PX())
    assert len(sl) == 4 and np.allclose(co[sl, 0], 1.0) and np.allclose(co[ma, 0], 0.0) and np.allclose(co[sl, 1], co[ma, 1])
    sl2, ma2 = periodic_vertex_pairs(mesh, PXY())
    assert len(sl2) == 4 + 5 - 1 and not set(sl2) & set(ma2)      # right edge + top edge, the corner (1, 1) once
    corner = {tuple(co[s]): tuple(co[m]) for s, m in zip(sl2, ma2)}
    assert corner[(1.0, 1.0)] == (0.0, 0.0) and corner[(1.0, 0.0)] == (0.0, 0.0) and corner[(0.0, 1.0)] == (0.0, 0.0)
    V = FunctionSpace(mesh, "CG", 1, constrained_domain=PX())
    assert V.dim() == mesh.num_vertices() and np.array_equal(V.periodic_pairs()[0], sl)
    pairs = V._periodic_couplings()
    assert pairs.shape[1] == 2 and set(pairs[:, 0]) <= set(ma)
    V2 = FunctionSpace(mesh, "CG", 2, constrained_domain=PX())          # P2: the edge nodes of the slave side are tied as well
    s2, m2 = V2.periodic_pairs()
    X2 = V2.node_coordinates()
    assert len(s2) == 4 + 3 and np.allclose(X2[s2, 0], 1.0) and np.allclose(X2[m2, 0], 0.0) and np.allclose(X2[s2, 1], X2[m2, 1])

    class Shifted(PX):
        def map(self, x, y):
            y[0], y[1] = x[0] - 1.0, x[1] + 0.01
    with pytest.raises(SolverError):
        periodic_vertex_pairs(mesh, Shifted())


def test_dolfin_hdf5_mesh_files(tmp_path):
    """settings['mesh'] = 'case.h5' (SolverBase._read_hdf5_mesh, :203-221): /mesh, /boundaries, /subdomains in DOLFIN's
    HDF5File layout, written here through the same libhdf5 and read back; markers are matched by vertex tuples, so a file
    whose entities are listed in another order (and with their vertices permuted) gives the same MeshFunctions."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd import SolverBase as SB, case, hdf5io
    from fenicssolver_amd.fem import UnitCubeMesh, UnitSquareMesh, MeshFunction, AutoSubDomain, Constant, SolverError, near
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
    for mesh in (UnitCubeMesh(3, 2, 2), UnitSquareMesh(4, 3)):
        td = mesh.topology().dim()
        fm = MeshFunction("size_t", mesh, td - 1)
        fm.set_all(0)
        AutoSubDomain(lambda x: near(x[0], 0.0)).mark(fm, 5)
        AutoSubDomain(lambda x: near(x[1], 1.0)).mark(fm, 7)
        cm = MeshFunction("size_t", mesh, td)
        cm.set_all(0)
        cm.array()[::3] = 2
        path = str(tmp_path / ("m%d.h5" % td))
        hdf5io.write_dolfin_mesh(path, mesh, fm, cm)
        with hdf5io.H5File(path) as f:
            assert sorted(f.keys("/")) == ["boundaries", "mesh", "subdomains"] and f.has("/mesh/topology") and not f.has("/mesh/nope")
            assert f.read("/mesh/coordinates").shape == mesh.coordinates().shape
        b = case.read_mesh_file(path)
        assert np.array_equal(b.mesh.coordinates(), mesh.coordinates()) and np.array_equal(b.mesh.cells(), mesh.cells())
        assert np.array_equal(b.facet_markers.array(), fm.array()) and np.array_equal(b.cell_markers.array(), cm.array())
        # the same content, entities shuffled and their vertices reversed
        rng = np.random.default_rng(1)
        sel = np.nonzero(fm.array())[0]
        pf = rng.permutation(len(sel))
        with hdf5io.H5File(path, "w") as f:
            f.write("/mesh/coordinates", mesh.coordinates())
            f.write("/mesh/topology", mesh.cells())
            f.write("/boundaries/topology", mesh.facets()[sel][pf][:, ::-1])
            f.write("/boundaries/values", fm.array()[sel][pf])
        b2 = case.read_mesh_file(path)
        assert np.array_equal(b2.facet_markers.array(), fm.array()) and b2.cell_markers is None
    # through the solver class: markers come from the file, not from the SubDomains
    s = copy.deepcopy(SB.default_case_settings)
    s.update(mesh=str(tmp_path / "m3.h5"), scalar_name="temperature", report_settings=dict(logging_level=50, logging_file=None, plotting_freq=0, saving_freq=0))
    bcs = OrderedDict()
    bcs["left"] = {'boundary_id': 5, 'type': 'Dirichlet', 'value': Constant(350)}
    bcs["back"] = {'boundary_id': 7, 'type': 'Dirichlet', 'value': 300}
    s['boundary_conditions'] = bcs
    s['material'] = {'thermal_conductivity': 20.0, 'density': 1.0, 'specific_heat_capacity': 1.0}
    solver = ScalarTransportSolver(s)
    assert np.count_nonzero(solver.boundary_facets.array() == 5) == 2 * 2 * 2
    F, dbc = solver.generate_form(0, None, None, None, None)
    assert [len(b.dofs) for b in dbc] == [9, 12]
    # a marker entity that is not in the mesh is an error, not a silent drop
    with hdf5io.H5File(str(tmp_path / "bad.h5"), "w") as f:
        f.write("/mesh/coordinates", mesh.coordinates())
        f.write("/mesh/topology", mesh.cells())
        f.write("/boundaries/topology", np.array([[0, 19]]))
        f.write("/boundaries/values", np.array([3]))
    with pytest.raises(SolverError, match="not facets"):
        case.read_mesh_file(str(tmp_path / "bad.h5"))


def test_one_launch_iteration_kernel_keeps_four_waves_per_simd(tmp_path):
    """k_dict_cg_iter (fs_krylov.hip) is latency-bound: all of its 1024 workgroups have to be resident at once, which takes four
    waves per SIMD = at most 128 VGPRs.  The compiler's schedule for it is touchy (a cold fallback path inside the kernel once took
    it to 174 VGPRs and the launch from 19 to 25 us without any change to the hot path), so the register count is pinned here:
    cross-compiled for gfx950 with the Makefile's flags, resource usage from the compiler's own remarks."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src = os.path.join(ROOT, "fenicssolver_amd", "csrc", "fs_krylov.hip")
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-function",
                        "-c", src, "-o", str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=os.path.dirname(src), timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    blocks = re.split(r"remark: Function Name: ", out)
    seen = 0
    for b in blocks:
        if "k_dict_cg_iter" not in b.split("\n", 1)[0]:
            continue
        seen += 1
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert occ >= 4 and vgprs <= 128 and spill == 0, (vgprs, spill, occ)
    assert seen >= 1


def test_p2p_fallback_keeps_the_ranks_paired(monkeypatch):
    """once ANY rank reports the time-out of the peer-to-peer exchange every rank solves again over RCCL - also a rank
    whose own solve failed with a different error (stale ghosts make a breakdown as easily as a time-out); a failure nobody's
    transport caused is raised as it is, nothing switched off, nothing repeated."""
    from fenicssolver_amd import backend as B, _lib as L

    class Space:
        _p2p = True
        off = 0

        def enable_p2p_halo(self, on):
            self.off += not on

    def err(rc):
        e = L.BackendError("x")
        e.rc = rc
        return e

    monkeypatch.setattr(B, "_comm_up", True)
    # this rank: breakdown; another rank: time-out -> switched off, solved again, no exception
    monkeypatch.setattr(B, "comm_allgather", lambda v, n: [v[0], 1.0])
    sp, calls = Space(), []

    def solve_a():
        calls.append(1)
        if len(calls) == 1:
            raise err(L.FS_ERR_NUMERIC)
    B._with_p2p_fallback(sp, solve_a)
    assert len(calls) == 2 and sp.off == 1
    # nobody timed out: the local error is raised, nothing repeated
    monkeypatch.setattr(B, "comm_allgather", lambda v, n: [v[0], 0.0])
    sp, calls = Space(), []

    def solve_b():
        calls.append(1)
        raise err(L.FS_ERR_NUMERIC)
    with pytest.raises(L.BackendError):
        B._with_p2p_fallback(sp, solve_b)
    assert len(calls) == 1 and sp.off == 0
    # a failure that is not the transport's shows again in the second solve and is raised there
    monkeypatch.setattr(B, "comm_allgather", lambda v, n: [v[0], 1.0])
    sp, calls = Space(), []
    with pytest.raises(L.BackendError):
        B._with_p2p_fallback(sp, solve_b)
    assert len(calls) == 2 and sp.off == 1


def test_marching_window_launch_planning(tmp_path):
    """fs_box.h: box_recognize / box_cut / box_lds_bytes over 3 000 box shapes (even and odd strides, lines shorter and longer than a
    patch, one plane per chunk): patches tile the plane, every plane sits in exactly one chunk, every unit is taken by exactly one
    workgroup of the kernel's own loop, the window slots hold every position a step reads - compiled with hipcc, run on the CPU."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path / "box_plan_check")
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "fenicssolver_amd", "csrc"),
                        "-o", exe, os.path.join(ROOT, "tests", "cpp", "box_plan_check.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    out = subprocess.run([exe], stdout=subprocess.PIPE, timeout=120).stdout.decode()
    assert out.startswith("ok "), out


def test_cg2_stencil_tables_are_what_the_generator_writes(tmp_path):
    """fs_cg2_stencil.h (the compile-time loop structure of k_lat_march, fs_latmarch.h) is generated from the oracle's Kuhn box mesh and
    CG2 dof map (tools/gen_cg2_stencil.py): the committed header is byte for byte what the generator writes, eight parity classes of
    65 / 27 / 19 entries, 28.75 on average - the stored entries per row of a CG2 operator on a Kuhn mesh."""
    import subprocess, sys
    out = tmp_path / "fs_cg2_stencil.h"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_cg2_stencil.py"), str(out)], check=True, capture_output=True, timeout=300)
    with open(os.path.join(ROOT, "fenicssolver_amd", "csrc", "fs_cg2_stencil.h")) as fh:
        committed = fh.read()
    assert out.read_text() == committed
    import re
    counts = [int(v) for v in re.search(r"LM_CNT\[8\] = \{([^}]*)\}", committed).group(1).split(",")]
    assert sorted(counts) == [19, 19, 19, 27, 27, 27, 27, 65] and sum(counts) / 8.0 == 28.75

