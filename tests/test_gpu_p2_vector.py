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
    facets, _ = fo.facet_numbering(ce)
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
    facets, _ = fo.facet_numbering(ce)
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
