"""GPU parity tests of the HIP hot path against the CPU oracle, through the C-ABI.

Bars: connectivity / CSR structure bit-exact; fp64 values within the tolerance
written next to each assert (atomic scatter order and FMA contraction are the
only differences from the oracle's arithmetic).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu

RTOL_ASSEMBLY = 1e-12   # relative to max |A_ij|
RTOL_SPMV = 1e-13       # relative to |A| |x|
RTOL_SOLUTION = 1e-9    # relative to max |u| at equal Krylov tolerance


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def _assert_same_pattern(M, R):
    R = R.tocsr()
    R.sort_indices()
    assert M.shape == R.shape
    assert np.array_equal(M.indptr, R.indptr)
    assert np.array_equal(M.indices, R.indices)


# ---------------------------------------------------------------------------------------
def test_box_mesh_generator_matches_dolfin_order(gpu):
    for dims, p1 in (((3, 4, 5), (1.0, 1.0, 1.0)), ((7, 2, 3), (10.0, 1.0, 1.0))):
        m = gpu.DeviceMesh.box(*dims, p1=p1)
        xyz, cells, gid = m.get()
        co, ce = fo.box_mesh((0, 0, 0), p1, *dims)
        assert np.array_equal(xyz, co)            # bit-exact coordinates
        assert np.array_equal(cells, ce)          # bit-exact connectivity
        assert np.array_equal(gid, np.arange(len(co)))


def test_box_slab_has_owned_first_numbering(gpu):
    nx, ny, nz = 3, 2, 6
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), nx, ny, nz)
    plane = (nx + 1) * (ny + 1)
    for zb, ze in ((0, 3), (2, 5), (4, 7)):
        m = gpu.DeviceMesh.box(nx, ny, nz, zplanes=(zb, ze))
        xyz, cells, gid = m.get()
        nv, nc, n_owned = m.info()
        assert n_owned == (ze - zb) * plane
        assert np.array_equal(gid[:n_owned], np.arange(zb * plane, ze * plane))
        assert np.array_equal(xyz, co[gid])
        # local cells = every global cell touching an owned vertex, same relative order
        glob = gid[cells]
        owned_mask = (ce >= zb * plane) & (ce < ze * plane)
        expect = ce[owned_mask.any(axis=1)]
        assert np.array_equal(glob, expect)


@pytest.mark.parametrize("n", [1, 2, 5, 12])
def test_p1_poisson_assembly_structured(gpu, n):
    co, ce = fo.unit_cube_mesh(n)
    mesh = gpu.DeviceMesh.box(n, n, n)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    M = _csr(A)
    R = fo.assemble_p1_scalar(co, ce, 20.0)
    _assert_same_pattern(M, R)
    assert V.nnz == R.nnz
    assert np.abs(M.data - R.data).max() <= RTOL_ASSEMBLY * np.abs(R.data).max()


def test_p1_assembly_unstructured_data_mesh(gpu, data_dir):
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    assert V.nnz == 13315          # V + 2E
    A = gpu.DeviceMatrix(V)
    rng = np.random.default_rng(0)
    kcell = rng.uniform(0.5, 1.5, len(ce))
    ccell = rng.uniform(1.0, 2.0, len(ce))
    for kw, ref in (
        (dict(stiffness=20.0), fo.assemble_p1_scalar(co, ce, 20.0)),
        (dict(stiffness=("cell", kcell)), fo.assemble_p1_scalar(co, ce, kcell)),
        (dict(stiffness=2.0, mass=3.0), fo.assemble_p1_scalar(co, ce, 2.0, mass_coef=3.0)),
        (dict(stiffness=None, mass=("cell", ccell)),
         fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, ccell))),
        (dict(stiffness=("tensor", [[2.0, 0.3, 0.0], [0.3, 1.0, 0.1], [0.0, 0.1, 4.0]])),
         fo.assemble_p1_scalar(co, ce, np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.1], [0.0, 0.1, 4.0]]))),
    ):
        A.assemble(**kw)
        M = _csr(A)
        _assert_same_pattern(M, ref)
        assert np.abs(M.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max(), kw


def test_matrix_add_and_axpy(gpu):
    co, ce = fo.unit_cube_mesh(4)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    K = gpu.DeviceMatrix(V)
    Mm = gpu.DeviceMatrix(V)
    K.assemble(stiffness=1.5)
    Mm.assemble(mass=2.0)
    K.axpy(4.0, Mm)
    ref = fo.assemble_p1_scalar(co, ce, 1.5, mass_coef=8.0)
    got = _csr(K)
    assert np.abs(got.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max()
    K.assemble(stiffness=1.5)
    K.assemble(mass=8.0, add=True)
    got = _csr(K)
    assert np.abs(got.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max()


@pytest.mark.parametrize("symmetric", [False, True])
def test_dirichlet_matches_oracle(gpu, data_dir, symmetric):
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    d1 = fo.dirichlet_dofs_p1(facets, fm, 1)
    d2 = fo.dirichlet_dofs_p1(facets, fm, 2)
    dofs = np.concatenate([d1, d2])
    vals = np.concatenate([np.full(len(d1), 350.0), np.full(len(d2), 300.0)])
    rng = np.random.default_rng(1)
    b0 = rng.standard_normal(len(co))
    A0 = fo.assemble_p1_scalar(co, ce, 20.0)
    Ar, br = fo.apply_dirichlet(A0, b0, dofs, vals, symmetric=symmetric)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    b = gpu.DeviceVector(V.n_owned, b0)
    A.apply_dirichlet(b, dofs, vals, symmetric=symmetric)
    M = _csr(A)
    _assert_same_pattern(M, Ar)
    assert np.abs(M.data - Ar.data).max() <= RTOL_ASSEMBLY * np.abs(Ar.data).max()
    assert np.abs(b.get() - br).max() <= 1e-12 * np.abs(br).max()
    # identity rows are exact
    assert np.all(M.diagonal()[dofs] == 1.0)
    assert np.all(b.get()[dofs] == vals)


def test_dirichlet_later_entries_win(gpu):
    co, ce = fo.unit_cube_mesh(2)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, [0, 1, 0], [1.0, 2.0, 3.0], symmetric=False)
    got = b.get()
    assert got[0] == 3.0 and got[1] == 2.0


def test_spmv_matches_oracle(gpu, data_dir):
    rng = np.random.default_rng(2)
    for co, ce in (fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml")), fo.unit_cube_mesh(9)):
        mesh = gpu.DeviceMesh(co, ce)
        V = gpu.DeviceSpace(mesh, 1)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=3.0, mass=1.0)
        R = _csr(A)   # use the device values so only the SpMV is under test
        xh = rng.standard_normal(V.n_local)
        x = gpu.DeviceVector(V.n_local, xh)
        y = gpu.DeviceVector(V.n_owned)
        A.spmv(x, y)
        ref = R @ xh
        scale = (abs(R) @ np.abs(xh)).max()
        assert np.abs(y.get() - ref).max() <= RTOL_SPMV * scale


def test_vector_ops(gpu):
    rng = np.random.default_rng(3)
    a = rng.standard_normal(100003)
    b = rng.standard_normal(100003)
    x = gpu.DeviceVector(a.size, a)
    y = gpu.DeviceVector(b.size, b)
    assert abs(x.dot(y) - float(a @ b)) <= 1e-12 * float(np.abs(a) @ np.abs(b))
    y.axpy(2.5, x)
    assert np.allclose(y.get(), b + 2.5 * a, rtol=0, atol=1e-14 * 10)
    y.fill(7.0)
    assert np.all(y.get() == 7.0)


def test_rhs_source_and_facet_terms(gpu, data_dir):
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    b = gpu.DeviceVector(V.n_owned)
    rng = np.random.default_rng(4)
    gpu.assemble_vector(V, b, source=2.5)
    ref = fo.assemble_p1_source(co, ce, 2.5)
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    fc = rng.uniform(0, 1, len(ce))
    gpu.assemble_vector(V, b, source=("cell", fc))
    ref = fo.assemble_p1_source(co, ce, fc)
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    fn = rng.uniform(0, 1, len(co))
    gpu.assemble_vector(V, b, source=("nodal", fn))
    ref = fo.assemble_p1_source(co, ce, f_nodal=fn)
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    # facet load on marker 1, accumulated on top
    tri = facets[fm == 1]
    gpu.assemble_facet_vector(V, b, tri, 36.0)
    ref = ref + fo.assemble_p1_facet_load(co, facets, fm, 1, 36.0)
    assert np.abs(b.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    # Robin matrix on marker 2
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    A.add_facet_mass(facets[fm == 2], 100.0)
    R = fo.assemble_p1_scalar(co, ce, 1.0) + fo.assemble_p1_facet_mass(co, facets, fm, 2, 100.0)
    R = R.tocsr()
    M = _csr(A)
    assert abs(M - R).max() <= RTOL_ASSEMBLY * abs(R).max()


# ---- the solve -------------------------------------------------------------------------------
def _solve_heat(gpu, mesh, dofs, vals, k=20.0, rtol=1e-8, precond="jacobi"):
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=k)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=rtol, max_iter=5000, precond=precond)
    return x.get(), st


def test_config1_heat_transfer_data_mesh(gpu, data_dir):
    """data/TestHeatTransfer.json on data/mesh.xml: T = 350 - 2.5 z."""
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    d1 = fo.dirichlet_dofs_p1(facets, fm, 1)
    d2 = fo.dirichlet_dofs_p1(facets, fm, 2)
    dofs = np.concatenate([d1, d2])
    vals = np.concatenate([np.full(len(d1), 350.0), np.full(len(d2), 300.0)])
    T, st = _solve_heat(gpu, gpu.DeviceMesh(co, ce), dofs, vals, rtol=1e-8)
    A0 = fo.assemble_p1_scalar(co, ce, 20.0)
    Ab, bb = fo.apply_dirichlet(A0, np.zeros(len(co)), dofs, vals, True)
    xo, ito, _ = fo.pcg_jacobi_single_reduction(Ab, bb, rtol=1e-8)
    assert st["converged"] == 1
    assert abs(st["iterations"] - ito) <= 1          # same recurrence, same count (93)
    assert st["true_rel_residual"] <= 1.5e-8
    assert np.abs(T - xo).max() <= 1e-7 * 350.0        # both stopped at 1e-8: agree far below that
    assert np.abs(T - (350.0 - 2.5 * co[:, 2])).max() <= 1e-4
    # tight tolerance reproduces the exact discrete solution (the reference's LU answer)
    T, st = _solve_heat(gpu, gpu.DeviceMesh(co, ce), dofs, vals, rtol=1e-13)
    assert np.abs(T - (350.0 - 2.5 * co[:, 2])).max() <= 1e-9


@pytest.mark.parametrize("n,axis", [(12, 2), (24, 2), (24, 0)])
def test_config2_family_cg_parity(gpu, n, axis):
    P = fo.heat_box_problem(n, axis=axis)
    mesh = gpu.DeviceMesh.box(n, n, n)
    T, st = _solve_heat(gpu, mesh, P["dofs"], P["vals"], rtol=1e-8)
    xo, ito, hist = fo.pcg_jacobi_single_reduction(P["A"], P["b"], rtol=1e-8)
    assert st["converged"] == 1
    assert abs(st["iterations"] - ito) <= 1
    assert np.abs(T - xo).max() <= RTOL_SOLUTION * 350.0 * 100
    assert np.abs(T - P["exact"]).max() <= 1e-3
    # residual history follows the oracle's while well above round-off
    h = gpu.krylov_history()
    m = min(len(h), len(hist), 20)
    assert np.allclose(h[:m], hist[:m], rtol=1e-6)
    # unpreconditioned CG also converges to the same answer
    T2, st2 = _solve_heat(gpu, mesh, P["dofs"], P["vals"], rtol=1e-10, precond="none")
    assert st2["converged"] == 1
    assert np.abs(T2 - P["exact"]).max() <= 1e-5


@pytest.mark.parametrize("n,axis,rtol", [(12, 2, 1e-8), (24, 0, 1e-8), (24, 2, 1e-12), (40, 2, 1e-10)])
def test_pipelined_cg_follows_the_single_reduction_recurrence(gpu, n, axis, rtol):
    """fs_krylov_opts.pipelined = 1 (Ghysels-Vanroose): in exact arithmetic the iterates of CG - same count within +2,
    same residual history while well above round-off, same solution <= 1e-9 relative, and the oracle's answer."""
    P = fo.heat_box_problem(n, axis=axis)
    mesh = gpu.DeviceMesh.box(n, n, n)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, P["dofs"], P["vals"], symmetric=True)
    x0, x1 = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
    s0 = gpu.krylov_solve(A, b, x0, rtol=rtol, max_iter=5000, pipelined=False)
    h0 = gpu.krylov_history()
    s1 = gpu.krylov_solve(A, b, x1, rtol=rtol, max_iter=5000, pipelined=True)
    h1 = gpu.krylov_history()
    assert s0["converged"] == 1 and s1["converged"] == 1
    assert -1 <= s1["iterations"] - s0["iterations"] <= 2, (s0["iterations"], s1["iterations"])
    assert s1["true_rel_residual"] <= 1.5 * rtol
    m = min(len(h0), len(h1), 25)
    assert np.allclose(h0[:m], h1[:m], rtol=1e-6)
    scale = np.abs(x0.get()).max()
    assert np.abs(x1.get() - x0.get()).max() <= max(RTOL_SOLUTION, 30 * rtol) * scale
    xo, ito, _ = fo.pcg_jacobi_single_reduction(P["A"], P["b"], rtol=rtol)       # the oracle's recurrence and answer
    assert -1 <= s1["iterations"] - ito <= 2
    xp, itp, hp = fo.pcg_jacobi_pipelined(P["A"], P["b"], rtol=rtol)             # the oracle's restatement of k_pcg_update
    assert abs(s1["iterations"] - itp) <= 1
    mp = min(len(h1), len(hp), 25)
    assert np.allclose(h1[:mp], hp[:mp], rtol=1e-6)
    assert np.abs(x1.get() - xo).max() <= max(RTOL_SOLUTION, 30 * rtol) * scale
    if rtol <= 1e-10:            # the linear profile is the exact discrete solution
        assert np.abs(x1.get() - P["exact"]).max() <= 1e-6 * scale
    # a nonzero initial guess and the preconditioned norm go through the same pass set-up
    x2 = gpu.DeviceVector(V.n_owned)
    x2.set(np.full(V.n_owned, 320.0))
    s2 = gpu.krylov_solve(A, b, x2, rtol=rtol, max_iter=5000, pipelined=True, nonzero_guess=True, norm="preconditioned")
    assert s2["converged"] == 1
    assert np.abs(x2.get() - x0.get()).max() <= max(RTOL_SOLUTION, 300 * rtol) * scale


def test_pipelined_cg_needs_the_scaled_jacobi_recurrence(gpu):
    mesh = gpu.DeviceMesh.box(3, 3, 3)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0, mass=1.0)
    b = gpu.DeviceVector(V.n_owned)
    b.fill(1.0)
    x = gpu.DeviceVector(V.n_owned)
    with pytest.raises(gpu.BackendError):
        gpu.krylov_solve(A, b, x, precond="none", pipelined=True)


def test_manufactured_solution_converges_second_order(gpu):
    """-k lap u = f with u = sin(pi x) sin(pi y) sin(pi z): L2-ish error ~ h^2 (Appendix C6)."""
    errs = []
    k = 20.0
    for n in (8, 16, 32):
        mesh = gpu.DeviceMesh.box(n, n, n)
        xyz, cells, _ = mesh.get()
        V = gpu.DeviceSpace(mesh, 1)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=k)
        u = np.sin(np.pi * xyz[:, 0]) * np.sin(np.pi * xyz[:, 1]) * np.sin(np.pi * xyz[:, 2])
        f = 3 * np.pi ** 2 * k * u
        b = gpu.DeviceVector(V.n_owned)
        gpu.assemble_vector(V, b, source=("nodal", f))
        on_b = np.nonzero(((xyz == 0.0) | (xyz == 1.0)).any(axis=1))[0]
        A.apply_dirichlet(b, on_b, 0.0, symmetric=True)
        x = gpu.DeviceVector(V.n_owned)
        st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
        assert st["converged"] == 1
        errs.append(np.sqrt(np.mean((x.get() - u) ** 2)))
    r1 = np.log2(errs[0] / errs[1])
    r2 = np.log2(errs[1] / errs[2])
    assert 1.7 <= r1 <= 2.3 and 1.7 <= r2 <= 2.3, (errs, r1, r2)


def test_slab_rows_equal_global_rows(gpu):
    """A slab (owned planes + ghost layer) assembles exactly the owned rows of the global matrix."""
    nx, ny, nz = 4, 3, 8
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 2), nx, ny, nz)
    G = fo.assemble_p1_scalar(co, ce, 20.0).tocsr()
    plane = (nx + 1) * (ny + 1)
    for zb, ze in ((0, 4), (4, 9), (3, 6)):
        mesh = gpu.DeviceMesh.box(nx, ny, nz, p1=(1.0, 1.0, 2.0), zplanes=(zb, ze))
        _, _, gid = mesh.get()
        V = gpu.DeviceSpace(mesh, 1)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=20.0)
        M = _csr(A)
        assert M.shape == ((ze - zb) * plane, len(gid))
        # map local columns to global and compare row by row
        Mg = sp.csr_matrix((M.data, gid[M.indices], M.indptr), shape=(M.shape[0], len(co)))
        Mg.sort_indices()
        ref = G[zb * plane:ze * plane]
        ref.sort_indices()
        assert np.array_equal(Mg.indptr, ref.indptr)
        assert np.array_equal(Mg.indices, ref.indices)
        assert np.abs(Mg.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max()


# ---- vector P1 elasticity ----------------------------------------------------------------------
def test_elasticity_assembly_and_nullspace(gpu):
    E, nu = 2e11, 0.27
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 8, 2, 2)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=fo.lame(E, nu))
    M = _csr(A)
    R = fo.assemble_p1_elasticity(co, ce, E, nu)
    # the device pattern is the full 3x3 block pattern (explicit zeros kept)
    assert M.shape == R.shape
    assert abs(M - R).max() <= RTOL_ASSEMBLY * abs(R).max()
    # rigid-body modes span the null space: K r = 0 (Appendix C5)
    ns = fo.rigid_body_modes(co)
    x = gpu.DeviceVector(V.n_local)
    y = gpu.DeviceVector(V.n_owned)
    scale = abs(R).max()
    for r in ns:
        x.set(r)
        A.spmv(x, y)
        assert np.abs(y.get()).max() <= 1e-10 * scale * np.abs(r).max()


def test_elasticity_cantilever_solve(gpu):
    E, nu = 2e11, 0.27
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), 10, 2, 2)
    n = len(co)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=fo.lame(E, nu))
    b = gpu.DeviceVector(V.n_owned)
    f = (0.0, 0.0, -7800.0 * 10.0)
    gpu.assemble_vector(V, b, vector_value=f)
    left = np.nonzero(co[:, 0] == 0.0)[0]
    dofs = (left[:, None] * 3 + np.arange(3)[None, :]).ravel()
    A.apply_dirichlet(b, dofs, 0.0, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=20000)
    assert st["converged"] == 1
    R = fo.assemble_p1_elasticity(co, ce, E, nu)
    rb = fo.assemble_p1_vector_source(co, ce, f)
    Ab, bb = fo.apply_dirichlet(R, rb, dofs, 0.0, True)
    ref = fo.solve_direct(Ab, bb)
    assert np.abs(x.get() - ref).max() <= 1e-6 * np.abs(ref).max()
    assert ref.reshape(n, 3)[:, 2].min() < 0.0


def test_errors_are_reported_not_silent(gpu):
    co, ce = fo.unit_cube_mesh(2)
    bad = ce.copy()
    bad[0, 0] = 10 ** 6
    with pytest.raises(gpu.BackendError):
        gpu.DeviceMesh(co, bad)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)       # all-zero matrix: Jacobi undefined
    b = gpu.DeviceVector(V.n_owned, np.ones(V.n_owned))
    x = gpu.DeviceVector(V.n_owned)
    with pytest.raises(gpu.BackendError):
        gpu.krylov_solve(A, b, x)


# ---- non-symmetric path: advection + BiCGStab ---------------------------------------------------
def test_advection_assembly_and_bicgstab(gpu, data_dir):
    rng = np.random.default_rng(5)
    for co, ce in (fo.unit_cube_mesh(6), fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))):
        mesh = gpu.DeviceMesh(co, ce)
        V = gpu.DeviceSpace(mesh, 1)
        A = gpu.DeviceMatrix(V)
        vconst = np.array([0.3, -0.2, 0.5])
        vcell = rng.uniform(-1, 1, (len(ce), 3))
        for vel in (vconst, vcell):
            A.assemble(stiffness=0.7, mass=0.1, advection=vel, advection_scale=2.5)
            ref = fo.assemble_matrix(len(co), ce, fo.p1_stiffness_local(co, ce, 0.7) + fo.p1_mass_local(co, ce, 0.1)
                                     + fo.p1_advection_local(co, ce, vel, 2.5))
            M = _csr(A)
            _assert_same_pattern(M, ref)
            assert np.abs(M.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max()
        # solve the (non-symmetric) system against the oracle's sparse LU
        b0 = rng.standard_normal(len(co))
        dofs = np.nonzero(co[:, 2] == co[:, 2].min())[0]
        Ab, bb = fo.apply_dirichlet(ref, b0, dofs, 1.0, True)
        xref = fo.solve_direct(Ab, bb)
        b = gpu.DeviceVector(V.n_owned, b0)
        A.apply_dirichlet(b, dofs, 1.0, symmetric=True)
        x = gpu.DeviceVector(V.n_owned)
        st = gpu.krylov_solve(A, b, x, rtol=1e-12, max_iter=5000, method="bicgstab")
        assert st["converged"] == 1 and st["true_rel_residual"] <= 1e-11
        assert np.abs(x.get() - xref).max() <= 1e-8 * np.abs(xref).max()


def test_matrix_free_product_matches_assembled_operator(gpu, data_dir):
    """fs_operator_apply (north_star's matrix-free product): y = K(form) x without forming K, against the
    ORACLE's assembled matrix for diffusion, reaction, tensor diffusion, per-cell coefficients and advection."""
    rng = np.random.default_rng(11)
    for co, ce in (fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml")), fo.unit_cube_mesh(7)):
        mesh = gpu.DeviceMesh(co, ce)
        V = gpu.DeviceSpace(mesh, 1)
        xh = rng.standard_normal(V.n_local)
        x = gpu.DeviceVector(V.n_local, xh)
        y = gpu.DeviceVector(V.n_owned)
        kcell = rng.uniform(0.5, 2.0, len(ce))
        vcell = rng.uniform(-1, 1, (len(ce), 3))
        T = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.1], [0.0, 0.1, 0.5]])
        cases = (
            (dict(stiffness=3.0), fo.p1_stiffness_local(co, ce, 3.0)),
            (dict(stiffness=3.0, mass=0.4), fo.p1_stiffness_local(co, ce, 3.0) + fo.p1_mass_local(co, ce, 0.4)),
            (dict(stiffness=("cell", kcell)), fo.p1_stiffness_local(co, ce, kcell)),
            (dict(stiffness=("tensor", T)), fo.p1_stiffness_local(co, ce, T)),
            (dict(stiffness=0.7, mass=0.1, advection=vcell, advection_scale=2.5),
             fo.p1_stiffness_local(co, ce, 0.7) + fo.p1_mass_local(co, ce, 0.1) + fo.p1_advection_local(co, ce, vcell, 2.5)),
        )
        for kw, local in cases:
            R = fo.assemble_matrix(len(co), ce, local).tocsr()
            gpu.apply_operator(V, x, y, **kw)
            ref = R @ xh
            scale = (abs(R) @ np.abs(xh)).max()
            assert np.abs(y.get() - ref).max() <= 1e-13 * scale, kw
    with pytest.raises(gpu.BackendError):
        gpu.apply_operator(V, x, x, stiffness=1.0)                    # in place
    V3 = gpu.DeviceSpace(mesh, 3)
    with pytest.raises(gpu.BackendError):
        gpu.apply_operator(V3, gpu.DeviceVector(V3.n_local * 3), gpu.DeviceVector(V3.n_owned * 3), stiffness=1.0)


def test_matrix_free_product_on_cg2_spaces(gpu, data_dir):
    """fs_operator_apply on scalar CG2 spaces (round 6, VERDICT r5 missing #5): y = K(form) x by the row-gather walk of the CG2
    assembly with every local row multiplied into x - against the product with the ASSEMBLED operator of the same form
    (constant and per-cell stiffness, mass), on the reference's file mesh and on a box (general and snapped geometry)."""
    rng = np.random.default_rng(12)
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    for mesh in (gpu.DeviceMesh(co, ce), gpu.DeviceMesh.box(9, 7, 5, (0.0, 0.0, 0.0), (1.0, 0.7, 1.3))):
        V = gpu.DeviceSpace(mesh, 1, degree=2)
        nc = mesh.info()[1]
        xh = rng.standard_normal(V.n_local)
        x = gpu.DeviceVector(V.n_local, xh)
        y, yr = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
        kcell = rng.uniform(0.5, 2.0, nc)
        mcell = rng.uniform(0.1, 1.0, nc)
        for kw in (dict(stiffness=3.0), dict(stiffness=3.0, mass=0.4), dict(stiffness=("cell", kcell), mass=("cell", mcell)), dict(mass=2.0)):
            A = gpu.DeviceMatrix(V)
            A.assemble(**kw)
            A.spmv(x, yr)
            gpu.apply_operator(V, x, y, **kw)
            ref = yr.get()
            assert np.abs(ref).max() > 0 and np.abs(y.get() - ref).max() <= 1e-12 * np.abs(ref).max(), kw
    with pytest.raises(gpu.BackendError):
        gpu.apply_operator(V, x, y, stiffness=1.0, advection=(1.0, 0.0, 0.0))      # no advection on CG2


def test_per_cell_tensor_stiffness(gpu):
    """FS_COEF_CELL_TENSOR: one 3x3 conductivity tensor per cell, assembled and matrix-free."""
    rng = np.random.default_rng(31)
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 1.1), 4, 3, 3)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    B = rng.standard_normal((len(ce), 3, 3))
    Kc = np.einsum("cij,ckj->cik", B, B) + 0.5 * np.eye(3)[None]             # SPD per cell
    A.assemble(stiffness=("cell_tensor", Kc), mass=0.2)
    ref = fo.assemble_matrix(len(co), ce, fo.p1_stiffness_local(co, ce, Kc) + fo.p1_mass_local(co, ce, 0.2))
    M = _csr(A)
    _assert_same_pattern(M, ref)
    assert np.abs(M.data - ref.data).max() <= RTOL_ASSEMBLY * np.abs(ref.data).max()
    xh = rng.standard_normal(V.n_local)
    y = gpu.DeviceVector(V.n_owned)
    gpu.apply_operator(V, gpu.DeviceVector(V.n_local, xh), y, stiffness=("cell_tensor", Kc), mass=0.2)
    assert np.abs(y.get() - ref @ xh).max() <= 1e-13 * (abs(ref) @ np.abs(xh)).max()
    V2 = gpu.DeviceSpace(mesh, 1, degree=2)
    with pytest.raises(gpu.BackendError):
        gpu.DeviceMatrix(V2).assemble(stiffness=("cell_tensor", Kc))            # CG2: constant or per-cell scalars only


def test_bicgstab_agrees_with_cg_on_spd(gpu):
    n = 12
    P = fo.heat_box_problem(n)
    mesh = gpu.DeviceMesh.box(n, n, n)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, P["dofs"], P["vals"], symmetric=True)
    x1, x2 = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
    s1 = gpu.krylov_solve(A, b, x1, rtol=1e-10, max_iter=3000, method="cg")
    s2 = gpu.krylov_solve(A, b, x2, rtol=1e-10, max_iter=3000, method="bicgstab")
    assert s1["converged"] == 1 and s2["converged"] == 1
    assert s2["iterations"] <= s1["iterations"]            # two SpMVs per BiCGStab iteration
    assert np.abs(x1.get() - x2.get()).max() <= 1e-7 * 350
    assert np.abs(x2.get() - P["exact"]).max() <= 1e-5


def test_supg_terms_match_oracle(gpu):
    """SUPG ("SPUG") test function q + tau (v . grad q): matrix (advection + mass), source and ds(i) extras."""
    from oracle import ns_oracle as nso
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 1.3), 4, 3, 5)
    rng = np.random.default_rng(5)
    co = co + 0.02 * rng.standard_normal(co.shape) * (np.abs(co - 0.5).max(axis=1) < 0.4)[:, None]   # distort the interior
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    pe = 7.0
    for vel in (np.array([0.3, -0.2, 0.5]), rng.standard_normal((len(ce), 3))):
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=0.6, mass=11.0, advection=vel, advection_scale=4.2, supg_pe=pe)
        ref = fo.assemble_p1_scalar(co, ce, 0.6) + fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, 11.0)) \
            + fo.assemble_matrix(len(co), ce, fo.p1_advection_local(co, ce, vel, 4.2)) \
            + fo.assemble_matrix(len(co), ce, fo.p1_supg_local(co, ce, vel, pe, 4.2, 11.0))
        assert abs(_csr(A) - ref).max() <= 1e-12 * abs(ref).max()
        # mass part only (old-step operator of the time stepping): advection_scale = 0
        A.assemble(mass=11.0, advection=vel, advection_scale=0.0, supg_pe=pe)
        ref = fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, 11.0)) \
            + fo.assemble_matrix(len(co), ce, fo.p1_supg_local(co, ce, vel, pe, 0.0, 11.0))
        assert abs(_csr(A) - ref).max() <= 1e-12 * abs(ref).max()
        b = gpu.DeviceVector(V.n_owned)
        fc = rng.uniform(1, 2, len(ce))
        gpu.assemble_vector(V, b, source=("cell", fc), supg=(vel, pe))
        refb = fo.assemble_p1_source(co, ce, fc) + fo.assemble_p1_supg_source(co, ce, vel, pe, fc)
        assert np.abs(b.get() - refb).max() <= 1e-12 * np.abs(refb).max()
        th = nso.TaylorHood(co, ce)
        fcells = nso.boundary_facet_cells(th, lambda x: x[2] < 0.05 or x[0] > 0.95)
        gval, hval = rng.uniform(1, 3, len(fcells)), rng.uniform(50, 90, len(fcells))
        A.assemble(stiffness=1.0)
        base = _csr(A)
        b.fill(0.0)
        gpu.assemble_facet_supg(V, A, b, fcells[:, 0], fcells[:, 1], vel, pe, g=gval, h=hval)
        dA, db = fo.supg_facet_terms(co, ce, fcells, vel, pe, gval, hval)
        assert abs((_csr(A) - base) - dA).max() <= 1e-12 * abs(dA).max()
        assert np.abs(b.get() - db).max() <= 1e-12 * np.abs(db).max()


def test_triangle_mesh_kernels_match_oracle(gpu):
    """2-D CG1 on triangles (the reference's runnable examples are 2-D): pattern, stiffness / mass / advection,
    sources, boundary-edge loads and Robin matrices, Dirichlet + CG, against the numpy oracle."""
    co, ce = fo.rectangle_mesh((0.0, 0.0), (1.0, 0.6), 7, 5)
    rng = np.random.default_rng(2)
    inner = (co[:, 0] > 0) & (co[:, 0] < 1) & (co[:, 1] > 0) & (co[:, 1] < 0.6)
    co = co + 0.02 * rng.standard_normal(co.shape) * inner[:, None]
    n = len(co)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    assert V.n_owned == n
    kc = rng.uniform(0.5, 2.0, len(ce))
    vel = rng.standard_normal((len(ce), 3))
    vel[:, 2] = 0.0
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=("cell", kc), mass=3.0, advection=vel, advection_scale=1.7)
    ref = fo.assemble_generic(n, ce, fo.tri_stiffness_local(co, ce, kc) + fo.tri_mass_local(co, ce, 3.0)
                              + fo.tri_advection_local(co, ce, vel[:, :2], 1.7))
    M = _csr(A)
    _assert_same_pattern(M, fo.assemble_generic(n, ce, np.ones((len(ce), 3, 3))))
    assert abs(M - ref).max() <= RTOL_ASSEMBLY * abs(ref).max()
    b = gpu.DeviceVector(n)
    fn = np.sin(3 * co[:, 0]) + co[:, 1]
    gpu.assemble_vector(V, b, source=("nodal", fn))
    assert np.abs(b.get() - fo.assemble_tri_source(co, ce, f_nodal=fn)).max() <= 1e-14
    gpu.assemble_vector(V, b, source=("cell", kc))
    assert np.abs(b.get() - fo.assemble_tri_source(co, ce, kc)).max() <= 1e-14
    # boundary edges
    edges, cf, cnt = fo.tri_edge_numbering(ce)
    fm = fo.mark_edges(co, ce, lambda x, ob: ob and abs(x[0] - 1.0) < 1e-12, 1)
    fm = fo.mark_edges(co, ce, lambda x, ob: ob and abs(x[1]) < 1e-12, 2, fm)
    e1, e2 = edges[fm == 1], edges[fm == 2]
    b.fill(0.0)
    gpu.assemble_facet_vector(V, b, e1, 36.0)
    assert np.abs(b.get() - fo.assemble_edge_load(co, edges, fm, 1, 36.0)).max() <= 1e-13
    A.assemble(stiffness=1.0)
    base = _csr(A)
    A.add_facet_mass(e2, 100.0)
    assert abs((_csr(A) - base) - fo.assemble_edge_mass(co, edges, fm, 2, 100.0)).max() <= 1e-12
    # Dirichlet + CG: the linear function is reproduced exactly
    A.assemble(stiffness=2.5)
    b.fill(0.0)
    bnd = np.nonzero(~inner)[0].astype(np.int32)
    exact = 3.0 * co[:, 0] - 2.0 * co[:, 1] + 1.0
    A.apply_dirichlet(b, bnd, exact[bnd], symmetric=True)
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-13, max_iter=2000)
    assert st["converged"] == 1 and np.abs(x.get() - exact).max() <= 1e-10


def test_locality_order_of_a_shuffled_mesh(gpu, data_dir):
    """fs_mesh_locality_order: both outputs are permutations; uploading the mesh in that order and un-permuting gives the SAME
    operator bit for bit in structure and <= 1e-12 in value, and the same solution (parity on the permuted data/mesh.xml)."""
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    rng = np.random.default_rng(3)
    p = rng.permutation(len(co))                      # file id of every original vertex
    co_f = np.empty_like(co)
    co_f[p] = co
    ce_f = np.sort(p[ce], axis=1)[rng.permutation(len(ce))].astype(np.int32)
    vo, cord = gpu.locality_order(co_f, ce_f)
    assert np.array_equal(np.sort(vo), np.arange(len(co))) and np.array_equal(np.sort(cord), np.arange(len(ce)))
    dev_of_file = np.empty(len(co), dtype=np.int64)
    dev_of_file[vo] = np.arange(len(co))
    # locality: consecutive vertices are close (the file order has mean distance ~ domain size)
    d_new = np.linalg.norm(np.diff(co_f[vo], axis=0), axis=1).mean()
    d_old = np.linalg.norm(np.diff(co_f, axis=0), axis=1).mean()
    assert d_new < 0.25 * d_old
    mesh = gpu.DeviceMesh(co_f[vo], np.sort(dev_of_file[ce_f[cord]], axis=1).astype(np.int32))
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    M = _csr(A)
    q = dev_of_file[p]                                # device id of every original vertex
    Mo = M[q][:, q].tocsr()
    Mo.sort_indices()
    R = fo.assemble_p1_scalar(co, ce, 20.0)
    _assert_same_pattern(Mo, R)
    assert np.abs(Mo.data - R.tocsr().data).max() <= RTOL_ASSEMBLY * np.abs(R.data).max()
    # config 1 on the renumbered upload: T = 350 - 2.5 z in the original numbering
    _, fm = fo.read_dolfin_xml_meshfunction(os.path.join(data_dir, "mesh_facet_region.xml"))
    facets, _, _ = fo.facet_numbering(ce)
    d1, d2 = fo.dirichlet_dofs_p1(facets, fm, 1), fo.dirichlet_dofs_p1(facets, fm, 2)
    dofs = q[np.concatenate([d1, d2])].astype(np.int32)
    vals = np.concatenate([np.full(len(d1), 350.0), np.full(len(d2), 300.0)])
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=1e-13, max_iter=5000)
    assert st["converged"] == 1
    assert np.abs(x.get()[q] - (350.0 - 2.5 * co[:, 2])).max() <= 1e-9


def _box_system(gpu, mesh, n, conductivity=20.0, mass=None):
    """Heat-box operator and load with the Dirichlet pair on the z-faces, on the given device mesh of the n^3 cube."""
    P = fo.heat_box_problem(n)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=conductivity, mass=mass)
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, source=3.0)
    A.apply_dirichlet(b, P["dofs"], P["vals"], symmetric=True)
    return P, V, A, b


@pytest.mark.parametrize("n,pipelined", [(20, False), (33, False), (20, True)])
def test_row_dictionary_product_is_the_streaming_product(gpu, n, pipelined):
    """Uniform box + constant coefficients: the scaled operator has a few dozen DISTINCT rows (the box assembly snaps its edge
    vectors to the grid spacing, so equal stencils are equal bit for bit) and the CG product runs from class numbers + a dictionary
    in LDS (fs_krylov_stats.row_classes).  Same offsets, same summation order: every product equals the streaming kernel's bit for
    bit (test_row_dictionary_product_bits).  Since round 4 a lane of the dictionary kernel holds two rows of a pair of slices, so
    the three fused dot products are summed in another order than the streaming kernel's: iteration count equal, residual history
    and solution equal to rounding."""
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n, mass=0.7)
    runs = []
    try:
        for on in (1, 0):
            gpu.set_option("row_dictionary", on)
            x = gpu.DeviceVector(V.n_local)
            st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000, pipelined=pipelined)
            runs.append((st, x.get()[:V.n_owned].copy(), gpu.krylov_history().copy()))
    finally:
        gpu.set_option("row_dictionary", 1)
    (s1, x1, h1), (s0, x0, h0) = runs
    assert 0 < s1["row_classes"] <= 512 and s0["row_classes"] == 0
    assert s1["converged"] == 1 and s0["converged"] == 1 and s1["iterations"] == s0["iterations"]
    assert len(h1) == len(h0) and np.allclose(h1, h0, rtol=1e-7, atol=0.0)
    assert np.abs(x1 - x0).max() <= 1e-11 * np.abs(x0).max()
    assert s1["true_rel_residual"] <= 2e-10


@pytest.mark.parametrize("dims,mass", [((20, 20, 20), 0.7), ((33, 33, 33), None), ((70, 9, 11), 0.3), ((7, 40, 5), None), ((130, 4, 3), 1.0)])
def test_row_dictionary_product_bits(gpu, dims, mass):
    """fs_spmv_dictionary = fs_spmv BIT FOR BIT, over a chain of dependent products (each input is the previous output, rescaled on
    the device side of the API: a stale cached line of x - the kernel reads lane 63's neighbours through the scalar cache - would
    show).  Shapes with long and short mesh lines: pairs of slices inside a line, across line ends, single slices at both ends."""
    nx, ny, nz = dims
    mesh = gpu.DeviceMesh.box(nx, ny, nz)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0, mass=mass)
    rng = np.random.default_rng(5)
    xa = gpu.DeviceVector(V.n_local)
    xb = gpu.DeviceVector(V.n_local)
    ya = gpu.DeviceVector(V.n_owned)
    yb = gpu.DeviceVector(V.n_owned)
    x0 = rng.standard_normal(V.n_local)
    xa.set(x0)
    xb.set(x0)
    for it in range(12):
        nc = A.spmv_dictionary(xa, ya)
        assert nc > 0
        A.spmv(xb, yb)
        a, bb = ya.get(), yb.get()
        assert np.array_equal(a, bb), (it, np.abs(a - bb).max())
        nrm = np.abs(a).max()
        nxt = np.zeros(V.n_local)
        nxt[:V.n_owned] = a / nrm
        xa.set(nxt)
        xb.set(nxt)


@pytest.mark.parametrize("dims,mass", [((20, 20, 20), 0.7), ((33, 33, 33), None), ((70, 9, 11), 0.3), ((7, 40, 5), None), ((130, 4, 3), 1.0),
                                       ((215, 6, 5), None), ((40, 41, 30), 0.2), ((399, 3, 2), None)])
def test_marching_window_product_bits(gpu, dims, mass):
    """k_box_spmv (round 6: windows of x marching through the mesh planes of a P1 box, fs_box.h) = the streaming product = the
    work-item dictionary product, BIT FOR BIT, over a chain of dependent products.  Shapes with even and odd rows per line / per
    plane (the loader's 16-byte groups start at even indices: an odd plane stride shifts every other window by one), lines longer
    than a patch, patches that end inside a line, chunks of one plane."""
    nx, ny, nz = dims
    mesh = gpu.DeviceMesh.box(nx, ny, nz)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0, mass=mass)
    rng = np.random.default_rng(6)
    xs = [gpu.DeviceVector(V.n_local) for _ in range(3)]
    ys = [gpu.DeviceVector(V.n_owned) for _ in range(3)]
    x0 = rng.standard_normal(V.n_local)
    for v in xs:
        v.set(x0)
    try:
        gpu.set_option("box_min_rows", 0)
        for it in range(8):
            gpu.set_option("box_spmv", 1)
            assert A.spmv_dictionary(xs[0], ys[0]) > 0
            assert gpu.last_product_kind() == 3, "the marching-window product did not run"
            gpu.set_option("box_spmv", 0)
            assert A.spmv_dictionary(xs[1], ys[1]) > 0
            assert gpu.last_product_kind() == 1
            A.spmv(xs[2], ys[2])
            a, b, c = (v.get() for v in ys)
            assert np.array_equal(a, c), (it, np.abs(a - c).max(), int((a != c).sum()))
            assert np.array_equal(b, c), (it, np.abs(b - c).max())
            nxt = np.zeros(V.n_local)
            nxt[:V.n_owned] = a / np.abs(a).max()
            for v in xs:
                v.set(nxt)
    finally:
        gpu.set_option("box_spmv", 1)
        gpu.set_option("box_min_rows", 1500000)


@pytest.mark.parametrize("method,scale", [("cg", True), ("cg", False), ("bicgstab", True)])
def test_solves_through_the_marching_window_product(gpu, method, scale):
    """CG on the scaled operator (three fused sums of the product: z.z, w.z, sum d z^2), CG without the scaling (r.z, w.z, r.r) and
    BiCGStab (w.r, w.w, r.r; status word only) through k_box_spmv against the same solves through k_dict_spmv: same iteration
    counts, solutions equal to rounding (the partial sums are added in another order: patches, not work items)."""
    n = 47
    mesh = gpu.DeviceMesh.box(n, n, n)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    nn = (n + 1) ** 3
    idx = np.arange(nn)
    dofs = idx[(idx // ((n + 1) ** 2) == 0) | (idx // ((n + 1) ** 2) == n)].astype(np.int32)
    vals = np.where(dofs < (n + 1) ** 2, 350.0, 300.0)
    res = {}
    try:
        gpu.set_option("box_min_rows", 0)
        gpu.set_option("cg_fused", 0)
        for box in (1, 0):
            gpu.set_option("box_spmv", box)
            A.assemble(stiffness=20.0, mass=0.3 if method == "bicgstab" else None)
            b = gpu.DeviceVector(V.n_owned)
            gpu.assemble_vector(V, b, source=1.0)
            A.apply_dirichlet(b, dofs, vals, True)
            x = gpu.DeviceVector(V.n_owned)
            st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=3000, method=method, diagonal_scale=scale)
            assert st["converged"] == 1, st
            if method == "cg" and scale:
                assert st["row_classes"] > 0, st
            if st["row_classes"] > 0:           # (solves that do not take the dictionary form run the streaming kernels either way)
                assert gpu.last_product_kind() == (3 if box else 1), (box, gpu.last_product_kind())
            res[box] = (st, x.get())
    finally:
        gpu.set_option("box_spmv", 1)
        gpu.set_option("box_min_rows", 1500000)
        gpu.set_option("cg_fused", -1)
    (s1, x1), (s0, x0) = res[1], res[0]
    assert abs(s1["iterations"] - s0["iterations"]) <= (0 if method == "cg" else 2), (s1, s0)
    assert np.abs(x1 - x0).max() <= 1e-9 * np.abs(x0).max()
    assert s1["true_rel_residual"] <= 5e-10


@pytest.mark.parametrize("dims,ahead", [((47, 47, 47), None), ((40, 41, 30), None), ((99, 20, 12), None)])
def test_one_launch_iteration_in_marching_window_form(gpu, dims, ahead):
    """k_box_cg_iter (option "box_iter", opt-in: fs_krylov_boxiter.inc) - update k + product k + 1 of a P1 box operator with the windows
    of r, w, s through an LDS ring - against k_dict_cg_iter on the same solve: same iteration count, solutions equal to rounding
    (the dot partials have another geometry), true residual at the tolerance; shapes with odd and even strides."""
    nx, ny, nz = dims
    mesh = gpu.DeviceMesh.box(nx, ny, nz)
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    nn = (nx + 1) * (ny + 1)
    dofs = np.concatenate([np.arange(nn), np.arange(nz * nn, (nz + 1) * nn)]).astype(np.int32)
    vals = np.concatenate([np.full(nn, 350.0), np.full(nn, 300.0)])
    res = {}
    try:
        gpu.set_option("box_iter_min_rows", 0)
        gpu.set_option("cg_fused", 1)
        for on in (1, 0):
            gpu.set_option("box_iter", on)
            A.assemble(stiffness=20.0)
            b = gpu.DeviceVector(V.n_owned)
            gpu.assemble_vector(V, b, source=1.0)
            A.apply_dirichlet(b, dofs, vals, True)
            x = gpu.DeviceVector(V.n_owned)
            st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=3000)
            assert st["converged"] == 1 and st["fused_iteration"] == 1 and st["row_classes"] > 0, st
            assert st["product_kind"] == (3 if on else 1), st          # (3: the marching-window form ran)
            res[on] = (st, x.get())
    finally:
        gpu.set_option("box_iter", 0)
        gpu.set_option("box_iter_min_rows", 400000)
        gpu.set_option("cg_fused", -1)
    (s1, x1), (s0, x0) = res[1], res[0]
    assert s1["iterations"] == s0["iterations"], (s1, s0)
    assert np.abs(x1 - x0).max() <= 1e-10 * np.abs(x0).max()
    assert s1["true_rel_residual"] <= 2e-10


def test_block_row_dictionary_product_bits_and_the_amg_solve(gpu):
    """Vector P1 space on a uniform box (round 4): the 3 x 3 block rows of the elasticity operator repeat (33 classes at any size;
    the elasticity kernel snaps its edge vectors like the scalar ones) and the products of fs_amg_solve - four per V-cycle on the
    fine level, one per CG iteration - run from class numbers + class rows (k_dict_spmv3).  fs_spmv_dictionary = fs_spmv bit
    for bit over a chain of dependent vectors; the AMG-PCG solve with and without it: same iteration count, same solution."""
    nx, ny, nz = 70, 46, 46                      # 156 839 nodes: above the size from which the block form is used
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (2.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=(1.0, 1.5))
    nodes = np.arange((nx + 1) * (ny + 1) * (nz + 1))
    left = nodes[nodes % (nx + 1) == 0]
    dofs = (left[:, None] * 3 + np.arange(3)).ravel().astype(np.int32)
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=(0.0, 0.0, -1.0))
    A.apply_dirichlet(b, dofs, 0.0, True)
    rng = np.random.default_rng(9)
    xa, xb = gpu.DeviceVector(V.n_local), gpu.DeviceVector(V.n_local)
    ya, yb = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
    x0 = rng.standard_normal(V.n_local)
    xa.set(x0)
    xb.set(x0)
    for it in range(6):
        nc = A.spmv_dictionary(xa, ya)
        assert 0 < nc <= 64
        A.spmv(xb, yb)
        a, bb = ya.get(), yb.get()
        assert np.array_equal(a, bb), (it, np.abs(a - bb).max())
        nxt = a / np.abs(a).max()
        xa.set(nxt)
        xb.set(nxt)
    runs = []
    try:
        for on in (1, 0):
            gpu.set_option("row_dictionary", on)
            amg = gpu.AMG(A, nullspace="rigid_body")
            x = gpu.DeviceVector(V.n_local)
            st = amg.solve(b, x, rtol=1e-9)
            runs.append((st, x.get()[:V.n_owned].copy()))
            amg.close()
    finally:
        gpu.set_option("row_dictionary", 1)
    (s1, x1), (s0, x0_) = runs
    assert s1["converged"] == 1 and s0["converged"] == 1 and s1["iterations"] == s0["iterations"]
    assert s1["row_classes"] > 0 and s0["row_classes"] == 0
    assert np.abs(x1 - x0_).max() <= 1e-10 * np.abs(x0_).max()


def test_row_dictionary_state_does_not_outlive_its_call(gpu):
    """the class table is keyed on (value pointer, matrix, space) and dropped when the solve returns - a solve,
    a trim of the pool, then ANOTHER space of the same row count whose arrays land on the freed addresses multiplies through the
    streaming kernels, with its own values."""
    n = 24
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n)
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    assert st["converged"] == 1 and st["row_classes"] > 0
    del A, b, x, V
    gpu.trim_memory()
    rng = np.random.default_rng(11)
    mesh2 = gpu.DeviceMesh.box(n, n, n)
    V2 = gpu.DeviceSpace(mesh2, 1)
    A2 = gpu.DeviceMatrix(V2)
    A2.assemble(stiffness=("cell", 1.0 + rng.random(6 * n ** 3)))
    x2 = gpu.DeviceVector(V2.n_local)
    y2 = gpu.DeviceVector(V2.n_owned)
    xv = rng.standard_normal(V2.n_local)
    x2.set(xv)
    A2.spmv(x2, y2)
    rp, ci, va, shape = A2.to_csr()
    M = sp.csr_matrix((va, ci, rp), shape=shape)
    ref = M @ xv[:shape[1]]
    assert np.abs(y2.get() - ref).max() <= 1e-12 * np.abs(ref).max()
    assert A2.spmv_dictionary(x2, y2) == 0          # rows do not repeat: the streaming product, same result
    assert np.abs(y2.get() - ref).max() <= 1e-12 * np.abs(ref).max()


def test_box_snap_off_assembles_the_same_operator_and_the_dictionary_steps_aside(gpu):
    """A/B of the translation-invariant assembly (option box_snap): with and without snapping the operator is the oracle's to 1e-12;
    without it (spacings that are not dyadic) equal stencils differ in the last bits and the product stays with the streaming kernels."""
    n = 12
    dims = (1.0, 0.7, 1.3)
    co, ce = fo.box_mesh((0, 0, 0), dims, n, n, n)
    Ao = fo.assemble_p1_scalar(co, ce, k=20.0, mass_coef=2.0)
    got = {}
    try:
        for snap in (1, 0):
            gpu.set_option("box_snap", snap)
            mesh = gpu.DeviceMesh.box(n, n, n, (0.0, 0.0, 0.0), dims)
            V = gpu.DeviceSpace(mesh, 1)
            A = gpu.DeviceMatrix(V)
            A.assemble(stiffness=20.0, mass=2.0)
            rp, ci, va, shape = A.to_csr()
            M = sp.csr_matrix((va, ci, rp), shape=shape)
            assert abs(M - Ao).max() <= 1e-12 * abs(Ao).max()
            x = gpu.DeviceVector(V.n_local)
            y = gpu.DeviceVector(V.n_owned)
            x.set(np.linspace(0.0, 1.0, V.n_local))
            got[snap] = A.spmv_dictionary(x, y)
    finally:
        gpu.set_option("box_snap", 1)
    assert got[1] > 0 and got[0] == 0


def test_row_dictionary_is_not_used_where_rows_do_not_repeat(gpu):
    """The form is found from the VALUES of every solve and verified row by row: a per-cell coefficient, or the same cube uploaded
    as a general mesh (no snapping: equal stencils differ in the last bits), leaves the streaming kernels in use; an operator
    that changes from one kind to the other between two solves is followed."""
    n = 16
    mesh = gpu.DeviceMesh.box(n, n, n)
    rng = np.random.default_rng(3)
    nc = 6 * n ** 3
    P, V, A, b = _box_system(gpu, mesh, n, conductivity=("cell", 1.0 + rng.random(nc)))
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-9, max_iter=5000)
    assert st["converged"] == 1 and st["row_classes"] == 0
    # the same matrix object re-assembled with a constant coefficient: a new matrix is needed for the form to be tried again
    # (a matrix whose rows did not repeat is not hashed again), a fresh one gets it
    A2 = gpu.DeviceMatrix(V)
    A2.assemble(stiffness=20.0)
    b2 = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b2, source=3.0)
    A2.apply_dirichlet(b2, P["dofs"], P["vals"], symmetric=True)
    st2 = gpu.krylov_solve(A2, b2, x, rtol=1e-9, max_iter=5000)
    assert st2["converged"] == 1 and st2["row_classes"] > 0
    # general upload of a box whose spacings are not dyadic: coordinates i * h carry rounding noise and nothing snaps it away
    # (with h = 1/16 the differences are exact and the rows repeat anyway)
    dims = (1.0, 0.7, 1.3)
    co, ce = fo.box_mesh((0, 0, 0), dims, n, n, n)
    gm = gpu.DeviceMesh(co, ce)
    P3, V3, A3, b3 = _box_system(gpu, gm, n)
    x3 = gpu.DeviceVector(V3.n_local)
    st3 = gpu.krylov_solve(A3, b3, x3, rtol=1e-9, max_iter=5000)
    assert st3["converged"] == 1 and st3["row_classes"] == 0
    # the device generator of the same box: snapped, repeated rows - and the same solution to rounding
    bm = gpu.DeviceMesh.box(n, n, n, (0.0, 0.0, 0.0), dims)
    P4, V4, A4, b4 = _box_system(gpu, bm, n)
    x4 = gpu.DeviceVector(V4.n_local)
    st4 = gpu.krylov_solve(A4, b4, x4, rtol=1e-9, max_iter=5000)
    assert st4["converged"] == 1 and st4["row_classes"] > 0
    assert np.abs(x4.get()[:V4.n_owned] - x3.get()[:V3.n_owned]).max() <= 1e-8 * np.abs(x3.get()).max()


def test_class_table_is_kept_only_for_a_matrix_that_still_equals_it_row_for_row(gpu):
    """The class table of a call outlives it, and the next call on the same space first compares EVERY row of its matrix with
    the row's old class, bit for bit (fs_krylov_stats.classes_kept): the same system solved again keeps the table - same bits
    of the solution; one more Dirichlet row (15 rows of a million-entry operator change), or another coefficient (all of them),
    fails the comparison, the classes are found from scratch and the result is that of the streaming kernels."""
    n = 24
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n)

    def solve(AA, bb):
        x = gpu.DeviceVector(V.n_local)
        st = gpu.krylov_solve(AA, bb, x, rtol=1e-10, max_iter=5000)
        assert st["converged"] == 1
        return st, x.get()[:V.n_owned].copy()

    def streaming(AA, bb):
        gpu.set_option("row_dictionary", 0)
        try:
            st, x = solve(AA, bb)
        finally:
            gpu.set_option("row_dictionary", 1)
        assert st["row_classes"] == 0
        return x

    s1, x1 = solve(A, b)
    assert s1["row_classes"] > 0 and s1["classes_kept"] == 0           # a space never seen before
    s2, x2 = solve(A, b)
    assert s2["classes_kept"] == 1 and s2["row_classes"] == s1["row_classes"] and s2["iterations"] == s1["iterations"]
    assert np.array_equal(x2, x1)
    # one interior node pinned: its row and its neighbours' change, nothing else
    node = np.array([(n + 1) ** 2 * (n // 2) + (n + 1) * (n // 2) + n // 2], dtype=P["dofs"].dtype)
    A.apply_dirichlet(b, node, np.array([400.0]), symmetric=True)
    s3, x3 = solve(A, b)
    assert s3["row_classes"] > s1["row_classes"] and s3["classes_kept"] == 0
    assert abs(x3[node[0]] - 400.0) <= 1e-9 * 400.0
    ref = streaming(A, b)
    assert np.abs(x3 - ref).max() <= 1e-9 * np.abs(ref).max()
    # ... and the product on the changed matrix, bit for bit (a stale table would be off in those 15 rows)
    xv = gpu.DeviceVector(V.n_local)
    ya = gpu.DeviceVector(V.n_owned)
    yb = gpu.DeviceVector(V.n_owned)
    xv.set(np.random.default_rng(11).standard_normal(V.n_local))
    assert A.spmv_dictionary(xv, ya) > 0
    A.spmv(xv, yb)
    assert np.array_equal(ya.get(), yb.get())
    # another coefficient on the same space: every row differs
    A.assemble(stiffness=23.0)
    b5 = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b5, source=3.0)
    A.apply_dirichlet(b5, P["dofs"], P["vals"], symmetric=True)
    kept = []
    for rep in range(4):                # (after a failed comparison the next attempts are skipped: 1, 2, 4 ... calls)
        s5, x5 = solve(A, b5)
        kept.append(s5["classes_kept"])
        assert s5["row_classes"] == s1["row_classes"]
        if rep == 0:
            ref5 = streaming(A, b5)
        assert np.abs(x5 - ref5).max() <= 1e-9 * np.abs(ref5).max()
    assert kept[0] == 0 and kept[-1] == 1


def test_pattern_built_row_by_row_is_the_pattern_of_the_sorted_keys(gpu, tmp_path):
    """CG1 spaces get their sparsity pattern row by row from the sorted (vertex, cell) incidences (k_row_columns: a small set per
    row in LDS) instead of from 12 sorted keys per cell; FS_PATTERN_BY_ROWS=0 keeps the sorted-keys path.  Two processes, one per
    path: row pointers, column indices, assembled values and a product of a box, a vector space, a shuffled (file-like) cube,
    a triangle mesh and the CG2 spaces of a box and of the triangles are the same arrays, bit for bit; a triangle fan whose centre
    has more neighbours than the per-row set holds falls back to the sorted keys."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for tag, env in (("rows", {}), ("keys", {"FS_PATTERN_BY_ROWS": "0"})):
        f = str(tmp_path / (tag + ".npz"))
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "pattern_worker.py"), f], env=dict(os.environ, FS_SPACE_DEBUG="1", **env),
                           cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        log = p.stdout.decode()
        assert p.returncode == 0, log[-2000:]
        assert log.count("sparsity pattern row by row: yes") == (6 if tag == "rows" else 0), log[-2000:]
        assert log.count("sparsity pattern row by row: no") == (1 if tag == "rows" else 0), log[-2000:]       # (the fan)
        files.append(np.load(f))
    a, b = files
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 28
    for k in a.files:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


def test_box_assembly_is_translation_invariant_and_matches_the_oracle(gpu):
    """fs_mesh_create_box meshes: interior rows of the P1 operator are identical BIT FOR BIT (edge vectors snapped to the grid
    spacing), and the values are the oracle's to rounding."""
    n = 12
    mesh = gpu.DeviceMesh.box(n, n, n, (0.0, 0.0, 0.0), (1.0, 0.7, 1.3))
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0, mass=2.0)
    rp, ci, va, shape = A.to_csr()
    M = sp.csr_matrix((va, ci, rp), shape=shape)
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 1.3), n, n, n)
    Ao = fo.assemble_p1_scalar(co, ce, k=20.0, mass_coef=2.0)
    assert abs(M - Ao).max() <= 1e-12 * abs(Ao).max()
    cnt = np.diff(rp)
    rows = np.nonzero(cnt == 15)[0]
    vals = va[rp[rows][:, None] + np.arange(15)]
    assert len(rows) == (n - 1) ** 3 and len(np.unique(vals, axis=0)) == 1


@pytest.mark.parametrize("dims,p1", [((12, 12, 12), (1.0, 0.7, 1.3)), ((33, 5, 9), (2.0, 1.0, 1.0)), ((7, 40, 5), (1.0, 1.0, 1.0))])
def test_box_assembly_fast_path_bits(gpu, dims, p1):
    """k_assemble_p1_box_gather (round 6: the P1 scalar assembly of a box mesh from the reference rows of its six cell types) writes
    the values of the general row-gather kernel BIT FOR BIT: constant and per-cell stiffness, with and without a mass term, A = and
    A += forms.  Option "box_assembly" switches between the two."""
    nx, ny, nz = dims
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), p1)
    V = gpu.DeviceSpace(mesh, 1)
    rng = np.random.default_rng(3)
    kcell = 1.0 + rng.random(6 * nx * ny * nz)
    mcell = 0.5 + rng.random(6 * nx * ny * nz)
    got = {}
    try:
        for fast in (1, 0):
            gpu.set_option("box_assembly", fast)
            out = []
            for kw in (dict(stiffness=20.0), dict(stiffness=20.0, mass=2.0), dict(stiffness=("cell", kcell)),
                       dict(stiffness=("cell", kcell), mass=("cell", mcell))):
                A = gpu.DeviceMatrix(V)
                A.assemble(**kw)
                out.append(A.to_csr()[2].copy())
                A.assemble(add=True, **kw)              # A += the same form
                out.append(A.to_csr()[2].copy())
            got[fast] = out
    finally:
        gpu.set_option("box_assembly", 1)
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b), (np.abs(a - b).max(), int((a != b).sum()))
    co, ce = fo.box_mesh((0, 0, 0), p1, nx, ny, nz)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=("cell", kcell), mass=2.0)
    rp, ci, va, shape = A.to_csr()
    Ao = fo.assemble_p1_scalar(co, ce, k=kcell, mass_coef=2.0)
    assert abs(sp.csr_matrix((va, ci, rp), shape=shape) - Ao).max() <= 1e-12 * abs(Ao).max()


@pytest.mark.parametrize("dims,p1", [((8, 8, 8), (1.0, 0.7, 1.3)), ((17, 4, 6), (2.0, 1.0, 1.0))])
def test_box_assembly_fast_path_bits_cg2(gpu, dims, p1):
    """k_assemble_p2_box_gather against the general CG2 row-gather kernel, bit for bit (constant / per-cell stiffness, mass, A +=)."""
    nx, ny, nz = dims
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), p1)
    V = gpu.DeviceSpace(mesh, 1, degree=2)
    rng = np.random.default_rng(4)
    kcell = 1.0 + rng.random(6 * nx * ny * nz)
    mcell = 0.5 + rng.random(6 * nx * ny * nz)
    got = {}
    try:
        for fast in (1, 0):
            gpu.set_option("box_assembly", fast)
            out = []
            for kw in (dict(stiffness=20.0), dict(stiffness=20.0, mass=2.0), dict(stiffness=("cell", kcell), mass=("cell", mcell)), dict(mass=3.0)):
                A = gpu.DeviceMatrix(V)
                A.assemble(**kw)
                out.append(A.to_csr()[2].copy())
                A.assemble(add=True, **kw)
                out.append(A.to_csr()[2].copy())
            got[fast] = out
    finally:
        gpu.set_option("box_assembly", 1)
    for a, b in zip(got[1], got[0]):
        assert np.abs(a).max() > 0 and np.array_equal(a, b), (np.abs(a - b).max(), int((a != b).sum()))


def test_row_dictionary_buffers_may_move_between_solves(gpu):
    """The captured CG batches bake the dictionary's buffers in: a larger space in between re-allocates them, and the batches of the
    first space must be captured again (they are keyed on those buffers) - same solution before and after."""
    def solve(n):
        mesh = gpu.DeviceMesh.box(n, n, n)
        P, V, A, b = _box_system(gpu, mesh, n)
        x = gpu.DeviceVector(V.n_local)
        st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
        assert st["converged"] == 1 and st["row_classes"] > 0 and st["iterations"] > 64      # beyond the first (uncaptured) batch
        return (mesh, V, A, b), x.get()[:V.n_owned].copy(), st
    keep, x_small, st_small = solve(24)
    _, x_big, _ = solve(40)                      # larger class / descriptor arrays
    mesh, V, A, b = keep
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    assert st["iterations"] == st_small["iterations"] and np.array_equal(x.get()[:V.n_owned], x_small)


@pytest.mark.parametrize("n,rtol", [(12, 1e-8), (24, 1e-10), (40, 1e-12)])
def test_one_launch_cg_iteration_is_the_two_launch_iteration_bit_for_bit(gpu, n, rtol):
    """k_dict_cg_iter (fs_krylov.hip): launch k = update of iteration k + product of iteration k + 1, the new residual on the
    neighbour columns recomputed from the old r, w, s with the owner's two fmas.  Same operations on the same operands, same
    partial-sum geometry: iteration count, residual history and solution EQUAL the two-launch iteration's, bit for bit - through
    the plain launches of the first batch, the captured batches after it (n = 40: more than 64 iterations), and a restart from a
    nonzero guess; and both are the oracle's Jacobi-PCG."""
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n, mass=0.7 if n == 24 else None)
    got = {}
    try:
        for fused in (1, 0):
            gpu.set_option("cg_fused", fused)
            x = gpu.DeviceVector(V.n_local)
            st = gpu.krylov_solve(A, b, x, rtol=rtol, max_iter=5000)
            hist = gpu.krylov_history()
            # continue from a perturbed solution (nonzero guess: the first product of the pass is the plain kernel's)
            x1 = x.get().copy()
            x1[:V.n_owned] *= 1.0 + 1e-3 * np.cos(np.arange(V.n_owned))
            xg = gpu.DeviceVector(V.n_local)
            xg.set(x1)
            st2 = gpu.krylov_solve(A, b, xg, rtol=rtol, max_iter=5000, nonzero_guess=True)
            got[fused] = (st, x.get()[:V.n_owned].copy(), np.array(hist), st2, xg.get()[:V.n_owned].copy())
    finally:
        gpu.set_option("cg_fused", -1)
    (s1, x1, h1, r1, y1), (s0, x0, h0, r0, y0) = got[1], got[0]
    assert s1["fused_iteration"] == 1 and s0["fused_iteration"] == 0 and s1["row_classes"] > 0
    assert s1["converged"] == 1 and s1["iterations"] == s0["iterations"] and (n < 40 or s1["iterations"] > 64)
    assert np.array_equal(h1, h0) and np.array_equal(x1, x0)
    assert r1["iterations"] == r0["iterations"] and np.array_equal(y1, y0) and r1["fused_iteration"] == 1
    rp, ci, va, shape = A.to_csr()                  # (the constrained operator the solves ran on)
    xo, ito, _ = fo.pcg_jacobi_single_reduction(sp.csr_matrix((va, ci, rp), shape=shape), b.get()[:shape[0]], rtol=rtol)
    assert abs(s1["iterations"] - ito) <= 1
    assert np.abs(x1 - xo).max() <= 1e-7 * np.abs(xo).max()


def test_host_follows_the_one_launch_iteration_through_pinned_progress_words(gpu):
    """Round 5: the leader lane of k_dict_cg_iter writes the iteration in progress and - once the recurrence stops - the status word
    into pinned host memory; the host keeps cg_ahead .. cg_ahead + cg_sub launches enqueued instead of batches of 32 with the status
    word copied back behind each.  Same iterates, same histories, bit for bit, for several (cg_sub, cg_ahead); far fewer launches
    behind the last iteration (fs_krylov_stats.launches); the iteration limit and a solve converged from the start behave alike.
    And the class table kept from the first solve is compared with the later matrices WITHOUT the scaled copy being written
    (FS_LAZY_SCALE_COPY: values scaled on the fly in k_dict_finish) - the solves above are that path from the second one on."""
    n = 40
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n)
    got = {}
    try:
        for mode in ((0, 16, 6), (1, 16, 6), (1, 2, 1), (1, 8, 40), (1, 64, 3)):
            gpu.set_option("cg_mirror", mode[0]); gpu.set_option("cg_sub", mode[1]); gpu.set_option("cg_ahead", mode[2])
            x = gpu.DeviceVector(V.n_local)
            st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
            hist = np.array(gpu.krylov_history())
            y = gpu.DeviceVector(V.n_local)
            lim = gpu.krylov_solve(A, b, y, rtol=1e-14, max_iter=37)
            z = gpu.DeviceVector(V.n_local)
            zero = gpu.krylov_solve(A, b, z, rtol=1.0, max_iter=50)
            got[mode] = (st, x.get()[:V.n_owned].copy(), hist, lim, y.get()[:V.n_owned].copy(), zero)
    finally:
        gpu.set_option("cg_mirror", 1); gpu.set_option("cg_sub", 16); gpu.set_option("cg_ahead", 6)
    s0, x0, h0, l0, y0, z0 = got[(0, 16, 6)]
    assert s0["fused_iteration"] == 1 and s0["converged"] == 1 and s0["iterations"] > 64 and s0["classes_kept"] == 0
    assert s0["launches"] - s0["iterations"] >= 1 and got[(1, 16, 6)][0]["classes_kept"] == 1 and got[(0, 16, 6)][3]["classes_kept"] == 1
    for mode, (st, x, h, lim, y, zero) in got.items():
        assert st["fused_iteration"] == 1 and st["iterations"] == s0["iterations"] and st["converged"] == 1, mode
        assert np.array_equal(h, h0) and np.array_equal(x, x0), mode
        assert lim["iterations"] == 37 and lim["converged"] == 0 and np.array_equal(y, y0), mode
        assert zero["iterations"] == 0 and zero["converged"] == 1, mode
        if mode[0]:
            assert 1 <= st["launches"] - st["iterations"] <= mode[1] + mode[2] + 1, (mode, st["launches"], st["iterations"])
    assert got[(1, 16, 6)][0]["launches"] < s0["launches"]


def test_one_launch_cg_iteration_stops_at_the_iteration_limit_like_the_two_launch_one(gpu):
    """max_iter below what the tolerance needs: both iterations stop after exactly max_iter steps with the same iterate (status 3
    path of k_dict_cg_iter: the launch that sees iter == limit only checks); and a solve that is converged from the start
    (zero right-hand side rows apart from the boundary lift, tolerance 1) takes zero iterations on both paths."""
    n = 20
    mesh = gpu.DeviceMesh.box(n, n, n)
    P, V, A, b = _box_system(gpu, mesh, n)
    got = {}
    try:
        for fused in (1, 0):
            gpu.set_option("cg_fused", fused)
            x = gpu.DeviceVector(V.n_local)
            st = gpu.krylov_solve(A, b, x, rtol=1e-14, max_iter=37)
            y = gpu.DeviceVector(V.n_local)
            st0 = gpu.krylov_solve(A, b, y, rtol=1.0, max_iter=50)
            got[fused] = (st, x.get()[:V.n_owned].copy(), st0)
    finally:
        gpu.set_option("cg_fused", -1)
    (s1, x1, z1), (s0, x0, z0) = got[1], got[0]
    assert s1["fused_iteration"] == 1 and s0["fused_iteration"] == 0
    assert s1["iterations"] == s0["iterations"] == 37 and s1["converged"] == s0["converged"] == 0
    assert np.array_equal(x1, x0)
    assert z1["iterations"] == z0["iterations"] == 0 and z1["converged"] == z0["converged"] == 1
