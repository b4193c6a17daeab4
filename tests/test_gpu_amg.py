"""Smoothed-aggregation AMG (fs_amg_*; the PETSc GAMG role of SolverBase.solve_amg, SolverBase.py:643-672).

There is no reference arithmetic to pin an AMG hierarchy to (aggregation in GAMG is itself
implementation-defined), so the hierarchy is held to the properties that define the method, with
scipy as the checker:
  * Galerkin:    A_{l+1} == P_l^T A_l P_l                       (the two row-wise SpGEMMs)
  * near-null space: P_l B_{l+1} == B_l wherever A_l B_l == 0    (tentative QR + prolongator smoothing)
  * the V-cycle is a symmetric positive definite operator        (CG may be used outside)
  * the preconditioned solve returns the oracle's direct solution; iteration counts are mesh independent
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def _poisson(gpu, n, seed=0, variable=True):
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    rng = np.random.default_rng(seed)
    kc = rng.uniform(0.5, 1.5, len(ce)) if variable else 1.0
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=("cell", kc) if variable else 1.0)
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, source=1.0)
    bot, top = np.nonzero(co[:, 2] == 0)[0], np.nonzero(co[:, 2] == 1)[0]
    dofs = np.concatenate([bot, top]).astype(np.int32)
    vals = np.concatenate([np.full(len(bot), 1.0), np.full(len(top), 2.0)])
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    K = fo.assemble_p1_scalar(co, ce, kc)
    rhs = fo.assemble_p1_source(co, ce, 1.0)
    Ab, bb = fo.apply_dirichlet(K, rhs, dofs, vals, True)
    return V, A, b, Ab.tocsr(), bb


def _elasticity(gpu, dims=(12, 3, 3), clamp_components=None):
    co, ce = fo.box_mesh((0, 0, 0), (4.0, 1.0, 1.0), *dims)
    E, nu = 2e11, 0.27
    mu, lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, ncomp=3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=(mu, lam))
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=[0.0, -7.8e4, 0.0])
    left = np.nonzero(co[:, 0] == 0)[0]
    comps = np.arange(3) if clamp_components is None else np.asarray(clamp_components)
    dofs = (left[:, None] * 3 + comps).ravel().astype(np.int32)
    if clamp_components is not None:       # pin the remaining rigid-body modes (y, z translation, x rotation)
        xm = co[:, 0].max()
        na = np.nonzero((co[:, 0] == xm) & (co[:, 1] == 0) & (co[:, 2] == 0))[0][0]
        nb_ = np.nonzero((co[:, 0] == xm) & (co[:, 1] == 1) & (co[:, 2] == 0))[0][0]
        extra = np.array([na * 3 + 1, na * 3 + 2, nb_ * 3 + 2], dtype=np.int32)
        dofs = np.unique(np.concatenate([dofs, extra])).astype(np.int32)
    vals = np.zeros(len(dofs))
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    K = fo.assemble_p1_elasticity(co, ce, E, nu)
    rhs = fo.assemble_p1_vector_source(co, ce, (0.0, -7.8e4, 0.0))
    Ab, bb = fo.apply_dirichlet(K, rhs, dofs, vals, True)
    return V, A, b, Ab.tocsr(), bb, fo.rigid_body_modes(co)


def _check_hierarchy(amg, A0, B0, nb):
    info = amg.info()
    assert info["levels"] >= 2
    A = A0
    B = B0
    for l in range(info["levels"] - 1):
        Al = amg.level_matrix(l, "A")
        scale = abs(Al).max()
        assert abs(Al - A).max() <= 1e-12 * scale, l          # level 0: the assembled matrix itself
        P = amg.level_matrix(l, "P")
        Ac = amg.level_matrix(l + 1, "A")
        ref = (P.T @ Al @ P).tocsr()
        # dead coarse dofs get a unit diagonal; everything else is the Galerkin product
        d = abs(Ac - ref)
        dead = np.asarray(abs(ref).sum(axis=1)).ravel() == 0
        d = d.tolil()
        for k in np.nonzero(dead)[0]:
            assert Ac[k, k] == 1.0
            d[k, k] = 0.0
        assert d.tocsr().max() <= 1e-11 * abs(ref).max(), l
        assert abs(Ac - Ac.T).max() <= 1e-11 * abs(Ac).max(), l
        # near-null space carried to the next level
        Bl = amg.level_nullspace(l, nb)
        if l == 0:
            ident = np.asarray(abs(Al - sp.diags(Al.diagonal())).sum(axis=1)).ravel() == 0
            assert np.allclose(Bl[~ident], B[~ident], rtol=0, atol=1e-13 * np.abs(B).max())
        Bc = amg.level_nullspace(l + 1, nb)
        resid = Al @ Bl
        interp = P @ Bc
        li = amg.level_info(l)
        bs = li["block_size"]
        # rows whose whole node sees A B == 0 (interior, away from eliminated dofs) and that are aggregated
        ident_rows = np.asarray(abs(Al - sp.diags(Al.diagonal())).sum(axis=1)).ravel() == 0
        ok = (np.abs(resid) <= 1e-9 * (abs(Al) @ np.abs(Bl)) + 1e-300).all(axis=1) & ~ident_rows
        ok_node = ok.reshape(-1, bs).all(axis=1)
        nbr = (abs(Al) > 0).astype(np.int8)
        node_of = np.repeat(np.arange(len(ok_node)), bs)
        G = sp.csr_matrix((np.ones(nbr.nnz), (node_of[nbr.tocoo().row], node_of[nbr.tocoo().col])),
                          shape=(len(ok_node), len(ok_node)))
        all_nbrs_ok = np.asarray(G @ (~ok_node).astype(float)).ravel() == 0
        rows = np.repeat(ok_node & all_nbrs_ok, bs)
        assert rows.sum() > 0 or l > 0
        if rows.any():
            assert np.abs(interp[rows] - Bl[rows]).max() <= 1e-8 * np.abs(Bl).max(), l
        A = Ac
    return info


def test_amg_poisson_hierarchy_and_solve(gpu):
    V, A, b, Ab, bb = _poisson(gpu, 14)
    amg = gpu.AMG(A, coarse_size=20)
    info = _check_hierarchy(amg, Ab, np.ones((V.n_owned, 1)), 1)
    assert info["levels"] >= 3 and info["operator_complexity"] < 2.0
    x = gpu.DeviceVector(V.n_local)
    st = amg.solve(b, x, rtol=1e-10)
    ref = fo.solve_direct(Ab, bb)
    assert st["converged"] == 1 and st["iterations"] <= 25
    assert st["true_rel_residual"] <= 2e-10
    assert np.abs(x.get()[:V.n_owned] - ref).max() <= 1e-8 * np.abs(ref).max()


def test_amg_vcycle_is_symmetric_positive_definite(gpu):
    V, A, b, Ab, bb = _poisson(gpu, 5)
    amg = gpu.AMG(A, coarse_size=20)
    n = V.n_owned
    M = np.empty((n, n))
    r = gpu.DeviceVector(n)
    z = gpu.DeviceVector(V.n_local)
    e = np.zeros(n)
    for i in range(n):
        e[:] = 0.0
        e[i] = 1.0
        r.set(e)
        amg.apply(r, z)
        M[:, i] = z.get()[:n]
    assert np.abs(M - M.T).max() <= 1e-10 * np.abs(M).max()
    w = np.linalg.eigvalsh(0.5 * (M + M.T))
    assert w.min() > 0
    # spectrum of the preconditioned operator: clustered well away from 0
    ev = np.linalg.eigvals(M @ Ab.toarray()).real
    assert ev.min() > 0.3 and ev.max() < 1.6


def test_amg_iterations_do_not_grow_with_the_mesh(gpu):
    its = []
    for n in (8, 16, 32):
        V, A, b, Ab, bb = _poisson(gpu, n, variable=False)
        amg = gpu.AMG(A)
        x = gpu.DeviceVector(V.n_local)
        st = amg.solve(b, x, rtol=1e-8)
        assert st["converged"] == 1 and st["true_rel_residual"] <= 2e-8
        its.append(st["iterations"])
        xj = gpu.DeviceVector(V.n_local)
        sj = gpu.krylov_solve(A, b, xj, rtol=1e-8, max_iter=5000)
        assert np.abs(x.get() - xj.get()).max() <= 1e-6 * np.abs(xj.get()).max()
    assert max(its) <= 20 and its[-1] <= its[0] + 6, its


@pytest.mark.parametrize("clamp", [None, (0,)])
def test_amg_elasticity_rigid_body_modes(gpu, clamp):
    V, A, b, Ab, bb, rbm = _elasticity(gpu, clamp_components=clamp)
    amg = gpu.AMG(A, nullspace=rbm, coarse_size=100)
    info = _check_hierarchy(amg, Ab, rbm.T.copy(), 6)
    assert amg.level_info(1)["block_size"] == 6
    x = gpu.DeviceVector(V.n_local)
    st = amg.solve(b, x, rtol=1e-10)
    ref = fo.solve_direct(Ab, bb)
    assert st["converged"] == 1 and st["iterations"] <= 40, st
    assert np.abs(x.get()[:V.n_owned] - ref).max() <= 1e-7 * np.abs(ref).max()
    # the same solve with Jacobi needs an order of magnitude more iterations
    xj = gpu.DeviceVector(V.n_local)
    sj = gpu.krylov_solve(A, b, xj, rtol=1e-10, max_iter=20000)
    assert sj["iterations"] > 5 * st["iterations"]


def test_fp32_storage_of_coarse_and_transfer_operators_leaves_the_solve_alone(gpu):
    """The V-cycle streams its 6 x 6-block coarse operators and its transfer operators rounded to fp32 (fs_amg.hip: half the bytes
    of what bounds it); vectors, accumulation and the CG outside stay fp64.  Held here: the preconditioner is still symmetric
    positive definite (to the rounding of its stored numbers), the iteration count is that of fp64 storage +- 1, the solution the
    oracle's, and the inspection hook keeps returning the fp64 Galerkin operators."""
    V, A, b, Ab, bb, rbm = _elasticity(gpu, dims=(8, 3, 3))
    ref = fo.solve_direct(Ab, bb)
    n = V.n_owned
    out = {}
    try:
        for fp32 in (0, 1):
            gpu.set_option("amg_coarse_fp32", fp32)
            amg = gpu.AMG(A, nullspace=rbm, coarse_size=60)
            assert amg.level_info(1)["block_size"] == 6
            _check_hierarchy(amg, Ab, rbm.T.copy(), 6)
            x = gpu.DeviceVector(V.n_local)
            st = amg.solve(b, x, rtol=1e-10)
            assert st["converged"] == 1
            assert np.abs(x.get()[:n] - ref).max() <= 1e-7 * np.abs(ref).max()
            M = np.empty((n, n))
            r = gpu.DeviceVector(n)
            z = gpu.DeviceVector(V.n_local)
            e = np.zeros(n)
            for i in range(n):
                e[:] = 0.0
                e[i] = 1.0
                r.set(e)
                amg.apply(r, z)
                M[:, i] = z.get()[:n]
            out[fp32] = (st["iterations"], M)
            amg.close()
    finally:
        gpu.set_option("amg_coarse_fp32", 1)
    (it64, M64), (it32, M32) = out[0], out[1]
    assert abs(it32 - it64) <= 1, (it64, it32)
    assert np.abs(M64 - M64.T).max() <= 1e-10 * np.abs(M64).max()
    assert np.abs(M32 - M32.T).max() <= 1e-6 * np.abs(M32).max()          # fp32 rounding of the two Galerkin summation orders
    assert np.linalg.eigvalsh(0.5 * (M32 + M32.T)).min() > 0
    assert 0 < np.abs(M32 - M64).max() <= 1e-5 * np.abs(M64).max()       # the storage IS different, by fp32 rounding only


def test_amg_rejects_bad_input(gpu):
    from fenicssolver_amd._lib import BackendError
    V, A, b, Ab, bb = _poisson(gpu, 4)
    with pytest.raises(ValueError):
        gpu.AMG(A, nullspace=np.ones((1, 3)))
    with pytest.raises(BackendError):
        gpu.AMG(A, nullspace=np.ones((2, V.n_owned)))      # 2 vectors: not 1, 3 or 6


def test_solve_amg_is_the_default_of_the_elasticity_solver(gpu):
    """LinearElasticitySolver.solve_form -> solve_amg in 3D (LinearElasticitySolver.py:247-253): AMG with the
    rigid-body modes of build_nullspace unless solver_parameters name another preconditioner."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 40, 4, 4)

    def make(**params):
        bcs = OrderedDict()
        bcs["fixed"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0)), 'boundary_id': 1, 'type': 'Dirichlet',
                        'value': Constant((0, 0, 0))}
        bcs["tip"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 10)), 'boundary_id': 2, 'type': 'stress',
                      'value': Constant((0, 0, 1e6))}
        s = copy.deepcopy(SB.default_case_settings)
        s['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800}
        s['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
        s['boundary_conditions'] = bcs
        s['solver_settings']['solver_parameters'] = dict({'krylov_relative_tolerance': 1e-10}, **params)
        s['report_settings'] = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
        return LinearElasticitySolver(s)

    a = make()
    ua = a.solve().vector().array()
    assert a.last_solve_stats['amg_levels'] >= 2 and a.last_solve_stats['iterations'] <= 60
    j = make(preconditioner='jacobi')
    uj = j.solve().vector().array()
    assert 'amg_levels' not in j.last_solve_stats and j.last_solve_stats['iterations'] > 200
    assert np.abs(ua - uj).max() <= 1e-6 * np.abs(uj).max()


def test_device_rigid_body_modes_equal_the_host_built_near_null_space(gpu):
    """nullspace='rigid_body' builds the six modes of SolverBase.build_nullspace on the device from the node coordinates:
    level-0 near-null space identical to the un-normalised host vectors, same hierarchy and the same iteration count as
    the (orthonormalised) host basis the reference passes - the tentative prolongator orthonormalises per aggregate."""
    co, ce = fo.box_mesh((0, 0, 0), (6.0, 1.0, 1.2), 24, 4, 5)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=fo.lame(2e11, 0.27))
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=(0.0, 0.0, -7.8e4))
    left = np.nonzero(co[:, 0] == 0.0)[0]
    A.apply_dirichlet(b, (left[:, None] * 3 + np.arange(3)).ravel().astype(np.int32), 0.0, symmetric=True)
    n = len(co)
    raw = np.zeros((6, n, 3))
    raw[0, :, 0] = raw[1, :, 1] = raw[2, :, 2] = 1.0
    raw[3, :, 0], raw[3, :, 1] = -co[:, 1], co[:, 0]
    raw[4, :, 0], raw[4, :, 2] = co[:, 2], -co[:, 0]
    raw[5, :, 2], raw[5, :, 1] = co[:, 1], -co[:, 2]
    dev = gpu.AMG(A, nullspace="rigid_body")
    host = gpu.AMG(A, nullspace=fo.rigid_body_modes(co))
    assert dev.info()["levels"] == host.info()["levels"]
    for l in range(dev.info()["levels"]):
        assert dev.level_info(l)["n_nodes"] == host.level_info(l)["n_nodes"]
        assert dev.level_info(l)["nnz_blocks"] == host.level_info(l)["nnz_blocks"]
    xd, xh = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
    sd = dev.solve(b, xd, rtol=1e-10)
    sh = host.solve(b, xh, rtol=1e-10)
    assert sd["converged"] == 1 and abs(sd["iterations"] - sh["iterations"]) <= 1
    assert np.abs(xd.get() - xh.get()).max() <= 1e-7 * np.abs(xh.get()).max()
    # the device near-null space itself (level 0, [dof][6]) against the host vectors
    import ctypes as C
    from fenicssolver_amd import _lib as L
    got = np.empty(V.n_owned * 6)
    L.check(L.load().fs_amg_level_get(dev.h, 0, 2, None, None, L.p_f64(got)), "fs_amg_level_get")
    fixed = np.zeros(3 * n, dtype=bool)
    fixed[(left[:, None] * 3 + np.arange(3)).ravel()] = True
    want = raw.reshape(6, -1).T.copy()
    got = got.reshape(-1, 6)
    free = ~fixed
    assert np.abs(got[free] - want[free]).max() <= 1e-14 * np.abs(want).max()
    dev.close()
    host.close()
    from fenicssolver_amd._lib import BackendError
    with pytest.raises(BackendError):
        Q = gpu.DeviceSpace(mesh, 1)
        K = gpu.DeviceMatrix(Q)
        K.assemble(stiffness=1.0)
        gpu.AMG(K, nullspace="rigid_body")


def test_hierarchy_is_reused_across_quasi_static_time_steps(gpu):
    """Transient elasticity with solving_dynamics False re-solves the static problem with time-dependent loads
    (LinearElasticitySolver.py:216-220, examples/test_linear_elasticity.py:118-121): same operator every step, so the AMG
    hierarchy of the first step serves the others; a changed Dirichlet set rebuilds it."""
    import copy
    import math
    from collections import OrderedDict
    from fenicssolver_amd.fem import BoxMesh, Point, VectorFunctionSpace, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 30, 3, 3)
    bcs = OrderedDict()
    bcs["fixed"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 0)), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}
    bcs["tensile"] = {'boundary': AutoSubDomain(lambda x: near(x[0], 10)), 'boundary_id': 2, 'type': 'stress',
                      'value': lambda t: Constant((1e8 * math.sin(100 * math.pi * 2 * t), 0, 0))}
    s = copy.deepcopy(SB.default_case_settings)
    s['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800}
    s['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    s['boundary_conditions'] = bcs
    s['solver_settings']['transient_settings'] = {'transient': True, 'starting_time': 0.0, 'time_step': 0.001, 'ending_time': 0.0035}
    s['report_settings'] = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = LinearElasticitySolver(s)
    seen = []
    orig = solver._device_solve

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append((solver.last_solve_stats['amg_reused'], solver.last_solve_stats['iterations'], float(np.abs(out.vector().array()).max())))
        return out
    solver._device_solve = spy
    solver.solve()
    assert [r for r, _, _ in seen] == [False, True, True, True]
    assert all(it <= 60 for _, it, _ in seen)
    # t = starting_time + dt (step - 1) (SolverBase.py:440-465): -dt, 0, dt, 2 dt -> the load, hence the field, differs every step
    assert seen[1][2] == 0.0 and len({round(v, 14) for _, _, v in seen}) == 3          # |sin(-w dt)| = |sin(w dt)|
