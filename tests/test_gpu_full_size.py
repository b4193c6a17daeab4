"""BASELINE.json's configurations at FULL size, checked through size-independent properties
(the oracle cannot assemble 10 M-DOF problems in seconds): exact discrete solutions the
set-ups imply, linearity in the data, symmetry of the operator, true-residual convergence and
the iteration anchors of SURVEY.md Appendix C8."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _heat_cube(gpu, n, degree=1, lo=350.0, hi=300.0, k=20.0, rtol=1e-8):
    mesh = gpu.DeviceMesh.box(n, n, n)
    V = gpu.DeviceSpace(mesh, 1, degree=degree)
    if degree == 1:
        P = (n + 1) ** 2
        dofs = np.concatenate([np.arange(P), np.arange(n * P, (n + 1) * P)])
        vals = np.concatenate([np.full(P, lo), np.full(P, hi)])
        z = None
    else:
        xyz, _, _ = mesh.get()
        ed = V.edges().astype(np.int64)
        z = np.concatenate([xyz[:, 2], 0.5 * (xyz[ed[:, 0], 2] + xyz[ed[:, 1], 2])])
        b0, b1 = np.nonzero(z == 0.0)[0], np.nonzero(z == 1.0)[0]
        dofs = np.concatenate([b0, b1])
        vals = np.concatenate([np.full(len(b0), lo), np.full(len(b1), hi)])
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=k)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=rtol, max_iter=50000)
    return mesh, V, A, x, st, z


def test_config2_p1_poisson_1m_dof(gpu):
    """configs[1]: unit cube n=99, 1 000 000 DOF, 5 821 794 tets, 14 761 198 nnz."""
    n = 99
    mesh, V, A, x, st, _ = _heat_cube(gpu, n)
    assert (V.n_owned, V.nnz) == (1000000, 14761198) and mesh.info()[1] == 5821794
    assert st["converged"] == 1 and st["true_rel_residual"] <= 1.02e-8
    assert st["iterations"] == 293                     # == the C oracle's PCG count on the same problem
    P = (n + 1) ** 2
    zc = np.repeat(np.arange(n + 1) / n, P)
    T = x.get()
    assert np.abs(T - (350.0 - 50.0 * zc)).max() <= 5e-4      # exact discrete solution is linear in z
    # linearity in the boundary data: doubling the Dirichlet values doubles the solution
    _, _, _, x2, st2, _ = _heat_cube(gpu, n, lo=700.0, hi=600.0)
    assert st2["iterations"] == st["iterations"]
    assert np.abs(x2.get() - 2.0 * T).max() <= 1e-9 * 700.0
    # symmetry of the assembled operator after symmetric elimination: <A u, v> == <u, A v>
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(V.n_owned), rng.standard_normal(V.n_owned)
    du, dv, w = gpu.DeviceVector(V.n_local, u), gpu.DeviceVector(V.n_local, v), gpu.DeviceVector(V.n_owned)
    A.spmv(du, w)
    auv = float(w.get() @ v)
    A.spmv(dv, w)
    uav = float(w.get() @ u)
    assert abs(auv - uav) <= 1e-10 * abs(auv)
    # the solve is deterministic: same bits on a second run (fixed-order reductions, atomic-free assembly)
    _, _, _, x3, _, _ = _heat_cube(gpu, n)
    assert np.array_equal(x3.get(), T)


def test_config2_family_10m_dof_hbm_resident(gpu):
    n = 215
    mesh, V, A, x, st, _ = _heat_cube(gpu, n)
    assert V.n_owned == 216 ** 3 and st["converged"] == 1 and st["true_rel_residual"] <= 1.02e-8
    if not os.environ.get("FS_DISABLE_DIA"):
        assert V.n_dia_slices == V.n_slices             # every slice of the Kuhn cube is stored in DIA form
    zc = np.repeat(np.arange(n + 1) / n, (n + 1) ** 2)
    assert np.abs(x.get() - (350.0 - 50.0 * zc)).max() <= 2e-3
    # round 6: the solve's products went through the marching-window kernel (launch shape 6 x 2: mesh lines of 216 rows) ...
    assert st["row_classes"] > 0 and st["product_kind"] == 3
    _marching_windows_equal_work_items(gpu, V, A)


def _marching_windows_equal_work_items(gpu, V, A):
    """... and k_box_spmv = k_dict_spmv BIT FOR BIT on the assembled, Dirichlet-constrained operator at this size (the production launch
    geometry: hundreds of patches x chunks of planes), on a vector of pseudo-random numbers."""
    rng = np.random.default_rng(11)
    xv = gpu.DeviceVector(V.n_local)
    xv.set(rng.standard_normal(V.n_local))
    ya, yb = gpu.DeviceVector(V.n_owned), gpu.DeviceVector(V.n_owned)
    try:
        gpu.set_option("box_spmv", 1)
        assert A.spmv_dictionary(xv, ya) > 0 and gpu.last_product_kind() == 3
        gpu.set_option("box_spmv", 0)
        assert A.spmv_dictionary(xv, yb) > 0 and gpu.last_product_kind() == 1
    finally:
        gpu.set_option("box_spmv", 1)
    a, b = ya.get(), yb.get()
    assert np.array_equal(a, b), (np.abs(a - b).max(), int((a != b).sum()))


def test_largest_single_gpu_p1_problem_86m_dof(gpu):
    """Maximum size of the 32-bit connectivity (cell-vertex incidences < 2^31): unit cube n=440, 85.8 M DOF, 511 M tets,
    6.2e9 pattern keys through the 64-bit sort - about a third of the 288 GB of one MI355X for a moment.  One size
    further the space constructor must refuse loudly."""
    n = 440
    mesh, V, A, x, st, _ = _heat_cube(gpu, n)
    assert V.n_owned == 441 ** 3 and mesh.info()[1] == 6 * n ** 3
    assert st["converged"] == 1 and st["true_rel_residual"] <= 1.02e-8
    T = x.get()
    P = (n + 1) ** 2
    zmean = T.reshape(n + 1, P).mean(axis=1)
    assert np.abs(zmean - (350.0 - 50.0 * np.arange(n + 1) / n)).max() <= 5e-3     # linear in z, plane by plane
    assert T.min() >= 300.0 - 1e-6 and T.max() <= 350.0 + 1e-6                       # discrete maximum principle
    # (launch shape 8 x 3 of the marching-window product: mesh lines of 441 rows, an ODD number of rows per plane)
    assert st["product_kind"] == 3
    _marching_windows_equal_work_items(gpu, V, A)
    del A, x, V, mesh
    gpu.trim_memory()
    big = gpu.DeviceMesh.box(480, 480, 480)                                           # 663 M tets: 4 nc >= 2^31
    with pytest.raises(gpu.BackendError):
        gpu.DeviceSpace(big, 1)
    del big
    gpu.trim_memory()


def test_config3_elasticity_cantilever_5m_dof(gpu):
    """configs[2]: BoxMesh((0,0,0),(10,1,1),472,59,59), vector P1, E=2e11, nu=0.27, clamped at x=0,
    body force; 5 108 400 DOF.  Euler-Bernoulli tip deflection q L^4 / (8 E I)."""
    nx, ny, nz = 472, 59, 59
    E, nu = 2e11, 0.27
    mu, lm = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0, 0, 0), (10.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 3)
    assert V.n_owned == 5108400 and mesh.info()[1] == 9858192
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=(mu, lm))
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=(0.0, 0.0, -7800.0 * 10.0))
    nodes = np.arange((nx + 1) * (ny + 1) * (nz + 1))
    left = nodes[nodes % (nx + 1) == 0]
    A.apply_dirichlet(b, (left[:, None] * 3 + np.arange(3)).ravel(), 0.0, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    st = gpu.krylov_solve(A, b, x, rtol=1e-8, max_iter=100000)
    assert st["converged"] == 1 and st["true_rel_residual"] <= 2e-8
    u = x.get().reshape(-1, 3)
    beam = -7800.0 * 10.0 * 10.0 ** 4 / (8 * E * (1.0 / 12.0))
    tip = u[nodes % (nx + 1) == nx, 2].mean()
    assert abs(tip - beam) <= 0.01 * abs(beam)
    assert np.abs(u[left]).max() == 0.0                     # clamped face
    assert abs(u[:, 1].mean()) <= 1e-3 * abs(tip)           # no net sideways motion (the Kuhn split is not mirror-symmetric)
    # the solve_amg path (SolverBase.py:643-672): same system, smoothed-aggregation AMG with the six rigid-body modes
    xyz = mesh.get(True, False, False)[0]
    ns = np.zeros((6, len(xyz), 3))
    ns[0, :, 0] = ns[1, :, 1] = ns[2, :, 2] = 1.0
    ns[3, :, 0], ns[3, :, 1] = -xyz[:, 1], xyz[:, 0]
    ns[4, :, 0], ns[4, :, 2] = xyz[:, 2], -xyz[:, 0]
    ns[5, :, 2], ns[5, :, 1] = xyz[:, 1], -xyz[:, 2]
    amg = gpu.AMG(A, nullspace=ns.reshape(6, -1))
    xa = gpu.DeviceVector(V.n_owned)
    sa = amg.solve(b, xa, rtol=1e-8)
    assert sa["converged"] == 1 and sa["iterations"] <= 60 and sa["true_rel_residual"] <= 4e-8
    assert sa["iterations"] * 50 < st["iterations"]          # two orders of magnitude fewer iterations than Jacobi-CG
    ua = xa.get().reshape(-1, 3)
    assert np.abs(ua - u).max() <= 2e-5 * np.abs(u).max()    # both solved to 1e-8 of a 1e9-conditioned system
    info = amg.info()
    assert info["levels"] >= 4 and info["operator_complexity"] < 1.8


def test_config4_p2_poisson_10m_dof_single_gpu(gpu):
    """configs[3] on ONE GPU: unit cube n=107, P2, 215^3 = 9 938 375 DOF, 7 350 258 tets."""
    n = 107
    mesh, V, A, x, st, z = _heat_cube(gpu, n, degree=2)
    assert V.n_owned == 215 ** 3 and mesh.info()[1] == 7350258
    assert st["converged"] == 1 and st["true_rel_residual"] <= 1.02e-8
    assert np.abs(x.get() - (350.0 - 50.0 * z)).max() <= 3e-3   # P2 reproduces the linear profile
    # the CG2 operator of the uniform cube has a few hundred distinct rows (every slice is in DIA form at this size): the product
    # ran from class numbers + a dictionary (too large for LDS as a whole: every work item - a mesh line - brings its classes
    # into its wave's region) - and follows the streaming product
    assert 0 < st["row_classes"] <= 4096
    x1, h1, it1 = x.get().copy(), gpu.krylov_history().copy(), st["iterations"]
    try:
        gpu.set_option("row_dictionary", 0)
        b = gpu.DeviceVector(V.n_owned)
        dofs = np.nonzero((z == 0.0) | (z == 1.0))[0]
        A.assemble(stiffness=20.0)
        A.apply_dirichlet(b, dofs, np.where(z[dofs] == 0.0, 350.0, 300.0), symmetric=True)
        x0 = gpu.DeviceVector(V.n_owned)
        st0 = gpu.krylov_solve(A, b, x0, rtol=1e-8, max_iter=50000)
    finally:
        gpu.set_option("row_dictionary", 1)
    # (same per-row arithmetic; the dot products are partitioned over a different number of workgroups at this size - the
    # streaming product is the two-launch pair kernel here -, so the recurrences agree to rounding, not to the bit as they
    # do where both forms use one launch geometry: tests/test_gpu_kernels.py::test_row_dictionary_product_is_the_streaming_product)
    assert st0["row_classes"] == 0 and abs(st0["iterations"] - it1) <= 1
    h0 = gpu.krylov_history()
    m = min(len(h0), len(h1), 200)
    assert np.allclose(h0[:m], h1[:m], rtol=1e-8)
    assert np.abs(x0.get() - x1).max() <= 1e-6


def test_config5_taylor_hood_cavity_2m_velocity_dofs(gpu):
    """configs[4]: lid-driven cavity, unit cube n=43, P2/P1: 1 975 509 velocity + 85 184 pressure dofs, 477 042 tets,
    nu=0.01, rho=1, dt=0.01, backward Euler, Newton per step.  ALL ten steps of the configuration; properties:
    Newton converges quadratically to DOLFIN's tolerances in every step (the residual includes the discrete continuity rows),
    boundary values are exact, the flow spins up monotonically (kinetic energy grows step by step, ever more slowly; the core
    moves with the lid)."""
    import copy
    import logging
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    n = 43
    mesh = UnitCubeMesh(n, n, n)
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 1.0)), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': 0.01}})
    s['solver_settings']['transient_settings'] = {'transient': True, 'starting_time': 0.0, 'time_step': 0.01,
                                                  'ending_time': 0.1 - 1e-9}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['report_settings'] = {"logging_level": logging.ERROR, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    W = solver.function_space
    assert mesh.num_cells() == 477042 and mesh.num_vertices() == 85184 and 3 * W.num_nodes() == 1975509
    energies = []
    solver.init_solver()
    solver.current_time, solver.current_step = 0.0, 0
    for step in range(10):
        solver.solve_current_step()
        h = solver.newton_history
        assert h[-1] <= max(1e-9 * h[0], 1e-10) and len(h) <= 5
        assert h[2] <= 1e-2 * h[1] if len(h) > 2 else True
        a = solver.w_current.vector().array().reshape(-1, 4)
        energies.append(float((a[:, :3] ** 2).sum()))
        solver.current_step += 1
        solver.current_time += 0.01
    co = W.node_coordinates()
    a = solver.w_current.vector().array().reshape(-1, 4)
    lid = co[:, 2] == 1.0
    wall = ((co[:, 0] == 0) | (co[:, 0] == 1) | (co[:, 1] == 0) | (co[:, 1] == 1) | (co[:, 2] == 0)) & ~lid
    assert np.all(a[lid, 0] == 1.0) and np.all(a[lid, 1:3] == 0.0) and np.all(a[wall, :3] == 0.0)
    assert all(e1 > e0 for e0, e1 in zip(energies, energies[1:]))
    growth = np.diff(energies)
    assert all(g1 < g0 for g0, g1 in zip(growth[1:], growth[2:]))       # the spin-up slows down towards the steady state
    near_lid = (co[:, 2] > 0.9) & (co[:, 2] < 1.0) & (np.abs(co[:, 0] - 0.5) < 0.2) & (np.abs(co[:, 1] - 0.5) < 0.2)
    assert a[near_lid, 0].mean() > 0.05            # fluid under the lid is dragged along +x
    assert np.abs(a[W.mesh().num_vertices():, 3]).max() == 0.0   # dummy pressure slots stay zero


# ---- the oracle itself at (scaled) BASELINE sizes: not only properties -----------------------------
def _device_csr(A):
    import scipy.sparse as sp
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def test_config2_full_size_against_the_c_oracle(gpu):
    """configs[1] at its FULL size against oracle/fem_oracle_c.c (the C restatement of DOLFIN's cell loop + KSPCG): sparsity
    bit-exact, 14.8 M matrix values <= 1e-12 relative, the same 293 iterations, solutions <= 1e-9 relative."""
    from oracle import c_oracle
    n = 99
    mesh, V, A, x, st, _ = _heat_cube(gpu, n)
    co, ce = c_oracle.box_mesh(n, n, n)
    rp, ci = c_oracle.csr_pattern(len(co), ce)
    A0 = gpu.DeviceMatrix(V)
    A0.assemble(stiffness=20.0)
    M = _device_csr(A0)
    assert np.array_equal(M.indptr, rp) and np.array_equal(M.indices, ci)            # bit-exact connectivity
    vals = c_oracle.assemble_p1(co, ce, 20.0, rp, ci)
    assert np.abs(M.data - vals).max() <= 1e-12 * np.abs(vals).max()
    ref = c_oracle.heat_box_solve(n, n, n, axis=2, rtol=1e-8)
    assert ref["iterations"] == st["iterations"] == 293
    assert np.abs(x.get() - ref["x"]).max() <= 1e-9 * np.abs(ref["x"]).max()


def test_config3_elasticity_against_the_oracle_at_90k_dof(gpu):
    """configs[2] scaled to what the numpy oracle assembles in seconds (118 x 15 x 15 cells of the same 10:1:1 bar, 91 392 DOF,
    159 300 tets): operator and load <= 1e-12 against the oracle; the AMG-PCG solution is held to the ORACLE's constrained
    system (||A_oracle x - b_oracle|| <= 2e-8 ||b||: sparse LU of a 3-D operator of this size takes minutes on the host, one
    product does not) and to beam theory."""
    from oracle import fem_oracle as fo
    nx, ny, nz = 118, 15, 15
    E, nu = 2e11, 0.27
    co, ce = fo.box_mesh((0, 0, 0), (10.0, 1.0, 1.0), nx, ny, nz)
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0, 0, 0), (10.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=fo.lame(E, nu))
    R = fo.assemble_p1_elasticity(co, ce, E, nu).tocsr()
    R.sort_indices()
    M = _device_csr(A)
    assert np.array_equal(M.indptr, R.indptr) and np.array_equal(M.indices, R.indices)
    assert np.abs(M.data - R.data).max() <= 1e-12 * np.abs(R.data).max()
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=(0.0, 0.0, -7800.0 * 10.0))
    bo = fo.assemble_p1_vector_source(co, ce, (0.0, 0.0, -7800.0 * 10.0))
    assert np.abs(b.get() - bo).max() <= 1e-12 * np.abs(bo).max()
    left = np.nonzero(co[:, 0] == 0.0)[0]
    dofs = (left[:, None] * 3 + np.arange(3)).ravel()
    A.apply_dirichlet(b, dofs, 0.0, symmetric=True)
    x = gpu.DeviceVector(V.n_owned)
    amg = gpu.AMG(A, nullspace="rigid_body")
    st = amg.solve(b, x, rtol=1e-12, max_iter=200)
    assert st["converged"] == 1 and st["iterations"] <= 60
    Ab, bb = fo.apply_dirichlet(R, bo, dofs, np.zeros(len(dofs)), True)
    xd = x.get()
    # (|A| |x| eps is 1e-9 |b| here: entries of 1e11 against displacements of 1e-3)
    assert np.linalg.norm(Ab @ xd - bb) <= 2e-8 * np.linalg.norm(bb)
    beam = -7800.0 * 10.0 * 10.0 ** 4 / (8 * E * (1.0 / 12.0))
    tip = xd.reshape(-1, 3)[co[:, 0] == 10.0, 2].mean()
    assert abs(tip - beam) <= 0.03 * abs(beam)


def test_config4_p2_against_the_oracle_at_118k_dof(gpu):
    """configs[3] scaled (unit cube n = 24, 117 649 DOF, 82 944 tets): CG2 numbering and sparsity bit-exact, values <= 1e-12
    against the oracle's 4-point-rule element matrices; the Jacobi-PCG solution against the oracle's own PCG at the same
    tolerance and against the oracle's constrained system."""
    from oracle import fem_oracle as fo
    n = 24
    co, ce = fo.unit_cube_mesh(n)
    mesh, V, A, x, st, z = _heat_cube(gpu, n, degree=2, rtol=1e-12)
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    assert np.array_equal(V.edges(), edges)
    nn = len(co) + len(edges)
    A0 = gpu.DeviceMatrix(V)
    A0.assemble(stiffness=20.0)
    M = _device_csr(A0)
    R = fo.assemble_generic(nn, cd, fo.p2_stiffness_local(co, ce, 20.0))
    assert np.array_equal(M.indptr, R.indptr) and np.array_equal(M.indices, R.indices)
    assert np.abs(M.data - R.data).max() <= 1e-12 * np.abs(R.data).max()
    b0, b1 = np.nonzero(z == 0.0)[0], np.nonzero(z == 1.0)[0]
    dofs = np.concatenate([b0, b1])
    vals = np.concatenate([np.full(len(b0), 350.0), np.full(len(b1), 300.0)])
    Ab, bb = fo.apply_dirichlet(R, np.zeros(nn), dofs, vals, True)
    xo, ito, _ = fo.pcg_jacobi_single_reduction(Ab, bb, rtol=1e-12)
    assert st["converged"] == 1 and ito - 2 <= st["iterations"] <= ito + 60       # (restarts from the true residual at 1e-12)
    xd = x.get()
    assert np.abs(xd - xo).max() <= 1e-9 * np.abs(xo).max()
    assert np.linalg.norm(Ab @ xd - bb) <= 2e-12 * np.linalg.norm(bb)
    assert np.abs(xd - (350.0 - 50.0 * z)).max() <= 1e-8


def test_config3_operator_at_full_size_against_the_c_oracle(gpu):
    """configs[2] at its FULL size (472 x 59 x 59 box, 5 108 400 DOF, 9.86 M tets) against the C oracle's restatement of DOLFIN's cell
    loop for inner(sigma(u), eps(v)) dx (oracle/fem_oracle_c.c orc_assemble_p1_elasticity, itself checked against the numpy oracle
    in the CPU suite): sparsity bit-exact, every stored value <= 1e-12 of the largest; the AMG-PCG solution of the solver path
    satisfies the ORACLE's constrained system."""
    from oracle import c_oracle
    nx, ny, nz = 472, 59, 59
    E, nu = 2e11, 0.27
    mu, lm = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    co, ce = c_oracle.box_mesh(nx, ny, nz, (0, 0, 0), (10.0, 1.0, 1.0))
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0, 0, 0), (10.0, 1.0, 1.0))
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=(mu, lm))
    M = _device_csr(A)
    cd12 = (ce.astype(np.int64)[:, :, None] * 3 + np.arange(3)).reshape(len(ce), 12).astype(np.int32)
    rp, ci = c_oracle.csr_pattern_generic(3 * len(co), cd12)
    assert M.shape[0] == 5108400 and np.array_equal(M.indptr, rp) and np.array_equal(M.indices, ci)
    vals = c_oracle.assemble_p1_elasticity(co, ce, mu, lm, rp, ci)
    assert np.abs(M.data - vals).max() <= 1e-12 * np.abs(vals).max()
    del M
    # the solve of the solver path against the oracle's operator: clamp x = 0, body force, AMG-PCG
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, vector_value=(0.0, 0.0, -7800.0 * 10.0))
    bo = b.get().copy()
    nodes = np.arange(len(co))
    left = nodes[nodes % (nx + 1) == 0]
    dofs = (left[:, None] * 3 + np.arange(3)).ravel()
    A.apply_dirichlet(b, dofs, 0.0, symmetric=True)
    x = gpu.DeviceVector(V.n_local)
    amg = gpu.AMG(A, nullspace="rigid_body")
    st = amg.solve(b, x, rtol=1e-10, max_iter=200)
    assert st["converged"] == 1
    import scipy.sparse as sp
    R = sp.csr_matrix((vals, ci, rp), shape=(3 * len(co), 3 * len(co)))
    xd = x.get()[:V.n_owned]
    r = R @ xd - bo                      # the unconstrained residual vanishes on the free rows (x = 0 on the clamped ones)
    r[dofs] = 0.0
    assert np.abs(xd[dofs]).max() == 0.0
    # (round-off floor of this residual: entries of 1e11 against displacements of 6e-3 - |A| |x| eps is 7e-8 per row, 0.118 being the
    # load of a row; measured 3.8e-8 of ||b||)
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(bo)
    amg.close()


def test_config4_operator_at_full_size_against_the_c_oracle(gpu):
    """configs[3] at its FULL size (unit cube n = 107, CG2, 9 938 375 DOF, 7.35 M tets) against the C oracle's CG2 cell loop (4-point rule,
    UFC edge order; orc_assemble_p2): node numbering and sparsity bit-exact, every stored value <= 1e-12 of the largest; the Jacobi-PCG
    solution of the timed path satisfies the oracle's constrained system to the solver's tolerance."""
    from oracle import c_oracle, fem_oracle as fo
    n = 107
    mesh, V, A0, x, st, z = _heat_cube(gpu, n, degree=2, rtol=1e-8)
    co, ce = c_oracle.box_mesh(n, n, n)
    # the oracle's own CG2 numbering (edges grouped by index difference, fem_oracle.p2_edge_order), from the connectivity alone
    cd, edges = fo.p2_cell_dofs(len(co), ce)
    assert np.array_equal(V.edges().astype(np.int64), edges.astype(np.int64))
    ndof = len(co) + len(edges)
    assert ndof == V.n_owned == (2 * n + 1) ** 3
    rp, ci = c_oracle.csr_pattern_generic(ndof, cd)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    M = _device_csr(A)
    assert np.array_equal(M.indptr, rp) and np.array_equal(M.indices, ci)
    vals = c_oracle.assemble_p2(co, ce, cd, 20.0, rp, ci)
    assert np.abs(M.data - vals).max() <= 1e-12 * np.abs(vals).max()
    del M
    import scipy.sparse as sp
    R = sp.csr_matrix((vals, ci, rp), shape=(ndof, ndof))
    T = x.get()
    r = R @ T                            # no load: the residual of the free rows is K T
    fixed = (z == 0.0) | (z == 1.0)
    r[fixed] = 0.0
    # ||b|| of the constrained system the solver stopped on: the load the eliminated columns put on the free rows + the Dirichlet rows
    bnorm = np.sqrt(np.linalg.norm(R[:, np.nonzero(fixed)[0]] @ T[fixed]) ** 2 + (T[fixed] ** 2).sum())
    assert st["converged"] == 1 and np.linalg.norm(r) <= 2e-8 * bnorm
    assert np.abs(T - (350.0 - 50.0 * z)).max() <= 2e-3
