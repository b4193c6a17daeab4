"""Taylor-Hood Navier-Stokes path (fs_assemble_navier_stokes / fs_saddle_solve; CoupledNavierStokesSolver.py:288-381,
215-245, 492-528) against the CPU oracle (oracle/ns_oracle.py) through the C-ABI.

Bars: matrix / right-hand side <= 1e-11 relative (atomic summation order, FMA); solutions of the linear systems
<= 1e-6 relative at a Krylov tolerance of 1e-10; Poiseuille flow (in the discrete space) reproduced to 1e-8."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import fem_oracle as fo, ns_oracle as ns

pytestmark = pytest.mark.gpu


def _setup(gpu, n=3, p1=(1.0, 1.0, 1.0)):
    co, ce = fo.box_mesh((0, 0, 0), p1, n, n, n)
    th = ns.TaylorHood(co, ce)
    mesh = gpu.DeviceMesh(co, ce)
    W = gpu.DeviceSpace(mesh, ncomp=4, degree=2)
    Q = gpu.DeviceSpace(mesh, ncomp=1, degree=1)
    assert W.n_owned == th.n
    # same edge-node numbering on both sides
    assert np.array_equal(W.edges().astype(np.int64), th.edges.astype(np.int64))
    return co, ce, th, mesh, W, Q


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


@pytest.mark.parametrize("newton,inv_dt", [(True, 7.0), (False, 0.0), (True, 0.0)])
def test_linearised_system_matches_oracle(gpu, newton, inv_dt):
    co, ce, th, mesh, W, Q = _setup(gpu, 3, (1.0, 0.8, 1.3))
    rng = np.random.default_rng(1)
    w0 = 0.3 * rng.standard_normal(th.n)
    wp = 0.3 * rng.standard_normal(th.n)
    w0[th.dummy_dofs()] = 0.0
    nu, rho, f = 0.07, 1.7, (0.1, -0.2, -9.8)
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    gpu.assemble_navier_stokes(J, g, gpu.DeviceVector(W.n_local, w0), gpu.DeviceVector(W.n_local, wp), nu=nu, rho=rho,
                               inv_dt=inv_dt, body_force=f, convection=True, newton=newton)
    Jr, gr = ns.ns_system(th, w0, nu, rho, inv_dt, wp, f, newton=newton)
    Jd = _csr(J)
    scale = abs(Jr).max()
    assert abs(Jd - Jr).max() <= 1e-11 * scale
    assert np.abs(g.get() - gr).max() <= 1e-11 * np.abs(gr).max()
    # Stokes (no convection) is the same operator with the convection terms off
    gpu.assemble_navier_stokes(J, g, None, None, nu=nu, rho=rho, inv_dt=0.0, body_force=f, convection=False, newton=False)
    Js, gs = ns.ns_system(th, np.zeros(th.n), nu, rho, 0.0, None, f, newton=False, convection=False)
    assert abs(_csr(J) - Js).max() <= 1e-11 * abs(Js).max()
    assert np.abs(g.get() - gs).max() <= 1e-11 * np.abs(gs).max()


def _pressure_operators(gpu, Q, pinned):
    Kp = gpu.DeviceMatrix(Q)
    Kp.assemble(stiffness=1.0)
    Kp.apply_dirichlet(None, np.asarray(pinned, dtype=np.int32), np.zeros(len(pinned)), symmetric=True)
    Mp = gpu.DeviceMatrix(Q)
    Mp.assemble(mass=1.0)
    return Kp, Mp


@pytest.mark.parametrize("inv_dt", [0.0, 100.0])
def test_saddle_solve_lid_driven_cavity_step(gpu, inv_dt):
    """One linearised step of the lid-driven cavity (BASELINE configs[4] set-up at a small size)."""
    co, ce, th, mesh, W, Q = _setup(gpu, 4)
    nu, rho = 0.01 if inv_dt else 0.1, 1.0
    X = th.node_coords
    bn = th.boundary_nodes(lambda x: True)
    lid = bn[X[bn, 2] == 1.0]
    vals = np.zeros((th.n_nodes, 4))
    vals[lid, 0] = 1.0
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])])
    bc_vals = vals.ravel()[bc_dofs]
    w0 = np.zeros(th.n)
    w0[bc_dofs] = bc_vals
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    dw0 = gpu.DeviceVector(W.n_local, w0)
    gpu.assemble_navier_stokes(J, g, dw0, gpu.DeviceVector(W.n_local, np.zeros(th.n)), nu=nu, rho=rho, inv_dt=inv_dt)
    J.apply_dirichlet(g, bc_dofs.astype(np.int32), bc_vals, symmetric=False)
    Jr, gr = ns.ns_system(th, w0, nu, rho, inv_dt, np.zeros(th.n))
    Jb, gb = ns.apply_dirichlet_rows(Jr, gr.copy(), bc_dofs, bc_vals)
    assert abs(_csr(J) - Jb).max() <= 1e-11 * abs(Jb).max()
    ref = spl.spsolve(Jb.tocsc(), gb)
    Kp, Mp = _pressure_operators(gpu, Q, [0])
    x = gpu.DeviceVector(W.n_local)
    st = gpu.saddle_solve(J, Kp if inv_dt else None, Mp, g, x, nu=nu, rho=rho, inv_dt=inv_dt, rtol=1e-10)
    assert st["converged"] == 1 and st["iterations"] <= (80 if inv_dt else 400), st   # steady: Jacobi is a weak A^-1
    sol = x.get()
    u_err = np.abs(sol.reshape(-1, 4)[:, :3] - ref.reshape(-1, 4)[:, :3]).max()
    p_err = np.abs(sol.reshape(-1, 4)[:th.nv, 3] - ref.reshape(-1, 4)[:th.nv, 3]).max()
    assert u_err <= 1e-6 * np.abs(ref.reshape(-1, 4)[:, :3]).max()
    assert p_err <= 1e-5 * np.abs(ref.reshape(-1, 4)[:th.nv, 3]).max()
    assert np.linalg.norm(Jb @ sol - gb) <= 2e-10 * np.linalg.norm(gb)


def test_newton_reproduces_poiseuille_flow(gpu):
    """u = (z(1-z), 0, 0), p = -2 nu rho x + c lies in the Taylor-Hood space and has (u.grad)u = 0: Newton on the
    device path must land on it (the oracle does, to 1e-12)."""
    co, ce, th, mesh, W, Q = _setup(gpu, 3)
    nu, rho = 0.3, 2.0
    X = th.node_coords
    exact = np.zeros((th.n_nodes, 4))
    exact[:, 0] = X[:, 2] * (1 - X[:, 2])
    exact[:th.nv, 3] = -2 * nu * rho * X[:th.nv, 0] + 5.0
    bn = th.boundary_nodes(lambda x: True)
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])]).astype(np.int32)
    bc_vals = exact.ravel()[bc_dofs]
    Kp, Mp = _pressure_operators(gpu, Q, [0])
    w = np.zeros(th.n)
    w[bc_dofs] = bc_vals
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    hist = []
    for it in range(8):
        dw = gpu.DeviceVector(W.n_local, w)
        gpu.assemble_navier_stokes(J, g, dw, None, nu=nu, rho=rho)
        r = gpu.DeviceVector(W.n_owned)
        J.spmv(dw, r)
        res = r.get() - g.get()
        res[bc_dofs] = 0.0
        hist.append(np.linalg.norm(res))
        if hist[-1] <= 1e-9 * hist[0]:
            break
        J.apply_dirichlet(g, bc_dofs, bc_vals, symmetric=False)
        x = gpu.DeviceVector(W.n_local, w)
        st = gpu.saddle_solve(J, None, Mp, g, x, nu=nu, rho=rho, rtol=1e-11, nonzero_guess=True)
        assert st["converged"] == 1
        w = x.get()
    assert len(hist) <= 6 and hist[-1] <= 1e-9 * hist[0], hist
    W4 = w.reshape(-1, 4)
    assert np.abs(W4[:, :3] - exact[:, :3]).max() <= 1e-8
    assert np.abs(W4[:th.nv, 3] - exact[:th.nv, 3]).max() <= 2e-6      # Krylov tolerance, pressure scale 5


# ---- the drop-in solver class -----------------------------------------------------------------------------
QUIET = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}


def _cavity_settings(n, transient, nu=0.01, lid=(1.0, 0.0, 0.0), t_end=0.02, body_source=None):
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    mesh = UnitCubeMesh(n, n, n)
    walls = AutoSubDomain(lambda x, on_boundary: on_boundary)      # the lid re-marks its facets afterwards
    top = AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 1.0))
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': walls, 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': top, 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant(lid)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s['solver_name'] = "CoupledNavierStokesSolver"
    s['mesh'] = mesh
    s['fe_degree'] = 1
    s['boundary_conditions'] = bcs
    s['body_source'] = body_source
    s['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 0}
    s['material'] = {'density': 1.0, 'kinematic_viscosity': nu}
    s['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01,
                                                  'ending_time': t_end}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['report_settings'] = dict(QUIET)
    return s, mesh


def _oracle_cavity(mesh, nu, lid, steps, dt, newton=True, body_force=None):
    co, ce = mesh.coordinates(), mesh.cells()
    th = ns.TaylorHood(co, ce)
    X = th.node_coords
    bn = th.boundary_nodes(lambda x: True)
    vals = np.zeros((th.n_nodes, 4))
    top = bn[X[bn, 2] == 1.0]
    # walls are marked first, the lid second: nodes shared by both take the lid value ("later wins")
    vals[top, :3] = lid
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])])
    bc_vals = vals.ravel()[bc_dofs]
    w = np.zeros(th.n)
    hist = None
    for k in range(max(steps, 1)):
        w, hist = ns.newton_solve(th, w, bc_dofs, bc_vals, nu, 1.0, (1.0 / dt) if steps else 0.0, w.copy(), body_force)
    return th, w, hist


def test_cavity_solver_class_steady_newton_matches_oracle(gpu):
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(4, transient=False, nu=0.1)
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve().vector().array()
    th, ref, hist = _oracle_cavity(mesh, 0.1, (1.0, 0.0, 0.0), 0, 0.01)
    assert len(solver.newton_history) <= len(hist) + 1 and solver.newton_history[-1] <= 1e-9 * solver.newton_history[0]
    W4, R4 = w.reshape(-1, 4), ref.reshape(-1, 4)
    assert np.abs(W4[:, :3] - R4[:, :3]).max() <= 1e-6
    assert np.abs(W4[:th.nv, 3] - R4[:th.nv, 3]).max() <= 1e-4 * max(np.abs(R4[:th.nv, 3]).max(), 1.0)
    u, p = solver.split()
    assert u.vector().size() == 3 * th.n_nodes and p.vector().size() == th.nv
    # discrete incompressibility: the continuity rows of the residual vanish, and so does the net boundary flux
    assert abs(W4[:, 2].sum()) < 1.0


def test_cavity_solver_class_transient_two_steps(gpu, tmp_path):
    """BASELINE configs[4] set-up (lid-driven cavity, nu = 0.01, rho = 1, dt = 0.01, backward Euler, Newton) at n = 4."""
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(4, transient=True, nu=0.01, t_end=0.02)
    s['report_settings'] = dict(QUIET, saving_freq=1, result_filename=str(tmp_path / "up.pvd"))
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve().vector().array()
    steps = solver.current_step
    th, ref, hist = _oracle_cavity(mesh, 0.01, (1.0, 0.0, 0.0), steps, 0.01)
    W4, R4 = w.reshape(-1, 4), ref.reshape(-1, 4)
    assert np.abs(W4[:, :3] - R4[:, :3]).max() <= 2e-6
    assert np.abs(W4[:th.nv, 3] - R4[:th.nv, 3]).max() <= 1e-4 * np.abs(R4[:th.nv, 3]).max()
    assert (tmp_path / "up.pvd").exists()


def test_picard_loop_agrees_with_newton(gpu):
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(3, transient=False, nu=0.2)
    a = CoupledNavierStokesSolver(s)
    wa = a.solve().vector().get_local()
    s2, _ = _cavity_settings(3, transient=False, nu=0.2)
    b = CoupledNavierStokesSolver(s2)
    b.using_nonlinear_solver = False
    wb = b.solve().vector().get_local()
    assert b.picard_iterations < 50
    assert np.abs(wa - wb).reshape(-1, 4)[:, :3].max() <= 5e-4      # Picard stops at |dw|_inf <= 1e-4


def test_unsupported_navier_stokes_settings_raise(gpu):
    from fenicssolver_amd.fem import SolverError, Constant
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(2, transient=False)
    s['boundary_conditions']['lid']['values'] = [{'variable': 'velocity', 'type': 'symmetry', 'value': None}]
    with pytest.raises(SolverError):
        CoupledNavierStokesSolver(s).solve()
    s, mesh = _cavity_settings(2, transient=False)
    s['advection_settings'] = {'stabilization_method': 'SUPG', 'Pe': 10}
    with pytest.raises(SolverError):
        CoupledNavierStokesSolver(s).solve()
    s, mesh = _cavity_settings(2, transient=False)
    s['advection_settings'] = {'stabilization_method': 'G2', 'kappa1': 4}          # Re missing
    with pytest.raises(SolverError):
        CoupledNavierStokesSolver(s).solve()


def test_pressure_boundary_terms_match_oracle_and_reproduce_poiseuille(gpu):
    """Pressure inlet / outlet (CoupledNavierStokesSolver.py:449-453): p n.v ds - nu ((grad u + grad u^T) n).v ds.
    With rho = 1 Poiseuille flow is a root of the form with these terms (checked on the oracle to 1e-13)."""
    co, ce, th, mesh, W, Q = _setup(gpu, 3)
    nu, rho = 0.3, 1.0
    fin = ns.boundary_facet_cells(th, lambda x: abs(x[0]) < 1e-12)
    fout = ns.boundary_facet_cells(th, lambda x: abs(x[0] - 1) < 1e-12)
    rng = np.random.default_rng(3)
    w0 = 0.2 * rng.standard_normal(th.n)
    w0[th.dummy_dofs()] = 0
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    gpu.assemble_navier_stokes(J, g, gpu.DeviceVector(W.n_local, w0), None, nu=nu, rho=rho)
    base, gbase = _csr(J), g.get()
    gpu.assemble_ns_pressure_boundary(J, g, fin[:, 0], fin[:, 1], nu, 5.0)
    gpu.assemble_ns_pressure_boundary(J, g, fout[:, 0], fout[:, 1], nu, None)       # farfield type: no load
    dJ1, dg1 = ns.pressure_boundary_terms(th, fin, nu, 5.0)
    dJ2, dg2 = ns.pressure_boundary_terms(th, fout, nu, None)
    assert abs((_csr(J) - base) - (dJ1 + dJ2)).max() <= 1e-12 * abs(base).max()
    assert np.abs((g.get() - gbase) - (dg1 + dg2)).max() <= 1e-12 * np.abs(gbase).max()
    assert np.abs(dg2).max() == 0.0 and np.abs(dg1).max() > 0.0

    # Newton with pressure Dirichlet at x = 0 and x = 1, exact velocity on the lateral walls
    X = th.node_coords
    exact = np.zeros((th.n_nodes, 4))
    exact[:, 0] = X[:, 2] * (1 - X[:, 2])
    exact[:th.nv, 3] = -2 * nu * rho * X[:th.nv, 0] + 5.0
    lateral = th.boundary_nodes(lambda x: min(abs(x[1]), abs(x[1] - 1), abs(x[2]), abs(x[2] - 1)) < 1e-12)
    vin, vout = np.nonzero(co[:, 0] == 0)[0], np.nonzero(co[:, 0] == 1)[0]
    bc_dofs = np.concatenate([th.velocity_dofs(lateral), th.pressure_dofs(vin), th.pressure_dofs(vout)]).astype(np.int32)
    bc_vals = exact.ravel()[bc_dofs]
    Kp, Mp = _pressure_operators(gpu, Q, np.concatenate([vin, vout]))
    w = np.zeros(th.n)
    w[bc_dofs] = bc_vals
    hist = []
    for it in range(8):
        dw = gpu.DeviceVector(W.n_local, w)
        gpu.assemble_navier_stokes(J, g, dw, None, nu=nu, rho=rho)
        gpu.assemble_ns_pressure_boundary(J, g, fin[:, 0], fin[:, 1], nu, 5.0)
        gpu.assemble_ns_pressure_boundary(J, g, fout[:, 0], fout[:, 1], nu, 5.0 - 2 * nu * rho)
        r = gpu.DeviceVector(W.n_owned)
        J.spmv(dw, r)
        res = r.get() - g.get()
        res[bc_dofs] = 0.0
        hist.append(np.linalg.norm(res))
        if hist[-1] <= 1e-9 * hist[0]:
            break
        J.apply_dirichlet(g, bc_dofs, bc_vals, symmetric=False)
        x = gpu.DeviceVector(W.n_local, w)
        st = gpu.saddle_solve(J, None, Mp, g, x, nu=nu, rho=rho, rtol=1e-11, nonzero_guess=True, velocity_sweeps=3)
        assert st["converged"] == 1
        w = x.get()
    W4 = w.reshape(-1, 4)
    assert hist[-1] <= 1e-9 * hist[0], hist
    assert np.abs(W4[:, :3] - exact[:, :3]).max() <= 1e-7
    assert np.abs(W4[:th.nv, 3] - exact[:th.nv, 3]).max() <= 1e-5


def test_channel_flow_with_pressure_inlet_and_outlet_through_the_solver_class(gpu):
    """Pressure-driven channel: no-slip on z = 0, 1, exact profile on the y faces, pressure Dirichlet at x = 0, 1
    (rho = 1, where the reference's boundary integrals are consistent): the solver class lands on Poiseuille flow."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    nu, dp = 0.3, 2 * 0.3
    mesh = UnitCubeMesh(3, 3, 3)
    prof = Expression(("x[2]*(1-x[2])", "0", "0"), degree=2)
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and (near(x[2], 0) or near(x[2], 1) or near(x[1], 0) or near(x[1], 1))),
                    'boundary_id': 1, 'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': prof}]}
    bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 0)), 'boundary_id': 2,
                    'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(dp)}]}
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 1)), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(0.0)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': nu}})
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['report_settings'] = dict(QUIET)
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve()
    u, p = solver.split()
    X = solver.function_space.node_coordinates()
    assert np.abs(u.node_values()[:, 0] - X[:, 2] * (1 - X[:, 2])).max() <= 1e-6
    assert np.abs(u.node_values()[:, 1:]).max() <= 1e-6
    co = mesh.coordinates()
    assert np.abs(p.vector().array() - dp * (1 - co[:, 0])).max() <= 1e-5


def test_ale_mesh_velocity_shifts_the_advecting_velocity_only(gpu):
    """reference_frame_settings {'type': 'ALE', 'mesh_velocity': w} (CoupledNavierStokesSolver.py:321-329): J gets
    (grad(.) (u0 - w)).v, the Newton terms keep u0; checked for Newton and Picard against the oracle."""
    co, ce, th, mesh, W, Q = _setup(gpu, 3, (1.0, 0.8, 1.3))
    rng = np.random.default_rng(2)
    w0 = 0.3 * rng.standard_normal(th.n)
    w0[th.dummy_dofs()] = 0.0
    nu, rho, wm = 0.05, 1.3, (0.4, -0.7, 0.25)
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    for newton in (True, False):
        gpu.assemble_navier_stokes(J, g, gpu.DeviceVector(W.n_local, w0), None, nu=nu, rho=rho, convection=True, newton=newton,
                                   mesh_velocity=wm)
        Jr, gr = ns.ns_system(th, w0, nu, rho, 0.0, None, None, newton=newton, mesh_velocity=wm)
        assert abs(_csr(J) - Jr).max() <= 1e-11 * abs(Jr).max()
        assert np.abs(g.get() - gr).max() <= 1e-11 * max(np.abs(gr).max(), 1.0)
        J0, _ = ns.ns_system(th, w0, nu, rho, 0.0, None, None, newton=newton)
        assert abs(Jr - J0).max() > 1e-3 * abs(Jr).max()            # the frame velocity does change the operator
    # J w0 - g is the residual of the ALE form: K w0 + (grad(u0) (u0 - w)).v
    gpu.assemble_navier_stokes(J, g, gpu.DeviceVector(W.n_local, w0), None, nu=nu, rho=rho, convection=True, newton=True, mesh_velocity=wm)
    r = _csr(J) @ w0 - g.get()
    rr = ns.residual(th, w0, nu, rho, mesh_velocity=wm)
    assert np.abs(r - rr).max() <= 1e-10 * np.abs(rr).max()


def _channel_solver(ale=None):
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import BoxMesh, Point, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    nu = 0.3
    mesh = BoxMesh(Point(0, 0, 0), Point(1, 1, 2), 3, 3, 6)
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
    if ale is not None:
        s['reference_frame_settings'] = {'type': 'ALE', 'mesh_velocity': ale}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-10}
    s['report_settings'] = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    return CoupledNavierStokesSolver(s), nu


def test_viscous_stress_projection_and_drag_lift(gpu):
    """viscous_stress = project(nu (grad u + grad u^T) - p I, CG1 tensors) (:149-155) against the oracle's sparse-LU
    projection; calc_drag_and_lift (:172-192, with the undefined self.ds read as the boundary markers) against the
    oracle's facet sums and against the analytic wall shear of the plane channel flow."""
    solver, nu = _channel_solver()
    w = solver.solve()
    co, ce = solver.mesh.coordinates(), solver.mesh.cells()
    th = ns.TaylorHood(co, ce)
    assert np.array_equal(solver.function_space.edge_nodes().astype(np.int64), th.edges.astype(np.int64))
    sig = solver.viscous_stress(w)
    ref = ns.viscous_stress_projection(th, w.vector().array(), nu)
    got = sig.node_values().reshape(-1, 3, 3)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-8 * np.abs(ref).max()
    assert np.abs(got - np.swapaxes(got, 1, 2)).max() == 0.0
    # the flow is u_z = x(1-x), p = 4 nu (1 - z/2): sigma_zx = nu (1 - 2x), sigma_ii = -p  (P1-representable: exact)
    X = co
    assert np.abs(got[:, 2, 0] - nu * (1 - 2 * X[:, 0])).max() <= 1e-6
    assert np.abs(got[:, 0, 0] + 4 * nu * (1 - X[:, 2] / 2)).max() <= 1e-6
    # forces on the walls (marker 1): drag along z, lift along x
    drag, lift = solver.calc_drag_and_lift(w, 2, 0, [1])
    F = ns.boundary_force(th, ref, lambda x: min(abs(x[0]), abs(x[0] - 1), abs(x[1]), abs(x[1] - 1)) < 1e-12)
    assert abs(drag - F[2]) <= 1e-9 * abs(F[2]) and abs(lift - F[0]) <= 1e-9 * max(abs(F[0]), abs(F[2]))
    # analytic: -int sigma n ds on x = 0 and x = 1 walls: -(sigma_zx * (-1))|x=0 * 2 - (sigma_zx * 1)|x=1 * 2 = 2 nu + 2 nu
    assert abs(drag - 4 * nu) <= 1e-6 and abs(lift) <= 1e-6
    with pytest.raises(Exception):
        solver.calc_drag_and_lift(w, 2, 0, [])
    # traction on the boundary vertices: sigma . n; on the wall x = 0 (n = -e_x) its z-component is -nu
    t = solver.boundary_traction(w).node_values()
    wall0 = np.nonzero((X[:, 0] == 0) & (X[:, 1] > 0) & (X[:, 1] < 1) & (X[:, 2] > 0) & (X[:, 2] < 2))[0]
    interior = np.nonzero((X[:, 0] > 0) & (X[:, 0] < 1) & (X[:, 1] > 0) & (X[:, 1] < 1) & (X[:, 2] > 0) & (X[:, 2] < 2))[0]
    assert np.abs(t[wall0, 2] + nu).max() <= 1e-6 and np.abs(t[interior]).max() == 0.0


def test_ale_frame_through_the_solver_api(gpu):
    """A frame moving along the channel axis changes nothing for a z-independent flow ((grad u) w = 0 for w = (0,0,c)),
    a frame moving across it does; both go through reference_frame_settings."""
    base, _ = _channel_solver()
    u0 = base.solve().vector().array().copy()
    along, _ = _channel_solver(ale=(0.0, 0.0, 0.8))
    u1 = along.solve().vector().array()
    assert np.abs(u1 - u0).max() <= 1e-7 * np.abs(u0).max()
    F, _ = along.generate_form(0, None, None, along.w_current, along.w_prev)
    assert F.describe()["mesh_velocity"] == [0.0, 0.0, 0.8]
    across, _ = _channel_solver(ale=(0.5, 0.0, 0.0))
    u2 = across.solve().vector().array()
    assert np.abs(u2 - u0).max() >= 1e-4 * np.abs(u0).max()
    th = ns.TaylorHood(base.mesh.coordinates(), base.mesh.cells())
    r = ns.residual(th, u2, 0.3, 1.0, mesh_velocity=(0.5, 0.0, 0.0))
    free = np.ones(th.n, dtype=bool)
    X = th.node_coords
    bn = th.boundary_nodes(lambda x: min(abs(x[0]), abs(x[0] - 1), abs(x[1]), abs(x[1] - 1)) < 1e-12)
    free[th.velocity_dofs(bn)] = False
    free[3::4] = True
    # momentum residual on the free velocity dofs of the interior (pressure-boundary integrals live on inlet / outlet nodes)
    inner = free.copy()
    io = th.boundary_nodes(lambda x: abs(x[2]) < 1e-12 or abs(x[2] - 2) < 1e-12)
    inner[th.velocity_dofs(io)] = False
    inner[3::4] = False
    assert np.abs(r[inner]).max() <= 1e-7 * max(np.abs(r).max(), 1e-3)


@pytest.mark.parametrize("mode,inv_dt", [(1, 0.0), (2, 0.0), (2, 25.0)])
def test_g2_streamline_term_matches_oracle(gpu, mode, inv_dt):
    """advection_settings {'stabilization_method': 'G2'} (CoupledNavierStokesSolver.py:334-363):
    F -= delta1 (a.grad u).(a.grad v) dx, delta1 = kappa1 h^2 (Re <= 1) or from |a|, h = 2 circumradius (and dt); with an ALE
    frame the advecting velocity a = u0 - w.  Same 14-point rule on both sides (the integrand is not a polynomial)."""
    co, ce, th, mesh, W, Q = _setup(gpu, 3, (1.0, 0.8, 1.3))
    rng = np.random.default_rng(4)
    co2 = co.copy()
    inner = np.all((co > 0) & (co < np.array([1.0, 0.8, 1.3])), axis=1)
    co2[inner] += 0.03 * rng.standard_normal((int(inner.sum()), 3))        # cells of different size and shape
    th = ns.TaylorHood(co2, ce)
    mesh = gpu.DeviceMesh(co2, ce)
    W = gpu.DeviceSpace(mesh, ncomp=4, degree=2)
    w0 = 0.4 * rng.standard_normal(th.n)
    w0[th.dummy_dofs()] = 0.0
    w0.reshape(-1, 4)[:5, :3] = 0.0                                         # a = 0 somewhere: delta1 must not blow up
    wp = 0.3 * rng.standard_normal(th.n)
    nu, rho, f, kappa1, wm = 0.05, 1.3, (0.0, 0.1, -1.0), 0.7, (0.2, -0.1, 0.05)
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    for newton in (True, False):
        gpu.assemble_navier_stokes(J, g, gpu.DeviceVector(W.n_local, w0), gpu.DeviceVector(W.n_local, wp), nu=nu, rho=rho,
                                   inv_dt=inv_dt, body_force=f, convection=True, newton=newton, mesh_velocity=wm, g2=(mode, kappa1))
        Jr, gr = ns.ns_system(th, w0, nu, rho, inv_dt, wp, f, newton=newton, mesh_velocity=wm, g2=(mode, kappa1))
        J0, g0 = ns.ns_system(th, w0, nu, rho, inv_dt, wp, f, newton=newton, mesh_velocity=wm)
        Jd = _csr(J)
        assert np.all(np.isfinite(Jd.data))
        assert abs(Jd - Jr).max() <= 1e-11 * abs(Jr).max()
        assert np.abs(g.get() - gr).max() <= 1e-11 * np.abs(gr).max() and np.array_equal(gr, g0)
        assert abs(Jr - J0).max() > 1e-4 * abs(Jr).max()                    # the term is there


def test_g2_through_the_solver_class(gpu):
    """The cavity with G2 settings: same Newton fixed point as the oracle's iteration with the term."""
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(3, transient=False, nu=0.05)
    s['advection_settings'] = {'stabilization_method': 'G2', 'Re': 20, 'kappa1': 0.5, 'kappa2': 0.5}
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve().vector().array()
    co, ce = mesh.coordinates(), mesh.cells()
    th = ns.TaylorHood(co, ce)
    bn = th.boundary_nodes(lambda x: True)
    vals = np.zeros((th.n_nodes, 4))
    vals[bn[th.node_coords[bn, 2] == 1.0], :3] = (1.0, 0.0, 0.0)
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])])
    ref, hist = ns.newton_solve(th, np.zeros(th.n), bc_dofs, vals.ravel()[bc_dofs], 0.05, 1.0, 0.0, None, None, g2=(2, 0.5))
    plain, _ = ns.newton_solve(th, np.zeros(th.n), bc_dofs, vals.ravel()[bc_dofs], 0.05, 1.0, 0.0, None, None)
    W4, R4 = w.reshape(-1, 4), ref.reshape(-1, 4)
    assert np.abs(W4[:, :3] - R4[:, :3]).max() <= 1e-6
    assert np.abs(R4[:, :3] - plain.reshape(-1, 4)[:, :3]).max() > 1e-4       # and it changes the flow


@pytest.mark.parametrize("newton,inv_dt", [(True, 0.0), (False, 5.0)])
def test_pressure_dependent_viscosity_matches_oracle(gpu, newton, inv_dt):
    """material 'Newtonian': False (CoupledNavierStokesSolver.viscosity :194-213, the branch without a temperature):
    nu(p) = nu0 (p / p_ref)^0.1 with the pressure of the current iterate, in the cell terms, the pressure-boundary terms
    and the stress projection.  Both element kernels, device vs oracle."""
    import os
    co, ce, th, mesh, W, Q = _setup(gpu, 3, (1.0, 0.8, 1.3))
    rng = np.random.default_rng(4)
    w0 = 0.3 * rng.standard_normal(th.n)
    w0.reshape(-1, 4)[:th.nv, 3] = 1.0e5 * (1.0 + 0.3 * rng.uniform(-1, 1, th.nv))       # absolute pressures around p_ref
    w0[th.dummy_dofs()] = 0.0
    wp = 0.3 * rng.standard_normal(th.n)
    nu, rho, f, law = 0.07, 1.7, (0.1, -0.2, -9.8), (1.0e5, 0.1)
    Jr, gr = ns.ns_system(th, w0, nu, rho, inv_dt, wp, f, newton=newton, viscosity_law=law)
    Jn, _ = ns.ns_system(th, w0, nu, rho, inv_dt, wp, f, newton=newton)
    assert abs(Jr - Jn).max() > 1e-3 * abs(Jn).max()          # the law matters at these pressures
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    dw = gpu.DeviceVector(W.n_local, w0)
    for mode in (None, "pair"):
        if mode:
            os.environ["FS_NS_ASSEMBLE"] = mode
        try:
            gpu.assemble_navier_stokes(J, g, dw, gpu.DeviceVector(W.n_local, wp), nu=nu, rho=rho, inv_dt=inv_dt,
                                       body_force=f, convection=True, newton=newton, viscosity_law=law)
        finally:
            os.environ.pop("FS_NS_ASSEMBLE", None)
        assert abs(_csr(J) - Jr).max() <= 1e-11 * abs(Jr).max()
        assert np.abs(g.get() - gr).max() <= 1e-11 * np.abs(gr).max()
        if mode is None and os.environ.get("FS_NS_ASSEMBLE"):
            break
    # pressure boundaries
    fin = ns.boundary_facet_cells(th, lambda x: abs(x[0]) < 1e-12)
    base, gbase = _csr(J), g.get()
    gpu.assemble_ns_pressure_boundary(J, g, fin[:, 0], fin[:, 1], nu, 5.0, viscosity_law=law, w0=dw)
    dJ, dg = ns.pressure_boundary_terms(th, fin, nu, 5.0, viscosity_law=law, w0=w0)
    dJn, _ = ns.pressure_boundary_terms(th, fin, nu, 5.0)
    assert abs(dJ - dJn).max() > 1e-3 * abs(dJn).max()
    assert abs((_csr(J) - base) - dJ).max() <= 1e-12 * abs(base).max()
    assert np.abs((g.get() - gbase) - dg).max() <= 1e-12 * np.abs(gbase).max()
    # stress projection: right-hand sides against the oracle's
    b9 = gpu.DeviceVector(9 * Q.n_owned)
    gpu.assemble_viscous_stress(W, dw, nu, Q, b9, viscosity_law=law)
    sig = ns.viscous_stress_projection(th, w0, nu, viscosity_law=law)
    M = fo.assemble_matrix(th.nv, th.cells, fo.p1_mass_local(th.coords, th.cells, 1.0))
    want = M @ sig.reshape(th.nv, 9)
    assert np.abs(b9.get().reshape(th.nv, 9) - want).max() <= 1e-10 * np.abs(want).max()
    with pytest.raises(gpu.BackendError):
        gpu.assemble_navier_stokes(J, g, None, None, nu=nu, rho=rho, convection=False, newton=False, viscosity_law=law)


def test_non_newtonian_channel_through_the_solver_class(gpu):
    """The solver class with material['Newtonian'] = False: pressure-driven channel at absolute pressures around p_ref;
    the converged iterate is a root of the ORACLE's residual with the same law, and differs from the Newtonian one."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, Expression, near, SolverError
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    nu, pref = 0.3, 10.0

    def run(newtonian, ref_pressure=pref):
        mesh = UnitCubeMesh(3, 3, 3)
        bcs = OrderedDict()
        bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and (near(x[2], 0) or near(x[2], 1) or near(x[1], 0) or near(x[1], 1))),
                        'boundary_id': 1, 'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
        bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 0)), 'boundary_id': 2,
                        'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(14.0)}]}
        bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 1)), 'boundary_id': 3,
                         'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(8.0)}]}
        s = copy.deepcopy(SB.default_case_settings)
        s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
                  'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 10.0},
                  'material': {'density': 1.0, 'kinematic_viscosity': nu, 'Newtonian': newtonian}})
        s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': ref_pressure}
        s['report_settings'] = dict(QUIET)
        solver = CoupledNavierStokesSolver(s)
        w = solver.solve()
        return solver, mesh, w.vector().array().copy()

    solver, mesh, w_nn = run(False)
    _, _, w_n = run(True)
    assert solver.viscosity_law() == (pref, 0.1) and solver.viscosity() == nu
    U_nn, U_n = w_nn.reshape(-1, 4)[:, 0], w_n.reshape(-1, 4)[:, 0]
    assert U_n.max() > 0.05 and np.abs(U_nn - U_n).max() > 2e-3 * U_n.max()
    # root of the oracle's residual (cell terms + pressure-boundary terms) on the free dofs
    th = ns.TaylorHood(mesh.coordinates(), mesh.cells())
    law = (pref, 0.1)
    K, rhs = ns.ns_system(th, w_nn, nu, 1.0, 0.0, None, None, newton=False, viscosity_law=law)
    for inside, val in ((lambda x: abs(x[0]) < 1e-12, 14.0), (lambda x: abs(x[0] - 1) < 1e-12, 8.0)):
        dJ, dg = ns.pressure_boundary_terms(th, ns.boundary_facet_cells(th, inside), nu, val, viscosity_law=law, w0=w_nn)
        K, rhs = K + dJ, rhs + dg
    r = K @ w_nn - rhs
    X = th.node_coords
    walls = th.boundary_nodes(lambda x: min(abs(x[1]), abs(x[1] - 1), abs(x[2]), abs(x[2] - 1)) < 1e-12)
    co = mesh.coordinates()
    fixed = np.concatenate([th.velocity_dofs(walls), th.pressure_dofs(np.nonzero(co[:, 0] == 0)[0]), th.pressure_dofs(np.nonzero(co[:, 0] == 1)[0]),
                            th.dummy_dofs()])
    r[fixed] = 0.0
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(rhs)
    with pytest.raises(SolverError):
        run(False, ref_pressure=0)


def test_time_dependent_boundary_values_and_form_parts(gpu):
    """translate_value's time-dependent forms on a velocity boundary (SolverBase.py:365-366 a sequence with one entry per time
    step, :376-377 a callable of the time), as examples/test_cfd_solver.py:127-129 passes them; and the reference's
    F_static / F_transient entry points (:288-381)."""
    from fenicssolver_amd.fem import Constant
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver

    def run(lid_value):
        s, mesh = _cavity_settings(3, transient=True, nu=0.05, t_end=0.03)
        s['boundary_conditions']['lid']['values'][0]['value'] = lid_value
        solver = CoupledNavierStokesSolver(s)
        return solver, mesh, solver.solve().vector().array().copy()

    ramp = [0.25, 0.5, 0.75, 1.0, 1.0]
    _, mesh, w_list = run([Constant((a, 0, 0)) for a in ramp])
    # get_current_time() of step k is starting_time + dt (k - 1), as in the reference (SolverBase.py:453-465)
    solver, _, w_call = run(lambda t: Constant((ramp[int(round(t / 0.01)) + 1], 0, 0)))
    assert np.abs(w_list - w_call).max() <= 1e-9          # (the right-hand side is summed with atomics: not bit-reproducible)
    _, _, w_const = run(Constant((1.0, 0, 0)))
    assert np.abs(w_list - w_const).reshape(-1, 4)[:, :3].max() > 1e-2           # the ramp is a different problem
    # oracle: the same ramp, step by step
    th = ns.TaylorHood(mesh.coordinates(), mesh.cells())
    X = th.node_coords
    bn = th.boundary_nodes(lambda x: True)
    top = bn[X[bn, 2] == 1.0]
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])])
    w = np.zeros(th.n)
    for k in range(solver.current_step):
        vals = np.zeros((th.n_nodes, 4))
        vals[top, 0] = ramp[k]
        w, _ = ns.newton_solve(th, w, bc_dofs, vals.ravel()[bc_dofs], 0.05, 1.0, 100.0, w.copy(), None)
    assert np.abs(w_list - w).reshape(-1, 4)[:, :3].max() <= 2e-6
    # form parts
    Fs = solver.F_static(None, None, solver.w_current)
    Ft = solver.F_transient(1, None, None, solver.w_current, solver.w_prev)
    assert Fs.inv_dt == 0.0 and Ft.inv_dt == pytest.approx(100.0) and Ft.w_prev is solver.w_prev
    assert Fs.describe()["nu"] == 0.05 and solver.transient_settings['transient'] is True


def _thermal_cavity(n, transient, t_end=0.02):
    from fenicssolver_amd.fem import Constant
    s, mesh = _cavity_settings(n, transient=transient, nu=0.05, t_end=t_end)
    s['solving_temperature'] = True
    s['material'] = {'density': 2.0, 'kinematic_viscosity': 0.05, 'specific_heat_capacity': 3.0, 'thermal_conductivity': 0.1}
    s['boundary_conditions']['walls']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(350.0)})
    s['boundary_conditions']['lid']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300.0)})
    s['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 0, 'temperature': 320.0}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0, 'temperature': 300.0}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-12}
    return s, mesh


def _thermal_operators(mesh, u_nodes, th):
    co, ce = mesh.coordinates(), mesh.cells()
    cap = 2.0 * 3.0
    V = fo.row_velocities(ce, u_nodes, cell_dofs=th.cell_nodes)            # the P2 velocity, integrated exactly
    K = fo.assemble_p1_scalar(co, ce, 0.1)
    C = fo.assemble_matrix(len(co), ce, fo.p1_advection_local(co, ce, V, cap))
    P = fo.assemble_interior_penalty(co, ce, 0.1 * cap)                     # IP, alpha = 0.1 (CoupledNavierStokesSolver.py:262)
    M = fo.assemble_matrix(len(co), ce, fo.p1_mass_local(co, ce, cap))
    top = np.nonzero(co[:, 2] == 1.0)[0]
    bnd = np.nonzero(np.any((co == 0.0) | (co == 1.0), axis=1))[0]
    vals = np.full(len(co), 350.0)
    vals[top] = 300.0                                                       # the lid is marked after the walls
    return K, C, P, M, bnd, vals[bnd]


def test_coupled_temperature_steady(gpu):
    """solving_temperature (CoupledNavierStokesSolver.py:236-239, 247-286): u, p, T = split(solver.solve()); the flow is the one
    of the same case without the temperature, T solves the IP-stabilised transport equation convected by that P2 velocity."""
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    from fenicssolver_amd.mixed import split
    s, mesh = _thermal_cavity(3, transient=False)
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve()
    u, p, T = split(w)
    assert T.vector().size() == mesh.num_vertices() and solver.temperature() is T
    s0, _ = _thermal_cavity(3, transient=False)
    s0['solving_temperature'] = False
    w0 = CoupledNavierStokesSolver(s0).solve()
    assert np.abs(w.vector().array() - w0.vector().array()).max() <= 1e-12            # one-way coupling
    th = ns.TaylorHood(mesh.coordinates(), mesh.cells())
    K, C, P, M, bnd, bvals = _thermal_operators(mesh, u.node_values(), th)
    A, b = fo.apply_dirichlet((K + C + P).tocsr(), np.zeros(th.nv), bnd, bvals, True)
    want = fo.solve_direct(A, b)
    Tv = T.vector().array()
    assert np.abs(Tv - want).max() <= 1e-6 * np.abs(want).max()
    assert Tv.min() >= 299.0 and Tv.max() <= 352.0 and np.ptp(Tv) > 10.0       # (no discrete maximum principle on 3^3 cells)


def test_coupled_temperature_transient_step_satisfies_the_crank_nicolson_equation(gpu, tmp_path):
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _thermal_cavity(3, transient=True, t_end=0.03)
    s['report_settings'] = dict(QUIET, saving_freq=1, result_filename=str(tmp_path / "upT.pvd"))
    solver = CoupledNavierStokesSolver(s)
    u, p, T = solver.split(solver.solve())
    Ts = solver._Tsolver
    Tn, Tp = T.vector().array(), Ts.w_prev.vector().array()
    th = ns.TaylorHood(mesh.coordinates(), mesh.cells())
    K, C, P, M, bnd, bvals = _thermal_operators(mesh, u.node_values(), th)
    dt = 0.01
    # (1/dt) M (T - T_prev) + 1/2 K T + 1/2 K T_prev + C T + P T = 0 on the free rows (ScalarTransportSolver.py:292-315)
    r = M @ (Tn - Tp) / dt + 0.5 * (K @ Tn) + 0.5 * (K @ Tp) + C @ Tn + P @ Tn
    r[bnd] = 0.0
    scale = np.abs(M @ Tn / dt).max()
    assert np.abs(r).max() <= 1e-8 * scale
    assert np.abs(Tn[bnd] - bvals).max() <= 1e-10 and np.abs(Tn - Tp).max() > 1e-3
    assert "temperature" in open(str(tmp_path / "upT000000.vtu")).read()


def test_boundary_pressure_that_varies_over_the_facets(gpu):
    """A pressure boundary value given as an Expression (hydrostatic outlet): the load inner(p_b n, v) ds with p_b through its P1
    interpolant on every facet (values at the facet's vertices), kernel against the oracle and through the solver class."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, Expression, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    co, ce, th, mesh, W, Q = _setup(gpu, 3)
    nu = 0.3
    fout = ns.boundary_facet_cells(th, lambda x: abs(x[0] - 1) < 1e-12)
    pfun = lambda x: 5.0 - 9.8 * x[2] + 0.5 * x[1]                              # noqa: E731
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    gpu.assemble_navier_stokes(J, g, None, None, nu=nu, rho=1.0, convection=False, newton=False)
    gbase = g.get()
    cells = ce.astype(np.int64)
    fv = np.array([[pfun(co[cells[c, v]]) for v in range(4) if v != o] for c, o in fout])
    gpu.assemble_ns_pressure_boundary(J, g, fout[:, 0], fout[:, 1], nu, fv)
    dJ, dg = ns.pressure_boundary_terms(th, fout, nu, pfun)
    assert np.abs((g.get() - gbase) - dg).max() <= 1e-12 * np.abs(dg).max()
    _, dg_c = ns.pressure_boundary_terms(th, fout, nu, 5.0 - 4.9 + 0.25)
    assert np.abs(dg - dg_c).max() > 1e-2 * np.abs(dg).max()                     # not the load of the mean pressure

    # solver class: fluid at rest under gravity, closed box except a hydrostatic outlet: u = 0, p = p0 - g z exactly
    m = UnitCubeMesh(3, 3, 3)
    bcs = OrderedDict()
    # every boundary facet is a wall first; the outlet re-marks its own facets afterwards (a facet is marked when ALL its
    # vertices are inside, so "on_boundary and not near(x[0], 1)" would leave the wall facets along the outlet rim unmarked)
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 1.0)), 'boundary_id': 2,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Expression("5.0 - 9.8*x[2]", degree=1)}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': m, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': Constant((0.0, 0.0, -9.8)), 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': nu}})
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-11}
    s['report_settings'] = dict(QUIET)
    solver = CoupledNavierStokesSolver(s)
    solver.solve()
    u, p = solver.split()
    assert np.abs(u.node_values()).max() <= 1e-8
    assert np.abs(p.vector().array() - (5.0 - 9.8 * m.coordinates()[:, 2])).max() <= 1e-6


def test_viscosity_depending_on_pressure_and_temperature(gpu):
    """material['Newtonian'] = False WITH solving_temperature (CoupledNavierStokesSolver.viscosity :199-203, round 4):
    nu (1 + 0.1 p/p_ref)(1 - 0.2 T/T_ref) on the current (u, p, T).  Kernels: cell terms, pressure-boundary traction and the stress
    projection with the law attached to the space, against the oracle.  Solver class: the converged (u, p, T) is a root of the ORACLE's
    flow residual with that law AND of the transport equation convected by that velocity - the reference's monolithic system."""
    import copy
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, AutoSubDomain, Constant, near, SolverError
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    from fenicssolver_amd.mixed import split
    co, ce, th, mesh, W, Q = _setup(gpu, 3)
    nu, rho, pref, tref = 0.3, 1.3, 10.0, 300.0
    rng = np.random.default_rng(5)
    Tv = 300.0 + 80.0 * rng.random(th.nv)
    w0 = 0.1 * rng.standard_normal(th.n)
    w0.reshape(-1, 4)[:th.nv, 3] = 10.0 + 4.0 * rng.random(th.nv)
    w0[th.dummy_dofs()] = 0.0
    law = ('pT', pref, 0.1, tref, 0.2, Tv)
    dT = gpu.DeviceVector(th.n_nodes, np.concatenate([Tv, np.zeros(th.n_nodes - th.nv)]))      # per node, read at the vertex nodes
    gpu.set_viscosity_law(W, law[:5], dT)
    J = gpu.DeviceMatrix(W)
    g = gpu.DeviceVector(W.n_owned)
    dw = gpu.DeviceVector(W.n_local, w0)
    gpu.assemble_navier_stokes(J, g, dw, None, nu=nu, rho=rho, convection=True, newton=True)
    fin = ns.boundary_facet_cells(th, lambda x: abs(x[0]) < 1e-12)
    gpu.assemble_ns_pressure_boundary(J, g, fin[:, 0] if hasattr(fin, 'shape') else np.array(fin)[:, 0], np.array(fin)[:, 1], nu,
                                      np.full(len(fin), 14.0), w0=dw)
    Kref, gref = ns.ns_system(th, w0, nu, rho, 0.0, None, None, newton=True, viscosity_law=law)
    dJ, dg = ns.pressure_boundary_terms(th, fin, nu, 14.0, viscosity_law=law, w0=w0)
    Kref, gref = (Kref + dJ).tocsr(), gref + dg
    rp, ci, va, shape = J.to_csr()
    import scipy.sparse as sps
    got = sps.csr_matrix((va, ci, rp), shape=shape)
    assert abs(got - Kref).max() <= 1e-11 * abs(Kref).max()
    assert np.abs(g.get() - gref).max() <= 1e-11 * np.abs(gref).max()
    Knewt, _ = ns.ns_system(th, w0, nu, rho, 0.0, None, None, newton=True)
    assert abs(Kref - Knewt).max() > 1e-2 * abs(Knewt).max()                 # the law is not a no-op
    bt = gpu.DeviceVector(9 * Q.n_owned)
    gpu.assemble_viscous_stress(W, dw, nu, Q, bt)
    sig = ns.viscous_stress_projection(th, w0, nu, viscosity_law=law)
    Mq = fo.assemble_matrix(th.nv, ce, fo.p1_mass_local(co, ce, 1.0))
    want = np.stack([Mq @ sig.reshape(th.nv, 9)[:, k] for k in range(9)], axis=1)
    assert np.abs(bt.get().reshape(th.nv, 9) - want).max() <= 1e-10 * np.abs(want).max()
    gpu.set_viscosity_law(W, None)
    gpu.assemble_navier_stokes(J, g, dw, None, nu=nu, rho=rho, convection=True, newton=True)
    rp, ci, va, shape = J.to_csr()
    assert abs(sps.csr_matrix((va, ci, rp), shape=shape) - Knewt).max() <= 1e-11 * abs(Knewt).max()      # detached again

    # ---- the solver class: heated channel driven by a pressure drop
    def run(newtonian):
        m = UnitCubeMesh(3, 3, 3)
        bcs = OrderedDict()
        bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and (near(x[2], 0) or near(x[2], 1) or near(x[1], 0) or near(x[1], 1))),
                        'boundary_id': 1, 'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))},
                                                     {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(420.0)}]}
        bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 0)), 'boundary_id': 2,
                        'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(14.0)},
                                   {'variable': "temperature", 'type': 'Dirichlet', 'value': Constant(300.0)}]}
        bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[0], 1)), 'boundary_id': 3,
                         'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(8.0)}]}
        s = copy.deepcopy(SB.default_case_settings)
        s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': m, 'fe_degree': 1, 'boundary_conditions': bcs, 'solving_temperature': True,
                  'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 10.0, 'temperature': 300.0},
                  'material': {'density': 1.0, 'kinematic_viscosity': nu, 'Newtonian': newtonian, 'specific_heat_capacity': 3.0,
                               'thermal_conductivity': 0.1}})
        s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': pref, 'temperature': tref}
        s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-12}
        s['report_settings'] = dict(QUIET)
        solver = CoupledNavierStokesSolver(s)
        return solver, m, solver.solve()

    solver, m, w = run(False)
    u, p, T = split(w)
    assert solver.temperature_law() == ('pT', pref, 0.1, tref, 0.2) and 2 <= solver.coupling_iterations <= 40
    wv, Tn = w.vector().array().copy(), T.vector().array().copy()
    _, _, w_n = run(True)
    assert np.abs(wv - w_n.vector().array()).reshape(-1, 4)[:, 0].max() > 2e-3 * np.abs(w_n.vector().array()).reshape(-1, 4)[:, 0].max()
    th2 = ns.TaylorHood(m.coordinates(), m.cells())
    law2 = ('pT', pref, 0.1, tref, 0.2, Tn)
    K, rhs = ns.ns_system(th2, wv, nu, 1.0, 0.0, None, None, newton=False, viscosity_law=law2)
    for inside, val in ((lambda x: abs(x[0]) < 1e-12, 14.0), (lambda x: abs(x[0] - 1) < 1e-12, 8.0)):
        dJ, dg = ns.pressure_boundary_terms(th2, ns.boundary_facet_cells(th2, inside), nu, val, viscosity_law=law2, w0=wv)
        K, rhs = K + dJ, rhs + dg
    r = K @ wv - rhs
    walls = th2.boundary_nodes(lambda x: min(abs(x[1]), abs(x[1] - 1), abs(x[2]), abs(x[2] - 1)) < 1e-12)
    mc = m.coordinates()
    fixed = np.concatenate([th2.velocity_dofs(walls), th2.pressure_dofs(np.nonzero(mc[:, 0] == 0)[0]), th2.pressure_dofs(np.nonzero(mc[:, 0] == 1)[0]),
                            th2.dummy_dofs()])
    r[fixed] = 0.0
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(rhs)
    # the temperature solves the IP-stabilised transport equation convected by THIS velocity
    cap = 1.0 * 3.0
    V = fo.row_velocities(m.cells(), u.node_values(), cell_dofs=th2.cell_nodes)
    A = (fo.assemble_p1_scalar(mc, m.cells(), 0.1) + fo.assemble_matrix(len(mc), m.cells(), fo.p1_advection_local(mc, m.cells(), V, cap))
         + fo.assemble_interior_penalty(mc, m.cells(), 0.1 * cap)).tocsr()
    wall_v = np.nonzero((mc[:, 1] == 0) | (mc[:, 1] == 1) | (mc[:, 2] == 0) | (mc[:, 2] == 1))[0]
    inlet_v = np.nonzero(mc[:, 0] == 0)[0]
    vals = np.full(len(mc), np.nan)
    vals[wall_v] = 420.0
    vals[inlet_v] = 300.0                                                   # the inlet is marked after the walls
    bnd = np.nonzero(~np.isnan(vals))[0]
    want = fo.solve_direct(*fo.apply_dirichlet(A, np.zeros(len(mc)), bnd, vals[bnd], False))
    assert np.abs(Tn - want).max() <= 1e-6 * np.abs(want).max()
    assert np.ptp(Tn) > 50.0


def test_cavity_first_time_step_against_the_oracle_newton_at_n8(gpu):
    """BASELINE configs[4] parameters (lid-driven cavity, nu = 0.01, rho = 1, dt = 0.01, backward Euler, Newton) at the largest size the
    numpy oracle's Newton loop (sparse LU of the 19 652-unknown saddle system per iteration) finishes in about half a minute: n = 8,
    one time step through the solver class against the oracle's own Newton iteration from the same start."""
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    s, mesh = _cavity_settings(8, transient=True, nu=0.01, t_end=0.01 - 1e-9)
    s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-11}
    solver = CoupledNavierStokesSolver(s)
    w = solver.solve().vector().array()
    assert solver.current_step == 1
    th, ref, hist = _oracle_cavity(mesh, 0.01, (1.0, 0.0, 0.0), 1, 0.01)
    assert th.n == 19652 and len(solver.newton_history) <= len(hist) + 1
    W4, R4 = w.reshape(-1, 4), ref.reshape(-1, 4)
    assert np.abs(W4[:, :3] - R4[:, :3]).max() <= 1e-6
    assert np.abs(W4[:th.nv, 3] - R4[:th.nv, 3]).max() <= 1e-4 * max(np.abs(R4[:th.nv, 3]).max(), 1.0)
    r = ns.residual(th, w, 0.01, 1.0, 100.0, np.zeros(th.n))
    bn = th.boundary_nodes(lambda x: True)
    r[np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0]), th.dummy_dofs()])] = 0.0
    assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(ns.ns_system(th, w, 0.01, 1.0, 100.0, np.zeros(th.n), None, newton=False)[0] @ w)
