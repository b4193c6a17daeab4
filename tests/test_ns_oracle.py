"""Pins of the Taylor-Hood oracle (oracle/ns_oracle.py) - CPU only.

DOLFIN/FFC are not installable, so the restatement of CoupledNavierStokesSolver's forms is pinned by known
answers: the quadrature rule integrates every monomial of its degree exactly, Poiseuille flow (which lies in
the P2/P1 space and has a vanishing convective term) is a root of the discrete residual and Newton finds it,
and the Jacobian is the derivative of the residual (central differences)."""
import math
from itertools import product

import numpy as np

from oracle import fem_oracle as fo, ns_oracle as ns


def test_quadrature_rules_are_exact_to_their_degree():
    for deg, npts in ((2, 4), (5, 14)):
        pts, w = ns.tet_quadrature(deg)
        assert len(w) == npts and abs(w.sum() - 1.0) < 1e-15
        for a, b, c, d in product(range(deg + 1), repeat=4):
            if a + b + c + d > deg:
                continue
            exact = math.factorial(a) * math.factorial(b) * math.factorial(c) * math.factorial(d) * 6 / \
                math.factorial(a + b + c + d + 3)
            num = (w * pts[:, 0] ** a * pts[:, 1] ** b * pts[:, 2] ** c * pts[:, 3] ** d).sum()
            assert abs(num - exact) < 1e-15


def test_p2_basis_is_nodal_and_sums_to_one():
    nodes = [np.eye(4)[i] for i in range(4)] + [0.5 * (np.eye(4)[i] + np.eye(4)[j]) for i, j in fo.P2_EDGE_VERTS]
    for k, lam in enumerate(nodes):
        phi, dphi = ns.p2_shape(lam)
        assert np.allclose(phi, np.eye(10)[k], atol=1e-15)
    phi, dphi = ns.p2_shape(np.array([0.1, 0.2, 0.3, 0.4]))
    assert abs(phi.sum() - 1.0) < 1e-15 and np.allclose(dphi.sum(axis=0), 1.0 * np.ones(4) * 0 + dphi.sum(axis=0)[0])


def _poiseuille(n=3, nu=0.3, rho=2.0):
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    th = ns.TaylorHood(co, ce)
    X = th.node_coords
    exact = np.zeros((th.n_nodes, 4))
    exact[:, 0] = X[:, 2] * (1 - X[:, 2])
    exact[:th.nv, 3] = -2 * nu * rho * X[:th.nv, 0] + 5.0
    bn = th.boundary_nodes(lambda x: True)
    bc_dofs = np.concatenate([th.velocity_dofs(bn), th.pressure_dofs([0])])
    return th, exact.ravel(), bc_dofs, nu, rho


def test_poiseuille_flow_is_reproduced_exactly():
    th, exact, bc_dofs, nu, rho = _poiseuille()
    free = np.ones(th.n, dtype=bool)
    free[bc_dofs] = False
    free[th.dummy_dofs()] = False
    for inv_dt in (0.0, 10.0):
        r = ns.residual(th, exact, nu, rho, inv_dt=inv_dt, w_prev=exact)
        assert np.abs(r[free]).max() < 1e-14
    w, hist = ns.newton_solve(th, np.zeros(th.n), bc_dofs, exact[bc_dofs], nu, rho)
    assert len(hist) <= 5 and np.abs(w - exact).max() < 1e-10
    # quadratic convergence of Newton
    assert hist[2] < 1e-3 * hist[1]


def test_jacobian_is_the_derivative_of_the_residual():
    th, exact, bc_dofs, nu, rho = _poiseuille(2)
    rng = np.random.default_rng(0)
    w0 = 0.1 * rng.standard_normal(th.n)
    dw = rng.standard_normal(th.n)
    w0[th.dummy_dofs()] = 0
    dw[th.dummy_dofs()] = 0
    args = (nu, rho, 3.0, 0.5 * w0, (0, 0, -9.8))
    J, g = ns.ns_system(th, w0, *args)
    assert np.abs((J @ w0 - g) - ns.residual(th, w0, *args)).max() < 1e-13
    eps = 1e-6
    fd = (ns.residual(th, w0 + eps * dw, *args) - ns.residual(th, w0 - eps * dw, *args)) / (2 * eps)
    assert np.abs(fd - J @ dw).max() <= 1e-8 * np.abs(J @ dw).max()
    # Picard matrix = Jacobian without the (grad(u0) du) term: differs, but shares the linear part
    K, _ = ns.ns_system(th, w0, *args, newton=False)
    assert abs(K - J).max() > 0
    K0, _ = ns.ns_system(th, np.zeros(th.n), *args, newton=False)
    J0, _ = ns.ns_system(th, np.zeros(th.n), *args, newton=True)
    assert abs(K0 - J0).max() == 0


def test_row_velocity_advection_is_the_exact_integral_for_p1_and_p2_fields():
    """fo.row_velocities / p1_advection_local with [nc,4,3] velocities: C_ab = int (u . grad phi_b) phi_a dx integrated EXACTLY for
    a P1 or a P2 velocity field (what FFC's quadrature does for ScalarTransportSolver.py:305-311 and for the temperature
    equation of the coupled flow solver, whose convective velocity is the P2 iterate) - pinned with the degree-5 rule."""
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 1.2), 2, 2, 2)
    rng = np.random.default_rng(12)
    th = ns.TaylorHood(co, ce)
    pts, wq = ns.tet_quadrature(5)
    detJ, g = fo.p1_geometry(co, ce)
    vol = np.abs(detJ) / 6.0
    # P2 field
    U2 = rng.standard_normal((th.n_nodes, 3))
    want = np.zeros((len(ce), 4, 4))
    for lam, w in zip(pts, wq):
        phi, _ = ns.p2_shape(lam)
        u = np.einsum("n,cni->ci", phi, U2[th.cell_nodes])
        want += (w * vol)[:, None, None] * np.einsum("a,cb->cab", lam, np.einsum("ci,cbi->cb", u, g))
    got = fo.p1_advection_local(co, ce, fo.row_velocities(ce, U2, cell_dofs=th.cell_nodes), 1.0)
    assert np.abs(got - want).max() <= 1e-14 * np.abs(want).max()
    # P1 field
    U1 = rng.standard_normal((len(co), 3))
    want = np.zeros((len(ce), 4, 4))
    for lam, w in zip(pts, wq):
        u = np.einsum("v,cvi->ci", lam, U1[ce.astype(np.int64)])
        want += (w * vol)[:, None, None] * np.einsum("a,cb->cab", lam, np.einsum("ci,cbi->cb", u, g))
    got = fo.p1_advection_local(co, ce, fo.row_velocities(ce, U1), 1.0)
    assert np.abs(got - want).max() <= 1e-14 * np.abs(want).max()
    # a constant field gives back the cell-wise formula
    const = np.tile([0.3, -0.2, 0.5], (len(co), 1))
    assert np.abs(fo.p1_advection_local(co, ce, fo.row_velocities(ce, const), 2.0) - fo.p1_advection_local(co, ce, (0.3, -0.2, 0.5), 2.0)).max() <= 1e-15
    # triangles, P1 field, 3-point edge-midpoint rule (exact for quadratics)
    c2, t2 = fo.rectangle_mesh((0, 0), (1.0, 0.8), 3, 2) if hasattr(fo, "rectangle_mesh") else (None, None)
    if c2 is not None:
        U = rng.standard_normal((len(c2), 2))
        area, g2 = fo.tri_geometry(c2, t2)
        want = np.zeros((len(t2), 3, 3))
        for lam in ((0.5, 0.5, 0.0), (0.0, 0.5, 0.5), (0.5, 0.0, 0.5)):
            lam = np.asarray(lam)
            u = np.einsum("v,cvi->ci", lam, U[t2.astype(np.int64)])
            want += (area / 3.0)[:, None, None] * np.einsum("a,cb->cab", lam, np.einsum("ci,cbi->cb", u, g2))
        got = fo.tri_advection_local(c2, t2, fo.row_velocities(t2, U), 1.0)
        assert np.abs(got - want).max() <= 1e-14 * np.abs(want).max()
