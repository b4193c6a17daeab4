"""Host-side numerics of the solver classes (no GPU): the boundary / coefficient integrals the Python layer works out itself
before it hands per-node loads or per-cell coefficients to the device, each against the oracle or a closed form."""
import numpy as np
import pytest

from oracle import fem_oracle as fo, ns_oracle as ns


def test_radiation_loads_equal_the_closed_form_monomial_integrals():
    from fenicssolver_amd.SolverBase import _radiation_loads, _radiation_loads_p2
    rng = np.random.default_rng(0)
    co = rng.standard_normal((12, 3))
    tri = np.array([[0, 1, 2], [3, 4, 5], [2, 6, 9], [7, 10, 11]])
    T = 300 + 60 * rng.random(12)
    a = fo.radiation_facet_loads(co, tri, T, 5.1e-8, 280.0)
    b = _radiation_loads(co, tri, T, 5.1e-8, 280.0)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    co2 = rng.standard_normal((12, 2))
    ed = np.array([[0, 1], [3, 4], [2, 6]])
    a = fo.radiation_facet_loads(co2, ed, T, 5.1e-8, 280.0)
    b = _radiation_loads(co2, ed, T, 5.1e-8, 280.0)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    # P2 facets: a P2 field that happens to be linear carries the same total radiated power as its P1 description
    tab = np.array([[0, 1, 2, 3, 4, 5]])
    Tl = np.concatenate([T[:3], [0.5 * (T[0] + T[1]), 0.5 * (T[0] + T[2]), 0.5 * (T[1] + T[2])]])
    p2 = _radiation_loads_p2(co[:6], np.array([[0, 1, 2]]), tab, Tl, 5.1e-8, 280.0)
    p1 = _radiation_loads(co[:6], np.array([[0, 1, 2]]), Tl, 5.1e-8, 280.0)
    assert abs(p2.sum() - p1.sum()) <= 1e-13 * abs(p1.sum())
    # ... and a genuinely quadratic field is integrated like an independent, finer rule does
    Tq = 300 + 60 * rng.random(6)
    g, w = np.polynomial.legendre.leggauss(10)
    g, w = 0.5 * (g + 1), 0.5 * w
    X = co[:3]
    area = 0.5 * np.linalg.norm(np.cross(X[1] - X[0], X[2] - X[0]))
    ref = np.zeros(6)
    for u, wu in zip(g, w):
        for v, wv in zip(g, w):
            lam = np.array([1 - u, u * (1 - v), u * v])
            phi = np.concatenate([lam * (2 * lam - 1), [4 * lam[0] * lam[1], 4 * lam[0] * lam[2], 4 * lam[1] * lam[2]]])
            ref += 2 * wu * wv * u * area * 5.1e-8 * (280.0 ** 4 - (Tq @ phi) ** 4) * phi
    got = _radiation_loads_p2(co[:6], np.array([[0, 1, 2]]), tab, Tq, 5.1e-8, 280.0)[0]
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_nodal_facet_loads_are_the_p1_mass_matrix_times_the_values():
    from fenicssolver_amd.SolverBase import _facet_nodal_loads
    rng = np.random.default_rng(1)
    co = rng.standard_normal((8, 3))
    tri = np.array([[0, 1, 2], [3, 5, 7]])
    gv = rng.standard_normal((2, 3))
    got = _facet_nodal_loads(co, tri, gv)
    area = fo.facet_areas(co, tri)
    M = (np.ones((3, 3)) + np.eye(3)) / 12.0
    assert np.abs(got - area[:, None] * (gv @ M)).max() <= 1e-15
    co2 = rng.standard_normal((8, 2))
    ed = np.array([[0, 1], [4, 6]])
    g2 = rng.standard_normal((2, 2))
    L = np.linalg.norm(co2[ed[:, 1]] - co2[ed[:, 0]], axis=1)
    assert np.abs(_facet_nodal_loads(co2, ed, g2) - L[:, None] * (g2 @ ((np.ones((2, 2)) + np.eye(2)) / 6.0))).max() <= 1e-15


def test_velocity_weights_of_the_solver_equal_the_oracle():
    from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver as S, _p2_shape_at, _TET14_POINTS
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 1.2), 2, 2, 1)
    th = ns.TaylorHood(co, ce)
    rng = np.random.default_rng(2)
    U2 = rng.standard_normal((th.n_nodes, 3))
    mine = 4.0 * np.einsum("na,cni->cai", S._P2_P1_MASS, U2[th.cell_nodes])
    assert np.abs(mine - fo.row_velocities(ce, U2, cell_dofs=th.cell_nodes)).max() <= 1e-14
    # the P2 shape table of the solver at the 14 points = the oracle's
    pts, _ = ns.tet_quadrature(5)
    assert np.abs(np.sort(pts, axis=0) - np.sort(_TET14_POINTS, axis=0)).max() <= 1e-15
    for lam, row in zip(_TET14_POINTS, _p2_shape_at(_TET14_POINTS)):
        assert np.abs(row - ns.p2_shape(lam)[0]).max() <= 1e-15


def test_p2_facet_weights_of_the_pressure_load():
    """int phi_node lambda_b ds / |F| used by LinearElasticitySolver._varying_pressure_load, against quadrature."""
    from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver as E
    W = np.zeros((6, 3))
    for bary, w in zip(ns.TRI_QP, ns.TRI_QW):
        phi = np.concatenate([bary * (2 * bary - 1), [4 * bary[0] * bary[1], 4 * bary[0] * bary[2], 4 * bary[1] * bary[2]]])
        W += w * np.outer(phi, bary)
    assert np.abs(W - E._W_TRI_P2).max() <= 1e-14
    g, w = np.polynomial.legendre.leggauss(4)
    g, w = 0.5 * (g + 1), 0.5 * w
    W2 = np.zeros((3, 2))
    for x, wx in zip(g, w):
        lam = np.array([1 - x, x])
        W2 += wx * np.outer(np.array([lam[0] * (2 * lam[0] - 1), lam[1] * (2 * lam[1] - 1), 4 * lam[0] * lam[1]]), lam)
    assert np.abs(W2 - E._W_SEG_P2).max() <= 1e-14


def test_facet_node_table_orders_vertices_then_edges():
    from fenicssolver_amd.fem import UnitCubeMesh, UnitSquareMesh, FunctionSpace
    m = UnitCubeMesh(2, 2, 2)
    V = FunctionSpace(m, "CG", 2)
    tri = m.facets()[m.exterior_facets()].astype(np.int64)
    tab = V.facet_node_table(tri)
    X = V.node_coordinates()
    assert tab.shape == (len(tri), 6) and np.array_equal(tab[:, :3], tri)
    for k, (i, j) in enumerate(((0, 1), (0, 2), (1, 2))):
        assert np.abs(X[tab[:, 3 + k]] - 0.5 * (X[tri[:, i]] + X[tri[:, j]])).max() <= 1e-15
    m2 = UnitSquareMesh(3, 2)
    V2 = FunctionSpace(m2, "CG", 2)
    ed = m2.facets()[m2.exterior_facets()].astype(np.int64)
    t2 = V2.facet_node_table(ed)
    X2 = V2.node_coordinates()
    assert t2.shape == (len(ed), 3) and np.abs(X2[t2[:, 2]] - 0.5 * (X2[ed[:, 0]] + X2[ed[:, 1]])).max() <= 1e-15
