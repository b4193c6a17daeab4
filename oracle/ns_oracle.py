"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product) for the Taylor-Hood path of
CoupledNavierStokesSolver: P2 velocity / P1 pressure on tetrahedra.

Parity unpinned against DOLFIN/FFC/PETSc themselves (not installable here, see fem_oracle.py); the
restatement is pinned by known answers: quadrature exactness, exact reproduction of Poiseuille flow
(quadratic velocity + linear pressure lie in the Taylor-Hood space), a finite-difference check of the
Jacobian against the residual, and the form goldens recorded from the reference's Python layer.

Forms restated (FenicsSolver/CoupledNavierStokesSolver.py):
  F_static    :288-365  F = 2 nu eps(u):eps(v) - (p/rho) div v + div u (q/rho) - f.v + (grad(u) a).v
  F_transient :367-381  F += (1/dt) (u - u_prev).v                      (backward Euler)
  generate_form :215-245  F = action(F, w_current);  J = derivative(F, w_current)   (Newton)
  solve_form  :492-528  Newton (NonlinearVariationalSolver) or Picard with under-relaxation 0.7

Numbering used here and on the device: one block of 4 dofs per P2 node (vertices first, then edge
mid-points in fem_oracle.p2_cell_dofs order): (u_x, u_y, u_z, p).  The pressure lives on vertex nodes only;
the pressure slot of an edge node is a dummy unknown with an identity row.
"""
import numpy as np
import scipy.sparse as sp

from . import fem_oracle as fo

# ---- quadrature on the reference tetrahedron (barycentric points, weights sum to 1) -------------
def tet_quadrature(degree):
    """Degree 2: 4 points; degree 5: the 14-point rule (Walkington / Grundmann-Moeller family)."""
    if degree <= 2:
        a, b = 0.5854101966249685, 0.1381966011250105
        pts = np.full((4, 4), b)
        np.fill_diagonal(pts, a)
        return pts, np.full(4, 0.25)
    pts, w = [], []
    for a, wt in ((0.3108859192633006, 0.1126879257180159), (0.0927352503108912, 0.0734930431163620)):
        for k in range(4):
            p = [a, a, a, a]
            p[k] = 1.0 - 3.0 * a
            pts.append(p)
            w.append(wt)
    b, wt = 0.0455037041256496, 0.0425460207770815
    c = 0.5 - b
    for p in ((b, b, c, c), (b, c, b, c), (b, c, c, b), (c, b, b, c), (c, b, c, b), (c, c, b, b)):
        pts.append(list(p))
        w.append(wt)
    return np.array(pts), np.array(w)


def p2_shape(lam):
    """P2 basis at barycentric point lam[4]: values [10] and d/d(lambda_k) [10,4] (UFC edge order)."""
    phi = np.zeros(10)
    dphi = np.zeros((10, 4))
    for i in range(4):
        phi[i] = lam[i] * (2.0 * lam[i] - 1.0)
        dphi[i, i] = 4.0 * lam[i] - 1.0
    for e, (i, j) in enumerate(fo.P2_EDGE_VERTS):
        phi[4 + e] = 4.0 * lam[i] * lam[j]
        dphi[4 + e, i] = 4.0 * lam[j]
        dphi[4 + e, j] = 4.0 * lam[i]
    return phi, dphi


class TaylorHood:
    """Mesh + P2/P1 numbering + cell geometry."""

    def __init__(self, coords, cells):
        self.coords = np.asarray(coords, dtype=np.float64)
        self.cells = np.asarray(cells, dtype=np.int64)
        self.nv = len(self.coords)
        cd, edges = fo.p2_cell_dofs(self.nv, cells)
        self.cell_nodes = cd.astype(np.int64)              # [nc,10]
        self.edges = edges
        self.node_coords = fo.p2_dof_coordinates(self.coords, edges.astype(np.int64))
        self.n_nodes = len(self.node_coords)
        self.n = 4 * self.n_nodes
        detJ, g = fo.p1_geometry(self.coords, cells)       # g[nc,4,3] = grad lambda_k
        self.vol = np.abs(detJ) / 6.0
        self.glam = g

    def velocity_dofs(self, nodes, comps=(0, 1, 2)):
        nodes = np.asarray(nodes, dtype=np.int64)
        return (nodes[:, None] * 4 + np.asarray(comps)[None, :]).ravel()

    def pressure_dofs(self, vertices):
        return np.asarray(vertices, dtype=np.int64) * 4 + 3

    def dummy_dofs(self):
        return np.arange(self.nv, self.n_nodes, dtype=np.int64) * 4 + 3

    def boundary_nodes(self, inside):
        """P2 nodes (vertices and edge mid-points) of the boundary facets whose mid-point satisfies inside(x)
        (DirichletBC topological search on marked facets)."""
        facets, _, cnt = fo.facet_numbering(self.cells)
        bf = facets[cnt == 1].astype(np.int64)
        mid = self.coords[bf].mean(axis=1)
        sel = bf[np.array([bool(inside(x)) for x in mid])]
        nodes = set(sel.ravel().tolist())
        key = {(int(a), int(b)): self.nv + k for k, (a, b) in enumerate(self.edges.astype(np.int64))}
        for t in sel:
            for a, b in ((t[0], t[1]), (t[0], t[2]), (t[1], t[2])):
                nodes.add(key[(min(a, b), max(a, b))])
        return np.array(sorted(nodes), dtype=np.int64)


def cell_h(coords, cells):
    """2 * Circumradius per tetrahedron (UFL's cell size in the G2 term, CoupledNavierStokesSolver.py:343)."""
    X = np.asarray(coords, dtype=np.float64)[np.asarray(cells, dtype=np.int64)]
    d = lambda p, q: np.linalg.norm(X[:, p] - X[:, q], axis=1)          # noqa: E731
    aA, bB, cC = d(0, 1) * d(2, 3), d(0, 2) * d(1, 3), d(0, 3) * d(1, 2)
    vol = np.abs(np.linalg.det(np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0], X[:, 3] - X[:, 0]], axis=1))) / 6.0
    prod = (aA + bB + cC) * (aA + bB - cC) * (aA - bB + cC) * (-aA + bB + cC)
    return 2.0 * np.sqrt(np.maximum(prod, 0.0)) / (24.0 * vol)


def g2_delta1(g2, h, U2, inv_dt):
    """delta1 of the reference's G2 term (:344-355): kappa1 h^2 for Re <= 1 (mode 1); otherwise kappa1/2 h/|a| (steady) or
    kappa1/2 / sqrt(1/dt^2 + 1/(|a|^2 h^2)) (transient); 0 where the advecting velocity vanishes (the term does)."""
    mode, kappa1 = g2
    if mode == 1:
        return kappa1 * h * h
    out = np.zeros_like(U2)
    ok = U2 > 0
    if inv_dt != 0.0:
        out[ok] = 0.5 * kappa1 / np.sqrt(inv_dt ** 2 + 1.0 / (U2[ok] * h[ok] ** 2))
    else:
        out[ok] = 0.5 * kappa1 * h[ok] / np.sqrt(U2[ok])
    return out


def viscosity_at(nu, law, p_q, T_q=None):
    """CoupledNavierStokesSolver.viscosity (:194-213): Newtonian -> nu; without a temperature nu * pow(p / p_ref, 0.1) with the
    CURRENT pressure (the reference evaluates it on up_0 / w_current, :306 and :401), law = (p_ref, exponent); with
    solving_temperature (:199-203) nu (1 + (p/p_ref) 0.1) (1 - (T/T_ref) 0.2), law = ('pT', p_ref, c_p, T_ref, c_T, T_vertices).
    p_q, T_q: pressure / temperature at the quadrature points."""
    if law is None:
        return np.full(np.shape(p_q), float(nu))
    if law[0] == 'pT':
        _, pref, cp, tref, ct = law[:5]
        return float(nu) * (1.0 + (np.asarray(p_q, dtype=np.float64) / float(pref)) * cp) * (1.0 - (np.asarray(T_q, dtype=np.float64) / float(tref)) * ct)
    return float(nu) * np.power(np.asarray(p_q, dtype=np.float64) / float(law[0]), float(law[1]))


def law_temperature(th, law):
    """[nc,4] vertex temperatures of every cell for the 'pT' law, None otherwise."""
    if law is None or law[0] != 'pT':
        return None
    return np.asarray(law[5], dtype=np.float64)[th.cells]


def ns_system(th, w0, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None, newton=True, convection=True,
              quad_degree=5, mesh_velocity=None, g2=None, viscosity_law=None):
    """Linearised system at the state w0:  J(w0) w_new = g(w0).

    J = 2 nu eps:eps + (1/dt) mass + (grad(.) u0).v [+ (grad(u0) .).v if newton] - (p/rho) div v + (q/rho) div u
    g = f.v + (1/dt) u_prev.v [+ (grad(u0) u0).v if newton]
    so that the Newton update solves J (w_new - w0) = -R(w0) with R(w) = K(w) w - rhs.
    mesh_velocity (constant 3-vector): the ALE frame of CoupledNavierStokesSolver.py:321-329 - the ADVECTING velocity is
    u0 - w_mesh: (grad(.) (u0 - w)).v in J; the Newton terms (grad(u0) .).v and (grad(u0) u0).v are unchanged
    (J w_new = J w0 - F(w0) with F's convective part (grad(u0) (u0 - w)).v).
    viscosity_law (p_ref, e): nu(p0) = nu (p0 / p_ref)^e frozen at the state w0 - Picard in the viscosity; K(w) w - rhs
    is still the exact residual, the reference's Newton (derivative of the form) converges to the same root.
    Returns (J csr [n,n], g [n]); dummy pressure rows are identity / zero.
    """
    nc = len(th.cells)
    pts, wq = tet_quadrature(quad_degree)
    W0 = np.asarray(w0, dtype=np.float64).reshape(th.n_nodes, 4)
    U0 = W0[th.cell_nodes][:, :, :3]                       # [nc,10,3]
    Up = None if w_prev is None else np.asarray(w_prev, dtype=np.float64).reshape(th.n_nodes, 4)[th.cell_nodes][:, :, :3]
    f = np.zeros(3) if body_force is None else np.asarray(body_force, dtype=np.float64)
    Ke = np.zeros((nc, 10, 4, 10, 4))
    ge = np.zeros((nc, 10, 4))
    h_cell = cell_h(th.coords, th.cells) if g2 is not None else None
    P0 = W0[th.cells][:, :, 3]                              # [nc,4] pressure at the cell vertices
    nu_const = nu
    T0 = law_temperature(th, viscosity_law)
    for lam, w in zip(pts, wq):
        nu = viscosity_at(nu_const, viscosity_law, P0 @ np.asarray(lam), None if T0 is None else T0 @ np.asarray(lam))     # [nc]
        phi, dphi = p2_shape(lam)
        gphi = np.einsum("ak,cki->cai", dphi, th.glam)     # [nc,10,3] physical gradients
        psi = lam                                           # P1 basis = barycentric coordinates
        wv = w * th.vol                                     # [nc]
        u0 = np.einsum("a,cai->ci", phi, U0)                # [nc,3]
        gu0 = np.einsum("cai,caj->cij", U0, gphi)           # [nc,3,3] d u0_i / d x_j
        # viscous: nu (delta_ij grad phi_a . grad phi_b + d_j phi_a d_i phi_b)
        gg = np.einsum("cak,cbk->cab", gphi, gphi)
        for i in range(3):
            Ke[:, :, i, :, i] += (nu * wv)[:, None, None] * gg
        Ke[:, :, :3, :, :3] += (nu * wv)[:, None, None, None, None] * np.einsum("caj,cbi->caibj", gphi, gphi)
        # mass
        mm = np.einsum("a,b->ab", phi, phi)
        for i in range(3):
            Ke[:, :, i, :, i] += (inv_dt * wv)[:, None, None] * mm[None]
        if convection:
            ua = u0 if mesh_velocity is None else u0 - np.asarray(mesh_velocity, dtype=np.float64)[None, :]
            adv = np.einsum("ck,cbk->cb", ua, gphi)         # (u0 - w_mesh) . grad phi_b
            cc = np.einsum("a,cb->cab", phi, adv)
            for i in range(3):
                Ke[:, :, i, :, i] += wv[:, None, None] * cc
            if g2 is not None:
                # F -= delta1 (a.grad u).(a.grad v) dx with a frozen at w0: the system is written for the new iterate, so the
                # right-hand side does not change and K(w) w - rhs is the exact residual of the term
                d1 = g2_delta1(g2, h_cell, np.einsum("ck,ck->c", ua, ua), inv_dt)
                ss = np.einsum("ca,cb->cab", adv, adv)
                for i in range(3):
                    Ke[:, :, i, :, i] -= (wv * d1)[:, None, None] * ss
            if newton:
                Ke[:, :, :3, :, :3] += wv[:, None, None, None, None] * np.einsum("ab,cij->caibj", mm, gu0)
                ge[:, :, :3] += wv[:, None, None] * np.einsum("a,ci->cai", phi, np.einsum("cij,cj->ci", gu0, u0))
        # pressure gradient / continuity (pressure basis on the 4 vertex nodes = local nodes 0..3)
        for m in range(4):
            Ke[:, :, :3, m, 3] += (-(1.0 / rho) * wv * psi[m])[:, None, None] * gphi
            Ke[:, m, 3, :, :3] += ((1.0 / rho) * wv * psi[m])[:, None, None] * gphi
        # loads
        ge[:, :, :3] += wv[:, None, None] * np.einsum("a,i->ai", phi, f)[None]
        if Up is not None and inv_dt != 0.0:
            up = np.einsum("a,cai->ci", phi, Up)
            ge[:, :, :3] += (inv_dt * wv)[:, None, None] * np.einsum("a,ci->cai", phi, up)
    dofs = (th.cell_nodes[:, :, None] * 4 + np.arange(4)[None, None, :]).reshape(nc, 40)
    rows = np.repeat(dofs, 40, axis=1).ravel()
    cols = np.tile(dofs, (1, 40)).ravel()
    J = sp.coo_matrix((Ke.reshape(nc, 1600).ravel(), (rows, cols)), shape=(th.n, th.n)).tocsr()
    g = np.zeros(th.n)
    np.add.at(g, dofs.ravel(), ge.reshape(nc, 40).ravel())
    dd = th.dummy_dofs()
    J = J + sp.coo_matrix((np.ones(len(dd)), (dd, dd)), shape=(th.n, th.n)).tocsr()
    return J, g


def apply_dirichlet_rows(J, g, dofs, vals):
    """DirichletBC.apply on a non-symmetric system: row -> identity, rhs -> value (columns kept)."""
    J = J.tolil()
    dofs = np.asarray(dofs, dtype=np.int64)
    vals = np.broadcast_to(np.asarray(vals, dtype=np.float64), dofs.shape)
    for d, v in zip(dofs, vals):
        J.rows[d] = [int(d)]
        J.data[d] = [1.0]
        g[d] = v
    return J.tocsr(), g


def residual(th, w, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None, mesh_velocity=None, g2=None, viscosity_law=None):
    """R(w) = K(w) w - rhs  (the nonlinear residual F of the reference after action(F, w))."""
    K, rhs = ns_system(th, w, nu, rho, inv_dt, w_prev, body_force, newton=False, mesh_velocity=mesh_velocity, g2=g2,
                       viscosity_law=viscosity_law)
    return K @ w - rhs


def viscous_stress_projection(th, w, nu, viscosity_law=None):
    """project(nu (grad u + grad u^T) - p I, TensorFunctionSpace(mesh, 'CG', 1)) (CoupledNavierStokesSolver.py:149-155):
    consistent P1 mass matrix, 4-point rule (integrand quadratic).  Returns sigma [nv, 3, 3]."""
    W = np.asarray(w, dtype=np.float64).reshape(th.n_nodes, 4)
    U = W[th.cell_nodes][:, :, :3]
    Pv = W[th.cells][:, :, 3]
    be = np.zeros((len(th.cells), 4, 9))
    pts, wq = tet_quadrature(5 if (viscosity_law is not None and viscosity_law[0] == 'pT') else 2)     # nu(p, T): quartic integrand
    for lam, wt in zip(pts, wq):
        lam = np.asarray(lam)
        _, dphi = p2_shape(lam)
        gphi = np.einsum("ak,cki->cai", dphi, th.glam)
        G = np.einsum("cai,caj->cij", U, gphi)
        pq = Pv @ lam
        T0 = law_temperature(th, viscosity_law)
        sig = viscosity_at(nu, viscosity_law, pq, None if T0 is None else T0 @ lam)[:, None, None] * (G + np.swapaxes(G, 1, 2)) \
            - pq[:, None, None] * np.eye(3)
        be += (wt * th.vol)[:, None, None] * lam[None, :, None] * sig.reshape(-1, 1, 9)
    M = fo.assemble_matrix(th.nv, th.cells, fo.p1_mass_local(th.coords, th.cells, 1.0))
    out = np.zeros((th.nv, 9))
    for k in range(9):
        out[:, k] = fo.solve_direct(M, fo.assemble_generic_vector(th.nv, th.cells, be[:, :, k]))
    return out.reshape(th.nv, 3, 3)


def boundary_force(th, sigma, inside):
    """-int sigma n ds over the boundary facets whose mid-point satisfies inside(x) (calc_drag_and_lift, :166-183; n the
    outward normal): with P1 sigma the facet integral is area * mean of the three vertex tensors."""
    facets, cell_facets, cnt = fo.facet_numbering(th.cells)
    bf = np.nonzero(cnt == 1)[0]
    owner = np.zeros(len(facets), dtype=np.int64)
    owner[cell_facets.ravel()] = np.repeat(np.arange(len(th.cells)), 4)
    F = np.zeros(3)
    for f in bf:
        tri = facets[f].astype(np.int64)
        X = th.coords[tri]
        if not inside(X.mean(axis=0)):
            continue
        nrm = 0.5 * np.cross(X[1] - X[0], X[2] - X[0])
        if nrm @ (X.mean(axis=0) - th.coords[th.cells[owner[f]]].mean(axis=0)) < 0:
            nrm = -nrm
        F -= sigma[tri].mean(axis=0) @ nrm
    return F


def newton_solve(th, w_init, bc_dofs, bc_vals, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None,
                 rtol=1e-9, atol=1e-10, max_it=50, newton=True, relax=1.0, g2=None, viscosity_law=None):
    """DOLFIN NewtonSolver semantics (relative 1e-9 / absolute 1e-10 on the residual 2-norm)."""
    import scipy.sparse.linalg as spl
    w = np.array(w_init, dtype=np.float64)
    w[bc_dofs] = bc_vals
    free = np.ones(th.n, dtype=bool)
    free[bc_dofs] = False
    r0 = None
    history = []
    for it in range(max_it + 1):
        J, g = ns_system(th, w, nu, rho, inv_dt, w_prev, body_force, newton=newton, g2=g2, viscosity_law=viscosity_law)
        r = (J @ w - g)
        r[~free] = 0.0
        rn = np.linalg.norm(r)
        history.append(rn)
        r0 = rn if r0 is None else r0
        if rn <= atol or rn <= rtol * r0:
            return w, history
        Jb, gb = apply_dirichlet_rows(J, g.copy(), bc_dofs, bc_vals)
        w_new = spl.spsolve(Jb.tocsc(), gb)
        w = w + relax * (w_new - w)
    raise RuntimeError("Newton did not converge: %r" % history)


# ---- pressure boundaries (CoupledNavierStokesSolver.py:449-453, 459-460) -----------------------------------
# 6-point degree-4 rule on the triangle (barycentric, weights sum to 1)
TRI_QP = np.array([[0.108103018168070, 0.445948490915965, 0.445948490915965],
                   [0.445948490915965, 0.108103018168070, 0.445948490915965],
                   [0.445948490915965, 0.445948490915965, 0.108103018168070],
                   [0.816847572980459, 0.091576213509771, 0.091576213509771],
                   [0.091576213509771, 0.816847572980459, 0.091576213509771],
                   [0.091576213509771, 0.091576213509771, 0.816847572980459]])
TRI_QW = np.array([0.223381589678011] * 3 + [0.109951743655322] * 3)


def boundary_facet_cells(th, inside):
    """(cell, opposite local vertex) of the boundary facets whose mid-point satisfies inside(x)."""
    out = []
    opp = ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2))
    facets, cf, cnt = fo.facet_numbering(th.cells)
    for c in range(len(th.cells)):
        for o in range(4):
            if cnt[cf[c, o]] != 1:
                continue
            if inside(th.coords[th.cells[c, list(opp[o])]].mean(axis=0)):
                out.append((c, o))
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def pressure_boundary_terms(th, facet_cells, nu, bvalue=None, viscosity_law=None, w0=None):
    """F += inner(bvalue*n, v)*ds - nu*inner((grad(u) + grad(u).T)*n, v)*ds on the given boundary facets.
    Returns (dJ csr, dg): the matrix of the viscous traction term and the load (moved to the right-hand side).
    bvalue None: the 'farfield' pressure type (traction term only)."""
    nf = len(facet_cells)
    Ke = np.zeros((nf, 10, 4, 10, 4))
    ge = np.zeros((nf, 10, 4))
    opp = ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2))
    nu_const = nu
    P0 = None if viscosity_law is None else np.asarray(w0, dtype=np.float64).reshape(th.n_nodes, 4)[th.cells][:, :, 3]
    for k, (c, o) in enumerate(facet_cells):
        gl = th.glam[c]
        gnorm = np.linalg.norm(gl[o])
        n = -gl[o] / gnorm
        area = 3.0 * th.vol[c] * gnorm
        for bary, w in zip(TRI_QP, TRI_QW):
            lam = np.zeros(4)
            lam[list(opp[o])] = bary
            T0 = law_temperature(th, viscosity_law)
            nu = nu_const if viscosity_law is None else float(viscosity_at(nu_const, viscosity_law, P0[c] @ lam,
                                                                             None if T0 is None else T0[c] @ lam))
            phi, dphi = p2_shape(lam)
            gphi = dphi @ gl                          # [10,3]
            wv = w * area
            gn = gphi @ n                              # grad phi_b . n
            for i in range(3):
                Ke[k, :, i, :, i] += -nu * wv * np.outer(phi, gn)
            Ke[k, :, :3, :, :3] += -nu * wv * np.einsum("a,bi,j->aibj", phi, gphi, n)
            if bvalue is not None:
                # a number, or a function of the point (a boundary pressure that varies: evaluated through its P1 interpolant
                # on the facet, i.e. from its values at the facet's vertices - what DOLFIN does with a degree-1 Expression)
                if callable(bvalue):
                    pv = np.array([bvalue(th.coords[th.cells[c, v]]) for v in opp[o]])
                    pb = float(pv @ bary)
                else:
                    pb = float(bvalue)
                ge[k, :, :3] -= wv * pb * np.outer(phi, n)
    cells = facet_cells[:, 0]
    dofs = (th.cell_nodes[cells][:, :, None] * 4 + np.arange(4)[None, None, :]).reshape(nf, 40)
    rows = np.repeat(dofs, 40, axis=1).ravel()
    cols = np.tile(dofs, (1, 40)).ravel()
    dJ = sp.coo_matrix((Ke.reshape(nf, 1600).ravel(), (rows, cols)), shape=(th.n, th.n)).tocsr()
    dg = np.zeros(th.n)
    np.add.at(dg, dofs.ravel(), ge.reshape(nf, 40).ravel())
    return dJ, dg
