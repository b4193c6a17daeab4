"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product) for the Taylor-Hood path of CoupledNavierStokesSolver on
TRIANGLES: P2 velocity / P1 pressure in two dimensions - the case the reference's own CFD example runs
(examples/test_cfd_solver.py:83-170 on UnitSquareMesh(40, 100); the class is dimension-free,
FenicsSolver/CoupledNavierStokesSolver.py:84-102, 288-381).

Parity unpinned against DOLFIN/FFC/PETSc themselves (not installable here); pinned by known answers: quadrature exactness,
exact reproduction of plane Poiseuille flow (quadratic velocity + linear pressure lie in the Taylor-Hood space), a
finite-difference check of the Jacobian against the residual, and the form goldens recorded from the reference's Python layer
for a 2-D case (tests/golden/reference_forms.json: navier_stokes_2d_*).

Numbering, as on the device: one block of 4 unknowns per P2 node (vertices first, then edge mid-points in
fem_oracle.tri_p2_cell_dofs order): (u_x, u_y, -, p).  The third slot is a dummy unknown with an identity row, as is the pressure
slot of an edge node.  Quadrature: the 12-point degree-6 rule of Dunavant - deliberately NOT the 7-point rule of the device
kernel, so that agreement to 1e-11 also checks that rule's exactness for the degree-5 integrands (the G2 term, degree 6 in the
advecting velocity, is integrated with the device's 7 points on request: quad='radon7')."""
import numpy as np
import scipy.sparse as sp

from . import fem_oracle as fo
from .ns_oracle import apply_dirichlet_rows, g2_delta1, viscosity_at, law_temperature   # noqa: F401  (shared, dimension-free helpers)


def tri_quadrature(name="dunavant12"):
    """Barycentric points [nq,3] and weights (sum 1)."""
    if name == "radon7":
        s15 = np.sqrt(15.0)
        a1, w1 = (6.0 - s15) / 21.0, (155.0 - s15) / 1200.0
        a2, w2 = (6.0 + s15) / 21.0, (155.0 + s15) / 1200.0
        pts, w = [[1 / 3.0] * 3], [0.225]
        for a, wt in ((a1, w1), (a2, w2)):
            for k in range(3):
                p = [a, a, a]
                p[k] = 1.0 - 2.0 * a
                pts.append(p)
                w.append(wt)
        return np.array(pts), np.array(w)
    pts, w = [], []
    for a, wt in ((0.249286745170910, 0.116786275726379), (0.063089014491502, 0.050844906370207)):
        for k in range(3):
            p = [a, a, a]
            p[k] = 1.0 - 2.0 * a
            pts.append(p)
            w.append(wt)
    a, b = 0.053145049844817, 0.310352451033784
    c = 1.0 - a - b
    for p in ((a, b, c), (a, c, b), (b, a, c), (b, c, a), (c, a, b), (c, b, a)):
        pts.append(list(p))
        w.append(0.082851075618374)
    return np.array(pts), np.array(w)


class TaylorHood2D:
    def __init__(self, coords, cells):
        self.coords = np.asarray(coords, dtype=np.float64)[:, :2]
        self.cells = np.asarray(cells, dtype=np.int64)
        self.nv = len(self.coords)
        cd, edges = fo.tri_p2_cell_dofs(self.nv, cells)
        self.cell_nodes = cd.astype(np.int64)                       # [nc,6]
        self.edges = edges.astype(np.int64)
        self.node_coords = np.concatenate([self.coords, 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])])
        self.n_nodes = len(self.node_coords)
        self.n = 4 * self.n_nodes
        self.area, self.glam = fo.tri_geometry(self.coords, self.cells)     # glam [nc,3,2]

    def velocity_dofs(self, nodes, comps=(0, 1)):
        nodes = np.asarray(nodes, dtype=np.int64)
        return (nodes[:, None] * 4 + np.asarray(comps)[None, :]).ravel()

    def pressure_dofs(self, vertices):
        return np.asarray(vertices, dtype=np.int64) * 4 + 3

    def dummy_dofs(self):
        return np.concatenate([np.arange(self.n_nodes, dtype=np.int64) * 4 + 2, np.arange(self.nv, self.n_nodes, dtype=np.int64) * 4 + 3])

    def boundary_nodes(self, inside):
        """P2 nodes (two vertices + the mid-point) of the boundary edges whose mid-point satisfies inside(x)."""
        edges, _, cnt = fo.tri_edge_numbering(self.cells)
        be = edges[cnt == 1].astype(np.int64)
        mid = self.coords[be].mean(axis=1)
        sel = be[np.array([bool(inside(x)) for x in mid])]
        key = {(int(a), int(b)): self.nv + k for k, (a, b) in enumerate(self.edges)}
        nodes = set(sel.ravel().tolist())
        for a, b in sel:
            nodes.add(key[(min(a, b), max(a, b))])
        return np.array(sorted(nodes), dtype=np.int64)


def cell_h(th):
    """2 * circumradius of every triangle = |e0||e1||e2| / (2 area) (UFL's 2*Circumradius(mesh), :343)."""
    X = th.coords[th.cells]
    d = lambda p, q: np.linalg.norm(X[:, p] - X[:, q], axis=1)          # noqa: E731
    return d(0, 1) * d(0, 2) * d(1, 2) / (2.0 * th.area)


def ns_system(th, w0, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None, newton=True, convection=True, quad="dunavant12",
              mesh_velocity=None, g2=None, viscosity_law=None):
    """J(w0) w_new = g(w0), the 2-D counterpart of ns_oracle.ns_system (same terms, same signs, same layout of 4 per node)."""
    nc = len(th.cells)
    pts, wq = tri_quadrature(quad)
    W0 = np.asarray(w0, dtype=np.float64).reshape(th.n_nodes, 4)
    U0 = W0[th.cell_nodes][:, :, :2]
    Up = None if w_prev is None else np.asarray(w_prev, dtype=np.float64).reshape(th.n_nodes, 4)[th.cell_nodes][:, :, :2]
    f = np.zeros(2) if body_force is None else np.asarray(body_force, dtype=np.float64)[:2]
    Ke = np.zeros((nc, 6, 4, 6, 4))
    ge = np.zeros((nc, 6, 4))
    h_cell = cell_h(th) if g2 is not None else None
    P0 = W0[th.cells][:, :, 3]
    nu_const = nu
    for lam, w in zip(pts, wq):
        T0 = law_temperature(th, viscosity_law)
        nu = viscosity_at(nu_const, viscosity_law, P0 @ np.asarray(lam), None if T0 is None else T0 @ np.asarray(lam))
        phi, dphi = fo.tri_p2_shape(lam)
        gphi = np.einsum("ak,cki->cai", dphi, th.glam)              # [nc,6,2]
        wv = w * th.area
        u0 = np.einsum("a,cai->ci", phi, U0)
        gu0 = np.einsum("cai,caj->cij", U0, gphi)
        gg = np.einsum("cak,cbk->cab", gphi, gphi)
        mm = np.einsum("a,b->ab", phi, phi)
        for i in range(2):
            Ke[:, :, i, :, i] += (nu * wv)[:, None, None] * gg + (inv_dt * wv)[:, None, None] * mm[None]
        Ke[:, :, :2, :, :2] += (nu * wv)[:, None, None, None, None] * np.einsum("caj,cbi->caibj", gphi, gphi)
        if convection:
            ua = u0 if mesh_velocity is None else u0 - np.asarray(mesh_velocity, dtype=np.float64)[None, :2]
            adv = np.einsum("ck,cbk->cb", ua, gphi)
            cc = np.einsum("a,cb->cab", phi, adv)
            for i in range(2):
                Ke[:, :, i, :, i] += wv[:, None, None] * cc
            if g2 is not None:
                d1 = g2_delta1(g2, h_cell, np.einsum("ck,ck->c", ua, ua), inv_dt)
                ss = np.einsum("ca,cb->cab", adv, adv)
                for i in range(2):
                    Ke[:, :, i, :, i] -= (wv * d1)[:, None, None] * ss
            if newton:
                Ke[:, :, :2, :, :2] += wv[:, None, None, None, None] * np.einsum("ab,cij->caibj", mm, gu0)
                ge[:, :, :2] += wv[:, None, None] * np.einsum("a,ci->cai", phi, np.einsum("cij,cj->ci", gu0, u0))
        for m in range(3):
            Ke[:, :, :2, m, 3] += (-(1.0 / rho) * wv * lam[m])[:, None, None] * gphi
            Ke[:, m, 3, :, :2] += ((1.0 / rho) * wv * lam[m])[:, None, None] * gphi
        ge[:, :, :2] += wv[:, None, None] * np.einsum("a,i->ai", phi, f)[None]
        if Up is not None and inv_dt != 0.0:
            up = np.einsum("a,cai->ci", phi, Up)
            ge[:, :, :2] += (inv_dt * wv)[:, None, None] * np.einsum("a,ci->cai", phi, up)
    dofs = (th.cell_nodes[:, :, None] * 4 + np.arange(4)[None, None, :]).reshape(nc, 24)
    rows = np.repeat(dofs, 24, axis=1).ravel()
    cols = np.tile(dofs, (1, 24)).ravel()
    J = sp.coo_matrix((Ke.reshape(nc, 576).ravel(), (rows, cols)), shape=(th.n, th.n)).tocsr()
    g = np.zeros(th.n)
    np.add.at(g, dofs.ravel(), ge.reshape(nc, 24).ravel())
    dd = th.dummy_dofs()
    J = J + sp.coo_matrix((np.ones(len(dd)), (dd, dd)), shape=(th.n, th.n)).tocsr()
    return J, g


def residual(th, w, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None, mesh_velocity=None, g2=None, viscosity_law=None, quad="dunavant12"):
    K, rhs = ns_system(th, w, nu, rho, inv_dt, w_prev, body_force, newton=False, mesh_velocity=mesh_velocity, g2=g2,
                       viscosity_law=viscosity_law, quad=quad)
    return K @ w - rhs


def boundary_edge_cells(th, inside):
    """(cell, opposite local vertex) of the boundary edges whose mid-point satisfies inside(x), ascending edge id."""
    edges, cf, cnt = fo.tri_edge_numbering(th.cells)
    out = []
    for c in range(len(th.cells)):
        for o in range(3):
            e = cf[c, o]
            if cnt[e] == 1 and inside(th.coords[edges[e].astype(np.int64)].mean(axis=0)):
                out.append((e, c, o))
    out.sort()
    return np.array([(c, o) for _, c, o in out], dtype=np.int64).reshape(-1, 2)


def pressure_boundary_terms(th, facet_cells, nu, bvalue=None, viscosity_law=None, w0=None):
    """F += inner(bvalue*n, v)*ds - nu*inner((grad(u) + grad(u).T)*n, v)*ds on boundary edges (:449-453, 459-460): (dJ, dg).
    4-point Gauss-Legendre on the edge (the device uses 3 points: both exact for these integrands)."""
    nf = len(facet_cells)
    Ke = np.zeros((nf, 6, 4, 6, 4))
    ge = np.zeros((nf, 6, 4))
    xg, wg = np.polynomial.legendre.leggauss(4)
    sg, wg = 0.5 * (xg + 1.0), 0.5 * wg
    nu_const = nu
    P0 = None if viscosity_law is None else np.asarray(w0, dtype=np.float64).reshape(th.n_nodes, 4)[th.cells][:, :, 3]
    for k, (c, o) in enumerate(facet_cells):
        gl = th.glam[c]
        gnorm = np.linalg.norm(gl[o])
        n = -gl[o] / gnorm
        length = 2.0 * th.area[c] * gnorm
        vi, vj = [v for v in range(3) if v != o]
        for s, w in zip(sg, wg):
            lam = np.zeros(3)
            lam[vi], lam[vj] = 1.0 - s, s
            T0 = law_temperature(th, viscosity_law)
            nu = nu_const if viscosity_law is None else float(viscosity_at(nu_const, viscosity_law, P0[c] @ lam,
                                                                             None if T0 is None else T0[c] @ lam))
            phi, dphi = fo.tri_p2_shape(lam)
            gphi = dphi @ gl
            wv = w * length
            gn = gphi @ n
            for i in range(2):
                Ke[k, :, i, :, i] += -nu * wv * np.outer(phi, gn)
            Ke[k, :, :2, :, :2] += -nu * wv * np.einsum("a,bi,j->aibj", phi, gphi, n)
            if bvalue is not None:
                if callable(bvalue):
                    pb = (1.0 - s) * bvalue(th.coords[th.cells[c, vi]]) + s * bvalue(th.coords[th.cells[c, vj]])
                else:
                    pb = float(bvalue)
                ge[k, :, :2] -= wv * pb * np.outer(phi, n)
    cells = facet_cells[:, 0]
    dofs = (th.cell_nodes[cells][:, :, None] * 4 + np.arange(4)[None, None, :]).reshape(nf, 24)
    rows = np.repeat(dofs, 24, axis=1).ravel()
    cols = np.tile(dofs, (1, 24)).ravel()
    dJ = sp.coo_matrix((Ke.reshape(nf, 576).ravel(), (rows, cols)), shape=(th.n, th.n)).tocsr()
    dg = np.zeros(th.n)
    np.add.at(dg, dofs.ravel(), ge.reshape(nf, 24).ravel())
    return dJ, dg


def newton_solve(th, w_init, bc_dofs, bc_vals, nu, rho=1.0, inv_dt=0.0, w_prev=None, body_force=None, rtol=1e-9, atol=1e-10,
                 max_it=50, newton=True, relax=1.0, g2=None, viscosity_law=None, extra=None):
    """DOLFIN NewtonSolver semantics; extra(w) -> (dJ, dg) adds boundary terms (pressure boundaries)."""
    import scipy.sparse.linalg as spl
    w = np.array(w_init, dtype=np.float64)
    w[bc_dofs] = bc_vals
    free = np.ones(th.n, dtype=bool)
    free[bc_dofs] = False
    free[th.dummy_dofs()] = False
    r0, history = None, []
    for it in range(max_it + 1):
        J, g = ns_system(th, w, nu, rho, inv_dt, w_prev, body_force, newton=newton, g2=g2, viscosity_law=viscosity_law)
        if extra is not None:
            dJ, dg = extra(w)
            J, g = J + dJ, g + dg
        r = J @ w - g
        r[~free] = 0.0
        rn = np.linalg.norm(r)
        history.append(rn)
        r0 = rn if r0 is None else r0
        if rn <= atol or rn <= rtol * r0:
            return w, history
        Jb, gb = apply_dirichlet_rows(J, g.copy(), bc_dofs, bc_vals)
        w = w + relax * (spl.spsolve(Jb.tocsc(), gb) - w)
    raise RuntimeError("Newton did not converge: %r" % history)


def viscous_stress_projection(th, w, nu, viscosity_law=None):
    """project(nu (grad u + grad u^T) - p I, TensorFunctionSpace(mesh, 'CG', 1)) (CoupledNavierStokesSolver.py:149-155) on
    triangles: consistent P1 mass matrix, loads by the 12-point rule.  Returns sigma [nv, 2, 2]."""
    W = np.asarray(w, dtype=np.float64).reshape(th.n_nodes, 4)
    U = W[th.cell_nodes][:, :, :2]
    Pv = W[th.cells][:, :, 3]
    be = np.zeros((len(th.cells), 3, 4))
    pts, wq = tri_quadrature("dunavant12")
    for lam, wt in zip(pts, wq):
        lam = np.asarray(lam)
        _, dphi = fo.tri_p2_shape(lam)
        gphi = np.einsum("ak,cki->cai", dphi, th.glam)
        G = np.einsum("cai,caj->cij", U, gphi)
        pq = Pv @ lam
        T0 = law_temperature(th, viscosity_law)
        sig = viscosity_at(nu, viscosity_law, pq, None if T0 is None else T0 @ np.asarray(lam))[:, None, None] * (G + np.swapaxes(G, 1, 2)) \
            - pq[:, None, None] * np.eye(2)
        be += (wt * th.area)[:, None, None] * lam[None, :, None] * sig.reshape(-1, 1, 4)
    M = fo.assemble_matrix(th.nv, th.cells, fo.tri_mass_local(th.coords, th.cells, 1.0))
    out = np.zeros((th.nv, 4))
    for k in range(4):
        out[:, k] = fo.solve_direct(M, fo.assemble_generic_vector(th.nv, th.cells, be[:, :, k]))
    return out.reshape(th.nv, 2, 2)


def boundary_force(th, sigma, inside):
    """-int sigma n ds over the boundary edges whose mid-point satisfies inside(x) (calc_drag_and_lift, :166-183; n the outward
    normal): with P1 sigma the edge integral is length * mean of the two vertex tensors."""
    edges, cell_edges, cnt = fo.tri_edge_numbering(th.cells)
    owner = np.zeros(len(edges), dtype=np.int64)
    owner[cell_edges.ravel()] = np.repeat(np.arange(len(th.cells)), 3)
    F = np.zeros(2)
    for e in np.nonzero(cnt == 1)[0]:
        ed = edges[e].astype(np.int64)
        X = th.coords[ed]
        if not inside(X.mean(axis=0)):
            continue
        t = X[1] - X[0]
        nrm = np.array([t[1], -t[0]])
        if nrm @ (X.mean(axis=0) - th.coords[th.cells[owner[e]]].mean(axis=0)) < 0:
            nrm = -nrm
        F -= sigma[ed].mean(axis=0) @ nrm
    return F
