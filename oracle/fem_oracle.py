"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the FenicsSolver hot path.

This module is a numpy/scipy restatement of what the reference delegates to
DOLFIN/FFC/PETSc when ``SolverBase.solve()`` runs
(/root/reference/FenicsSolver/SolverBase.py:484-490, 544-546, 592-672):
mesh ingest, P1 dof tables, cell-by-cell assembly of the forms built in
ScalarTransportSolver.generate_form (ScalarTransportSolver.py:228-359) and
LinearElasticitySolver.generate_form (LinearElasticitySolver.py:206-245),
Dirichlet application and the linear solve.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  It is never on the product path.

PARITY PINNING.  The arithmetic of the reference path lives in un-vendored
third-party code (DOLFIN/FFC/UFL/FIAT 2017.1-2019.1, PETSc; unpinned, see
SURVEY.md section 8c) that cannot be built or imported here, and the
reference's own example scripts assert no number.  The oracle is therefore
pinned by the known-answer set-ups the reference ships:
  * data/TestHeatTransfer.json + data/mesh*.xml  -> T = 350 - 2.5 z exactly
    (tests/golden/data, tests/test_oracle_*.py),
  * lexicographic facet numbering reproduces the 100+100 marked boundary
    facets of data/mesh_facet_region.xml,
  * exact reference-tet element matrices,
  * patch tests / rigid-body null space / structure counts (Appendix C4-C7).
Against DOLFIN itself parity is "unpinned" (stated in DESIGN.md).
"""
from __future__ import annotations

import re
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------
# mesh ingest  (SolverBase.py:223-258 -> dolfin.Mesh(xml), MeshFunction(xml))
# --------------------------------------------------------------------------

_VERT_RE = re.compile(
    r'<vertex\s+index="(\d+)"\s+x="([^"]+)"\s+y="([^"]+)"(?:\s+z="([^"]+)")?')
_TET_RE = re.compile(
    r'<tetrahedron\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"\s+v3="(\d+)"')
_TRI_RE = re.compile(
    r'<triangle\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"')
_ENT_RE = re.compile(r'<entity\s+index="(\d+)"\s+value="(-?\d+)"')
_MF_RE = re.compile(r'<mesh_function\s+type="(\w+)"\s+dim="(\d+)"\s+size="(\d+)"')


def read_dolfin_xml_mesh(path):
    """DOLFIN-XML mesh reader.  Returns (coords[N,gdim] f64, cells[Nc,nv] i32).

    Each cell's vertex list is sorted ascending, as ``dolfin.Mesh(xml)`` does
    through ``mesh.order()``.
    """
    text = open(path, "r").read()
    verts = _VERT_RE.findall(text)
    has_z = any(v[3] != "" for v in verts)
    gdim = 3 if has_z else 2
    coords = np.zeros((len(verts), gdim), dtype=np.float64)
    for idx, x, y, z in verts:
        i = int(idx)
        coords[i, 0] = float(x)
        coords[i, 1] = float(y)
        if gdim == 3:
            coords[i, 2] = float(z)
    tets = _TET_RE.findall(text)
    if tets:
        cells = np.zeros((len(tets), 4), dtype=np.int32)
        for t in tets:
            cells[int(t[0])] = [int(t[1]), int(t[2]), int(t[3]), int(t[4])]
    else:
        tris = _TRI_RE.findall(text)
        cells = np.zeros((len(tris), 3), dtype=np.int32)
        for t in tris:
            cells[int(t[0])] = [int(t[1]), int(t[2]), int(t[3])]
    cells.sort(axis=1)
    return coords, cells


def read_dolfin_xml_meshfunction(path):
    """Old-style ``<mesh_function>`` reader -> (dim, values[int64])."""
    text = open(path, "r").read()
    m = _MF_RE.search(text)
    dim, size = int(m.group(2)), int(m.group(3))
    vals = np.zeros(size, dtype=np.int64)
    for idx, v in _ENT_RE.findall(text):
        vals[int(idx)] = int(v)
    return dim, vals


# --------------------------------------------------------------------------
# structured generators (dolfin.BoxMesh / UnitCubeMesh ordering, Appendix D-8)
# --------------------------------------------------------------------------

def box_mesh(p0, p1, nx, ny, nz):
    """``BoxMesh(Point(p0), Point(p1), nx, ny, nz)``: vertices x-fastest,
    cells iz->iy->ix, six tets per hex around the v0-v7 diagonal, each cell's
    vertices sorted ascending (examples/test_linear_elasticity.py:42)."""
    # DOLFIN BoxMesh.cpp evaluates  a + (i*(b - a))/n  in this order; the device
    # generator uses the same expression so coordinates are bit-identical.
    x = p0[0] + (np.arange(nx + 1, dtype=np.float64) * (p1[0] - p0[0])) / float(nx)
    y = p0[1] + (np.arange(ny + 1, dtype=np.float64) * (p1[1] - p0[1])) / float(ny)
    z = p0[2] + (np.arange(nz + 1, dtype=np.float64) * (p1[2] - p0[2])) / float(nz)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (iz * (ny + 1) * (nx + 1) + iy * (nx + 1) + ix).ravel().astype(np.int64)
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    v4 = v0 + (nx + 1) * (ny + 1)
    v5 = v1 + (nx + 1) * (ny + 1)
    v6 = v2 + (nx + 1) * (ny + 1)
    v7 = v3 + (nx + 1) * (ny + 1)
    tets = np.stack([
        np.stack([v0, v1, v3, v7], 1),
        np.stack([v0, v1, v7, v5], 1),
        np.stack([v0, v5, v7, v4], 1),
        np.stack([v0, v3, v2, v7], 1),
        np.stack([v0, v6, v4, v7], 1),
        np.stack([v0, v2, v6, v7], 1),
    ], axis=1)  # [nhex, 6, 4]
    cells = tets.reshape(-1, 4)
    cells = np.sort(cells, axis=1).astype(np.int32)
    return coords, cells


def unit_cube_mesh(n):
    return box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n)


# --------------------------------------------------------------------------
# topology: facets / edges numbered lexicographically
# --------------------------------------------------------------------------

def facet_numbering(cells):
    """Facets = sorted vertex triples ranked lexicographically.

    Returns (facets[Nf,3], cell_facets[Nc,4], facet_cell_count[Nf]) where
    local facet i of a cell is the one opposite local vertex i (UFC).
    """
    cells = np.asarray(cells, dtype=np.int64)
    nc = cells.shape[0]
    opp = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]
    tri = np.stack([cells[:, list(o)] for o in opp], axis=1).reshape(-1, 3)
    tri = np.sort(tri, axis=1)
    uniq, inv, cnt = np.unique(tri, axis=0, return_inverse=True, return_counts=True)
    return uniq.astype(np.int32), inv.reshape(nc, 4).astype(np.int32), cnt.astype(np.int32)


def edge_numbering(cells):
    """Edges = sorted vertex pairs ranked lexicographically.

    UFC local edges of a tet: e0=(v2,v3) e1=(v1,v3) e2=(v1,v2) e3=(v0,v3)
    e4=(v0,v2) e5=(v0,v1).
    """
    cells = np.asarray(cells, dtype=np.int64)
    nc = cells.shape[0]
    loc = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
    ed = np.stack([cells[:, list(e)] for e in loc], axis=1).reshape(-1, 2)
    ed = np.sort(ed, axis=1)
    uniq, inv = np.unique(ed, axis=0, return_inverse=True)
    return uniq.astype(np.int32), inv.reshape(nc, 6).astype(np.int32)


def boundary_facets(cells):
    facets, cell_facets, cnt = facet_numbering(cells)
    return np.nonzero(cnt == 1)[0].astype(np.int32), facets


def mark_facets(coords, cells, inside, marker_id, markers=None, eps=3e-16):
    """``SubDomain.mark(facets, id)`` (SolverBase.py:277-283): a facet is
    marked when all its vertices and its midpoint satisfy ``inside(x,
    on_boundary)`` (Appendix D-4).  ``inside`` receives (x[3] array, bool)."""
    facets, _, cnt = facet_numbering(cells)
    if markers is None:
        markers = np.zeros(len(facets), dtype=np.int64)
    on_b = cnt == 1
    for f in range(len(facets)):
        vs = facets[f]
        ok = True
        for v in vs:
            if not inside(coords[v], bool(on_b[f])):
                ok = False
                break
        if ok and inside(coords[vs].mean(axis=0), bool(on_b[f])):
            markers[f] = marker_id
    return markers


def dirichlet_dofs_p1(facets, facet_markers, marker_id):
    """Topological DirichletBC on P1: every vertex in the closure of the
    facets carrying ``marker_id`` (ScalarTransportSolver.py:169-175,
    Appendix D-3).  Sorted ascending."""
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    return np.unique(np.asarray(facets)[sel].ravel()).astype(np.int32)


# --------------------------------------------------------------------------
# P1 element geometry
# --------------------------------------------------------------------------

def p1_geometry(coords, cells):
    """Returns (detJ[Nc], grads[Nc,4,3]) of the barycentric basis."""
    c = np.asarray(coords, dtype=np.float64)[np.asarray(cells, dtype=np.int64)]
    J = np.stack([c[:, 1] - c[:, 0], c[:, 2] - c[:, 0], c[:, 3] - c[:, 0]], axis=2)  # columns
    detJ = np.linalg.det(J)
    Jinv = np.linalg.inv(J)  # rows of Jinv are grads of lambda_1..3
    g = np.zeros((c.shape[0], 4, 3))
    g[:, 1:, :] = Jinv
    g[:, 0, :] = -Jinv.sum(axis=1)
    return detJ, g


def p1_stiffness_local(coords, cells, k=1.0):
    """Ke[a,b] = |detJ|/6 * grad_a . K . grad_b  for a(T,q)=int k grad T.grad q
    (ScalarTransportSolver.py:284-285).  k: scalar, per-cell array or 3x3."""
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    k = np.asarray(k, dtype=np.float64)
    if k.ndim == 2 and k.shape == (3, 3):
        Kg = np.einsum("ij,caj->cai", k, g)
        Ke = np.einsum("cai,cbi->cab", g, Kg)
        return Ke * vol[:, None, None]
    if k.ndim == 3:                                     # one 3x3 tensor per cell (a tensor Expression of degree 0)
        return vol[:, None, None] * np.einsum("cai,cij,cbj->cab", g, k, g)
    Ke = np.einsum("cai,cbi->cab", g, g)
    if k.ndim == 0:
        return Ke * (vol * float(k))[:, None, None]
    return Ke * (vol * k)[:, None, None]


def p1_mass_local(coords, cells, c=1.0):
    """Me[a,b] = |detJ|/120 * (1+delta_ab) * c  (exact P1 mass matrix)."""
    detJ, _ = p1_geometry(coords, cells)
    base = (np.ones((4, 4)) + np.eye(4)) / 120.0
    c = np.asarray(c, dtype=np.float64)
    w = np.abs(detJ) * (float(c) if c.ndim == 0 else c)
    return w[:, None, None] * base[None]


# --------------------------------------------------------------------------
# CSR pattern + assembly  (DOLFIN Assembler + PETSc AIJ, SolverBase.py:595,608-612)
# --------------------------------------------------------------------------

def csr_pattern(n, cells, ncomp=1):
    """Sorted-column CSR pattern of the P1 (vector: node-interleaved) space."""
    cells = np.asarray(cells, dtype=np.int64)
    nv = cells.shape[1]
    r = np.repeat(cells, nv, axis=1).ravel()
    c = np.tile(cells, (1, nv)).ravel()
    key = np.unique(r * n + c)
    rows = key // n
    cols = key % n
    if ncomp > 1:
        # expand every node pair to an ncomp x ncomp block, dof = node*ncomp+comp
        rr = (rows[:, None, None] * ncomp + np.arange(ncomp)[None, :, None])
        cc = (cols[:, None, None] * ncomp + np.arange(ncomp)[None, None, :])
        rr = np.broadcast_to(rr, (len(rows), ncomp, ncomp)).ravel()
        cc = np.broadcast_to(cc, (len(rows), ncomp, ncomp)).ravel()
        nn = n * ncomp
        key = np.unique(rr * nn + cc)
        rows = key // nn
        cols = key % nn
        n = nn
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), cols.astype(np.int32)


def assemble_matrix(n, cells, Ke):
    """Scatter-add local matrices (sum of duplicates) into sorted CSR."""
    cells = np.asarray(cells, dtype=np.int64)
    nv = cells.shape[1]
    r = np.repeat(cells, nv, axis=1).ravel()
    c = np.tile(cells, (1, nv)).ravel()
    A = sp.coo_matrix((Ke.ravel(), (r, c)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def assemble_p1_scalar(coords, cells, k=1.0, mass_coef=None):
    """A = K(k) [+ mass_coef * M]."""
    n = len(coords)
    Ke = p1_stiffness_local(coords, cells, k)
    if mass_coef is not None:
        Ke = Ke + p1_mass_local(coords, cells, mass_coef)
    return assemble_matrix(n, cells, Ke)


def p1_advection_local(coords, cells, velocity, scale=1.0):
    """Ce[a,b] = scale * vol/4 * (v . grad phi_b): Galerkin  inner(velocity, grad(T))*Tq*capacity*dx
    (ScalarTransportSolver.py:311) with a cell-wise constant velocity.  velocity: [3] or [nc,3]."""
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    v = np.asarray(velocity, dtype=np.float64)
    if v.ndim == 3:                                    # row velocities [nc,4,3]: V_a = (4/vol) int v phi_a dx (exact for a field)
        vg = np.einsum("cai,cbi->cab", v, g)
        return scale * 0.25 * vol[:, None, None] * vg
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(cells), 3))
    vg = np.einsum("ci,cbi->cb", v, g)                 # v . grad phi_b
    return scale * 0.25 * vol[:, None, None] * np.broadcast_to(vg[:, None, :], (len(cells), 4, 4))


# int phi^P2_n phi^P1_a dx on a tetrahedron of unit volume (n: 4 vertices, then the 6 UFC edges; a: vertex)
def _p2_p1_mass_unit():
    M = np.zeros((10, 4))
    for n in range(4):
        for a in range(4):
            M[n, a] = 0.0 if n == a else -1.0 / 60.0
    for e, (i, j) in enumerate(P2_EDGE_VERTS):
        for a in range(4):
            M[4 + e, a] = 1.0 / 15.0 if a in (i, j) else 1.0 / 30.0
    return M


def row_velocities(cells, nodal_velocity, cell_dofs=None):
    """V[c,a,:] = (d+1)/|K| int_K u phi_a dx for a P1 (nodal values at the vertices) or P2 (cell_dofs [nc,10] given) velocity
    field u on tetrahedra, or a P1 field on triangles: the weights with which  inner(u, grad(T)) * q * dx
    (ScalarTransportSolver.py:305-311) is integrated EXACTLY for a finite-element velocity; a constant u gives V = u."""
    U = np.asarray(nodal_velocity, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    if cell_dofs is not None:
        Uc = U[np.asarray(cell_dofs, dtype=np.int64)]                    # [nc,10,dim]
        return 4.0 * np.einsum("na,cni->cai", _p2_p1_mass_unit(), Uc)
    Uc = U[cells]                                                        # [nc,d+1,dim]
    nv = cells.shape[1]
    return (Uc.sum(axis=1, keepdims=True) + Uc) / (nv + 1.0)


def assemble_p1_source(coords, cells, f=None, f_nodal=None, cell_markers=None, subdomain_id=None):
    """b_a = int f phi_a dx.  Constant/per-cell f: |detJ|/24 each vertex
    (ScalarTransportSolver.py:213-226 body source S*q*dx[(id)]).  Nodal f
    (interpolated Expression/Function): b_e = M_e f_e."""
    n = len(coords)
    cells64 = np.asarray(cells, dtype=np.int64)
    detJ, _ = p1_geometry(coords, cells)
    b = np.zeros(n)
    mask = np.ones(len(cells), dtype=bool)
    if subdomain_id is not None:
        mask = np.asarray(cell_markers) == subdomain_id
    if f_nodal is not None:
        Me = p1_mass_local(coords, cells, 1.0)
        fe = np.asarray(f_nodal)[cells64]
        be = np.einsum("cab,cb->ca", Me, fe)
    else:
        fv = np.asarray(f, dtype=np.float64)
        w = np.abs(detJ) / 24.0 * (float(fv) if fv.ndim == 0 else fv)
        be = np.repeat(w[:, None], 4, axis=1)
    be = be * mask[:, None]
    np.add.at(b, cells64.ravel(), be.ravel())
    return b


def facet_areas(coords, facets):
    p = np.asarray(coords)[np.asarray(facets, dtype=np.int64)]
    return 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1)


def assemble_p1_facet_load(coords, facets, facet_markers, marker_id, g):
    """b_a += int_{ds(id)} g phi_a ds = g*area/3 per facet vertex
    (flux / Neumann terms, ScalarTransportSolver.py:176-200)."""
    n = len(coords)
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    b = np.zeros(n)
    if len(sel) == 0:
        return b
    ar = facet_areas(coords, np.asarray(facets)[sel])
    np.add.at(b, np.asarray(facets, dtype=np.int64)[sel].ravel(),
              np.repeat(g * ar / 3.0, 3))
    return b


def assemble_p1_facet_mass(coords, facets, facet_markers, marker_id, h):
    """A_ab += int_{ds(id)} h phi_a phi_b ds = h*area/12*(1+delta_ab)
    (HTC/Robin term  htc*(Ta-T)*q*ds, ScalarTransportSolver.py:201-208)."""
    n = len(coords)
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    f = np.asarray(facets, dtype=np.int64)[sel]
    ar = facet_areas(coords, f)
    base = (np.ones((3, 3)) + np.eye(3)) / 12.0
    Me = (h * ar)[:, None, None] * base[None]
    r = np.repeat(f, 3, axis=1).ravel()
    c = np.tile(f, (1, 3)).ravel()
    A = sp.coo_matrix((Me.ravel(), (r, c)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    return A


# --------------------------------------------------------------------------
# linear elasticity, vector P1, node-interleaved dofs (LinearElasticitySolver.py:62-69, 215)
# --------------------------------------------------------------------------

def lame(E, nu):
    mu = E / (2.0 * (1.0 + nu))
    lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    return mu, lmbda


def p1_elasticity_local(coords, cells, E, nu):
    """Ke[(a,i),(b,j)] = vol * ( lmbda g_ai g_bj + mu g_aj g_bi + mu delta_ij g_a.g_b )
    from inner(sigma(u), grad(v)) with sigma = 2 mu sym(grad u) + lmbda div u I."""
    mu, lmbda = lame(E, nu)
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    gg = np.einsum("cak,cbk->cab", g, g)
    Ke = (lmbda * np.einsum("cai,cbj->caibj", g, g)
          + mu * np.einsum("caj,cbi->caibj", g, g)
          + mu * np.einsum("cab,ij->caibj", gg, np.eye(3)))
    Ke = Ke * vol[:, None, None, None, None]
    return Ke.reshape(len(cells), 12, 12)


def assemble_p1_elasticity(coords, cells, E, nu):
    n = len(coords)
    Ke = p1_elasticity_local(coords, cells, E, nu)
    cells64 = np.asarray(cells, dtype=np.int64)
    dofs = (cells64[:, :, None] * 3 + np.arange(3)[None, None, :]).reshape(len(cells), 12)
    r = np.repeat(dofs, 12, axis=1).ravel()
    c = np.tile(dofs, (1, 12)).ravel()
    A = sp.coo_matrix((Ke.ravel(), (r, c)), shape=(3 * n, 3 * n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def assemble_p1_vector_source(coords, cells, f):
    """b_(a,i) = int f_i phi_a dx for constant vector f (body force,
    LinearElasticitySolver.py:227-228)."""
    n = len(coords)
    detJ, _ = p1_geometry(coords, cells)
    w = np.abs(detJ) / 24.0
    nodal = np.zeros(n)
    np.add.at(nodal, np.asarray(cells, dtype=np.int64).ravel(), np.repeat(w, 4))
    return (nodal[:, None] * np.asarray(f, dtype=np.float64)[None, :]).ravel()


def rigid_body_modes(coords):
    """The 6 vectors of SolverBase.build_nullspace (SolverBase.py:674-706),
    before orthonormalisation, node-interleaved."""
    n = len(coords)
    x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
    ns = np.zeros((6, n, 3))
    ns[0, :, 0] = 1.0
    ns[1, :, 1] = 1.0
    ns[2, :, 2] = 1.0
    ns[3, :, 0] = -y
    ns[3, :, 1] = x
    ns[4, :, 0] = z
    ns[4, :, 2] = -x
    ns[5, :, 2] = y
    ns[5, :, 1] = -z
    return ns.reshape(6, 3 * n)


# --------------------------------------------------------------------------
# Dirichlet  (DirichletBC.apply / assemble_system, SolverBase.py:598-602, 608-612, 644)
# --------------------------------------------------------------------------

def apply_dirichlet(A, b, dofs, vals, symmetric=True):
    """symmetric=False: rows -> identity, b[i]=g (LinearVariationalSolver).
    symmetric=True : additionally b -= A[:,i] g and columns zeroed
    (assemble_system).  The CSR pattern is preserved (explicit zeros kept)."""
    A = A.tocsr().copy()
    b = np.array(b, dtype=np.float64, copy=True)
    n = A.shape[0]
    dofs = np.asarray(dofs, dtype=np.int64)
    vals = np.broadcast_to(np.asarray(vals, dtype=np.float64), dofs.shape)
    flag = np.zeros(n, dtype=bool)
    g = np.zeros(n)
    flag[dofs] = True
    g[dofs] = vals  # later entries win on duplicates, as later BCs do in DOLFIN
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    cols = A.indices
    data = A.data
    if symmetric:
        colmask = flag[cols] & ~flag[rows]
        np.subtract.at(b, rows[colmask], data[colmask] * g[cols[colmask]])
        data[colmask] = 0.0
    rowmask = flag[rows]
    data[rowmask] = 0.0
    data[rowmask & (rows == cols)] = 1.0
    b[flag] = g[flag]
    return A, b


# --------------------------------------------------------------------------
# Krylov  (PETSc KSPCG + PCJACOBI behind SolverBase.py:663-670)
# --------------------------------------------------------------------------

def pcg_jacobi(A, b, rtol=1e-8, maxit=10000, x0=None):
    """Textbook Jacobi-PCG from x0=0, stops when ||r||_2 <= rtol ||b||_2
    (unpreconditioned norm: BASELINE.json 'CG solve to 1e-8').
    Returns (x, iterations, residual_history)."""
    n = A.shape[0]
    dinv = 1.0 / A.diagonal()
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    r = b - A @ x
    bb = float(b @ b)
    thresh = rtol * rtol * bb
    hist = [float(r @ r)]
    if hist[0] <= thresh:
        return x, 0, hist
    z = dinv * r
    p = z.copy()
    rz = float(r @ z)
    it = 0
    while it < maxit:
        q = A @ p
        alpha = rz / float(p @ q)
        x += alpha * p
        r -= alpha * q
        it += 1
        rr = float(r @ r)
        hist.append(rr)
        if rr <= thresh:
            break
        z = dinv * r
        rz_new = float(r @ z)
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it, hist


def pcg_jacobi_single_reduction(A, b, rtol=1e-8, maxit=10000):
    """Chronopoulos-Gear single-reduction Jacobi-PCG (PETSc
    KSPCGUseSingleReduction): the recurrence the HIP solver runs.

    iteration k:  w = A z_k ; gamma=r.z, delta=w.z, rho=r.r (one reduction)
                  stop if rho <= rtol^2 b.b
                  beta=gamma/gamma_old, alpha=gamma/(delta-beta*gamma/alpha_old)
                  p=z+beta p ; s=w+beta s ; x+=alpha p ; r-=alpha s ; z=D^-1 r
    """
    n = A.shape[0]
    dinv = 1.0 / A.diagonal()
    x = np.zeros(n)
    r = np.array(b, dtype=np.float64, copy=True)
    z = dinv * r
    p = np.zeros(n)
    s = np.zeros(n)
    bb = float(b @ b)
    thresh = rtol * rtol * bb
    gamma_old = 1.0
    alpha_old = 1.0
    hist = []
    it = 0
    while True:
        w = A @ z
        gamma = float(r @ z)
        delta = float(w @ z)
        rho = float(r @ r)
        hist.append(rho)
        if rho <= thresh or it >= maxit:
            break
        if it == 0:
            beta = 0.0
            alpha = gamma / delta
        else:
            beta = gamma / gamma_old
            alpha = gamma / (delta - beta * gamma / alpha_old)
        p = z + beta * p
        s = w + beta * s
        x += alpha * p
        r -= alpha * s
        z = dinv * r
        gamma_old = gamma
        alpha_old = alpha
        it += 1
    return x, it, hist


def pcg_jacobi_pipelined(A, b, rtol=1e-8, maxit=10000):
    """The pipelined recurrence of fs_krylov.hip (k_pcg_update; Ghysels & Vanroose 2014) on the symmetrically scaled
    system Ah = D^-1/2 A D^-1/2 - what fs_krylov_opts.pipelined = 1 runs.  PETSc's counterpart is KSPPIPECG
    (the reference reaches PETSc through SolverBase.py:663-670).

    iteration i:  gamma = r.r, delta = w.r, rho = sum d r^2 (= the unscaled ||r||^2) of r_i, w_i = Ah r_i - reduced WHILE
                  n = Ah w runs;  stop if rho <= rtol^2 b.b
                  beta = gamma/gamma_old, alpha = gamma/(delta - beta*gamma/alpha_old)
                  z = n + beta z ; s = w + beta s ; p = r + beta p ; x += alpha p ; r -= alpha s ; w -= alpha z
    Returns (x, iterations, history of rho)."""
    d = A.diagonal()
    sc = 1.0 / np.sqrt(d)
    Ah = sp.diags(sc) @ A @ sp.diags(sc)
    n = A.shape[0]
    bh = sc * np.asarray(b, dtype=np.float64)
    x = np.zeros(n)
    r = bh.copy()
    w = Ah @ r
    z = np.zeros(n)
    s = np.zeros(n)
    p = np.zeros(n)
    thresh = rtol * rtol * float(b @ b)
    gamma_old = alpha_old = 1.0
    hist = []
    it = 0
    while True:
        gamma, delta, rho = float(r @ r), float(w @ r), float(d @ (r * r))
        hist.append(rho)
        if rho <= thresh or it >= maxit:
            break
        nv = Ah @ w
        beta = 0.0 if it == 0 else gamma / gamma_old
        alpha = gamma / delta if it == 0 else gamma / (delta - beta * gamma / alpha_old)
        z = nv + beta * z
        s = w + beta * s
        p = r + beta * p
        x += alpha * p
        r -= alpha * s
        w -= alpha * z
        gamma_old, alpha_old = gamma, alpha
        it += 1
    return sc * x, it, np.array(hist)


def solve_direct(A, b):
    """Sparse LU: the reference's *default* linear solve
    (LinearVariationalSolver linear_solver='default')."""
    return spla.spsolve(A.tocsc(), b)


# --------------------------------------------------------------------------
# whole-path helpers used by tests / bench cpu baseline
# --------------------------------------------------------------------------

def heat_box_problem(n, k=20.0, t_lo=350.0, t_hi=300.0, axis=2, dims=None, p1=None):
    """Config-2 family: unit cube (or box) P1, T=t_lo on face axis=0, t_hi on
    the opposite face, natural elsewhere, f=0.  Returns dict with A,b (BC
    applied symmetrically), dofs, exact solution."""
    if dims is None:
        dims = (n, n, n)
    if p1 is None:
        p1 = (1.0, 1.0, 1.0)
    coords, cells = box_mesh((0.0, 0.0, 0.0), p1, *dims)
    A = assemble_p1_scalar(coords, cells, k)
    b = np.zeros(len(coords))
    lo = np.nonzero(coords[:, axis] == 0.0)[0]
    hi = np.nonzero(coords[:, axis] == p1[axis])[0]
    dofs = np.concatenate([lo, hi]).astype(np.int32)
    vals = np.concatenate([np.full(len(lo), t_lo), np.full(len(hi), t_hi)])
    Abc, bbc = apply_dirichlet(A, b, dofs, vals, symmetric=True)
    exact = t_lo + (t_hi - t_lo) * coords[:, axis] / p1[axis]
    return dict(coords=coords, cells=cells, A0=A, A=Abc, b=bbc, dofs=dofs, vals=vals, exact=exact)


# --------------------------------------------------------------------------
# P2 (quadratic Lagrange) tetrahedra: dofs = vertices, then edges (lexicographic edge numbering);
# local order = 4 vertices, then UFC edges e0=(v2,v3) e1=(v1,v3) e2=(v1,v2) e3=(v0,v3) e4=(v0,v2) e5=(v0,v1)
# (SURVEY Appendix C3).  FFC integrates grad.grad of P2 with the 4-point degree-2 rule (Appendix D-5).
# --------------------------------------------------------------------------
P2_EDGE_VERTS = ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
_QA = (5.0 - np.sqrt(5.0)) / 20.0
_QB = (5.0 + 3.0 * np.sqrt(5.0)) / 20.0
P2_QUAD_POINTS = np.array([[_QB, _QA, _QA, _QA], [_QA, _QB, _QA, _QA], [_QA, _QA, _QB, _QA], [_QA, _QA, _QA, _QB]])


P2_MAX_EDGE_CLASSES = 16


def p2_edge_order(edges):
    """Order of the edge nodes of a P2 space.  DOLFIN's dof numbering is not reproducible (Appendix D-7), so
    the build fixes its own: lexicographic by (v0, v1) in general; when the mesh has at most 16 distinct
    index differences v1 - v0 (structured meshes: 7) edges are grouped by that difference first, which makes
    consecutive rows share their column offsets.  Returns the permutation of the lexicographic edge list."""
    e = np.asarray(edges, dtype=np.int64)
    delta = e[:, 1] - e[:, 0]
    if len(np.unique(delta)) <= P2_MAX_EDGE_CLASSES:
        return np.lexsort((e[:, 0], delta))
    return np.arange(len(e))


def p2_cell_dofs(n_vertices, cells):
    """[nc,10] global dofs of every cell and the edge table [ne,2] in edge-node order."""
    edges, cell_edges = edge_numbering(cells)
    order = p2_edge_order(edges)
    rank = np.empty(len(edges), dtype=np.int64)
    rank[order] = np.arange(len(edges))
    cd = np.concatenate([np.asarray(cells, dtype=np.int64), n_vertices + rank[cell_edges.astype(np.int64)]], axis=1)
    return cd.astype(np.int32), edges[order]


def p2_dof_coordinates(coords, edges):
    return np.concatenate([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])], axis=0)


def p2_basis_gradients(g, lam):
    """grad of the 10 P2 basis functions at barycentric point lam[4]; g[nc,4,3] -> [nc,10,3]."""
    out = np.zeros((g.shape[0], 10, 3))
    for i in range(4):
        out[:, i, :] = (4.0 * lam[i] - 1.0) * g[:, i, :]
    for e, (i, j) in enumerate(P2_EDGE_VERTS):
        out[:, 4 + e, :] = 4.0 * (lam[i] * g[:, j, :] + lam[j] * g[:, i, :])
    return out


def p2_advection_local(coords, cells, velocity, scale=1.0):
    """Ce[a,b] = scale * int phi_a (v . grad phi_b) dx for the P2 basis and a constant or per-cell velocity
    (inner(velocity, grad(T))*Tq*capacity*dx, ScalarTransportSolver.py:305-311 with fe_degree 2): cubic integrand, integrated
    with the 14-point degree-5 rule of the Navier-Stokes oracle (the device uses Keast's 5-point degree-3 rule)."""
    from oracle import ns_oracle as nso
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    v = np.asarray(velocity, dtype=np.float64)
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(vol), 3))
    pts, wq = nso.tet_quadrature(5)
    Ce = np.zeros((len(vol), 10, 10))
    for lam, w in zip(pts, wq):
        phi, dphi = nso.p2_shape(lam)                       # [10], [10,4] (derivatives w.r.t. the barycentric coordinates)
        gphi = np.einsum("ak,cki->cai", dphi, g)            # physical gradients
        Ce += (scale * w * vol)[:, None, None] * np.einsum("a,cb->cab", phi, np.einsum("ci,cbi->cb", v, gphi))
    return Ce


def p2_stiffness_local_qp(coords, cells, k_of_point):
    """Ke[a,b] = int k(x) grad phi_a . grad phi_b dx with k given by a function of the barycentric point -> [nc] values
    (a conductivity depending on the P2 temperature iterate, evaluated at the quadrature points): 14-point degree-5 rule."""
    from oracle import ns_oracle as nso
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    pts, wq = nso.tet_quadrature(5)
    Ke = np.zeros((len(vol), 10, 10))
    for lam, w in zip(pts, wq):
        _, dphi = nso.p2_shape(lam)
        gphi = np.einsum("ak,cki->cai", dphi, g)
        Ke += (w * vol * k_of_point(np.asarray(lam)))[:, None, None] * np.einsum("cai,cbi->cab", gphi, gphi)
    return Ke


def p2_stiffness_local(coords, cells, k=1.0):
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    Ke = np.zeros((len(cells), 10, 10))
    for lam in P2_QUAD_POINTS:
        gp = p2_basis_gradients(g, lam)
        Ke += 0.25 * np.einsum("cai,cbi->cab", gp, gp)
    kk = np.asarray(k, dtype=np.float64)
    return Ke * (vol * (float(kk) if kk.ndim == 0 else kk))[:, None, None]


def p2_mass_reference():
    """Exact P2 mass matrix of a tetrahedron of unit volume (x V for any affine image): 1/420 *
    {6 same vertex, 1 vertex-vertex, 32 same edge, 16 edges sharing a vertex, 8 opposite edges,
    -4 vertex on the edge, -6 vertex off the edge}."""
    M = np.zeros((10, 10))
    for a in range(4):
        for b in range(4):
            M[a, b] = 6.0 if a == b else 1.0
    for e, (i, j) in enumerate(P2_EDGE_VERTS):
        for f, (k, l) in enumerate(P2_EDGE_VERTS):
            shared = len({i, j} & {k, l})
            M[4 + e, 4 + f] = 32.0 if shared == 2 else (16.0 if shared == 1 else 8.0)
        for a in range(4):
            M[a, 4 + e] = M[4 + e, a] = -4.0 if a in (i, j) else -6.0
    return M / 420.0


def p2_mass_local(coords, cells, c=1.0):
    detJ, _ = p1_geometry(coords, cells)
    cc = np.asarray(c, dtype=np.float64)
    w = np.abs(detJ) / 6.0 * (float(cc) if cc.ndim == 0 else cc)
    return w[:, None, None] * p2_mass_reference()[None]


def p2_source_local(coords, cells, f=1.0):
    """int f phi_a dx for constant f: -V/20 per vertex dof, V/5 per edge dof."""
    detJ, _ = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    w = np.concatenate([np.full(4, -1.0 / 20.0), np.full(6, 1.0 / 5.0)])
    ff = np.asarray(f, dtype=np.float64)
    return (vol * (float(ff) if ff.ndim == 0 else ff))[:, None] * w[None, :]


def p2_elasticity_local(coords, cells, E, nu):
    """30x30 element matrix of inner(sigma(u), grad(v)) on P2 tetrahedra (the reference's elasticity example runs
    VectorFunctionSpace(mesh, 'CG', 2): examples/test_linear_elasticity.py:105-106; form LinearElasticitySolver.py:62-69, 215).
    Ke[(a,i),(b,j)] = int lmbda d_i phi_a d_j phi_b + mu d_j phi_a d_i phi_b + mu delta_ij grad phi_a . grad phi_b dx;
    the integrand is quadratic: the 4-point rule FFC picks (Appendix C3, D-5) is exact.  Dof order per cell:
    node-major (a = 4 vertices then 6 UFC edges), component-minor, as vector spaces interleave (Appendix D-7)."""
    mu, lmbda = lame(E, nu)
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    Ke = np.zeros((len(cells), 10, 3, 10, 3))
    eye = np.eye(3)
    for lam in P2_QUAD_POINTS:
        gp = p2_basis_gradients(g, lam)
        gg = np.einsum("cak,cbk->cab", gp, gp)
        Ke += 0.25 * (lmbda * np.einsum("cai,cbj->caibj", gp, gp) + mu * np.einsum("caj,cbi->caibj", gp, gp)
                      + mu * np.einsum("cab,ij->caibj", gg, eye))
    return (Ke * vol[:, None, None, None, None]).reshape(len(cells), 30, 30)


def p2_vector_cell_dofs(cell_dofs):
    """[nc,30] dofs of a 3-vector P2 space from the node table [nc,10]: dof = node*3 + component."""
    cd = np.asarray(cell_dofs, dtype=np.int64)
    return (cd[:, :, None] * 3 + np.arange(3)[None, None, :]).reshape(len(cd), 30)


def assemble_p2_elasticity(coords, cells, E, nu):
    cd, edges = p2_cell_dofs(len(coords), cells)
    n_nodes = len(coords) + len(edges)
    return assemble_generic(3 * n_nodes, p2_vector_cell_dofs(cd), p2_elasticity_local(coords, cells, E, nu)), cd, edges


def assemble_p2_vector_source(coords, cells, f):
    """b_(a,i) = int f_i phi_a dx for a constant vector f on the 3-vector P2 space (-V/20 per vertex, V/5 per edge node)."""
    cd, edges = p2_cell_dofs(len(coords), cells)
    nodal = assemble_generic_vector(len(coords) + len(edges), cd, p2_source_local(coords, cells, 1.0))
    return (nodal[:, None] * np.asarray(f, dtype=np.float64)[None, :]).ravel()


def p2_div_load_local(coords, cells, c_vertex=None, c_const=None):
    """be[(a,i)] = int c d_i phi_a dx (the thermal-stress load E alpha (T - T0)/(1-2nu) I : grad v,
    LinearElasticitySolver.py:78-85, 231-238) with c constant or P1 (given by its vertex values [nv]): the integrand is
    at most quadratic, 4-point rule exact."""
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    ce = np.asarray(cells, dtype=np.int64)
    be = np.zeros((len(cells), 10, 3))
    for lam in P2_QUAD_POINTS:
        gp = p2_basis_gradients(g, lam)
        cq = np.full(len(cells), float(c_const)) if c_vertex is None else np.asarray(c_vertex)[ce] @ np.asarray(lam)
        be += 0.25 * cq[:, None, None] * gp
    return (be * vol[:, None, None]).reshape(len(cells), 30)


def assemble_p2_facet_vector_load(coords, edges, facets, facet_markers, marker_id, g):
    """b_(a,i) = int g_i phi_a ds(marker_id), constant vector g, 3-vector P2 space: g_i * area / 3 on the three edge
    nodes of every marked facet, nothing on its vertices (LinearElasticitySolver.py:165-196 tractions)."""
    nv = len(coords)
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    f = np.asarray(facets, dtype=np.int64)[sel]
    area = facet_areas(coords, f)
    ekey = np.asarray(edges, dtype=np.int64)
    ekey = ekey[:, 0] * nv + ekey[:, 1]
    sorter = np.argsort(ekey)
    b = np.zeros((nv + len(edges), 3))
    for i, j in ((0, 1), (0, 2), (1, 2)):
        eid = sorter[np.searchsorted(ekey[sorter], f[:, i] * nv + f[:, j])]
        np.add.at(b, nv + eid, (area / 3.0)[:, None] * np.asarray(g, dtype=np.float64)[None, :])
    return b.ravel()


def von_mises_at(G, E, nu):
    """sqrt(3/2 s:s), s = dev(2 mu sym(G) + lmbda tr(G) I), for displacement gradients G[..., 3, 3]
    (LinearElasticitySolver.py:62-76)."""
    mu, lmbda = lame(E, nu)
    tr = np.trace(G, axis1=-2, axis2=-1)
    sg = mu * (G + np.swapaxes(G, -1, -2)) + lmbda * tr[..., None, None] * np.eye(3)
    dev = sg - np.trace(sg, axis1=-2, axis2=-1)[..., None, None] / 3.0 * np.eye(3)
    return np.sqrt(1.5 * np.einsum("...ij,...ij->...", dev, dev))


def von_mises_projection(coords, cells, u, E, nu, degree=1, cell_dofs=None):
    """project(von_Mises(u), FunctionSpace(mesh, 'P', 1)) (LinearElasticitySolver.py:71-76): solve M w = b with the
    consistent P1 mass matrix, b_a = int vm lambda_a dx.  u: [n_nodes, 3].  P1 displacement: vm is constant per cell;
    P2 (cell_dofs [nc,10]): 4-point degree-2 rule.  Returns (w, b)."""
    ce = np.asarray(cells, dtype=np.int64)
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    be = np.zeros((len(ce), 4))
    if degree == 1:
        G = np.einsum("cni,cnk->cik", np.asarray(u)[ce], g)
        be[:] = (0.25 * vol * von_mises_at(G, E, nu))[:, None]
    else:
        un = np.asarray(u)[np.asarray(cell_dofs, dtype=np.int64)]          # [nc,10,3]
        for lam in P2_QUAD_POINTS:
            gp = p2_basis_gradients(g, lam)
            G = np.einsum("cni,cnk->cik", un, gp)
            be += 0.25 * (vol * von_mises_at(G, E, nu))[:, None] * np.asarray(lam)[None, :]
    b = assemble_generic_vector(len(coords), ce, be)
    M = assemble_matrix(len(coords), cells, p1_mass_local(coords, cells, 1.0))
    return solve_direct(M, b), b


def assemble_generic(n_dofs, cell_dofs, Ke):
    cd = np.asarray(cell_dofs, dtype=np.int64)
    nd = cd.shape[1]
    r = np.repeat(cd, nd, axis=1).ravel()
    c = np.tile(cd, (1, nd)).ravel()
    A = sp.coo_matrix((Ke.ravel(), (r, c)), shape=(n_dofs, n_dofs)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def assemble_generic_vector(n_dofs, cell_dofs, be):
    b = np.zeros(n_dofs)
    np.add.at(b, np.asarray(cell_dofs, dtype=np.int64).ravel(), be.ravel())
    return b


def p2_facet_dofs(n_vertices, edges, facets, facet_markers, marker_id):
    """Topological DirichletBC on P2: vertices and edges in the closure of the marked facets (Appendix D-3)."""
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    f = np.asarray(facets, dtype=np.int64)[sel]
    verts = np.unique(f.ravel())
    nv = int(n_vertices)
    ekey = np.asarray(edges, dtype=np.int64)
    ekey = ekey[:, 0] * nv + ekey[:, 1]
    sorter = np.argsort(ekey)
    fe = np.concatenate([f[:, [0, 1]], f[:, [0, 2]], f[:, [1, 2]]], axis=0)   # facets hold ascending vertices
    fkey = np.unique(fe[:, 0] * nv + fe[:, 1])
    eid = sorter[np.searchsorted(ekey[sorter], fkey)]
    assert np.array_equal(ekey[eid], fkey)
    return np.concatenate([verts, np.sort(nv + eid)]).astype(np.int32)


# ---- SUPG ("SPUG", ScalarTransportSolver.py:259-270): test function q + tau (v . grad q) --------------------
def tet_circumradius(coords, cells):
    """R of every tetrahedron by solving |x - p_i|^2 = R^2 for the circumcentre (independent of the edge-product
    formula the device uses)."""
    c = np.asarray(coords, dtype=np.float64)[np.asarray(cells, dtype=np.int64)]
    A = 2.0 * (c[:, 1:] - c[:, :1])                                    # [nc,3,3]
    rhs = (c[:, 1:] ** 2).sum(axis=2) - (c[:, :1] ** 2).sum(axis=2)    # [nc,3]
    centre = np.linalg.solve(A, rhs[:, :, None])[:, :, 0]
    return np.linalg.norm(centre - c[:, 0], axis=1)


def supg_weights(coords, cells, velocity, pe):
    """w[c,a] = tau_c (v_c . grad phi_a), tau = 0.5 h / (4/(Pe h) + 2 |v|), h = 2 * Circumradius (:262-266)."""
    detJ, g = p1_geometry(coords, cells)
    v = np.asarray(velocity, dtype=np.float64)
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(cells), 3))
    h = 2.0 * tet_circumradius(coords, cells)
    tau = 0.5 * h / (4.0 / (pe * h) + 2.0 * np.linalg.norm(v, axis=1))
    return tau[:, None] * np.einsum("ci,cai->ca", v, g), v


def p1_supg_local(coords, cells, velocity, pe, advection_scale=0.0, mass_coef=0.0):
    """Extra element matrix of the SUPG test-function part: w_a * (scale (v . grad phi_b) vol + mass vol/4)."""
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    w, v = supg_weights(coords, cells, velocity, pe)
    vg = np.einsum("ci,cbi->cb", v, g)
    m = np.broadcast_to(np.asarray(mass_coef, dtype=np.float64), (len(cells),))
    return w[:, :, None] * (advection_scale * vg[:, None, :] * vol[:, None, None] + (0.25 * m * vol)[:, None, None])


def assemble_p1_supg_source(coords, cells, velocity, pe, f=None, f_nodal=None):
    """int f tau (v . grad phi_a) dx for a constant / per-cell source f, or for the P1 interpolant of a source Function given by its
    vertex values f_nodal (ScalarTransportSolver.py:213-226 with Tq of :259-276) - that one by the 4-point degree-2 rule."""
    detJ, g = p1_geometry(coords, cells)
    vol = np.abs(detJ) / 6.0
    w, _ = supg_weights(coords, cells, velocity, pe)
    cells = np.asarray(cells, dtype=np.int64)
    if f_nodal is not None:
        fe = np.asarray(f_nodal, dtype=np.float64)[cells]                                          # [nc, 4]
        a_, b_ = 0.5854101966249685, 0.1381966011250105
        lam = np.full((4, 4), b_) + (a_ - b_) * np.eye(4)
        mean = sum(0.25 * (fe @ lam[q]) for q in range(4))                                         # int S_h dx / |K|
        ff = mean
    else:
        ff = np.broadcast_to(np.asarray(f, dtype=np.float64), (len(cells),))
    b = np.zeros(len(coords))
    np.add.at(b, cells.ravel(), (w * (ff * vol)[:, None]).ravel())
    return b


def supg_facet_terms(coords, cells, facet_cells, velocity, pe, g=None, h=None):
    """(dA, db) of the SUPG part of ds terms over boundary facets given as (cell, opposite local vertex):
    db[a] += g area w_a ;  dA[a, b on facet] += h (area/3) w_a   for the four vertices a of the cell."""
    import scipy.sparse as sp
    coords = np.asarray(coords, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    w, _ = supg_weights(coords, cells, velocity, pe)
    n = len(coords)
    db = np.zeros(n)
    rows, cols, vals = [], [], []
    opp = ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2))
    for k, (c, o) in enumerate(np.asarray(facet_cells, dtype=np.int64)):
        tri = cells[c, list(opp[o])]
        p = coords[tri]
        area = 0.5 * np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
        gk = 0.0 if g is None else float(np.broadcast_to(g, (len(facet_cells),))[k])
        hk = 0.0 if h is None else float(np.broadcast_to(h, (len(facet_cells),))[k])
        for a in range(4):
            db[cells[c, a]] += gk * area * w[c, a]
            for bnode in tri:
                rows.append(cells[c, a]); cols.append(bnode); vals.append(hk * area / 3.0 * w[c, a])
    dA = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    return dA, db


# ---- 2-D P1 (triangles): the reference's runnable examples are 2-D (examples/test_heat_transfer.py:34,
# test_electrostatics.py:35 use UnitSquareMesh(40, 40)) ---------------------------------------------------
def rectangle_mesh(p0, p1, nx, ny):
    """RectangleMesh(Point(p0), Point(p1), nx, ny) / UnitSquareMesh with the default "right" diagonal: vertices
    x-fastest, per square (iy outer, ix inner) the cells (v0, v1, v3) and (v0, v2, v3) [upstream DOLFIN RectangleMesh]."""
    xs = np.array([p0[0] + (i * (p1[0] - p0[0])) / nx for i in range(nx + 1)])
    ys = np.array([p0[1] + (j * (p1[1] - p0[1])) / ny for j in range(ny + 1)])
    coords = np.stack([np.tile(xs, ny + 1), np.repeat(ys, nx + 1)], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1, v2 = v0 + 1, v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.stack([np.stack([v0, v1, v3], axis=1), np.stack([v0, v2, v3], axis=1)], axis=1).reshape(-1, 3)
    return coords, np.sort(cells, axis=1).astype(np.int32)


def tri_edge_numbering(cells):
    """Facets of a triangle mesh = edges, ranked lexicographically; local facet i is opposite local vertex i.
    Returns (edges [ne,2], cell_facets [nc,3], count [ne])."""
    cells = np.asarray(cells, dtype=np.int64)
    opp = [(1, 2), (0, 2), (0, 1)]
    ed = np.sort(np.stack([cells[:, list(o)] for o in opp], axis=1).reshape(-1, 2), axis=1)
    uniq, inv, cnt = np.unique(ed, axis=0, return_inverse=True, return_counts=True)
    return uniq.astype(np.int32), inv.reshape(len(cells), 3).astype(np.int32), cnt.astype(np.int32)


def tri_geometry(coords, cells):
    """(area [nc], grad phi [nc,3,2]) of P1 on triangles."""
    c = np.asarray(coords, dtype=np.float64)[np.asarray(cells, dtype=np.int64)]
    e1, e2 = c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    g = np.zeros((len(c), 3, 2))
    g[:, 1] = np.stack([e2[:, 1], -e2[:, 0]], axis=1) / det[:, None]
    g[:, 2] = np.stack([-e1[:, 1], e1[:, 0]], axis=1) / det[:, None]
    g[:, 0] = -(g[:, 1] + g[:, 2])
    return 0.5 * np.abs(det), g


def tri_stiffness_local(coords, cells, k=1.0):
    area, g = tri_geometry(coords, cells)
    k = np.asarray(k, dtype=np.float64)
    if k.ndim == 3:                                     # one 2x2 tensor per cell (a tensor Expression of degree 0)
        return area[:, None, None] * np.einsum("cai,cij,cbj->cab", g, k[:, :2, :2], g)
    if k.ndim == 2 and k.shape in ((2, 2), (3, 3)):
        return area[:, None, None] * np.einsum("cai,ij,cbj->cab", g, k[:2, :2], g)
    kk = np.broadcast_to(k, (len(area),))
    return (kk * area)[:, None, None] * np.einsum("cai,cbi->cab", g, g)


def tri_mass_local(coords, cells, c=1.0):
    area, _ = tri_geometry(coords, cells)
    cc = np.broadcast_to(np.asarray(c, dtype=np.float64), (len(area),))
    return (cc * area / 12.0)[:, None, None] * (np.ones((3, 3)) + np.eye(3))[None]


def tri_advection_local(coords, cells, velocity, scale=1.0):
    area, g = tri_geometry(coords, cells)
    v = np.asarray(velocity, dtype=np.float64)
    if v.ndim == 3:                                    # row velocities [nc,3,2] (row_velocities)
        return scale * (area / 3.0)[:, None, None] * np.einsum("cai,cbi->cab", v[:, :, :2], g)
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(area), 2))
    vg = np.einsum("ci,cbi->cb", v, g)
    return scale * (area / 3.0)[:, None, None] * np.broadcast_to(vg[:, None, :], (len(area), 3, 3))


def assemble_tri_source(coords, cells, f=1.0, f_nodal=None):
    area, _ = tri_geometry(coords, cells)
    cells = np.asarray(cells, dtype=np.int64)
    b = np.zeros(len(coords))
    if f_nodal is not None:
        fe = np.asarray(f_nodal, dtype=np.float64)[cells]
        be = (area / 12.0)[:, None] * (fe.sum(axis=1, keepdims=True) + fe)
    else:
        be = np.broadcast_to((np.broadcast_to(np.asarray(f, dtype=np.float64), area.shape) * area / 3.0)[:, None], (len(area), 3))
    np.add.at(b, cells.ravel(), be.ravel())
    return b


def assemble_edge_load(coords, edges, markers, marker_id, g):
    """int g phi_a ds over the marked boundary edges: g * length / 2 on both end points."""
    e = np.asarray(edges, dtype=np.int64)[np.asarray(markers) == marker_id]
    length = np.linalg.norm(np.asarray(coords)[e[:, 1]] - np.asarray(coords)[e[:, 0]], axis=1)
    b = np.zeros(len(coords))
    np.add.at(b, e.ravel(), np.repeat(np.broadcast_to(g, length.shape) * length / 2.0, 2))
    return b


def assemble_edge_mass(coords, edges, markers, marker_id, h):
    """int h T q ds: h * length/6 * [[2,1],[1,2]] per marked edge."""
    import scipy.sparse as sp
    e = np.asarray(edges, dtype=np.int64)[np.asarray(markers) == marker_id]
    length = np.linalg.norm(np.asarray(coords)[e[:, 1]] - np.asarray(coords)[e[:, 0]], axis=1)
    w = np.broadcast_to(h, length.shape) * length / 6.0
    rows = np.concatenate([e[:, 0], e[:, 0], e[:, 1], e[:, 1]])
    cols = np.concatenate([e[:, 0], e[:, 1], e[:, 0], e[:, 1]])
    vals = np.concatenate([2 * w, w, w, 2 * w])
    n = len(coords)
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


def mark_edges(coords, cells, inside, marker_id, markers=None, eps=3e-16):
    """SubDomain.mark on the facets (edges) of a triangle mesh: every vertex and the mid-point inside."""
    edges, _, cnt = tri_edge_numbering(cells)
    if markers is None:
        markers = np.zeros(len(edges), dtype=np.int64)
    co = np.asarray(coords, dtype=np.float64)
    for i, (a, b) in enumerate(edges):
        ob = cnt[i] == 1
        pts = (co[a], co[b], 0.5 * (co[a] + co[b]))
        if all(inside(p, ob) for p in pts):
            markers[i] = marker_id
    return markers


# ---- interior penalty (ScalarTransportSolver.py:312-315) --------------------------------------------------------
def assemble_interior_penalty(coords, cells, coefficient):
    """+ coefficient * avg(h)^2 * jump(grad T, n) * jump(grad q, n) dS over the interior facets, h = 2 * Circumradius,
    jump(w, n) = w+ . n+ + w- . n-.  Written as UFL reads: per facet, per side, grad phi_i . n of the side; the facet
    normal comes from the facet's own vertices (cross product, oriented away from the cell's opposite vertex) and
    the facet area from the same cross product - independent of the barycentric-gradient shortcut of the device."""
    import scipy.sparse as sp
    co = np.asarray(coords, dtype=np.float64)
    ce = np.asarray(cells, dtype=np.int64)
    n = len(co)
    _, g = p1_geometry(co, ce)
    h = 2.0 * tet_circumradius(co, ce)
    facets, cell_facets, cnt = facet_numbering(ce)
    sides = {}
    for c in range(len(ce)):
        for i in range(4):
            sides.setdefault(int(cell_facets[c, i]), []).append((c, i))
    rows, cols, vals = [], [], []
    for f, ss in sides.items():
        if len(ss) != 2:
            continue
        tri = co[facets[f].astype(np.int64)]
        nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0])
        area = 0.5 * np.linalg.norm(nrm)
        nrm = nrm / np.linalg.norm(nrm)
        J = {}
        for c, i in ss:
            outward = nrm if np.dot(nrm, tri[0] - co[ce[c, i]]) > 0 else -nrm     # away from the opposite vertex
            for a in range(4):
                J[int(ce[c, a])] = J.get(int(ce[c, a]), 0.0) + float(np.dot(g[c, a], outward))
        w = coefficient * (0.5 * (h[ss[0][0]] + h[ss[1][0]])) ** 2 * area
        nodes = list(J)
        for a in nodes:
            for b in nodes:
                rows.append(a); cols.append(b); vals.append(w * J[a] * J[b])
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


# ---- SUPG on triangles (the same substitutions as on tetrahedra, ScalarTransportSolver.py:259-270) -----------------------
def tri_supg_weights(coords, cells, velocity, pe):
    """w[c,a] = tau_c (v_c . grad phi_a), tau = 0.5 h / (4/(Pe h) + 2 |v|), h = 2 * Circumradius."""
    area, g = tri_geometry(coords, cells)
    v = np.asarray(velocity, dtype=np.float64)[..., :2]
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(area), 2))
    h = 2.0 * tet_circumradius(np.asarray(coords, dtype=np.float64)[:, :2], cells)
    tau = 0.5 * h / (4.0 / (pe * h) + 2.0 * np.linalg.norm(v, axis=1))
    return tau[:, None] * np.einsum("ci,cai->ca", v, g), v


def tri_supg_local(coords, cells, velocity, pe, advection_scale=0.0, mass_coef=0.0):
    area, g = tri_geometry(coords, cells)
    w, v = tri_supg_weights(coords, cells, velocity, pe)
    vg = np.einsum("ci,cbi->cb", v, g)
    m = np.broadcast_to(np.asarray(mass_coef, dtype=np.float64), (len(area),))
    return w[:, :, None] * (advection_scale * vg[:, None, :] * area[:, None, None] + (m * area / 3.0)[:, None, None])


def assemble_tri_supg_source(coords, cells, velocity, pe, f=None, f_nodal=None):
    """2-D counterpart of assemble_p1_supg_source (f_nodal: vertex values of the source Function, 3-point edge-midpoint rule)."""
    area, _ = tri_geometry(coords, cells)
    w, _ = tri_supg_weights(coords, cells, velocity, pe)
    cells = np.asarray(cells, dtype=np.int64)
    if f_nodal is not None:
        fe = np.asarray(f_nodal, dtype=np.float64)[cells]
        lam = 0.5 * (1.0 - np.eye(3))
        ff = sum((fe @ lam[q]) / 3.0 for q in range(3))
    else:
        ff = np.broadcast_to(np.asarray(f, dtype=np.float64), (len(area),))
    b = np.zeros(len(coords))
    np.add.at(b, cells.ravel(), (w * (ff * area)[:, None]).ravel())
    return b


def tri_supg_facet_terms(coords, cells, facet_cells, velocity, pe, g=None, h=None):
    """db[a] += g |E| w_a ;  dA[a, b on the edge] += h (|E|/2) w_a  for the three vertices a of the cell behind each boundary edge."""
    import scipy.sparse as sp
    co = np.asarray(coords, dtype=np.float64)[:, :2]
    ce = np.asarray(cells, dtype=np.int64)
    w, _ = tri_supg_weights(co, ce, velocity, pe)
    n = len(co)
    db = np.zeros(n)
    rows, cols, vals = [], [], []
    opp = ((1, 2), (0, 2), (0, 1))
    for k, (c, o) in enumerate(np.asarray(facet_cells, dtype=np.int64)):
        ed = ce[c, list(opp[o])]
        length = np.linalg.norm(co[ed[1]] - co[ed[0]])
        gk = 0.0 if g is None else float(np.broadcast_to(g, (len(facet_cells),))[k])
        hk = 0.0 if h is None else float(np.broadcast_to(h, (len(facet_cells),))[k])
        for a in range(3):
            db[ce[c, a]] += gk * length * w[c, a]
            for bnode in ed:
                rows.append(ce[c, a]); cols.append(bnode); vals.append(hk * length / 2.0 * w[c, a])
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr(), db


def radiation_facet_loads(coords, facets, T, m, T_amb):
    """[nf, d] vertex loads int_F m (T_amb^4 - T_h^4) lambda_a ds of a P1 field T on boundary triangles / edges, for the
    reference's  m*(pow(T, 4) - pow(T_amb, 4))*Tq*ds  (ScalarTransportSolver.py:186-190): exact, by expanding T_h^4 in the
    barycentric monomials and  int lambda^alpha = |F| (d-1)! alpha! / (|alpha| + d - 1)!  - no quadrature involved."""
    from math import factorial
    from itertools import product
    co = np.asarray(coords, dtype=np.float64)
    fa = np.asarray(facets, dtype=np.int64)
    d = fa.shape[1]
    X = co[fa]
    if d == 3:
        e1, e2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
        c = np.cross(e1, e2) if X.shape[2] == 3 else (e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])[:, None]
        measure = 0.5 * np.linalg.norm(c, axis=1)
    else:
        measure = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    Tv = np.asarray(T, dtype=np.float64)[fa]                                  # [nf,d]
    out = np.zeros((len(fa), d))
    for a in range(d):
        acc = np.zeros(len(fa))
        for alpha in product(range(5), repeat=d):
            if sum(alpha) != 4:
                continue
            multinom = factorial(4)
            term = np.ones(len(fa))
            for k in range(d):
                multinom //= factorial(alpha[k])
                term = term * Tv[:, k] ** alpha[k]
            beta = list(alpha)
            beta[a] += 1
            integ = factorial(d - 1)
            for k in range(d):
                integ *= factorial(beta[k])
            acc += multinom * term * integ / factorial(sum(beta) + d - 1)
        out[:, a] = m * (T_amb ** 4 / d - acc) * measure                     # int lambda_a = |F| / d
    return out


def assemble_tri_interior_penalty(coords, cells, coefficient):
    """The same term on a triangle mesh: interior EDGES, normal = the edge direction turned by 90 degrees (oriented away from
    the cell's opposite vertex), |E| the edge length, h = 2 * Circumradius from the circumcentre."""
    import scipy.sparse as sp
    co = np.asarray(coords, dtype=np.float64)[:, :2]
    ce = np.asarray(cells, dtype=np.int64)
    n = len(co)
    _, g = tri_geometry(co, ce)
    h = 2.0 * tet_circumradius(co, ce)                   # (the circumcentre solve is dimension-free)
    edges, cell_edges, cnt = tri_edge_numbering(ce)
    sides = {}
    for c in range(len(ce)):
        for i in range(3):
            sides.setdefault(int(cell_edges[c, i]), []).append((c, i))
    rows, cols, vals = [], [], []
    for f, ss in sides.items():
        if len(ss) != 2:
            continue
        e = co[edges[f].astype(np.int64)]
        tvec = e[1] - e[0]
        length = np.linalg.norm(tvec)
        nrm = np.array([tvec[1], -tvec[0]]) / length
        J = {}
        for c, i in ss:
            outward = nrm if np.dot(nrm, e[0] - co[ce[c, i]]) > 0 else -nrm
            for a in range(3):
                J[int(ce[c, a])] = J.get(int(ce[c, a]), 0.0) + float(np.dot(g[c, a], outward))
        w = coefficient * (0.5 * (h[ss[0][0]] + h[ss[1][0]])) ** 2 * length
        nodes = list(J)
        for a in nodes:
            for b in nodes:
                rows.append(a); cols.append(b); vals.append(w * J[a] * J[b])
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


# ---- P2 Robin / HTC facet matrix (ScalarTransportSolver.py:201-208 with fe_degree 2) ---------------------------------
def assemble_p2_facet_mass(coords, edges, facets, facet_markers, marker_id, h):
    """int_F h phi_a phi_b ds on the marked facets for the P2 basis: by quadrature (the 6-point degree-4 rule, exact for
    the quartic integrand), not by the closed-form mass matrix the device uses.  Node of edge k = n_vertices + k."""
    import scipy.sparse as sp
    co = np.asarray(coords, dtype=np.float64)
    nv = len(co)
    ed = np.asarray(edges, dtype=np.int64)
    emap = {(int(a), int(b)): nv + k for k, (a, b) in enumerate(ed)}
    q = np.array([[0.108103018168070, 0.445948490915965, 0.445948490915965],
                  [0.445948490915965, 0.108103018168070, 0.445948490915965],
                  [0.445948490915965, 0.445948490915965, 0.108103018168070],
                  [0.816847572980459, 0.091576213509771, 0.091576213509771],
                  [0.091576213509771, 0.816847572980459, 0.091576213509771],
                  [0.091576213509771, 0.091576213509771, 0.816847572980459]])
    w = np.array([0.223381589678011] * 3 + [0.109951743655322] * 3)
    sel = np.nonzero(np.asarray(facet_markers) == marker_id)[0]
    hh = np.broadcast_to(np.asarray(h, dtype=np.float64), (len(sel),))
    rows, cols, vals = [], [], []
    for idx, f in enumerate(sel):
        v = [int(x) for x in facets[f]]
        nodes = v + [emap[tuple(sorted((v[0], v[1])))], emap[tuple(sorted((v[0], v[2])))], emap[tuple(sorted((v[1], v[2])))]]
        area = 0.5 * np.linalg.norm(np.cross(co[v[1]] - co[v[0]], co[v[2]] - co[v[0]]))
        M = np.zeros((6, 6))
        for lam, wq in zip(q, w):
            phi = np.array([lam[0] * (2 * lam[0] - 1), lam[1] * (2 * lam[1] - 1), lam[2] * (2 * lam[2] - 1),
                            4 * lam[0] * lam[1], 4 * lam[0] * lam[2], 4 * lam[1] * lam[2]])
            M += wq * np.outer(phi, phi)
        M *= area * hh[idx]
        for a in range(6):
            for b in range(6):
                rows.append(nodes[a]); cols.append(nodes[b]); vals.append(M[a, b])
    n = nv + len(ed)
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


# ---- 2-D vector P1 (plane-strain elasticity; LinearElasticitySolver.py:62-69 with dimension 2, which the reference
# hands to solve_linear_problem, :247-253) -------------------------------------------------------------------------
def tri_elasticity_local(coords, cells, E, nu):
    """Ke[(a,i),(b,j)] = area * (lmbda g_ai g_bj + mu g_aj g_bi + mu delta_ij g_a.g_b): inner(sigma(u), grad(v)) dx with
    sigma = 2 mu sym(grad u) + lmbda div u I on P1 triangles (constant gradients, exact)."""
    mu, lmbda = lame(E, nu)
    area, g = tri_geometry(coords, cells)
    gg = np.einsum("cak,cbk->cab", g, g)
    Ke = (lmbda * np.einsum("cai,cbj->caibj", g, g)
          + mu * np.einsum("caj,cbi->caibj", g, g)
          + mu * np.einsum("cab,ij->caibj", gg, np.eye(2)))
    return (Ke * area[:, None, None, None, None]).reshape(len(area), 6, 6)


def tri_vector_cell_dofs(cells):
    c = np.asarray(cells, dtype=np.int64)
    return (c[:, :, None] * 2 + np.arange(2)[None, None, :]).reshape(len(c), 6)


def assemble_tri_elasticity(coords, cells, E, nu, mass_coef=None):
    """Plane-strain stiffness (+ mass_coef * vector mass matrix: the inertia term of the dynamic form)."""
    Ke = tri_elasticity_local(coords, cells, E, nu)
    if mass_coef is not None:
        Me = tri_mass_local(coords, cells, mass_coef)                      # [nc,3,3]
        Ke = Ke + np.einsum("cab,ij->caibj", Me, np.eye(2)).reshape(len(Me), 6, 6)
    return assemble_generic(2 * len(coords), tri_vector_cell_dofs(cells), Ke)


def assemble_tri_vector_source(coords, cells, f, div_coef=None):
    """b_(a,i) = int f_i phi_a dx [+ int c d_i phi_a dx, c constant, per cell or nodal (averaged over the cell's
    vertices: one-point rule, as the 3-D thermal-stress load)]."""
    area, g = tri_geometry(coords, cells)
    c = np.asarray(cells, dtype=np.int64)
    be = np.broadcast_to((area / 3.0)[:, None, None] * np.asarray(f, dtype=np.float64)[None, None, :], (len(area), 3, 2)).copy()
    if div_coef is not None:
        dc = np.asarray(div_coef, dtype=np.float64)
        if dc.ndim == 0:
            cc = np.full(len(area), float(dc))
        elif len(dc) == len(area) and len(dc) != len(coords):
            cc = dc
        else:
            cc = dc[c].sum(axis=1) / 3.0
        be += (cc * area)[:, None, None] * g
    return assemble_generic_vector(2 * len(coords), tri_vector_cell_dofs(cells), be.reshape(len(area), 6))


def assemble_edge_vector_load(coords, edges, markers, marker_id, g):
    """int g . v ds over the marked boundary edges, g a constant 2-vector: g_i * length / 2 on both end points."""
    e = np.asarray(edges, dtype=np.int64)[np.asarray(markers) == marker_id]
    length = np.linalg.norm(np.asarray(coords)[e[:, 1]] - np.asarray(coords)[e[:, 0]], axis=1)
    b = np.zeros((len(coords), 2))
    for i in range(2):
        np.add.at(b[:, i], e.ravel(), np.repeat(g[i] * length / 2.0, 2))
    return b.ravel()


def tri_von_mises_projection(coords, cells, u, E, nu):
    """The 2-D case of von_mises_projection: LinearElasticitySolver.py:71-76 with dimension 2 - sigma the 2x2 tensor,
    s = sigma - tr(sigma)/3 Identity(2) (the reference keeps 1/3 in 2-D), vm constant per P1 triangle.  u: [n, 2]."""
    mu, lmbda = lame(E, nu)
    ce = np.asarray(cells, dtype=np.int64)
    area, g = tri_geometry(coords, cells)
    G = np.einsum("cni,cnk->cik", np.asarray(u)[ce], g)
    sg = mu * (G + np.swapaxes(G, -1, -2)) + lmbda * np.trace(G, axis1=-2, axis2=-1)[:, None, None] * np.eye(2)
    dev = sg - np.trace(sg, axis1=-2, axis2=-1)[:, None, None] / 3.0 * np.eye(2)
    vm = np.sqrt(1.5 * np.einsum("cij,cij->c", dev, dev))
    b = assemble_generic_vector(len(coords), ce, np.repeat((area / 3.0 * vm)[:, None], 3, axis=1))
    M = assemble_generic(len(coords), ce, tri_mass_local(coords, cells, 1.0))
    return solve_direct(M, b), b


# ---- periodic constraints (FunctionSpace(..., constrained_domain=pb), SolverBase.py:260-275) ----------------------------
def periodic_fold(A, b, slaves, masters, ncomp=1):
    """The system DOLFIN assembles on a space without the slave dofs, written on the full dof set: with P copying every
    master value to its slaves, (P^T A P + unit diagonal on the slave rows, P^T b with 0 on the slaves).  Solve it and
    call periodic_expand.  slaves / masters are node indices; dof = node*ncomp + component."""
    n = A.shape[0]
    sl = (np.asarray(slaves, dtype=np.int64)[:, None] * ncomp + np.arange(ncomp)[None, :]).ravel()
    ma = (np.asarray(masters, dtype=np.int64)[:, None] * ncomp + np.arange(ncomp)[None, :]).ravel()
    target = np.arange(n)
    target[sl] = ma
    keep = np.ones(n)
    keep[sl] = 0.0
    P = sp.csr_matrix((np.ones(n), (np.arange(n), target)), shape=(n, n))      # u_full = P u_folded
    Af = (P.T @ sp.csr_matrix(A) @ P + sp.diags(1.0 - keep)).tocsr()
    bf = P.T @ np.asarray(b, dtype=np.float64)
    bf[sl] = 0.0
    Af.sum_duplicates()
    Af.sort_indices()
    return Af, bf


def periodic_expand(u, slaves, masters, ncomp=1):
    u = np.array(u, dtype=np.float64).reshape(-1, ncomp)
    u[np.asarray(slaves, dtype=np.int64)] = u[np.asarray(masters, dtype=np.int64)]
    return u.reshape(-1)


# ---- scalar P2 on triangles (2-D meshes with fe_degree 2) ---------------------------------------------------------------
def tri_p2_cell_dofs(n_vertices, cells):
    """[nc,6] global dofs of every triangle (3 vertices, then the 3 UFC edges: edge i opposite vertex i) and the edge table
    [ne,2] in edge-node order (p2_edge_order: lexicographic, or grouped by index difference on structured meshes)."""
    edges, cell_edges, _ = tri_edge_numbering(cells)
    order = p2_edge_order(edges)
    rank = np.empty(len(edges), dtype=np.int64)
    rank[order] = np.arange(len(edges))
    cd = np.concatenate([np.asarray(cells, dtype=np.int64), n_vertices + rank[cell_edges.astype(np.int64)]], axis=1)
    return cd.astype(np.int32), edges[order]


TRI_P2_EDGES = ((1, 2), (0, 2), (0, 1))


def tri_p2_shape(lam):
    """(phi [6], dphi/dlambda [6,3]) of the P2 triangle at barycentric lam."""
    lam = np.asarray(lam, dtype=np.float64)
    phi = np.zeros(6)
    d = np.zeros((6, 3))
    for i in range(3):
        phi[i] = lam[i] * (2 * lam[i] - 1)
        d[i, i] = 4 * lam[i] - 1
    for e, (i, j) in enumerate(TRI_P2_EDGES):
        phi[3 + e] = 4 * lam[i] * lam[j]
        d[3 + e, i] = 4 * lam[j]
        d[3 + e, j] = 4 * lam[i]
    return phi, d


# degree-4 rule on the triangle (6 points, weights sum to 1): exact for the P2 mass matrix
_TRI_Q4 = (np.array([[0.108103018168070, 0.445948490915965, 0.445948490915965], [0.445948490915965, 0.108103018168070, 0.445948490915965],
                     [0.445948490915965, 0.445948490915965, 0.108103018168070], [0.816847572980459, 0.091576213509771, 0.091576213509771],
                     [0.091576213509771, 0.816847572980459, 0.091576213509771], [0.091576213509771, 0.091576213509771, 0.816847572980459]]),
           np.array([0.223381589678011] * 3 + [0.109951743655322] * 3))


def tri_p2_advection_local(coords, cells, velocity, scale=1.0):
    """Ce[a,b] = scale * int phi_a (v . grad phi_b) dx for the P2 basis on triangles (constant or per-cell velocity), with the
    degree-4 rule _TRI_Q4 (the device uses the 4-point degree-3 rule)."""
    area, g = tri_geometry(coords, cells)
    v = np.asarray(velocity, dtype=np.float64)[..., :2]
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(area), 2))
    pts, wq = _TRI_Q4
    Ce = np.zeros((len(area), 6, 6))
    for lam, w in zip(pts, wq):
        phi, dphi = tri_p2_shape(np.asarray(lam))
        gphi = np.einsum("ak,cki->cai", dphi, g)
        Ce += (scale * w * area)[:, None, None] * np.einsum("a,cb->cab", phi, np.einsum("ci,cbi->cb", v, gphi))
    return Ce


def tri_p2_stiffness_local(coords, cells, k=1.0):
    """Ke[a,b] = k int grad phi_a . grad phi_b dx; quadratic integrand, the degree-4 rule is exact."""
    area, g = tri_geometry(coords, cells)                       # g [nc,3,2] = grad lambda
    kk = np.broadcast_to(np.asarray(k, dtype=np.float64), (len(area),))
    Ke = np.zeros((len(area), 6, 6))
    for lam, w in zip(*_TRI_Q4):
        _, d = tri_p2_shape(lam)
        gp = np.einsum("ak,cki->cai", d, g)                     # [nc,6,2]
        Ke += w * np.einsum("cai,cbi->cab", gp, gp)
    return Ke * (kk * area)[:, None, None]


def tri_p2_mass_reference():
    """int phi_a phi_b over the unit-area triangle (= the 1/180 table of the device kernels), by the degree-4 rule."""
    M = np.zeros((6, 6))
    for lam, w in zip(*_TRI_Q4):
        phi, _ = tri_p2_shape(lam)
        M += w * np.outer(phi, phi)
    return M


def tri_p2_mass_local(coords, cells, c=1.0):
    area, _ = tri_geometry(coords, cells)
    cc = np.broadcast_to(np.asarray(c, dtype=np.float64), (len(area),))
    return (cc * area)[:, None, None] * tri_p2_mass_reference()[None]


def tri_p2_source_local(coords, cells, f=1.0):
    """int f phi_a dx for constant / per-cell f: 0 on the vertices, A/3 on the edge nodes."""
    area, _ = tri_geometry(coords, cells)
    ff = np.broadcast_to(np.asarray(f, dtype=np.float64), (len(area),))
    return (ff * area)[:, None] * np.array([0.0, 0.0, 0.0, 1.0 / 3, 1.0 / 3, 1.0 / 3])[None, :]


def tri_p2_edge_nodes(n_vertices, p2_edges, boundary_edges):
    """For boundary edges [n,2] (vertex pairs): their (a, c, mid) P2 nodes."""
    e = np.sort(np.asarray(boundary_edges, dtype=np.int64), axis=1)
    table = {(int(a), int(b)): i for i, (a, b) in enumerate(np.asarray(p2_edges, dtype=np.int64))}
    mid = np.array([n_vertices + table[(int(a), int(b))] for a, b in e], dtype=np.int64)
    be = np.asarray(boundary_edges, dtype=np.int64)
    return np.stack([be[:, 0], be[:, 1], mid], axis=1)


def assemble_tri_p2_edge_load(n_dofs, coords, nodes3, g):
    """int g phi ds over boundary edges given by their (a, c, mid) nodes: g |e| (1/6, 1/6, 4/6)."""
    co = np.asarray(coords, dtype=np.float64)
    length = np.linalg.norm(co[nodes3[:, 1]] - co[nodes3[:, 0]], axis=1)
    b = np.zeros(n_dofs)
    w = np.broadcast_to(g, length.shape) * length / 6.0
    np.add.at(b, nodes3[:, 0], w)
    np.add.at(b, nodes3[:, 1], w)
    np.add.at(b, nodes3[:, 2], 4.0 * w)
    return b


def assemble_tri_p2_edge_mass(n_dofs, coords, nodes3, h):
    """int h T q ds over boundary edges: h |e| / 30 [[4,-1,2],[-1,4,2],[2,2,16]] on (a, c, mid)."""
    co = np.asarray(coords, dtype=np.float64)
    length = np.linalg.norm(co[nodes3[:, 1]] - co[nodes3[:, 0]], axis=1)
    M3 = np.array([[4.0, -1.0, 2.0], [-1.0, 4.0, 2.0], [2.0, 2.0, 16.0]]) / 30.0
    Ke = (np.broadcast_to(h, length.shape) * length)[:, None, None] * M3[None]
    return assemble_generic(n_dofs, nodes3, Ke)


# ---- 2-vector P2 on triangles (plane strain with fe_degree 2) --------------------------------------------------------------
_TRI_MID = (np.array([[0.0, 0.5, 0.5], [0.5, 0.0, 0.5], [0.5, 0.5, 0.0]]), np.array([1.0 / 3] * 3))      # edge-midpoint rule, degree 2


def tri_p2_vector_cell_dofs(cell_dofs):
    cd = np.asarray(cell_dofs, dtype=np.int64)
    return (cd[:, :, None] * 2 + np.arange(2)[None, None, :]).reshape(len(cd), 12)


def tri_p2_elasticity_local(coords, cells, E, nu):
    """Ke[(a,i),(b,j)] = int lmbda d_i phi_a d_j phi_b + mu d_j phi_a d_i phi_b + mu delta_ij grad phi_a . grad phi_b dx on P2
    triangles; quadratic integrand, degree-4 rule."""
    mu, lmbda = lame(E, nu)
    area, g = tri_geometry(coords, cells)
    Ke = np.zeros((len(area), 6, 2, 6, 2))
    for lam, w in zip(*_TRI_Q4):
        _, d = tri_p2_shape(lam)
        gp = np.einsum("ak,cki->cai", d, g)
        gg = np.einsum("cak,cbk->cab", gp, gp)
        Ke += w * (lmbda * np.einsum("cai,cbj->caibj", gp, gp) + mu * np.einsum("caj,cbi->caibj", gp, gp)
                   + mu * np.einsum("cab,ij->caibj", gg, np.eye(2)))
    return (Ke * area[:, None, None, None, None]).reshape(len(area), 12, 12)


def assemble_tri_p2_elasticity(coords, cells, cell_dofs, n_nodes, E, nu, mass_coef=None):
    Ke = tri_p2_elasticity_local(coords, cells, E, nu)
    if mass_coef is not None:
        Me = tri_p2_mass_local(coords, cells, mass_coef)
        Ke = Ke + np.einsum("cab,ij->caibj", Me, np.eye(2)).reshape(len(Me), 12, 12)
    return assemble_generic(2 * n_nodes, tri_p2_vector_cell_dofs(cell_dofs), Ke)


def assemble_tri_p2_vector_source(coords, cells, cell_dofs, n_nodes, f, div_coef=None):
    """b_(a,i) = int f_i phi_a dx [+ int c d_i phi_a dx; c a number, per cell, or P1 given by its vertex values]."""
    area, g = tri_geometry(coords, cells)
    ce = np.asarray(cells, dtype=np.int64)
    be = np.zeros((len(area), 6, 2))
    be[:, 3:, :] = (area / 3.0)[:, None, None] * np.asarray(f, dtype=np.float64)[None, None, :]
    if div_coef is not None:
        dc = np.asarray(div_coef, dtype=np.float64)
        if dc.ndim == 0:
            cv = np.full((len(area), 3), float(dc))
        elif len(dc) == len(area) and len(dc) != len(coords):
            cv = np.repeat(dc[:, None], 3, axis=1)
        else:
            cv = dc[ce]
        for lam, w in zip(*_TRI_MID):
            _, d = tri_p2_shape(lam)
            gp = np.einsum("ak,cki->cai", d, g)
            be += (w * area * (cv @ lam))[:, None, None] * gp
    return assemble_generic_vector(2 * n_nodes, tri_p2_vector_cell_dofs(cell_dofs), be.reshape(len(area), 12))


def assemble_tri_p2_edge_vector_load(n_nodes, coords, nodes3, g):
    """int g . v ds over boundary edges given by their (a, c, mid) nodes, g a constant 2-vector."""
    co = np.asarray(coords, dtype=np.float64)
    length = np.linalg.norm(co[nodes3[:, 1]] - co[nodes3[:, 0]], axis=1)
    b = np.zeros((n_nodes, 2))
    for i in range(2):
        w = g[i] * length / 6.0
        np.add.at(b[:, i], nodes3[:, 0], w)
        np.add.at(b[:, i], nodes3[:, 1], w)
        np.add.at(b[:, i], nodes3[:, 2], 4.0 * w)
    return b.ravel()


def tri_p2_von_mises_projection(coords, cells, cell_dofs, u, E, nu):
    """project(von_Mises(u), P1) for a P2 displacement on triangles (u [n_nodes, 2]): the reference's 2-D expression (2x2
    tensor, deviator with 1/3) integrated against lambda_a with the 3-point edge-midpoint rule.  Returns (w, b)."""
    mu, lmbda = lame(E, nu)
    ce = np.asarray(cells, dtype=np.int64)
    area, g = tri_geometry(coords, cells)
    un = np.asarray(u)[np.asarray(cell_dofs, dtype=np.int64)]                 # [nc,6,2]
    be = np.zeros((len(ce), 3))
    for lam, w in zip(*_TRI_MID):
        _, d = tri_p2_shape(lam)
        gp = np.einsum("ak,cki->cai", d, g)
        G = np.einsum("cni,cnk->cik", un, gp)
        sg = mu * (G + np.swapaxes(G, -1, -2)) + lmbda * np.trace(G, axis1=-2, axis2=-1)[:, None, None] * np.eye(2)
        dev = sg - np.trace(sg, axis1=-2, axis2=-1)[:, None, None] / 3.0 * np.eye(2)
        vm = np.sqrt(1.5 * np.einsum("cij,cij->c", dev, dev))
        be += (w * area * vm)[:, None] * np.asarray(lam)[None, :]
    b = assemble_generic_vector(len(coords), ce, be)
    M = assemble_generic(len(coords), ce, tri_mass_local(coords, cells, 1.0))
    return solve_direct(M, b), b


# ---- interior penalty on P2 spaces (ScalarTransportSolver.py:312-315 is degree-agnostic) ---------------------------------
def assemble_p2_interior_penalty(coords, cells, coefficient, n_quad=4):
    """+ coefficient * avg(h)^2 * jump(grad T, n) * jump(grad q, n) dS for the P2 basis, tetrahedra (4-column cells) or
    triangles (3-column cells).  grad phi . n is linear along the facet, the integrand quadratic: integrated here with MORE
    points than needed (tets: the 6-point degree-4 rule; triangles: n_quad-point Gauss-Legendre) - the device uses the minimal
    exact rules.  Dofs as p2_cell_dofs / tri_p2_cell_dofs number them."""
    import scipy.sparse as sp
    co = np.asarray(coords, dtype=np.float64)
    ce = np.asarray(cells, dtype=np.int64)
    tdim = ce.shape[1] - 1
    if tdim == 3:
        cd, edges = p2_cell_dofs(len(co), ce)
        _, g = p1_geometry(co, ce)
        h = 2.0 * tet_circumradius(co, ce)
        facets, cell_facets, _ = facet_numbering(ce)
        loc_edges = P2_EDGE_VERTS
        qp = np.array([[0.108103018168070, 0.445948490915965, 0.445948490915965], [0.445948490915965, 0.108103018168070, 0.445948490915965],
                       [0.445948490915965, 0.445948490915965, 0.108103018168070], [0.816847572980459, 0.091576213509771, 0.091576213509771],
                       [0.091576213509771, 0.816847572980459, 0.091576213509771], [0.091576213509771, 0.091576213509771, 0.816847572980459]])
        qw = np.array([0.223381589678011] * 3 + [0.109951743655322] * 3)
    else:
        cd, edges = tri_p2_cell_dofs(len(co), ce)
        area_c, g = tri_geometry(co, ce)
        X = co[ce]
        d = lambda p, q: np.linalg.norm(X[:, p] - X[:, q], axis=1)          # noqa: E731
        h = d(0, 1) * d(1, 2) * d(2, 0) / (2.0 * area_c)
        facets, cell_facets, _ = tri_edge_numbering(ce)
        loc_edges = TRI_P2_EDGES
        xg, wg = np.polynomial.legendre.leggauss(n_quad)
        qp = np.stack([0.5 * (1.0 - xg), 0.5 * (1.0 + xg)], axis=1)
        qw = 0.5 * wg
    cd = cd.astype(np.int64)
    n = len(co) + len(edges)
    nvc = tdim + 1
    sides = {}
    for c in range(len(ce)):
        for i in range(nvc):
            sides.setdefault(int(cell_facets[c, i]), []).append((c, i))
    rows, cols, vals = [], [], []
    for f, ss in sides.items():
        if len(ss) != 2:
            continue
        fv = facets[f].astype(np.int64)
        P = co[fv]
        if tdim == 3:
            nrm = np.cross(P[1] - P[0], P[2] - P[0])
            meas = 0.5 * np.linalg.norm(nrm)
        else:
            t = P[1] - P[0]
            nrm = np.array([t[1], -t[0]])
            meas = np.linalg.norm(t)
        nrm = nrm / np.linalg.norm(nrm)
        w0 = coefficient * (0.5 * (h[ss[0][0]] + h[ss[1][0]])) ** 2 * meas
        Jq = []
        for bary in qp:
            J = {}
            for c, i in ss:
                outward = nrm if np.dot(nrm, P[0] - co[ce[c, i]]) > 0 else -nrm
                lam = np.zeros(nvc)
                for k, v in enumerate(fv):
                    lam[list(ce[c]).index(v)] = bary[k]
                dphi = np.zeros((len(cd[c]), nvc))
                for a in range(nvc):
                    dphi[a, a] = 4.0 * lam[a] - 1.0
                for e, (p_, q_) in enumerate(loc_edges):
                    dphi[nvc + e, p_] = 4.0 * lam[q_]
                    dphi[nvc + e, q_] = 4.0 * lam[p_]
                gphi = dphi @ g[c]
                for a in range(len(cd[c])):
                    J[int(cd[c, a])] = J.get(int(cd[c, a]), 0.0) + float(gphi[a] @ outward)
            Jq.append(J)
        nodes = list(Jq[0])
        for a in nodes:
            for b in nodes:
                rows.append(a); cols.append(b)
                vals.append(w0 * sum(wq * J[a] * J[b] for wq, J in zip(qw, Jq)))
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


# ---- SUPG on P2 spaces (ScalarTransportSolver.py:259-270 with fe_degree 2): Tq = q + tau (v . grad q) replaces the test function
# in EVERY term of the form (:278-311), so grad(Tq) = grad q + tau H_q v with the (cell-wise constant) Hessian H_q of the quadratic
# q.  Dimension-generic: simplices with nv = 3 or 4 vertices; quadrature of degree 4 (triangles) / 5 (tetrahedra), exact for all
# integrands below (the highest is the cubic  phi_b (v . grad q_a)).
def _p2_simplex_tools(coords, cells):
    from oracle import ns_oracle as nso
    cells = np.asarray(cells, dtype=np.int64)
    if cells.shape[1] == 4:
        detJ, g = p1_geometry(coords, cells)
        size = np.abs(detJ) / 6.0
        pts, wq = nso.tet_quadrature(5)
        shape = nso.p2_shape
        edges = P2_EDGE_VERTS
    else:
        size, g = tri_geometry(coords, cells)
        pts, wq = _TRI_Q4
        shape = tri_p2_shape
        edges = TRI_P2_EDGES
    return cells, size, g, pts, wq, shape, edges


def p2_hessians(g, edges):
    """H[c,a] = sum_kl d2 phi_a / d lambda_k d lambda_l  g_k g_l^T: vertex i: 4 g_i g_i^T, edge ij: 4 (g_i g_j^T + g_j g_i^T)."""
    nc, nv, d = g.shape
    H = np.zeros((nc, nv + len(edges), d, d))
    for i in range(nv):
        H[:, i] = 4.0 * np.einsum("ci,cj->cij", g[:, i], g[:, i])
    for e, (i, j) in enumerate(edges):
        H[:, nv + e] = 4.0 * (np.einsum("ci,cj->cij", g[:, i], g[:, j]) + np.einsum("ci,cj->cij", g[:, j], g[:, i]))
    return H


def p2_supg_tau(coords, cells, velocity, pe):
    """(tau [nc], v [nc,d]): tau = 0.5 h / (4/(Pe h) + 2 |v|), h = 2 * Circumradius (:262-266), constant or per-cell velocity."""
    cells = np.asarray(cells, dtype=np.int64)
    d = cells.shape[1] - 1
    v = np.asarray(velocity, dtype=np.float64)[..., :d]
    if v.ndim == 1:
        v = np.broadcast_to(v, (len(cells), d))
    h = 2.0 * tet_circumradius(np.asarray(coords, dtype=np.float64)[:, :d], cells)
    return 0.5 * h / (4.0 / (pe * h) + 2.0 * np.linalg.norm(v, axis=1)), v


def p2_supg_system_local(coords, cells, velocity, pe, stiffness=0.0, advection_scale=0.0, mass_coef=0.0):
    """Ke[a,b] = int  k grad phi_b . grad(Tq_a) + scale (v . grad phi_b) Tq_a + m phi_b Tq_a  dx,  Tq_a = q_a + tau (v . grad q_a);
    rows = test functions (k, m constant or per cell).  With pe = None the plain Galerkin matrix (tau = 0)."""
    cells, size, g, pts, wq, shape, edges = _p2_simplex_tools(coords, cells)
    nc = len(cells)
    if pe is None:
        d = cells.shape[1] - 1
        v = np.asarray(velocity, dtype=np.float64)[..., :d]
        v = np.broadcast_to(v, (nc, d)) if v.ndim == 1 else v
        tau = np.zeros(nc)
    else:
        tau, v = p2_supg_tau(coords, cells, velocity, pe)
    Hv = np.einsum("caij,cj->cai", p2_hessians(g, edges), v)                  # H_a v
    k = np.broadcast_to(np.asarray(stiffness, dtype=np.float64), (nc,))
    m = np.broadcast_to(np.asarray(mass_coef, dtype=np.float64), (nc,))
    nd = cells.shape[1] + len(edges)
    Ke = np.zeros((nc, nd, nd))
    for lam, w in zip(pts, wq):
        phi, dphi = shape(np.asarray(lam))
        gphi = np.einsum("ak,cki->cai", dphi, g)
        vg = np.einsum("ci,cai->ca", v, gphi)
        Tq = phi[None, :] + tau[:, None] * vg                                 # [nc, a]
        gTq = gphi + tau[:, None, None] * Hv                                  # [nc, a, d]
        Ke += (w * size * k)[:, None, None] * np.einsum("cai,cbi->cab", gTq, gphi)
        Ke += (w * size * advection_scale)[:, None, None] * Tq[:, :, None] * vg[:, None, :]
        Ke += (w * size * m)[:, None, None] * Tq[:, :, None] * phi[None, None, :]
    return Ke


def p2_supg_source_local(coords, cells, velocity, pe, f=None, f_nodal=None, cell_dofs=None):
    """be[a] = int f Tq_a dx for a constant / per-cell source f, or - f_nodal [n_dofs] with cell_dofs [nc, nd] - for the source
    Function's CG2 interpolant  S_h = sum_k S_k phi_k  (ScalarTransportSolver.py:213-226: get_body_source_items returns whatever
    translate_value made of settings['body_source'], :259-276: it is multiplied by Tq like every other term).  By quadrature
    (degree 4 / 5: the integrand S_h (v . grad q_a) is cubic), independently of the kernels' closed-form moments."""
    cells, size, g, pts, wq, shape, edges = _p2_simplex_tools(coords, cells)
    tau, v = p2_supg_tau(coords, cells, velocity, pe)
    be = np.zeros((len(cells), cells.shape[1] + len(edges)))
    if f_nodal is not None:
        Sk = np.asarray(f_nodal, dtype=np.float64)[np.asarray(cell_dofs, dtype=np.int64)]          # [nc, nd]
    else:
        ff = np.broadcast_to(np.asarray(f, dtype=np.float64), (len(cells),))
    for lam, w in zip(pts, wq):
        phi, dphi = shape(np.asarray(lam))
        vg = np.einsum("ci,cai->ca", v, np.einsum("ak,cki->cai", dphi, g))
        fq = Sk @ phi if f_nodal is not None else ff
        be += (w * size * fq)[:, None] * (phi[None, :] + tau[:, None] * vg)
    return be


def p2_supg_facet_terms(coords, cells, cell_dofs, n_dofs, facet_cells, velocity, pe, g=None, h=None):
    """(dA, db) of the SUPG part tau (v . grad q_a) of the boundary integrals  g Tq ds  and  h T Tq ds  over boundary facets given as
    (cell, opposite local vertex); a runs over ALL dofs of the cell behind the facet.  Facet quadrature: a Gauss rule along an edge,
    the degree-4 rule on a triangle (integrands are cubic)."""
    import scipy.sparse as sp
    cells, size, gg, _, _, shape, edges = _p2_simplex_tools(coords, cells)
    tau, v = p2_supg_tau(coords, cells, velocity, pe)
    nv = cells.shape[1]
    co = np.asarray(coords, dtype=np.float64)[:, :nv - 1]
    if nv == 4:
        fpts, fw = _TRI_Q4
    else:
        x, wg = np.polynomial.legendre.leggauss(4)
        fpts, fw = np.stack([0.5 * (1 - x), 0.5 * (1 + x)], axis=1), 0.5 * wg
    db = np.zeros(n_dofs)
    rows, cols, vals = [], [], []
    fc = np.asarray(facet_cells, dtype=np.int64)
    for kf, (c, o) in enumerate(fc):
        others = [i for i in range(nv) if i != o]
        p = co[cells[c, others]]
        meas = np.linalg.norm(p[1] - p[0]) if nv == 3 else 0.5 * np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
        gk = 0.0 if g is None else float(np.broadcast_to(g, (len(fc),))[kf])
        hk = 0.0 if h is None else float(np.broadcast_to(h, (len(fc),))[kf])
        for mu, w in zip(fpts, fw):
            lam = np.zeros(nv)
            lam[others] = mu
            phi, dphi = shape(lam)
            vg = (dphi @ gg[c]) @ v[c]                                        # v . grad q_a at the point, [nd]
            wa = w * meas * tau[c] * vg
            db[cell_dofs[c]] += gk * wa
            if hk:
                for a in range(len(phi)):
                    for bdof in range(len(phi)):
                        if phi[bdof] != 0.0:
                            rows.append(cell_dofs[c, a]); cols.append(cell_dofs[c, bdof]); vals.append(hk * wa[a] * phi[bdof])
    dA = sp.coo_matrix((vals, (rows, cols)), shape=(n_dofs, n_dofs)).tocsr()
    return dA, db
