"""Domain decomposition plans for one-process-per-GPU runs (host logic, numpy only).

DOLFIN partitions the mesh with SCOTCH under ``mpirun`` and PETSc builds the VecScatter
that refreshes ghost values (the reference only inherits this: SolverBase.py:102-118, 634).
Here the owner-computes plan is explicit:

  * every vertex (P1 dof node) has one owner rank;
  * a rank keeps every cell that touches one of its vertices (one ghost-cell layer), so
    assembly of the rows it owns needs no communication;
  * local numbering = owned vertices first (ascending global id), then ghosts grouped by
    owner rank (ascending), each group ascending by global id;
  * for every neighbour: the owned vertices it needs (send list, ascending global id — the
    same order the neighbour stores them as ghosts) and the number of ghosts it provides.

``slab_layout`` is the closed form of the same plan for the structured box meshes of
bench.py (z-slabs of whole vertex planes: <= 2 neighbours, contiguous send ranges);
``build_local_part`` handles any mesh/owner array and is what the tests compare it with.
"""
from __future__ import annotations

import numpy as np


def slab_owner(coords, n_parts, axis=2):
    """Owner array for slabs of (nearly) equal vertex count along `axis`.  Vertices sharing a
    coordinate stay together, so structured meshes split into whole planes."""
    c = np.asarray(coords)[:, axis]
    levels, inv = np.unique(c, return_inverse=True)
    counts = np.bincount(inv)
    cum = np.cumsum(counts)
    total = cum[-1]
    # plane p goes to the part whose share its midpoint falls in
    mid = cum - counts / 2.0
    owner_of_level = np.minimum((mid * n_parts / total).astype(np.int64), n_parts - 1)
    return owner_of_level[inv].astype(np.int32)


def rcb_owner(coords, n_parts):
    """Recursive coordinate bisection of the vertices (unstructured meshes)."""
    coords = np.asarray(coords)
    owner = np.zeros(len(coords), dtype=np.int32)

    def split(idx, lo, hi):
        if hi - lo <= 1:
            owner[idx] = lo
            return
        ext = coords[idx].max(axis=0) - coords[idx].min(axis=0)
        ax = int(np.argmax(ext))
        nl = (hi - lo) // 2
        k = int(round(len(idx) * nl / float(hi - lo)))
        order = idx[np.argsort(coords[idx, ax], kind="stable")]
        split(order[:k], lo, lo + nl)
        split(order[k:], lo + nl, hi)

    split(np.arange(len(coords)), 0, n_parts)
    return owner


class LocalPart:
    """One rank's share of a partitioned mesh (see module docstring for the numbering)."""

    def __init__(self, rank, l2g, n_owned, cells_local, cell_gids, neighbors, send_lists, recv_counts):
        self.rank = rank
        self.l2g = l2g                      # [n_local] global vertex id of every local vertex
        self.n_owned = n_owned
        self.cells = cells_local            # [nc_local, 4] local vertex ids (vertex order of the global cell)
        self.cell_gids = cell_gids          # [nc_local] global cell ids (ascending unless a locality order was given)
        self.neighbors = neighbors          # ranks, ascending
        self.send_lists = send_lists        # per neighbour: owned local ids, ascending global id
        self.recv_counts = recv_counts      # per neighbour: number of ghosts it owns

    @property
    def n_local(self):
        return len(self.l2g)

    def g2l(self, n_global):
        m = np.full(n_global, -1, dtype=np.int64)
        m[self.l2g] = np.arange(len(self.l2g))
        return m

    def dof_send_lists(self, ncomp):
        if ncomp == 1:
            return self.send_lists
        return [(s[:, None] * ncomp + np.arange(ncomp)[None, :]).ravel().astype(np.int32) for s in self.send_lists]


def _local_cell_mask(cells, owner, q, face_pairs, master_cells_of=None):
    """Cells of rank q's part: those holding a q-owned vertex and - with face_pairs, the (cell, cell) pairs of the interior
    facets - their neighbours across a facet (the second layer interior-facet integrals need: a row of an owned vertex a takes
    contributions from every facet of the cells around a, and the cell on the far side need not touch any owned vertex).
    master_cells_of ([n_global] master vertex of every slave, -1 elsewhere; CG2 spaces with a periodic constraint): the cells
    around the masters of the slave vertices of those cells as well - the master of a slave EDGE node is an edge node, and an edge
    exists on the device only inside a cell (a P1 space makes do with the master vertex alone, without cells)."""
    m = (owner[cells] == q).any(axis=1)
    if face_pairs is not None:
        a, b = face_pairs[:, 0], face_pairs[:, 1]
        m2 = m.copy()
        m2[b[m[a]]] = True
        m2[a[m[b]]] = True
        m = m2
    if master_cells_of is not None:
        v = np.unique(cells[m])
        ma = master_cells_of[v]
        need = np.zeros(len(owner), dtype=bool)
        need[ma[ma >= 0]] = True
        m = m | need[cells].any(axis=1)
    return m


def _with_masters(verts, master_of):
    """verts plus the masters of the slaves among them (tied vertex pairs of a periodic constraint), ascending."""
    m = master_of[verts]
    return np.union1d(verts, m[m >= 0])


def build_local_part(cells, owner, rank, vertex_rank=None, cell_rank=None, face_pairs=None, tied=None, tied_cells=False):
    """vertex_rank / cell_rank (optional, [n_global] each): a locality order (backend.locality_order) - the owned vertices
    and the local cells are then numbered by it instead of by their global ids (the order a mesh file happens to have).
    face_pairs [nf,2] (optional, global cell ids of the two cells of every interior facet): two cell layers instead of one.
    tied (optional, (slaves, masters) global vertex ids of a periodic constraint; `owner` must give a slave its master's rank):
    the folded operator P^T A P moves every column of a slave onto its master, so a part holds the master of every slave among
    its vertices as well - as an extra ghost vertex without cells where the master is no mesh neighbour of anything local (the
    far side of the domain).  tied_cells (CG2 spaces): the part also takes the CELLS around those masters (see _local_cell_mask)."""
    cells = np.asarray(cells, dtype=np.int64)
    owner = np.asarray(owner)
    if face_pairs is not None:
        face_pairs = np.asarray(face_pairs, dtype=np.int64).reshape(-1, 2)
    n_global = len(owner)
    master_of = None
    if tied is not None:
        master_of = np.full(n_global, -1, dtype=np.int64)
        master_of[np.asarray(tied[0], dtype=np.int64)] = np.asarray(tied[1], dtype=np.int64)
        if not np.array_equal(owner[np.asarray(tied[0], dtype=np.int64)], owner[np.asarray(tied[1], dtype=np.int64)]):
            raise ValueError("build_local_part: a tied (slave, master) pair must live on one rank")
    mco = master_of if tied_cells else None
    keep = np.nonzero(_local_cell_mask(cells, owner, rank, face_pairs, mco))[0]
    if cell_rank is not None:
        keep = keep[np.argsort(np.asarray(cell_rank)[keep], kind="stable")]
    lc = cells[keep]
    verts = np.unique(lc)
    if master_of is not None:
        verts = _with_masters(verts, master_of)
    mine = verts[owner[verts] == rank]
    if vertex_rank is not None:
        mine = mine[np.argsort(np.asarray(vertex_rank)[mine], kind="stable")]
    ghosts = verts[owner[verts] != rank]
    gorder = np.lexsort((ghosts, owner[ghosts]))          # by owner rank, then global id
    ghosts = ghosts[gorder]
    l2g = np.concatenate([mine, ghosts])
    g2l = np.full(n_global, -1, dtype=np.int64)
    g2l[l2g] = np.arange(len(l2g))
    neighbors, recv_counts = np.unique(owner[ghosts], return_counts=True)
    # what each neighbour q needs from me: my vertices in the cells of q's part (ascending global id = q's ghost order)
    send_lists = []
    nb_all = set(neighbors.tolist())
    general = face_pairs is not None or master_of is not None      # (q's whole part is worked out, not only its cells around my vertices)
    if not general:
        cand = cells[(owner[cells] == rank).any(axis=1)]
        others = sorted(nb_all | set(np.unique(owner[cand]).tolist()) - {rank})
    else:
        others = [q for q in range(int(owner.max()) + 1) if q != rank]
    for q in others:
        if not general:
            touch = cand[(owner[cand] == q).any(axis=1)]
        else:
            touch = cells[_local_cell_mask(cells, owner, q, face_pairs, mco)]
        v = np.unique(touch)
        if master_of is not None:
            v = _with_masters(v, master_of)
        v = v[owner[v] == rank]
        if general and len(v) == 0 and q not in nb_all:
            continue
        if q not in nb_all or (general and len(v) == 0):
            # q needs my vertices but I need none of q's (or the reverse): cannot happen, the overlap is symmetric
            raise AssertionError("asymmetric neighbourhood between ranks %d and %d" % (rank, q))
        send_lists.append(g2l[v].astype(np.int32))
    return LocalPart(rank, l2g, len(mine), g2l[lc].astype(np.int32), keep, [int(q) for q in neighbors],
                     send_lists, [int(c) for c in recv_counts])


def slab_ranges(n_planes, world, planes_per_rank=None):
    """[zb, ze) owned vertex planes of every rank."""
    if planes_per_rank is not None:
        return [(r * planes_per_rank, (r + 1) * planes_per_rank) for r in range(world)]
    cuts = [n_planes * r // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def slab_layout(nx, ny, nz, zplanes, rank, world):
    """Closed-form plan of the z-slab [zplanes) of BoxMesh(nx,ny,nz) as fs_mesh_create_box numbers it:
    owned planes first (x-fastest), then the lower ghost plane, then the upper ghost plane.
    Returns dict(n_owned, n_local, planes (global z index of every local plane), l2g, neighbors,
    send_lists, recv_counts)."""
    P = (nx + 1) * (ny + 1)
    zb, ze = zplanes
    has_lo, has_hi = zb > 0, ze < nz + 1
    planes = list(range(zb, ze)) + ([zb - 1] if has_lo else []) + ([ze] if has_hi else [])
    n_owned = (ze - zb) * P
    l2g = np.concatenate([np.arange(iz * P, (iz + 1) * P, dtype=np.int64) for iz in planes])
    neighbors, send_lists, recv_counts = [], [], []
    if has_lo:
        neighbors.append(rank - 1)
        send_lists.append(np.arange(0, P, dtype=np.int32))
        recv_counts.append(P)
    if has_hi:
        neighbors.append(rank + 1)
        send_lists.append(np.arange(n_owned - P, n_owned, dtype=np.int32))
        recv_counts.append(P)
    return dict(n_owned=n_owned, n_local=n_owned + (has_lo + has_hi) * P, planes=planes, l2g=l2g,
                neighbors=neighbors, send_lists=send_lists, recv_counts=recv_counts, plane_size=P)


def slab_dirichlet(nx, ny, nz, layout, axis, lo_value=350.0, hi_value=300.0):
    """Local dofs/values of the Dirichlet face pair of the box heat problem (axis = 0, 1 or 2),
    ghosts included (the elimination needs the values of constrained ghost columns)."""
    P = layout["plane_size"]
    inplane = np.arange(P)
    ix, iy = inplane % (nx + 1), inplane // (nx + 1)
    dofs, vals = [], []
    for lp, iz in enumerate(layout["planes"]):
        if axis == 2:
            if iz == 0:
                dofs.append(lp * P + inplane)
                vals.append(np.full(P, lo_value))
            elif iz == nz:
                dofs.append(lp * P + inplane)
                vals.append(np.full(P, hi_value))
        else:
            c, m = (ix, nx) if axis == 0 else (iy, ny)
            lo, hi = inplane[c == 0], inplane[c == m]
            dofs += [lp * P + lo, lp * P + hi]
            vals += [np.full(lo.size, lo_value), np.full(hi.size, hi_value)]
    if not dofs:
        return np.zeros(0, dtype=np.int32), np.zeros(0)
    return np.concatenate(dofs).astype(np.int32), np.concatenate(vals)


# ---- CG2 (P2) spaces on a decomposed mesh -------------------------------------------------------------------
_TET_EDGES = ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))


class P2Plan:
    """Node-level plan of a CG2 space on one rank's part.  Local node order (libfsamd.so, fs_space_create):
    [owned vertices | owned edges | ghost vertices | ghost edges]; an edge belongs to the rank owning its endpoint
    of smaller global id."""

    def __init__(self, n_owned_nodes, l2g_nodes, neighbors, send_lists, recv_lists):
        self.n_owned_nodes = n_owned_nodes
        self.l2g_nodes = l2g_nodes          # [n_local_nodes] global node id (vertices, then n_global_vertices + global edge index)
        self.neighbors = neighbors
        self.send_lists = send_lists        # per neighbour: local node ids (owned) in the agreed order
        self.recv_lists = recv_lists        # per neighbour: local node ids (ghost) in the same order
        self.recv_counts = [len(r) for r in recv_lists]


def _edge_keys(g0, g1, n_global):
    lo, hi = np.minimum(g0, g1), np.maximum(g0, g1)
    return lo.astype(np.int64) * n_global + hi


def build_p2_plan(cells, owner, rank, part, local_edges, global_edges, order_id=None, part_cells_of=None):
    """cells [nc,4] / owner [nv]: the GLOBAL mesh and vertex owners; part: this rank's LocalPart; local_edges [ne,2]: the
    device's edge table in node order (local vertex ids, owned edges first); global_edges [ne_g,2]: the host's global
    edge-node table (FunctionSpace.edge_nodes()).  No communication: every rank derives both sides of each exchange
    from the global mesh, ordered by (vertices by global id, then edges by (g0, g1)).
    order_id ([n_global], optional): the ids the DEVICE mesh was given as global ids (fs_mesh_set_global_ids) where they are not
    the mesh's own - an edge belongs to the owner of its end point of smaller ORDER id (periodic CG2 spaces: a slave vertex is
    ordered right behind its master, so that a slave edge and its master edge are owned through corresponding end points).
    part_cells_of (rank -> boolean mask over the global cells, optional): the cells of another rank's part where they are not
    just the cells around its vertices."""
    cells = np.asarray(cells, dtype=np.int64)
    owner = np.asarray(owner)
    n_global = len(owner)
    oid = np.arange(n_global, dtype=np.int64) if order_id is None else np.asarray(order_id, dtype=np.int64)
    nv, nvo = part.n_local, part.n_owned
    le = np.asarray(local_edges, dtype=np.int64).reshape(-1, 2)
    ga, gb = part.l2g[le[:, 0]], part.l2g[le[:, 1]]
    g0 = np.minimum(ga, gb)
    g1 = np.maximum(ga, gb)
    e_owner = owner[np.where(oid[ga] < oid[gb], ga, gb)]
    neo = int((e_owner == rank).sum())
    if not (np.all(e_owner[:neo] == rank) and np.all(e_owner[neo:] != rank)):
        raise AssertionError("device edge table is not ordered owned-first")
    ne = len(le)
    node_of_vertex = np.where(np.arange(nv) < nvo, np.arange(nv), np.arange(nv) + neo)
    node_of_edge = np.where(np.arange(ne) < neo, nvo + np.arange(ne), nv + np.arange(ne))
    # global node ids
    gkey = _edge_keys(np.asarray(global_edges)[:, 0], np.asarray(global_edges)[:, 1], n_global)
    gsort = np.argsort(gkey)
    lkey = _edge_keys(g0, g1, n_global)
    gpos = gsort[np.searchsorted(gkey[gsort], lkey)]
    if not np.array_equal(gkey[gpos], lkey):
        raise AssertionError("a local edge is missing from the global edge table")
    l2g_nodes = np.empty(nv + ne, dtype=np.int64)
    l2g_nodes[node_of_vertex] = part.l2g
    l2g_nodes[node_of_edge] = n_global + gpos
    # exchanges
    lsort = np.argsort(lkey)
    send_lists, recv_lists = [], []
    mine_cells = cells[(owner[cells] == rank).any(axis=1)]
    for qi, q in enumerate(part.neighbors):
        # receive: ghost vertices owned by q (part order: by global id), then ghost edges owned by q by (g0, g1)
        gv = np.nonzero(owner[part.l2g] == q)[0]
        gv = gv[np.argsort(part.l2g[gv])]
        ge = np.nonzero(e_owner == q)[0]
        ge = ge[np.argsort(lkey[ge])]
        recv_lists.append(np.concatenate([node_of_vertex[gv], node_of_edge[ge]]).astype(np.int32))
        # send: my vertices (LocalPart order) and my edges that live in a cell local to q
        touch = mine_cells[(owner[mine_cells] == q).any(axis=1)] if part_cells_of is None else cells[part_cells_of(q)]
        a = np.concatenate([touch[:, i] for i, _ in _TET_EDGES])
        b = np.concatenate([touch[:, j] for _, j in _TET_EDGES])
        k, first = np.unique(_edge_keys(a, b, n_global), return_index=True)
        a, b = a[first], b[first]
        k = k[owner[np.where(oid[a] < oid[b], a, b)] == rank]   # owner = owner of the end point of smaller (order) id
        pos = lsort[np.searchsorted(lkey[lsort], k)]
        if not np.array_equal(lkey[pos], k):
            raise AssertionError("an edge to send is not a local edge")
        send_lists.append(np.concatenate([np.asarray(part.send_lists[qi], dtype=np.int64), node_of_edge[pos]]).astype(np.int32))
    return P2Plan(nvo + neo, l2g_nodes, list(part.neighbors), send_lists, recv_lists)


def build_p2_plan_local(cells, gid, owner, rank, neighbors, device_edges, n_global_vertices):
    """The CG2 node plan of one rank from ITS OWN part only - nothing of global size is touched (what DOLFIN's distributed
    dofmap builds from the local mesh and the shared-entity tables under mpirun).

    cells [nc,4]: this rank's cells as LOCAL vertex ids (every cell touching an owned vertex: one ghost-cell layer);
    gid [nv] / owner [nv]: global id and owning rank of every local vertex (owned vertices first); neighbors: the ranks
    owning ghost vertices; device_edges [ne,2]: the device's edge table in node order (local vertex ids, owned edges first;
    an edge belongs to the rank owning its endpoint of smaller GLOBAL id - fs_space_create applies the same rule).

    Why local data suffices: an edge this rank owns has its smaller endpoint here, so every cell holding it touches an owned
    vertex and is a local cell; the neighbour q stores that edge as a ghost exactly when one of those cells also touches a
    q-owned vertex - a property of local cells.  Both sides of an exchange order its nodes the same way: vertices by global id,
    then edges by the global ids (g0, g1) of their end points."""
    cells = np.asarray(cells, dtype=np.int64)
    gid = np.asarray(gid, dtype=np.int64)
    owner = np.asarray(owner)
    nv = len(gid)
    nvo = int((owner == rank).sum())
    if not np.all(owner[:nvo] == rank):
        raise AssertionError("local vertices are not numbered owned-first")
    le = np.asarray(device_edges, dtype=np.int64).reshape(-1, 2)
    ga, gb = gid[le[:, 0]], gid[le[:, 1]]
    first_is_small = ga < gb
    g0, g1 = np.where(first_is_small, ga, gb), np.where(first_is_small, gb, ga)
    e_owner = np.where(first_is_small, owner[le[:, 0]], owner[le[:, 1]])
    neo = int((e_owner == rank).sum())
    if not (np.all(e_owner[:neo] == rank) and np.all(e_owner[neo:] != rank)):
        raise AssertionError("device edge table is not ordered owned-first")
    ne = len(le)
    node_of_vertex = np.where(np.arange(nv) < nvo, np.arange(nv), np.arange(nv) + neo)
    node_of_edge = np.where(np.arange(ne) < neo, nvo + np.arange(ne), nv + np.arange(ne))
    ng = int(n_global_vertices)
    lkey = g0 * ng + g1
    lsort = np.argsort(lkey)
    cell_owner = owner[cells]
    mine_in_cell = cell_owner == rank
    send_lists, recv_lists = [], []
    for q in neighbors:
        gv = np.nonzero(owner == q)[0]
        gv = gv[np.argsort(gid[gv])]
        ge = np.nonzero(e_owner == q)[0]
        ge = ge[np.argsort(lkey[ge])]
        recv_lists.append(np.concatenate([node_of_vertex[gv], node_of_edge[ge]]).astype(np.int32))
        sel = (cell_owner == q).any(axis=1)
        touch, tmine = cells[sel], mine_in_cell[sel]
        sv = np.unique(touch[tmine])                               # my vertices in cells that are local to q as well
        sv = sv[np.argsort(gid[sv])]
        a = np.concatenate([touch[:, i] for i, _ in _TET_EDGES])
        b = np.concatenate([touch[:, j] for _, j in _TET_EDGES])
        ka, kb = gid[a], gid[b]
        small_a = ka < kb
        own = np.where(small_a, owner[a], owner[b]) == rank
        k = np.unique(np.where(small_a, ka, kb)[own] * ng + np.where(small_a, kb, ka)[own])
        pos = lsort[np.searchsorted(lkey[lsort], k)] if len(k) else np.zeros(0, dtype=np.int64)
        if len(k) and not np.array_equal(lkey[pos], k):
            raise AssertionError("an edge to send is not in the device's edge table")
        send_lists.append(np.concatenate([node_of_vertex[sv], node_of_edge[pos]]).astype(np.int32))
    plan = P2Plan(nvo + neo, None, [int(q) for q in neighbors], send_lists, recv_lists)
    plan.node_of_vertex, plan.node_of_edge = node_of_vertex, node_of_edge
    plan.edge_gid_pairs = (g0, g1)          # global end points of every local edge, in device edge order
    plan.n_owned_vertices, plan.n_owned_edges = nvo, neo
    return plan
