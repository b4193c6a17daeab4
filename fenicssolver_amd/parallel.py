"""One process per GPU: process-group plumbing for the solver API.

The reference becomes parallel by being started under ``mpirun`` - DOLFIN partitions the mesh
and PETSc communicates (SolverBase.py:102-118, 634).  Here a script that uses the solver
classes is started with ``python -m fenicssolver_amd.launch --nproc N script.py`` (or any
launcher that exports RANK / WORLD_SIZE / LOCAL_RANK, e.g. the driver's elastic launcher, mpirun, srun):
every rank owns a slab of the vertices (partition.py), assembles and solves its rows on its own
MI355X and exchanges halos / dot products over RCCL; the solution is gathered so that
``solver.result`` holds the full field on every rank, as DOLFIN's ``Function`` does for the
part a rank can see.

No ML framework, no MPI: the RCCL unique id travels through rendezvous.py, everything after that
(barrier, gathers) runs over the communicator itself.
"""
from __future__ import annotations

import numpy as np

from . import rendezvous

_state = {"ready": False}

# Tests flip this to send a single process through the decomposition code (one part, no communicator).
FORCE_DECOMPOSED_PATH = False


def world():
    return rendezvous.world()


def active():
    return world()[1] > 1 or FORCE_DECOMPOSED_PATH


def ensure_comm():
    """Select this rank's GPU (LOCAL_RANK) and bring up RCCL (idempotent)."""
    from . import backend
    rank, size, local_rank = world()
    if _state["ready"]:
        return rank, size
    backend.init(local_rank if size > 1 else 0)
    if size > 1:
        uid = rendezvous.exchange_unique_id(rank, size, backend.comm_unique_id)
        backend.comm_init(size, rank, uid)
        rendezvous.cleanup(rank, size)
    _state["ready"] = True
    return rank, size


def barrier():
    from . import backend
    backend.synchronize()
    if world()[1] > 1:
        backend.comm_allreduce_sum([0.0])


def max_over_ranks(value):
    from . import backend
    if world()[1] == 1:
        return float(value)
    return float(backend.comm_allgather([float(value)], 1).max())


def finalize():
    from . import backend
    if _state["ready"] and world()[1] > 1:
        backend.comm_finalize()
    _state["ready"] = False


_gid_cache = {}


def allgather_index_lists(ids):
    """Every rank's integer list (exact below 2^53: they travel as doubles through fs_comm_allgather)."""
    from . import backend
    ids = np.asarray(ids, dtype=np.int64)
    counts = backend.comm_allgather([float(len(ids))], 1)[:, 0].astype(np.int64)
    n_max = max(int(counts.max()), 1)
    got = backend.comm_allgather(ids.astype(np.float64), n_max)
    return [got[r, :counts[r]].astype(np.int64) for r in range(len(counts))]


def allgather_values(values):
    """Every rank's float list (one all-gather of the padded lists)."""
    from . import backend
    v = np.asarray(values, dtype=np.float64).ravel()
    counts = backend.comm_allgather([float(len(v))], 1)[:, 0].astype(np.int64)
    got = backend.comm_allgather(v, max(int(counts.max()), 1))
    return [got[r, :counts[r]].copy() for r in range(len(counts))]


def gather_owned(values_owned, gids_owned, n_global, ncomp=1):
    """Owned dof values of every rank -> the full vector (on every rank).  The values travel in one all-gather of the
    communicator (RCCL; fs_comm_allgather); the owners' global ids are exchanged once per layout and cached."""
    rank, size, _ = world()
    vals = np.asarray(values_owned, dtype=np.float64).reshape(-1, ncomp)
    gids = np.asarray(gids_owned)
    out = np.full((n_global, ncomp), np.nan)
    if size == 1:
        out[gids] = vals
    else:
        from . import backend
        key = (int(n_global), len(gids), int(gids[0]) if len(gids) else -1, int(gids[-1]) if len(gids) else -1)
        if key not in _gid_cache:
            if len(_gid_cache) > 16:
                _gid_cache.clear()
            _gid_cache[key] = allgather_index_lists(gids)
        parts = _gid_cache[key]
        n_max = max(len(g) for g in parts) * ncomp
        got = backend.comm_allgather(vals.reshape(-1), n_max)
        for r, g in enumerate(parts):
            out[g] = got[r, :len(g) * ncomp].reshape(-1, ncomp)
    if np.isnan(out).any():
        raise RuntimeError("gather_owned: some dofs are owned by no rank")
    return out.reshape(-1)


class _Part:
    pass


class LocalView:
    """The Localizer of a DISTRIBUTED mesh (fem.BoxMesh(distributed=True)): host arrays are already this rank's - owned
    entries first, ghosts after - so every map is the identity; what remains is to know which entries are owned."""
    is_local_view = True
    is_identity = True       # host order == device order (P1 on a distributed box); LocalNodeView (CG2) permutes

    def __init__(self, n_owned, n_local, n_cells, l2g, n_global, ncomp):
        self.n_owned, self.n_local, self.ncomp = int(n_owned), int(n_local), int(ncomp)
        self.l2g, self.n_global = np.asarray(l2g, dtype=np.int64), int(n_global)
        self.part = _Part()
        self.part.n_owned = self.n_owned
        self.part.cell_gids = np.arange(int(n_cells))        # "global" cell ids of the host mesh = its own (local) ids

    def owned_gids(self):
        return self.l2g[:self.n_owned]

    def cells(self, arr):
        return np.asarray(arr)

    def nodes(self, arr):
        return np.asarray(arr)

    def dofs(self, dofs, vals):
        return np.asarray(dofs, dtype=np.int32), np.asarray(vals, dtype=np.float64)

    def facets(self, tri):
        t = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        mask = (t < self.n_owned).any(axis=1)                  # facets that touch a row of this rank
        return t[mask].astype(np.int32), mask

    def spec(self, spec):
        return spec


class LocalNodeView(LocalView):
    """The view of a CG2 space on a DISTRIBUTED mesh.  Host arrays are this rank's (local vertices, then the edges of the
    local cells in the host's edge order); the device numbers the same nodes [owned vertices | owned edges | ghost vertices |
    ghost edges] (fs_space_create), so every map is a permutation of local data - l2h[device node] = host node - and still
    nothing of global size exists on any rank.  Vertex ids (facet lists, cells) are the same on both sides."""
    is_identity = False

    def __init__(self, n_owned_nodes, l2h, n_cells, n_owned_vertices, ncomp, vertex_gids, n_global_vertices, edge_gid_pairs):
        self.l2h = np.asarray(l2h, dtype=np.int64)
        self.h2l = np.empty(len(self.l2h), dtype=np.int64)
        self.h2l[self.l2h] = np.arange(len(self.l2h))
        self.n_owned, self.n_local, self.ncomp = int(n_owned_nodes), len(self.l2h), int(ncomp)
        self.n_owned_vertices = int(n_owned_vertices)
        self.vertex_gids, self.n_global_vertices = np.asarray(vertex_gids, dtype=np.int64), int(n_global_vertices)
        self.edge_gid_pairs = edge_gid_pairs           # (g0, g1) of every local edge, in DEVICE edge order
        self.part = _Part()
        self.part.n_owned = self.n_owned_vertices
        self.part.cell_gids = np.arange(int(n_cells))

    def owned_gids(self):
        raise RuntimeError("a CG2 space on a distributed mesh has no global node numbering: use parallel.gather_nodes(u)")

    def nodes(self, arr):
        a = np.asarray(arr)
        if a.shape[0] == self.n_local:
            return a[self.l2h]
        return a.reshape(self.n_local, -1)[self.l2h].reshape(-1)

    def to_host(self, device_values):
        """device order (owned + ghost entries, ncomp per node) -> the host's node order"""
        a = np.asarray(device_values)[:self.n_local * self.ncomp].reshape(self.n_local, -1)
        return a[self.h2l].reshape(-1)

    def dofs(self, dofs, vals):
        d = np.asarray(dofs, dtype=np.int64)
        return (self.h2l[d // self.ncomp] * self.ncomp + d % self.ncomp).astype(np.int32), np.asarray(vals, dtype=np.float64)

    def facets(self, tri):
        t = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        mask = (t < self.n_owned_vertices).any(axis=1)
        return t[mask].astype(np.int32), mask

    def spec(self, spec):
        if isinstance(spec, tuple) and spec[0] == "nodal":
            return (spec[0], self.nodes(spec[1]))
        return spec


def gather_nodes(u):
    """The owned nodal values of a Function on a distributed mesh from every rank, with keys that name the nodes globally:
    (vertex_gids [n], vertex_values [n, ncomp], edge_keys [m, 2] = global end points (g0 < g1), edge_values [m, ncomp]).
    P1 spaces return empty edge arrays."""
    V = u.function_space()
    loc = V.localizer()
    nc = V._ncomp
    vals = np.asarray(u.vector().get_local()).reshape(-1, nc)
    if not isinstance(loc, LocalNodeView):
        if loc is None or not getattr(loc, "is_local_view", False):
            nv = V.mesh().num_vertices()
            ed = V.edge_nodes().astype(np.int64) if V.degree() == 2 else np.zeros((0, 2), dtype=np.int64)
            return np.arange(nv), vals[:nv], ed, vals[nv:]
        g = gather_owned(vals[:loc.n_owned].reshape(-1), loc.owned_gids(), loc.n_global, nc).reshape(-1, nc)
        return np.arange(loc.n_global), g, np.zeros((0, 2), dtype=np.int64), np.zeros((0, nc))
    dev = vals[loc.l2h]                                    # device order: [owned vertices | owned edges | ghosts]
    nvo = loc.n_owned_vertices
    neo = loc.n_owned - nvo
    vg = allgather_index_lists(loc.vertex_gids[:nvo])
    vv = allgather_values(dev[:nvo].reshape(-1))
    g0, g1 = loc.edge_gid_pairs
    e0, e1 = allgather_index_lists(g0[:neo]), allgather_index_lists(g1[:neo])
    ev = allgather_values(dev[nvo:nvo + neo].reshape(-1))
    return (np.concatenate(vg), np.concatenate(vv).reshape(-1, nc), np.stack([np.concatenate(e0), np.concatenate(e1)], axis=1),
            np.concatenate(ev).reshape(-1, nc))


def gather_function(u):
    """The global nodal array [n_global (, ncomp)] of a Function on a distributed mesh, on every rank (one all-gather);
    on a replicated mesh the Function already is global."""
    V = u.function_space()
    loc = V.localizer()
    vals = u.node_values()
    if loc is None or not getattr(loc, "is_local_view", False):
        return vals.copy()
    n = V._ncomp
    out = gather_owned(np.asarray(vals).reshape(-1, n)[:loc.n_owned].reshape(-1), loc.owned_gids(), loc.n_global, n)
    return out if n == 1 else out.reshape(-1, n)


class Localizer:
    """Maps global host arrays (per cell / per node / dof lists / facet lists) to one rank's part."""

    def __init__(self, part, n_global_vertices, ncomp, p2_plan=None, n_global_nodes=None):
        """p2_plan: partition.P2Plan of a CG2 space (node-level maps); P1 spaces use the vertex maps of the part."""
        self.part = part
        self.ncomp = ncomp
        self.n_global_vertices = n_global_vertices
        self.g2l_vertex = part.g2l(n_global_vertices)
        if p2_plan is None:
            self.n_global = n_global_vertices
            self.l2g = part.l2g
            self.n_owned = part.n_owned
        else:
            self.n_global = int(n_global_nodes)
            self.l2g = p2_plan.l2g_nodes
            self.n_owned = p2_plan.n_owned_nodes
        self.g2l = np.full(self.n_global, -1, dtype=np.int64)
        self.g2l[self.l2g] = np.arange(len(self.l2g))

    def owned_gids(self):
        """Global node ids of the rows this rank owns, in local order."""
        return self.l2g[:self.n_owned]

    def cells(self, arr):
        a = np.asarray(arr)
        return a[self.part.cell_gids]

    def nodes(self, arr):
        """Nodal array [n_global] or dof array [n_global*ncomp] -> local (owned + ghost) order."""
        a = np.asarray(arr)
        if a.shape[0] == self.n_global:
            return a[self.l2g]
        return a.reshape(self.n_global, -1)[self.l2g].reshape(-1)

    def dofs(self, dofs, vals):
        d = np.asarray(dofs, dtype=np.int64)
        node, comp = d // self.ncomp, d % self.ncomp
        loc = self.g2l[node]
        keep = loc >= 0
        return (loc[keep] * self.ncomp + comp[keep]).astype(np.int32), np.asarray(vals, dtype=np.float64)[keep]

    def tied_pairs(self, slaves, masters):
        """(slave, master) NODE pairs of a periodic constraint -> the pairs whose slave is local (owned or ghost), in local
        node numbers (the part holds the master of every local slave, partition.build_local_part(tied=...))."""
        sl, ma = self.g2l[np.asarray(slaves, dtype=np.int64)], self.g2l[np.asarray(masters, dtype=np.int64)]
        keep = sl >= 0
        if (ma[keep] < 0).any():
            raise RuntimeError("a local slave node's master is not in the part")
        return sl[keep].astype(np.int32), ma[keep].astype(np.int32)

    def facets(self, tri):
        """Facets with at least one owned vertex (their cell is local, so all three vertices are);
        returns (local vertex triples, mask into the input)."""
        t = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
        loc = self.g2l_vertex[t]              # facets are given, and handed to the device, as vertex triples
        mask = ((loc >= 0) & (loc < self.part.n_owned)).any(axis=1)
        sel = loc[mask]
        if (sel < 0).any():
            raise RuntimeError("a boundary facet touching an owned vertex has a non-local vertex")
        return sel.astype(np.int32), mask

    def spec(self, spec):
        """backend coefficient spec (None | number | (kind, array)) -> local."""
        if isinstance(spec, tuple):
            kind, arr = spec
            if kind in ("cell", "cell_tensor", "cell_qp"):
                return (kind, self.cells(arr))
            if kind == "nodal":
                return (kind, self.nodes(arr))
        return spec
