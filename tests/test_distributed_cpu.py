"""world_size-2/3 `gloo` tests of the N>1 path on CPU: partition plans + the distributed
single-reduction CG sequence (halo -> SpMV+dots -> one all-reduce -> update) reproduce the
single-rank solution; the closed-form slab plan equals the general plan."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fenicssolver_amd import partition
from oracle import fem_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_plan_equals_general_plan():
    nx, ny, nz = 4, 3, 10
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 2), nx, ny, nz)
    P = (nx + 1) * (ny + 1)
    for world in (2, 3, 4):
        ranges = partition.slab_ranges(nz + 1, world)
        owner = np.zeros(len(co), dtype=np.int32)
        for r, (zb, ze) in enumerate(ranges):
            owner[zb * P:ze * P] = r
        seen = np.zeros(len(co), dtype=int)
        for r, zr in enumerate(ranges):
            lay = partition.slab_layout(nx, ny, nz, zr, r, world)
            part = partition.build_local_part(ce, owner, r)
            assert np.array_equal(lay["l2g"], part.l2g) and lay["n_owned"] == part.n_owned
            assert lay["neighbors"] == part.neighbors and lay["recv_counts"] == part.recv_counts
            for a, b in zip(lay["send_lists"], part.send_lists):
                assert np.array_equal(a, b)
            seen[part.l2g[:part.n_owned]] += 1
            # ghost order on the receiver == send order on the owner (ascending global id)
            off = part.n_owned
            for q, cnt in zip(part.neighbors, part.recv_counts):
                other = partition.build_local_part(ce, owner, q)
                sl = other.send_lists[other.neighbors.index(r)]
                assert np.array_equal(other.l2g[sl], part.l2g[off:off + cnt])
                off += cnt
            # every cell touching an owned vertex is local, and only those
            touch = (owner[ce.astype(np.int64)] == r).any(axis=1)
            assert np.array_equal(part.cell_gids, np.nonzero(touch)[0])
        assert np.all(seen == 1)


def test_owner_functions_cover_all_vertices(data_dir):
    co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
    for world in (2, 5, 8):
        o = partition.rcb_owner(co, world)
        cnt = np.bincount(o, minlength=world)
        assert cnt.min() > 0 and cnt.max() - cnt.min() <= world
        o2 = partition.slab_owner(co, world, axis=2)
        assert set(np.unique(o2)) == set(range(world))
    # unstructured plan: symmetric neighbourhoods, consistent orders
    owner = partition.rcb_owner(co, 4)
    parts = [partition.build_local_part(ce, owner, r) for r in range(4)]
    for p in parts:
        off = p.n_owned
        for q, cnt in zip(p.neighbors, p.recv_counts):
            sl = parts[q].send_lists[parts[q].neighbors.index(p.rank)]
            assert np.array_equal(parts[q].l2g[sl], p.l2g[off:off + cnt])
            off += cnt
        assert off == p.n_local


def _run(world, mode, tmp_path, recurrence="single_reduction"):
    out = str(tmp_path / ("dist_%s_%s_%d.npz" % (mode, recurrence, world)))
    port = 29500 + (os.getpid() % 2000) + world
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), mode, out, recurrence]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("world,mode", [(2, "slab"), (3, "slab"), (2, "rcb"), (3, "general_slab")])
def test_distributed_cg_matches_single_rank(world, mode, tmp_path):
    r = _run(world, mode, tmp_path)
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 1.8), 5, 4, 9)
    A = fo.assemble_p1_scalar(co, ce, 20.0)
    lo, hi = np.nonzero(co[:, 0] == 0.0)[0], np.nonzero(co[:, 0] == 1.0)[0]
    Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), np.concatenate([lo, hi]),
                                np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), True)
    x1, it1, _ = fo.pcg_jacobi_single_reduction(Ab, bb, rtol=1e-10)
    assert not np.isnan(r["x"]).any()
    assert abs(int(r["iterations"]) - it1) <= 1
    # reduction order differs between 1 and N ranks: agreement to 1e-12 relative
    assert np.abs(r["x"] - x1).max() <= 1e-10 * np.abs(x1).max()
    assert np.abs(r["x"] - (350.0 - 50.0 * co[:, 0])).max() <= 1e-6


@pytest.mark.parametrize("world,mode", [(2, "slab"), (3, "slab"), (3, "rcb")])
def test_distributed_pipelined_cg_matches_single_rank(world, mode, tmp_path):
    """The pipelined recurrence (all-reduce in flight under the product) on 2-3 gloo ranks: the single-rank oracle's solution
    <= 1e-9 relative, its iteration count within +2 - and the oracle's own restatement of the pipelined recurrence agrees
    with its single-reduction one."""
    r = _run(world, mode, tmp_path, recurrence="pipelined")
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 1.8), 5, 4, 9)
    A = fo.assemble_p1_scalar(co, ce, 20.0)
    lo, hi = np.nonzero(co[:, 0] == 0.0)[0], np.nonzero(co[:, 0] == 1.0)[0]
    Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), np.concatenate([lo, hi]),
                                np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), True)
    x1, it1, h1 = fo.pcg_jacobi_single_reduction(Ab, bb, rtol=1e-10)
    x2, it2, h2 = fo.pcg_jacobi_pipelined(Ab, bb, rtol=1e-10)
    assert -1 <= it2 - it1 <= 2 and np.abs(x2 - x1).max() <= 1e-9 * np.abs(x1).max()
    m = min(len(h1), len(h2), 15)
    assert np.allclose(h1[:m], h2[:m], rtol=1e-8)
    assert not np.isnan(r["x"]).any()
    assert -1 <= int(r["iterations"]) - it1 <= 2
    assert np.abs(r["x"] - x1).max() <= 1e-9 * np.abs(x1).max()


def _device_like_edges(part, owner, rank):
    """The edge table fs_space_create builds for a part: unique local vertex pairs in key order, owned edges first
    (an edge belongs to the rank owning its endpoint of smaller global id)."""
    c = part.cells.astype(np.int64)
    pairs = np.concatenate([np.sort(c[:, [i, j]], axis=1) for i, j in partition._TET_EDGES])
    pairs = np.unique(pairs, axis=0)                          # lexicographic (v0, v1) = the ungrouped key order
    g = part.l2g[pairs]
    vmin = np.where(g[:, 0] < g[:, 1], pairs[:, 0], pairs[:, 1])
    ghost = vmin >= part.n_owned
    return np.concatenate([pairs[~ghost], pairs[ghost]])


@pytest.mark.parametrize("world", [2, 3, 4])
def test_p2_exchange_plan_is_consistent_between_ranks(world):
    """partition.build_p2_plan derives both sides of every exchange without communication: what rank p sends to q
    must be, value for value, what q expects from p - compared through global node ids - and every ghost node of
    every rank must be covered exactly once."""
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 2.0), 3, 4, 9)
    nvg = len(co)
    edges_g, _ = fo.edge_numbering(ce)
    owner = partition.slab_owner(co, world, axis=2)
    plans, parts = [], []
    for r in range(world):
        part = partition.build_local_part(ce, owner, r)
        le = _device_like_edges(part, owner, r)
        plans.append(partition.build_p2_plan(ce, owner, r, part, le, edges_g))
        parts.append(part)
    owned_total = 0
    seen = np.zeros(nvg + len(edges_g), dtype=int)
    for r, pl in enumerate(plans):
        owned_total += pl.n_owned_nodes
        seen[pl.l2g_nodes[:pl.n_owned_nodes]] += 1
        ghosts = np.concatenate(pl.recv_lists) if pl.recv_lists else np.zeros(0, dtype=np.int64)
        assert sorted(ghosts.tolist()) == list(range(pl.n_owned_nodes, len(pl.l2g_nodes)))      # each ghost exactly once
        for qi, q in enumerate(pl.neighbors):
            back = plans[q].neighbors.index(r)
            sent_gids = plans[q].l2g_nodes[plans[q].send_lists[back]]
            want_gids = pl.l2g_nodes[pl.recv_lists[qi]]
            assert np.array_equal(sent_gids, want_gids)
            assert np.all(plans[q].send_lists[back] < plans[q].n_owned_nodes)
    assert owned_total == nvg + len(edges_g) and np.all(seen == 1)         # the owned sets partition the P2 nodes


def _emulated_device_edges(part, owner, n_global):
    """What fs_space_create numbers on a part: the edges of the local cells, owned ones first (owner = owner of the end point
    of smaller global id), each group ascending by (local v0, local v1)."""
    c = part.cells.astype(np.int64)
    pairs = np.concatenate([np.sort(c[:, [i, j]], axis=1) for i, j in partition._TET_EDGES])
    pairs = np.unique(pairs, axis=0)
    g = part.l2g[pairs]
    small = np.where(g[:, 0] < g[:, 1], pairs[:, 0], pairs[:, 1])
    mine = owner[part.l2g[small]] == part.rank
    return np.concatenate([pairs[mine], pairs[~mine]]).astype(np.int32)


@pytest.mark.parametrize("mesh_kind,world", [("box", 2), ("box", 3), ("data", 4)])
def test_local_p2_plan_equals_the_plan_from_the_global_mesh(data_dir, mesh_kind, world):
    """partition.build_p2_plan_local sees only one rank's cells; it must produce the exchanges build_p2_plan derives from the
    global mesh, and sender and receiver must agree on the order of every exchange."""
    if mesh_kind == "box":
        co, ce = fo.box_mesh((0, 0, 0), (1, 1, 2), 3, 4, 9)
        owner = partition.slab_owner(co, world, axis=2)
    else:
        co, ce = fo.read_dolfin_xml_mesh(os.path.join(data_dir, "mesh.xml"))
        owner = partition.rcb_owner(co, world)
    ce = np.sort(ce, axis=1)
    ng = len(co)
    ge = np.unique(np.concatenate([ce[:, [i, j]] for i, j in partition._TET_EDGES]), axis=0)      # global edge table (v0 < v1)
    plans, parts = [], []
    for r in range(world):
        part = partition.build_local_part(ce, owner, r)
        dev_edges = _emulated_device_edges(part, owner, ng)
        ref = partition.build_p2_plan(ce, owner, r, part, dev_edges, ge)
        loc = partition.build_p2_plan_local(part.cells, part.l2g, owner[part.l2g], r, part.neighbors, dev_edges, ng)
        assert loc.n_owned_nodes == ref.n_owned_nodes and loc.neighbors == ref.neighbors
        for a, b in zip(loc.send_lists, ref.send_lists):
            assert np.array_equal(a, b)
        for a, b in zip(loc.recv_lists, ref.recv_lists):
            assert np.array_equal(a, b)
        plans.append((loc, ref))
        parts.append(part)
    # the k-th node a rank sends is the k-th node its neighbour receives (compared through global node ids)
    for r in range(world):
        loc, ref = plans[r]
        for qi, q in enumerate(loc.neighbors):
            other, oref = plans[q]
            back = other.neighbors.index(r)
            assert np.array_equal(ref.l2g_nodes[loc.send_lists[qi]], oref.l2g_nodes[other.recv_lists[back]])


@pytest.mark.parametrize("world", [2, 3, 5])
def test_two_layer_parts_for_interior_facet_integrals(world):
    """partition.build_local_part(face_pairs=...) (round 4: the interior-penalty term under decomposition).  For every rank: (1) every
    interior facet one of whose two cells holds an owned vertex has BOTH cells in the part - the rows of the owned vertices then get
    all their dS contributions from local data; (2) the exchange plan is consistent: what rank r sends to q is exactly q's ghosts
    owned by r, in q's ghost order."""
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 3.0), 3, 2, 9)
    ce = ce.astype(np.int64)
    _, cf, _ = fo.facet_numbering(ce)
    order = np.argsort(cf.ravel(), kind="stable")
    fid = cf.ravel()[order]
    dup = np.nonzero(fid[1:] == fid[:-1])[0]
    pairs = np.stack([order[dup] // 4, order[dup + 1] // 4], axis=1)
    owner = partition.slab_owner(co, world, axis=2)
    parts = [partition.build_local_part(ce, owner, r, face_pairs=pairs) for r in range(world)]
    one_layer = [partition.build_local_part(ce, owner, r) for r in range(world)]
    for r, part in enumerate(parts):
        local = np.zeros(len(ce), dtype=bool)
        local[part.cell_gids] = True
        touches = (owner[ce] == r).any(axis=1)
        need = touches[pairs[:, 0]] | touches[pairs[:, 1]]
        assert local[pairs[need]].all()
        assert len(part.cell_gids) > len(one_layer[r].cell_gids) or world == 1
        assert np.array_equal(part.l2g[:part.n_owned], one_layer[r].l2g[:one_layer[r].n_owned])      # the owned rows do not change
        off = part.n_owned
        for q, cnt in zip(part.neighbors, part.recv_counts):
            ghosts_from_q = part.l2g[off:off + cnt]
            off += cnt
            pq = parts[q]
            k = pq.neighbors.index(r)
            assert np.array_equal(pq.l2g[pq.send_lists[k]], ghosts_from_q)
        assert off == part.n_local


@pytest.mark.parametrize("world,axis", [(2, 2), (3, 2), (4, 2), (3, 0)])
def test_parts_of_a_periodic_space_hold_the_masters_of_their_slaves(world, axis):
    """partition.build_local_part(tied=...) (round 4: periodic_boundary under decomposition).  Slaves are owned by their masters'
    rank; every part holds the master of every slave among its vertices (as a ghost without cells where the master is at the far
    side of the domain); the exchange plan stays consistent (what r sends to q is q's ghosts owned by r, in q's ghost order); and
    the FOLDED global operator, restricted to the rows a rank owns, only has columns that are local to that rank."""
    nx, ny, nz = 3, 2, 9
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 3.0), nx, ny, nz)
    ce = ce.astype(np.int64)
    length = (1.0, 0.8, 3.0)[axis]
    slaves = np.nonzero(np.isclose(co[:, axis], length))[0]
    key = lambda idx: [tuple(np.round(np.delete(co[i], axis), 9)) for i in idx]
    lookup = {k: i for k, i in zip(key(np.nonzero(np.isclose(co[:, axis], 0.0))[0]), np.nonzero(np.isclose(co[:, axis], 0.0))[0])}
    masters = np.array([lookup[k] for k in key(slaves)])
    owner = np.array(partition.slab_owner(co, world, axis=2))
    owner[slaves] = owner[masters]
    parts = [partition.build_local_part(ce, owner, r, tied=(slaves, masters)) for r in range(world)]
    A = fo.assemble_p1_scalar(co, ce, 1.0, mass_coef=1.0)
    Af, _ = fo.periodic_fold(A, np.zeros(len(co)), slaves, masters, 1)
    Af = Af.tocsr()
    covered = np.zeros(len(co), dtype=int)
    for r, part in enumerate(parts):
        g2l = part.g2l(len(co))
        covered[part.l2g[:part.n_owned]] += 1
        loc_slaves = np.intersect1d(part.l2g, slaves)
        assert (g2l[masters[np.searchsorted(slaves, loc_slaves)]] >= 0).all()
        rows = part.l2g[:part.n_owned]
        cols = np.unique(Af[rows].indices)
        assert (g2l[cols] >= 0).all(), "a folded row of rank %d couples with a vertex outside its part" % r
        off = part.n_owned
        for q, cnt in zip(part.neighbors, part.recv_counts):
            ghosts_from_q = part.l2g[off:off + cnt]
            off += cnt
            pq = parts[q]
            assert np.array_equal(pq.l2g[pq.send_lists[pq.neighbors.index(r)]], ghosts_from_q)
        assert off == part.n_local
    assert np.all(covered == 1)


@pytest.mark.parametrize("world,axis", [(2, 2), (3, 2), (3, 1), (4, 0)])
def test_periodic_cg2_parts_own_slave_and_master_edges_together(world, axis):
    """Periodic constraints on CG2 spaces under decomposition (round 5; reference SolverBase.py:263-270 is degree-agnostic).  The
    master of a slave EDGE node is an edge node: build_local_part(tied_cells=True) brings the cells around the masters of a part's
    slave vertices, and with ORDER ids (a slave right behind its master) as the ids that decide who owns an edge, a slave edge and
    its master edge are owned by one rank.  Held: every local slave node has its master in the part; slave and master edge nodes
    have one owner; both sides of every exchange agree, node for node; the owned sets partition the nodes; and every cell pattern
    coupling of the folded operator with an owned row is local."""
    nx, ny, nz = 3, 3, 8
    box = (1.0, 0.9, 2.4)
    co, ce = fo.box_mesh((0, 0, 0), box, nx, ny, nz)
    ce = ce.astype(np.int64)
    nvg = len(co)
    slaves = np.nonzero(np.isclose(co[:, axis], box[axis]))[0]
    key = lambda idx: [tuple(np.round(np.delete(co[i], axis), 9)) for i in idx]
    m0 = np.nonzero(np.isclose(co[:, axis], 0.0))[0]
    lookup = {k: i for k, i in zip(key(m0), m0)}
    masters = np.array([lookup[k] for k in key(slaves)])
    cd, edges_g = fo.p2_cell_dofs(nvg, ce)
    edges_g = np.asarray(edges_g, dtype=np.int64)
    n_nodes = nvg + len(edges_g)
    # node-level pairs: an edge whose end points fold onto another edge's is tied to it
    fold_v = np.arange(nvg, dtype=np.int64)
    fold_v[slaves] = masters
    ekey = {tuple(sorted(e)): k for k, e in enumerate(edges_g.tolist())}
    node_fold = np.arange(n_nodes, dtype=np.int64)
    node_fold[slaves] = masters
    for k, (a, b) in enumerate(edges_g.tolist()):
        fa, fb = fold_v[a], fold_v[b]
        if (fa, fb) != (a, b) and fa != fb and tuple(sorted((fa, fb))) in ekey:
            node_fold[nvg + k] = nvg + ekey[tuple(sorted((fa, fb)))]
    assert (node_fold[nvg:] != np.arange(nvg, n_nodes)).sum() > 0
    owner = np.array(partition.slab_owner(co, world, axis=2))
    owner[slaves] = owner[masters]
    order_id = 2 * fold_v
    order_id[slaves] += 1
    master_of = np.full(nvg, -1, dtype=np.int64)
    master_of[slaves] = masters
    cells_of = lambda q: partition._local_cell_mask(ce, owner, q, None, master_of)
    plans, parts = [], []
    for r in range(world):
        part = partition.build_local_part(ce, owner, r, tied=(slaves, masters), tied_cells=True)
        assert np.array_equal(np.sort(part.cell_gids), np.nonzero(cells_of(r))[0])
        # the device's edge table: unique local vertex pairs, owned first - an edge belongs to the owner of its end of smaller ORDER id
        c = part.cells.astype(np.int64)
        pairs = np.unique(np.concatenate([np.sort(c[:, [i, j]], axis=1) for i, j in partition._TET_EDGES]), axis=0)
        oid = order_id[part.l2g[pairs]]
        vmin = np.where(oid[:, 0] < oid[:, 1], pairs[:, 0], pairs[:, 1])
        ghost = vmin >= part.n_owned
        le = np.concatenate([pairs[~ghost], pairs[ghost]])
        plans.append(partition.build_p2_plan(ce, owner, r, part, le, edges_g, order_id=order_id, part_cells_of=cells_of))
        parts.append(part)
    node_owner = np.full(n_nodes, -1)
    for r, pl in enumerate(plans):
        assert np.all(node_owner[pl.l2g_nodes[:pl.n_owned_nodes]] == -1)
        node_owner[pl.l2g_nodes[:pl.n_owned_nodes]] = r
    assert np.all(node_owner >= 0)                                         # the owned sets partition the nodes
    assert np.array_equal(node_owner, node_owner[node_fold])               # a slave lives with its master - edges too
    cells_nodes = cd.astype(np.int64)
    for r, pl in enumerate(plans):
        g2l = np.full(n_nodes, -1, dtype=np.int64)
        g2l[pl.l2g_nodes] = np.arange(len(pl.l2g_nodes))
        local = pl.l2g_nodes
        assert (g2l[node_fold[local]] >= 0).all(), "a local slave node's master is not in the part of rank %d" % r
        ghosts = np.concatenate(pl.recv_lists) if pl.recv_lists else np.zeros(0, dtype=np.int64)
        assert sorted(ghosts.tolist()) == list(range(pl.n_owned_nodes, len(pl.l2g_nodes)))
        for qi, q in enumerate(pl.neighbors):
            back = plans[q].neighbors.index(r)
            assert np.array_equal(plans[q].l2g_nodes[plans[q].send_lists[back]], pl.l2g_nodes[pl.recv_lists[qi]])
            assert np.all(plans[q].send_lists[back] < plans[q].n_owned_nodes)
        # folded pattern: rows fold(i), columns fold(j) for all (i, j) of a cell; those with an owned row must be local
        rows = node_fold[np.repeat(cells_nodes, 10, axis=1).ravel()]
        cols = node_fold[np.tile(cells_nodes, (1, 10)).ravel()]
        mine = node_owner[rows] == r
        assert (g2l[cols[mine]] >= 0).all(), "a folded row of rank %d couples with a node outside its part" % r

