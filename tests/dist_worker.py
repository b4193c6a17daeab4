"""Worker of tests/test_distributed_cpu.py: one process per rank, gloo backend.

Each rank builds its share of the problem from the partition plan (the product's host logic,
fenicssolver_amd/partition.py), assembles its owned rows with the oracle (the checker), and runs
the SAME distributed recurrence the HIP driver runs (fs_krylov.hip): halo exchange of z, local
SpMV + three local dots, ONE all-reduce of three doubles, fused update.  Rank 0 gathers the owned
solutions by global vertex id and writes them for the parent test to compare with the 1-rank oracle.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fenicssolver_amd import partition  # noqa: E402
from oracle import fem_oracle as fo  # noqa: E402


def halo_exchange(vec, n_owned, neighbors, send_lists, recv_counts):
    """ncclGroupStart; Send/Recv per neighbour; ncclGroupEnd  ==  isend/irecv + wait."""
    reqs, bufs = [], []
    off = n_owned
    for q, sl, rc in zip(neighbors, send_lists, recv_counts):
        sb = torch.from_numpy(np.ascontiguousarray(vec[sl]))
        rb = torch.empty(rc, dtype=torch.float64)
        reqs.append(dist.isend(sb, q))
        reqs.append(dist.irecv(rb, q))
        bufs.append((off, rb))
        off += rc
    for r in reqs:
        r.wait()
    for o, rb in bufs:
        vec[o:o + len(rb)] = rb.numpy()


def main():
    mode, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nx, ny, nz = 5, 4, 9
    p1 = (1.0, 0.8, 1.8)
    co, ce = fo.box_mesh((0, 0, 0), p1, nx, ny, nz)
    axis = 0
    if mode == "slab":
        zr = partition.slab_ranges(nz + 1, world)[rank]
        lay = partition.slab_layout(nx, ny, nz, zr, rank, world)
        l2g, n_owned = lay["l2g"], lay["n_owned"]
        neighbors, send_lists, recv_counts = lay["neighbors"], lay["send_lists"], lay["recv_counts"]
        dofs, vals = partition.slab_dirichlet(nx, ny, nz, lay, axis)
        g2l = np.full(len(co), -1, dtype=np.int64)
        g2l[l2g] = np.arange(len(l2g))
        P = (nx + 1) * (ny + 1)
        owned = (ce >= zr[0] * P) & (ce < zr[1] * P)
        cells_local = g2l[ce[owned.any(axis=1)].astype(np.int64)]
    else:
        owner = partition.rcb_owner(co, world) if mode == "rcb" else partition.slab_owner(co, world, axis=2)
        part = partition.build_local_part(ce, owner, rank)
        l2g, n_owned, cells_local = part.l2g, part.n_owned, part.cells
        neighbors, send_lists, recv_counts = part.neighbors, part.send_lists, part.recv_counts
        xl = co[l2g]
        lo, hi = np.nonzero(xl[:, axis] == 0.0)[0], np.nonzero(xl[:, axis] == p1[axis])[0]
        dofs = np.concatenate([lo, hi])
        vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
    n_local = len(l2g)
    assert cells_local.min() >= 0
    # local rows of the owned vertices (columns in local numbering), Dirichlet applied with ghost values
    A = fo.assemble_p1_scalar(co[l2g], cells_local, 20.0)
    A, b = fo.apply_dirichlet(A, np.zeros(n_local), dofs, vals, symmetric=True)
    A = A.tocsr()[:n_owned]
    b = b[:n_owned]
    dinv = 1.0 / A.diagonal()

    def allreduce(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t)
        return t.numpy()

    bb = allreduce([float(b @ b)])[0]
    thresh = (1e-10) ** 2 * bb
    if len(sys.argv) > 3 and sys.argv[3] == "pipelined":
        # fs_krylov_opts.pipelined: the sums of (r, w) are all-reduced asynchronously while n = Ah w (halo of w + local
        # product) runs; scaled system, the exchanged vector is w.  The ghost scale factors travel through the halo once.
        dfull = np.zeros(n_local)
        dfull[:n_owned] = A.diagonal()
        halo_exchange(dfull, n_owned, neighbors, send_lists, recv_counts)
        scl = 1.0 / np.sqrt(dfull)
        Ah = sp.diags(scl[:n_owned]) @ A @ sp.diags(scl)
        dv = dfull[:n_owned]
        x = np.zeros(n_owned)
        r = np.zeros(n_local)
        r[:n_owned] = scl[:n_owned] * b
        halo_exchange(r, n_owned, neighbors, send_lists, recv_counts)
        w = np.zeros(n_local)
        w[:n_owned] = Ah @ r
        r = r[:n_owned].copy()
        z, s, p = np.zeros(n_owned), np.zeros(n_owned), np.zeros(n_owned)
        gamma_old = alpha_old = 1.0
        it = 0
        while True:
            ro, wo = r, w[:n_owned]
            t = torch.tensor([float(ro @ ro), float(wo @ ro), float(dv @ (ro * ro))], dtype=torch.float64)
            work = dist.all_reduce(t, async_op=True)                   # ... in flight while the product runs
            halo_exchange(w, n_owned, neighbors, send_lists, recv_counts)
            nv = Ah @ w
            work.wait()
            g, d, rho = t.numpy()
            if rho <= thresh or it >= 2000:
                break
            beta = 0.0 if it == 0 else g / gamma_old
            alpha = g / d if it == 0 else g / (d - beta * g / alpha_old)
            z = nv + beta * z
            s = wo + beta * s
            p = ro + beta * p
            x += alpha * p
            r = ro - alpha * s
            w[:n_owned] = wo - alpha * z
            gamma_old, alpha_old = g, alpha
            it += 1
        x = scl[:n_owned] * x
        return finish(co, l2g, n_owned, x, it, n_local, send_lists, rank, world, out)
    x = np.zeros(n_owned)
    r = b.copy()
    z = np.zeros(n_local)
    z[:n_owned] = dinv * r
    p = np.zeros(n_owned)
    s = np.zeros(n_owned)
    gamma_old = alpha_old = 1.0
    it = 0
    while True:
        halo_exchange(z, n_owned, neighbors, send_lists, recv_counts)
        w = A @ z
        g, d, rho = allreduce([float(r @ z[:n_owned]), float(w @ z[:n_owned]), float(r @ r)])
        if rho <= thresh or it >= 2000:
            break
        beta = 0.0 if it == 0 else g / gamma_old
        alpha = g / d if it == 0 else g / (d - beta * g / alpha_old)
        p = z[:n_owned] + beta * p
        s = w + beta * s
        x += alpha * p
        r -= alpha * s
        z[:n_owned] = dinv * r
        gamma_old, alpha_old = g, alpha
        it += 1
    return finish(co, l2g, n_owned, x, it, n_local, send_lists, rank, world, out)


def finish(co, l2g, n_owned, x, it, n_local, send_lists, rank, world, out):
    gathered = [None] * world
    dist.all_gather_object(gathered, (l2g[:n_owned], x, it, n_local, [len(sl) for sl in send_lists]))
    if rank == 0:
        full = np.full(len(co), np.nan)
        for gid, xv, _, _, _ in gathered:
            assert np.all(np.isnan(full[gid]))          # every vertex owned exactly once
            full[gid] = xv
        np.savez(out, x=full, iterations=gathered[0][2], n_local=[g[3] for g in gathered])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
