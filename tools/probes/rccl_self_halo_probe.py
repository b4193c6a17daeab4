"""Per-iteration cost of the distributed CG code path under REAL RCCL with the one communicator a 1-GPU box allows: a z-periodic slab
whose ghost planes are fed by the rank itself (grouped ncclSend / ncclRecv to self on the communication stream) + the 3-double
ncclAllReduce.  Shows what the host-side enqueue of the collectives and the extra launches cost per iteration against the
single-GPU loop (plain launches / hipGraph), for both recurrences."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
nx = ny = n
nz = n + 2
pl = (nx + 1) * (ny + 1)


def problem(with_comm):
    slab = B.DeviceMesh.box(nx, ny, nz, zplanes=(1, nz))
    V = B.DeviceSpace(slab, 1)
    n_own = (nz - 1) * pl
    if with_comm:
        V.set_halo([0, 0], [np.arange(n_own - pl, n_own, dtype=np.int32), np.arange(0, pl, dtype=np.int32)], [pl, pl])
    A = B.DeviceMatrix(V)
    A.assemble(stiffness=20.0, mass=1.0)
    b = B.DeviceVector(V.n_owned, np.random.default_rng(0).standard_normal(V.n_owned))
    return V, A, b


def run(tag, V, A, b, **kw):
    x = B.DeviceVector(V.n_local)
    B.krylov_solve(A, b, x, rtol=1e-10, max_iter=300, **kw)
    B.krylov_solve(A, b, x, rtol=1e-13, max_iter=600, **kw)             # (the first solve of a process launches plainly: the graphs are built here)
    t0 = time.perf_counter()
    st = B.krylov_solve(A, b, x, rtol=1e-13, max_iter=600, **kw)          # fixed 600 iterations
    t1 = time.perf_counter()
    print("%-44s %d iterations  %.1f us / iteration (product %.1f us, update %.1f us)" % (
        tag, st["iterations"], (t1 - t0) * 1e6 / max(st["iterations"], 1), st["spmv_ms"] * 1e3, st["update_ms"] * 1e3), flush=True)


what = sys.argv[2] if len(sys.argv) > 2 else "all"          # all | one | rccl | p2p  (one variant alone: for a kernel trace)
if what in ("all", "one"):
    V, A, b = problem(False)
    print("rows", V.n_owned)
    run("no communicator (hipGraph batches)", V, A, b)
uid = B.comm_unique_id()
B.comm_init(1, 0, uid)
V2, A2, b2 = problem(True)
if what in ("all", "rccl"):
    run("RCCL 1 rank, self-halo, single-reduction", V2, A2, b2, pipelined=False)
if what == "all":
    run("RCCL 1 rank, self-halo, pipelined", V2, A2, b2, pipelined=True)
    a_ms, h_ms = B.comm_benchmark(V2, 200)
    print("in-stream all-reduce of 3 doubles %.1f us, ghost refresh (2 planes of %d doubles to self) %.1f us" % (a_ms * 1e3, pl, h_ms * 1e3))
if what in ("all", "p2p"):
    V2.enable_p2p_halo(True)
    run("peer-to-peer halo (to self), single-reduction", V2, A2, b2, pipelined=False)
if what == "all":
    run("peer-to-peer halo (to self), pipelined", V2, A2, b2, pipelined=True)
    a_ms, h_ms = B.comm_benchmark(V2, 200)
    print("peer-to-peer ghost refresh %.1f us" % (h_ms * 1e3))
if what in ("all", "p2p"):
    V2.enable_p2p_halo(False)
B.comm_finalize()
