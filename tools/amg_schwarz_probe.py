"""BASELINE configs[2] (P1 vector cantilever, 5.1 M DOF) under domain decomposition: iteration count of solve_amg's CG
against the number of parts.  The bar lies along z here (59 x 59 x 472 cells of a 1 x 1 x 10 bar, clamped at z = 0, loaded
along x) so that the device's z-slabs cut it across its length - the partition a user would choose.

  python -m fenicssolver_amd.launch --nproc N [--devices 0,0,..] tools/amg_schwarz_probe.py [scale]

(on one GPU: FS_RCCL_PATH=tests/shim/libfakerccl.so and --devices 0,0,...: the iteration counts are those of N GPUs, the
times are not).  Prints one line: parts, dofs, iterations, converged, true residual, tip deflection."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from fenicssolver_amd import backend as B, partition, parallel  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nx = ny = max(int(round(59 * scale)), 4)
nz = max(int(round(472 * scale)), 16)
rank, world = parallel.ensure_comm()
E, nu = 2e11, 0.27
mu, lm = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
zr = partition.slab_ranges(nz + 1, world)[rank]
mesh = B.DeviceMesh.box(nx, ny, nz, (0, 0, 0), (1.0, 1.0, 10.0), zplanes=zr)
V = B.DeviceSpace(mesh, 3)
lay = partition.slab_layout(nx, ny, nz, zr, rank, world)
if world > 1:
    sends = [(np.asarray(l, dtype=np.int64)[:, None] * 3 + np.arange(3)[None, :]).reshape(-1).astype(np.int32) for l in lay["send_lists"]]
    V.set_halo(lay["neighbors"], sends, [c * 3 for c in lay["recv_counts"]])
A = B.DeviceMatrix(V)
b = B.DeviceVector(V.n_owned)
x = B.DeviceVector(V.n_local)
P = lay["plane_size"]
clamped = [lp * P + np.arange(P) for lp, iz in enumerate(lay["planes"]) if iz == 0]
dofs = (np.concatenate(clamped)[:, None] * 3 + np.arange(3)).ravel().astype(np.int32) if clamped else np.zeros(0, dtype=np.int32)
A.assemble(lame=(mu, lm))
B.assemble_vector(V, b, vector_value=(-7800 * 10.0, 0.0, 0.0))
A.apply_dirichlet(b, dofs, 0.0, True)
kw = {}
if os.environ.get("AMG_COARSE"):
    kw["coarse_space"] = os.environ["AMG_COARSE"]
t0 = time.perf_counter()
amg = B.AMG(A, nullspace="rigid_body", **kw)
B.synchronize()
t1 = time.perf_counter()
st = amg.solve(b, x, rtol=1e-8, max_iter=2000)
t2 = time.perf_counter()
u = x.get()[:V.n_owned].reshape(-1, 3)
tip = parallel.max_over_ranks(float(-u[:, 0].min()))
if rank == 0:
    print("parts %d dofs %d iterations %d converged %d true_rel_residual %.2e tip %.5e (beam theory %.5e) setup %.1f ms solve %.1f ms %s"
          % (world, 3 * (nx + 1) * (ny + 1) * (nz + 1), st["iterations"], st["converged"], st["true_rel_residual"], tip,
             7800 * 10 * 10.0 ** 4 / (8 * E / 12), (t1 - t0) * 1e3, (t2 - t1) * 1e3, kw), flush=True)
parallel.barrier()
parallel.finalize()
