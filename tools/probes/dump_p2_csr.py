"""Runs ON THE GPU BOX: CSR of the P2 / P1 heat operator (Dirichlet rows applied, as the solver multiplies it) on a small box,
saved for offline analysis of the line / run / class structure (tools/probes/analyze_segments.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fenicssolver_amd import backend as B   # noqa: E402

B.init(0)
out = os.path.join(ROOT, "gpurun_out")
for degree, n in ((2, 20), (1, 40)):
    mesh = B.DeviceMesh.box(n, n, n)
    V = B.DeviceSpace(mesh, 1, degree=degree)
    A = B.DeviceMatrix(V)
    A.assemble(stiffness=20.0)
    xyz, cells, gid = mesh.get()
    if degree == 2:
        edges = V.edges().astype(np.int64)
        pos = np.concatenate([xyz, 0.5 * (xyz[edges[:, 0]] + xyz[edges[:, 1]])])
    else:
        pos = xyz
    lo = np.nonzero(pos[:, 2] == 0.0)[0]
    hi = np.nonzero(pos[:, 2] == 1.0)[0]
    b = B.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, np.concatenate([lo, hi]).astype(np.int32), np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), symmetric=True)
    rp, ci, va, shape = A.to_csr()
    np.savez_compressed(os.path.join(out, "csr_p%d_n%d.npz" % (degree, n)), rp=rp, ci=ci, va=va, pos=pos)
    print("degree", degree, "n", n, "rows", shape[0], "nnz", len(ci))
