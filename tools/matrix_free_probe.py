"""Matrix-free product (fs_operator_apply) against the assembled hybrid SELL/DIA product: milliseconds per product.
usage: python tools/matrix_free_probe.py [n ...]   (unit cube n x n x n, P1)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from fenicssolver_amd import backend as B

B.init(0)
for n in [int(a) for a in sys.argv[1:]] or [99, 215]:
    mesh = B.DeviceMesh.box(n, n, n, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    rng = np.random.default_rng(0)
    x = B.DeviceVector(V.n_local, rng.standard_normal(V.n_local))
    y = B.DeviceVector(V.n_owned)
    z = B.DeviceVector(V.n_owned)
    t_asm = A.spmv_benchmark(x, y, 50)
    t_mf = B.apply_operator(V, x, z, stiffness=1.0, reps=20)
    A.spmv(x, y)
    d = np.abs(y.get() - z.get()).max() / np.abs(y.get()).max()
    print("n=%d dofs=%d: assembled %.4f ms, matrix-free %.4f ms (x%.1f), rel. difference %.1e"
          % (n, V.n_owned, t_asm, t_mf, t_mf / t_asm, d), flush=True)
