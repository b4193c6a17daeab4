import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = 99
mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1)
A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
P = (n + 1) ** 2
dofs = np.concatenate([np.arange(P), np.arange(n * P, (n + 1) * P)]); vals = np.concatenate([np.full(P, 350.), np.full(P, 300.)])
for batch in (32, 16, 24, 48, 64, 32):
    B.set_option('cg_batch', batch)
    best = 1e9
    for rep in range(8):
        B.synchronize(); t0 = time.perf_counter()
        A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b, dofs, vals, True)
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000)
        best = min(best, time.perf_counter() - t0)
    print('batch %3d: step %.3f ms, %d iterations' % (batch, best * 1e3, st['iterations']), flush=True)
