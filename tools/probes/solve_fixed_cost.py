"""Fixed cost of a solve: BASELINE configs[1] solved with iteration limits 1, 33, 65 and without (solve_ms of the library)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fenicssolver_amd import backend as B
B.init(0)
n = 99
mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1); A = B.DeviceMatrix(V)
nv = (n + 1) ** 3; ids = np.arange(nv); iz = ids // ((n + 1) ** 2)
dofs = np.concatenate([ids[iz == 0], ids[iz == n]]).astype(np.int64)
vals = np.concatenate([np.full((iz == 0).sum(), 350.0), np.full((iz == n).sum(), 300.0)])
A.assemble(stiffness=20.0); b = B.DeviceVector(V.n_owned); A.apply_dirichlet(b, dofs, vals, symmetric=True)
x = B.DeviceVector(V.n_owned)
for mi in ((1, 1) if os.environ.get('FS_FIXED_ONLY') else (5000, 1, 5000, 1, 33, 65)):
    best = 1e9
    for rep in range(5):
        x.set(np.zeros(V.n_owned)); st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=mi); best = min(best, st["solve_ms"])
    print("max_iter %d: %d iterations, solve %.3f ms" % (mi, st["iterations"], best))
