import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fenicssolver_amd import backend as B
from fenicssolver_amd import partition
B.init(0)
n = 99
mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1)
lay = partition.slab_layout(n, n, n, (0, n + 1), 0, 1)
dofs, vals = partition.slab_dirichlet(n, n, n, lay, 2)
x = B.DeviceVector(V.n_owned)
for step in range(5):
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned)
    A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b, dofs, vals, symmetric=True)
    B.synchronize(); t0 = time.perf_counter()
    st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000)
    print("step %d (new matrix object): solve %.3f ms (lib %.3f), kept %d" % (step, (time.perf_counter() - t0) * 1e3, st["solve_ms"], st["classes_kept"]), flush=True)
