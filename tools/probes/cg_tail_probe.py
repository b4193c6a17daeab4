"""What the launches behind the last CG iteration cost: BASELINE configs[1] solved with the round-4 batches (status word copied back
behind every 32 launches) and with the progress words in pinned memory (cg_mirror) for several (cg_sub, cg_ahead) pairs.
Prints iterations, launches enqueued, the library's solve_ms (best and median of 9) and whether the solution bits are equal."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1); A = B.DeviceMatrix(V)
nv = (n + 1) ** 3; ids = np.arange(nv); iz = ids // ((n + 1) ** 2)
dofs = np.concatenate([ids[iz == 0], ids[iz == n]]).astype(np.int64)
vals = np.concatenate([np.full((iz == 0).sum(), 350.0), np.full((iz == n).sum(), 300.0)])
A.assemble(stiffness=20.0); b = B.DeviceVector(V.n_owned); A.apply_dirichlet(b, dofs, vals, symmetric=True)
x = B.DeviceVector(V.n_owned)
ref = None
for mirror, sub, ahead in (tuple(tuple(int(v) for v in a.split(',')) for a in sys.argv[2:]) or ((0, 8, 8), (1, 8, 8), (1, 16, 8), (1, 32, 8), (0, 8, 8), (1, 16, 8))):
    B.set_option("cg_mirror", mirror); B.set_option("cg_sub", sub); B.set_option("cg_ahead", ahead)
    ts = []
    for rep in range(11):
        x.fill(0.0); st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000)
        if rep >= 2: ts.append(st["solve_ms"])
    xs = x.get()
    if ref is None: ref = xs
    print("mirror %d sub %2d ahead %2d: %d iterations, %d launches, solve best %.3f median %.3f ms, fused %d, bits equal %s" % (
        mirror, sub, ahead, st["iterations"], st["launches"], min(ts), sorted(ts)[len(ts) // 2], st["fused_iteration"], bool(np.array_equal(xs, ref))))
