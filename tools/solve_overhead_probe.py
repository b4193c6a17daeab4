"""Fixed cost of one Krylov solve at configs[1]: wall time against the iteration count (max_iter cut), least squares."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1)
A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
P = (n + 1) ** 2
dofs = np.concatenate([np.arange(P), np.arange(n * P, (n + 1) * P)]); vals = np.concatenate([np.full(P, 350.), np.full(P, 300.)])
A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b, dofs, vals, True)
pts = []
for mi in (32, 64, 96, 128, 160, 192, 224, 256, 288, 100000):
    best = 1e9
    for rep in range(5):
        B.synchronize()
        t0 = time.perf_counter()
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=mi)
        best = min(best, time.perf_counter() - t0)
    pts.append((st['iterations'], best * 1e3))
    print('max_iter %6d iterations %4d converged %d  %.3f ms' % (mi, st['iterations'], st['converged'], best * 1e3), flush=True)
it = np.array([p[0] for p in pts[:-1]], float); ms = np.array([p[1] for p in pts[:-1]])
slope, icpt = np.polyfit(it, ms, 1)
print('fit over the cut solves: %.2f us per iteration + %.3f ms per solve; converged solve: %d iterations %.3f ms (fit predicts %.3f)' % (slope * 1e3, icpt, pts[-1][0], pts[-1][1], icpt + slope * pts[-1][0]))
for name in ('assemble', 'fill', 'dirichlet'):
    best = 1e9
    for rep in range(5):
        B.synchronize(); t0 = time.perf_counter()
        if name == 'assemble': A.assemble(stiffness=20.0)
        elif name == 'fill': b.fill(0.0)
        else: A.apply_dirichlet(b, dofs, vals, True)
        B.synchronize(); best = min(best, time.perf_counter() - t0)
    print('%-10s %.3f ms (synchronised)' % (name, best * 1e3))
