"""Pipelined against single-reduction CG on ONE GPU at the two bench sizes: iteration counts, true residuals, time per iteration
(what the extra 40 B/DOF of vector traffic cost where no collective hides behind the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
for n in (99, 215):
    mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh, 1); A = B.DeviceMatrix(V); A.assemble(stiffness=20.0)
    P = (n + 1) ** 2
    dofs = np.concatenate([np.arange(P), np.arange(n * P, (n + 1) * P)]).astype(np.int32)
    vals = np.concatenate([np.full(P, 350.0), np.full(P, 300.0)])
    b = B.DeviceVector(V.n_owned); A.apply_dirichlet(b, dofs, vals, True)
    for rtol in (1e-8, 1e-12):
        sol = {}
        for pipe in (False, True):
            x = B.DeviceVector(V.n_owned)
            B.krylov_solve(A, b, x, rtol=rtol, max_iter=20000, pipelined=pipe)
            t0 = time.perf_counter(); st = B.krylov_solve(A, b, x, rtol=rtol, max_iter=20000, pipelined=pipe); t1 = time.perf_counter()
            sol[pipe] = x.get()
            print("n=%d rtol %.0e %s: %d iterations, true residual %.2e, %.2f ms (%.1f us / iteration), product %.1f us update %.1f us"
                  % (n, rtol, "pipelined       " if pipe else "single-reduction", st["iterations"], st["true_rel_residual"], (t1 - t0) * 1e3,
                     (t1 - t0) * 1e6 / st["iterations"], st["spmv_ms"] * 1e3, st["update_ms"] * 1e3), flush=True)
        print("      max |x_pipelined - x_single| = %.2e" % np.abs(sol[True] - sol[False]).max())
