"""One-launch CG iteration (k_dict_cg_iter) against the two-launch iteration: same problem, both paths, iteration counts,
solution difference, time per iteration.  python tools/probes/fused_iter_probe.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fenicssolver_amd import backend as B
from fenicssolver_amd import _lib

B.init(0)
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [99, 215]:
    mesh = B.DeviceMesh.box(n, n, n)
    V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V)
    nv = (n + 1) ** 3
    ids = np.arange(nv)
    iz = ids // ((n + 1) ** 2)
    dofs = np.concatenate([ids[iz == 0], ids[iz == n]]).astype(np.int64)
    vals = np.concatenate([np.full((iz == 0).sum(), 350.0), np.full((iz == n).sum(), 300.0)])
    res = {}
    for mode in (0, 1, 0, 1):
        lib.fs_set_option(b"cg_fused", float(mode))
        A.assemble(stiffness=20.0)
        b = B.DeviceVector(V.n_owned)
        A.apply_dirichlet(b, dofs, vals, symmetric=True)
        x = B.DeviceVector(V.n_owned)
        best = None
        for rep in range(4):
            x.set(np.zeros(V.n_owned))
            st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=int(os.environ.get('FS_PROBE_MAXIT', '5000')))
            if best is None or st["solve_ms"] < best["solve_ms"]:
                best = st
        res[mode] = (best, x.get())
        best["iterations"] = max(best["iterations"], 1)
        print("n=%d fused=%d: %d iterations, solve %.3f ms = %.2f us/iteration, kernel %.2f us + update %.2f us, true rel res %.3e, fused flag %s"
              % (n, mode, best["iterations"], best["solve_ms"], 1e3 * best["solve_ms"] / best["iterations"], 1e3 * best["spmv_ms"],
                 1e3 * best["update_ms"], best["true_rel_residual"], best.get("fused_iteration")), flush=True)
    d = np.abs(res[0][1] - res[1][1]).max()
    print("n=%d: max |x_fused - x_two_launch| = %.3e, bitwise equal: %s" % (n, d, np.array_equal(res[0][1], res[1][1])), flush=True)
    del A, V, mesh
