"""CG batches as hipGraphs on/off (FS_CG_GRAPH) at 125 K / 1 M / 10 M DOF: solve time, iterations, bit-identical result."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, hashlib
    from fenicssolver_amd import backend as B
    B.init(0)
    n = int(sys.argv[2])
    mesh = B.DeviceMesh.box(n, n, n)
    V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V); A.assemble(stiffness=20.0)
    P = (n + 1) ** 2
    dofs = np.concatenate([np.arange(P), np.arange(V.n_owned - P, V.n_owned)]).astype(np.int32)
    vals = np.concatenate([np.full(P, 350.0), np.full(P, 300.0)])
    b = B.DeviceVector(V.n_owned); A.apply_dirichlet(b, dofs, vals, symmetric=True)
    x = B.DeviceVector(V.n_owned)
    ms = []
    for _ in range(8):
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000, precond="jacobi")
        ms.append(st["solve_ms"])
    print(json.dumps({"n": n, "graph": os.environ.get("FS_CG_GRAPH", "auto"), "iters": st["iterations"], "solve_ms_min": round(min(ms[2:]), 3),
                      "solve_ms_med": round(sorted(ms[2:])[3], 3), "spmv_ms": round(st["spmv_ms"], 5), "update_ms": round(st["update_ms"], 5),
                      "true_res": st["true_rel_residual"], "sha": hashlib.sha1(x.get().tobytes()).hexdigest()[:12]}), flush=True)
else:
    for n in (49, 99, 215):
        for g in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child", str(n)], env=dict(os.environ, FS_CG_GRAPH=g))
