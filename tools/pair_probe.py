"""Two-rows-per-lane DIA product on/off (FS_SPMV_PAIRS) at 1 M / 10 M DOF P1 and 10 M DOF P2."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from fenicssolver_amd import backend as B
    B.init(0)
    n, deg = int(sys.argv[2]), int(sys.argv[3])
    mesh = B.DeviceMesh.box(n, n, n)
    V = B.DeviceSpace(mesh, 1, degree=deg)
    A = B.DeviceMatrix(V); A.assemble(stiffness=20.0)
    b = B.DeviceVector(V.n_owned)
    if deg == 1:
        P = (n + 1) ** 2
        dofs = np.concatenate([np.arange(P), np.arange(V.n_owned - P, V.n_owned)]).astype(np.int32)
        vals = np.concatenate([np.full(P, 350.0), np.full(P, 300.0)])
        A.apply_dirichlet(b, dofs, vals, symmetric=True)
    else:
        B.assemble_vector(V, b, source=1.0)
        A.apply_dirichlet(b, np.arange((n + 1) ** 2, dtype=np.int32), np.zeros((n + 1) ** 2), symmetric=True)
    x = B.DeviceVector(V.n_owned)
    for _ in range(3):
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000, precond="jacobi")
    xs = x.get()
    print(json.dumps({"n": n, "deg": deg, "pairs": os.environ.get("FS_SPMV_PAIRS", "auto"), "iters": st["iterations"], "spmv_ms": round(st["spmv_ms"], 5),
                      "update_ms": round(st["update_ms"], 5), "solve_ms": round(st["solve_ms"], 2), "true_res": st["true_rel_residual"],
                      "xsum": float(xs.sum()), "x_mid": float(xs[len(xs) // 2])}), flush=True)
else:
    for n, deg in ((215, 1), (107, 2)):
        for pairs in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child", str(n), str(deg)], env=dict(os.environ, FS_SPMV_PAIRS=pairs, FS_SPACE_DEBUG="1"))
