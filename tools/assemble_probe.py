"""P1 gather assembly + CG product at 1 M / 10 M DOF under different slice orders (FS_SLICE_ORDER, FS_TILE_ROWS)."""
import os, sys, subprocess, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from fenicssolver_amd import backend as B
    B.init(0)
    n = int(sys.argv[2])
    mesh = B.DeviceMesh.box(n, n, n)
    V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V)
    for _ in range(3):
        A.assemble(stiffness=20.0)
    B.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        A.assemble(stiffness=20.0)
    B.synchronize()
    t_asm = (time.perf_counter() - t0) / reps * 1e3
    P = (n + 1) ** 2
    dofs = np.concatenate([np.arange(P), np.arange(V.n_owned - P, V.n_owned)]).astype(np.int32)
    vals = np.concatenate([np.full(P, 350.0), np.full(P, 300.0)])
    b = B.DeviceVector(V.n_owned); A.apply_dirichlet(b, dofs, vals, symmetric=True)
    x = B.DeviceVector(V.n_owned)
    for _ in range(2):
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000, precond="jacobi")
    print(json.dumps({"n": n, "order": os.environ.get("FS_SLICE_ORDER", "default"), "tile": os.environ.get("FS_TILE_ROWS"), "assemble_ms": round(t_asm, 3),
                      "iters": st["iterations"], "spmv_ms": round(st["spmv_ms"], 5), "update_ms": round(st["update_ms"], 5), "solve_ms": round(st["solve_ms"], 2),
                      "xsum": float(x.get().sum())}), flush=True)
else:
    for n in (99, 215):
        for order, tile in [("0", None)] + [("-2", t) for t in ("2048", "4096", "8192", "16384", "32768", "65536")]:
            env = dict(os.environ, FS_SLICE_ORDER=order)
            if tile:
                env["FS_TILE_ROWS"] = tile
            subprocess.run([sys.executable, __file__, "child", str(n)], env=env)
