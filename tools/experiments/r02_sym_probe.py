"""Symmetric mirror reads of the CG product on/off (FS_SPMV_SYM) at 1 M and 10 M DOF: kernel time, iterations, solution."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
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
    for _ in range(3):
        st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000, precond="jacobi")
    xs = x.get()
    print(json.dumps({"n": n, "sym": os.environ.get("FS_SPMV_SYM", "1"), "nt": os.environ.get("FS_SPMV_NT", "auto"), "iters": st["iterations"], "spmv_ms": st["spmv_ms"], "update_ms": st["update_ms"],
                      "solve_ms": st["solve_ms"], "true_res": st["true_rel_residual"], "sym_slices": st["sym_slices"], "mismatch": st["sym_mismatches"],
                      "streamed": st["spmv_streamed_bytes"], "csr": st["spmv_bytes"], "xsum": float(xs.sum()), "x_mid": float(xs[len(xs) // 2])}), flush=True)
else:
    combos = [("0", "0", None), ("1", "0", None)] + [("1", "-2", t) for t in ("2048", "4096", "8192", "16384", "32768")] + [("0", "-2", "8192")]
    for n in (99, 215):
        for sym, order, tile in combos:
            env = dict(os.environ, FS_SPMV_SYM=sym, FS_SLICE_ORDER=order, FS_SPACE_DEBUG="1")
            if tile:
                env["FS_TILE_ROWS"] = tile
            print("sym", sym, "order", order, "tile_rows", tile, flush=True)
            subprocess.run([sys.executable, __file__, "child", str(n)], env=env)
