"""Vector P2 cantilever (the reference's example scaled up): AMG-CG against Jacobi-CG through the solver class.
usage: python tools/p2_amg_probe.py nx ny nz [jacobi]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")), "tests"))
import numpy as np
from test_gpu_p2_vector import _example_solver

nx, ny, nz = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (40, 10, 10)
for pc in (("jacobi", "amg") if "jacobi" in sys.argv else ("amg",)):
    solver = _example_solver(nx, ny, nz, thermal=False, body=True)
    sp = solver.solver_settings.setdefault('solver_parameters', {})
    if pc == "jacobi":
        sp['preconditioner'] = 'jacobi'
    sp['maximum_iterations'] = 100000
    t0 = time.perf_counter()
    u = solver.solve()
    t1 = time.perf_counter()
    st = solver.last_solve_stats
    U = u.node_values()
    print("%s: dofs %d iterations %d solve %.1f ms (setup %.1f ms) total %.2f s true %.2e tip uz %.6e"
          % (pc, U.size, st['iterations'], st['solve_ms'], st.get('amg_setup_ms', 0.0), t1 - t0, st['true_rel_residual'], U[:, 2].min()), flush=True)
    print("   ", {k: v for k, v in st.items() if k.startswith('amg_')}, flush=True)
