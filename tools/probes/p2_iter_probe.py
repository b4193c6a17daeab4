"""BASELINE configs[3] (CG2 heat conduction, n = 107, 9.94 M DOF) through bench.P2Problem: one warm-up and one timed step; a
solve that breaks down (timing ablations with wrong numerics) is reported, not fatal - tools/probes/trace_probe.sh reads the trace.
python tools/probes/p2_iter_probe.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 107
prob = bench.P2Problem(n, (0, n + 1), 2, 0, 1)
for rep in range(2):
    try:
        t0 = time.perf_counter()
        st, asm = prob.step(1e-8)
        B.synchronize()
        print("n=%d P2: %d iterations, solve %.3f ms = %.2f us/iteration, product %.2f us + update %.2f us, classes %d"
              % (n, st["iterations"], st["solve_ms"], 1e3 * st["solve_ms"] / max(st["iterations"], 1), 1e3 * st["spmv_ms"], 1e3 * st["update_ms"], st["row_classes"]), flush=True)
    except Exception as e:          # noqa: BLE001
        print("solve failed (expected in a timing ablation):", str(e)[:120], flush=True)
