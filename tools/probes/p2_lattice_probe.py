"""CG2 heat conduction on the unit cube (BASELINE configs[3] at n = 107) solved in the space's numbering and in the solver's lattice
order (fs_lattice.hip): iterations, time per iteration, product / update kernel times, the two solutions against each other and
against the exact linear profile.   python tools/probes/p2_lattice_probe.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 107
prob = bench.P2Problem(n, (0, n + 1), 2, 0, 1)
sol = {}
for lat in (0, 1, 0, 1):
    B.set_option("lattice_order", lat)
    for rep in range(3):
        t0 = time.perf_counter()
        st, asm = prob.step(1e-8)
        B.synchronize()
        t = (time.perf_counter() - t0) * 1e3
    x = prob.x.get()[:prob.n_owned]
    sol[lat] = x
    print("n=%d lattice %d: %d iterations, step %.2f ms, solve %.3f ms = %.2f us/iteration, product %.2f us + update %.2f us, classes %d, lattice_order %d, "
          "max|x - exact| %.2e" % (n, lat, st["iterations"], t, st["solve_ms"], 1e3 * st["solve_ms"] / max(st["iterations"], 1), 1e3 * st["spmv_ms"],
                                   1e3 * st["update_ms"], st["row_classes"], st["lattice_order"], float(np.abs(x - prob.exact_owned).max())), flush=True)
print("max |x_lattice - x_plain| = %.3e" % float(np.abs(sol[1] - sol[0]).max()))
