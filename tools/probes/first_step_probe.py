"""Where the FIRST step of a process goes (bench.py: first_step_ms 50 ms against 7 ms steady at 1 M DOF): wall-clock of every call of
the first and of the second step.   python tools/probes/first_step_probe.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fenicssolver_amd import backend as B
from fenicssolver_amd import partition
t = time.perf_counter
def lap(what, t0):
    B.synchronize(); print("  %-34s %8.3f ms" % (what, (t() - t0) * 1e3), flush=True); return t()
t0 = t(); B.init(0); t0 = lap("fs_init", t0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
mesh = B.DeviceMesh.box(n, n, n); t0 = lap("mesh", t0)
V = B.DeviceSpace(mesh, 1); t0 = lap("space (pattern)", t0)
lay = partition.slab_layout(n, n, n, (0, n + 1), 0, 1)
dofs, vals = partition.slab_dirichlet(n, n, n, lay, 2)
A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned); t0 = lap("matrix + vectors", t0)
for step in range(3):
    print("step", step)
    t0 = t()
    A.assemble(stiffness=20.0); t0 = lap("assemble", t0)
    b.fill(0.0); A.apply_dirichlet(b, dofs, vals, symmetric=True); t0 = lap("dirichlet", t0)
    st = B.krylov_solve(A, b, x, rtol=1e-8, max_iter=20000); t0 = lap("solve (%d its, lib %.3f ms)" % (st["iterations"], st["solve_ms"]), t0)
