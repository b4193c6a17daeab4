"""Runs ON THE GPU BOX: cost of the first and of a second set-up (mesh + sparsity pattern) in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fenicssolver_amd import backend as B
t0 = time.perf_counter(); B.init(0); B.synchronize(); t1 = time.perf_counter()
print("init %.1f ms" % ((t1 - t0) * 1e3))
for n in [int(v) for v in os.environ.get('FS_SETUP_SIZES', '4,99,99,215,99').split(',')]:
    t0 = time.perf_counter(); m = B.DeviceMesh.box(n, n, n); B.synchronize(); t1 = time.perf_counter()
    V = B.DeviceSpace(m, 1); B.synchronize(); t2 = time.perf_counter()
    A = B.DeviceMatrix(V); A.assemble(stiffness=20.0); B.synchronize(); t3 = time.perf_counter()
    print("n %d: mesh %.2f ms, space %.2f ms, first assemble %.2f ms" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    del A, V, m
