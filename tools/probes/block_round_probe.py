"""Bare product of the 3x3-block operator of configs[2] with R blocks per round (libfsamd built with -DFS_BLOCK_ROUND=R)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fenicssolver_amd import _lib as L
if len(sys.argv) > 1 and sys.argv[1] != "default":
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfsamd_r%s.so" % sys.argv[1])
from fenicssolver_amd import backend as B
B.init(0)
mesh = B.DeviceMesh.box(472, 59, 59, (0, 0, 0), (10., 1., 1.)); V = B.DeviceSpace(mesh, 3); A = B.DeviceMatrix(V); A.assemble(lame=(1.0, 1.5))
x = B.DeviceVector(V.n_local, np.random.default_rng(0).standard_normal(V.n_local)); y = B.DeviceVector(V.n_owned)
ms = C.c_double()
best = 1e9
for rep in range(4):
    L.check(L.load().fs_spmv_benchmark(A.h, x.h, y.h, 50, C.byref(ms)), "b"); best = min(best, ms.value)
alg = V.nnz / 9 * 76 + V.n_owned / 3 * 52
print("R", sys.argv[1] if len(sys.argv) > 1 else "default", "bsr3 product %.1f us  %.2f TB/s algorithmic (%.3f of 8 TB/s)" % (best * 1e3, alg / best / 1e9, alg / best / 8e9))
