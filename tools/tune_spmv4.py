"""SpMV of the Taylor-Hood block matrix (bs = 4 on the CG2 pattern) at configs[4] size: launch parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = 43
mesh = B.DeviceMesh.box(n, n, n)
W = B.DeviceSpace(mesh, ncomp=4, degree=2)
J = B.DeviceMatrix(W); g = B.DeviceVector(W.n_owned)
B.assemble_navier_stokes(J, g, None, None, nu=0.01, rho=1.0, inv_dt=100.0, convection=False, newton=False)
x = B.DeviceVector(W.n_local, np.random.default_rng(0).standard_normal(W.n_local)); y = B.DeviceVector(W.n_owned)
print('dofs', W.n_owned, 'nnz', W.nnz, 'stored bytes', W.spmv_matrix_bytes, 'dia slices', W.n_dia_slices, '/', W.n_slices)
import os
for reps in (1, 2, 5, 20, 100):
    ms = [J.spmv_benchmark(x, y, reps) for _ in range(4)]
    print('row-split kernel, %3d back-to-back launches: %s ms each' % (reps, ['%.3f' % m for m in ms]))
os.environ['FS_SPMV4_GENERIC'] = '1'
for reps in (1, 20):
    ms = [J.spmv_benchmark(x, y, reps) for _ in range(3)]
    print('generic kernel,   %3d back-to-back launches: %s ms each' % (reps, ['%.3f' % m for m in ms]))
