"""Product of the Taylor-Hood block operator of configs[4] (bs = 4 on the CG2 pattern, n = 43): the entry-split kernel
(k_sell_spmv4_ksplit, default) against the block-row kernel (FS_SPMV4_KSPLIT=0), time per launch and the two results against each
other.  Run twice: python tools/probes/spmv4_probe.py ; FS_SPMV4_KSPLIT=0 python tools/probes/spmv4_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 43
mesh = B.DeviceMesh.box(n, n, n)
W = B.DeviceSpace(mesh, ncomp=4, degree=2)
J = B.DeviceMatrix(W); g = B.DeviceVector(W.n_owned)
w0 = B.DeviceVector(W.n_local, np.random.default_rng(1).standard_normal(W.n_local) * 0.1)
B.assemble_navier_stokes(J, g, w0, None, nu=0.01, rho=1.0, inv_dt=100.0, convection=True, newton=True)
x = B.DeviceVector(W.n_local, np.random.default_rng(0).standard_normal(W.n_local)); y = B.DeviceVector(W.n_owned)
ms = [J.spmv_benchmark(x, y, 50) for _ in range(4)]
yy = y.get()
print('FS_SPMV4_KSPLIT=%s: %s ms per launch; checksum %.15e |y|max %.6e' % (os.environ.get('FS_SPMV4_KSPLIT', '1'), ['%.4f' % m for m in ms], float(np.dot(yy, np.cos(np.arange(len(yy))))), float(np.abs(yy).max())))
np.save('/tmp/spmv4_%s.npy' % os.environ.get('FS_SPMV4_KSPLIT', '1'), yy)
if os.path.exists('/tmp/spmv4_0.npy') and os.path.exists('/tmp/spmv4_1.npy'):
    a, b = np.load('/tmp/spmv4_0.npy'), np.load('/tmp/spmv4_1.npy')
    print('max |y_ksplit - y_rows| / |y|max = %.3e' % float(np.abs(a - b).max() / np.abs(a).max()))
