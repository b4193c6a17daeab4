"""AMG-preconditioned CG on the bench workload (unit-cube Poisson, configs[1] and the 10 M-DOF case), next to Jacobi-CG."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
for n in (99, 215):
    mesh = B.DeviceMesh.box(n, n, n); V = B.DeviceSpace(mesh)
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned); xj = B.DeviceVector(V.n_owned)
    P = (n + 1) ** 2
    bot = np.arange(P); top = np.arange(P) + n * P
    dofs = np.concatenate([bot, top]).astype(np.int32); vals = np.concatenate([np.full(P, 350.0), np.full(P, 300.0)])
    A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b, dofs, vals, True)
    for rep in range(2):
        t0 = time.perf_counter(); amg = B.AMG(A); B.synchronize(); t1 = time.perf_counter()
        st = amg.solve(b, x, rtol=1e-8); t2 = time.perf_counter()
        info = amg.info()
        lv = [(amg.level_info(l)['n_nodes'], amg.level_info(l)['nnz_blocks']) for l in range(info['levels'])]
        print(n, 'dofs', V.n_owned, 'levels', lv, 'opc %.3f setup %.1f ms solve %.1f ms it %d true %.2e -> %.3g DOF/s' % (
            info['operator_complexity'], (t1 - t0) * 1e3, (t2 - t1) * 1e3, st['iterations'], st['true_rel_residual'], V.n_owned / (t2 - t0)), flush=True)
        amg.close()
    t0 = time.perf_counter(); sj = B.krylov_solve(A, b, xj, rtol=1e-8, max_iter=20000); t1 = time.perf_counter()
    print('   jacobi-CG', sj['iterations'], 'its %.1f ms' % ((t1 - t0) * 1e3), 'diff', np.abs(x.get() - xj.get()).max())
