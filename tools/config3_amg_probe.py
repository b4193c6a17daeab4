"""BASELINE.json configs[2] (P1 vector cantilever) with the AMG-preconditioned CG of solve_amg:
setup / solve split, iterations, agreement with the Jacobi-CG solution."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
E, nu = 2e11, 0.27
mu = E / (2 * (1 + nu)); lm = E * nu / ((1 + nu) * (1 - 2 * nu))
sizes = ((118, 15, 15), (236, 30, 30), (472, 59, 59))
if len(sys.argv) > 1:      # e.g. '2' = only the full configs[2] size
    sizes = [sizes[int(i)] for i in sys.argv[1].split(',')]
for dims in sizes:
    nx, ny, nz = dims
    mesh = B.DeviceMesh.box(nx, ny, nz, (0, 0, 0), (10., 1., 1.)); V = B.DeviceSpace(mesh, 3)
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
    nodes = np.arange((nx + 1) * (ny + 1) * (nz + 1)); left = nodes[nodes % (nx + 1) == 0]
    dofs = (left[:, None] * 3 + np.arange(3)).ravel()
    A.assemble(lame=(mu, lm)); B.assemble_vector(V, b, vector_value=(0, 0, -7800 * 10.)); A.apply_dirichlet(b, dofs, 0.0, True)
    xyz = mesh.get(True, False, False)[0]
    n = len(xyz)
    ns = np.zeros((6, n, 3))
    ns[0, :, 0] = 1; ns[1, :, 1] = 1; ns[2, :, 2] = 1
    ns[3, :, 0], ns[3, :, 1] = -xyz[:, 1], xyz[:, 0]
    ns[4, :, 0], ns[4, :, 2] = xyz[:, 2], -xyz[:, 0]
    ns[5, :, 2], ns[5, :, 1] = xyz[:, 1], -xyz[:, 2]
    variants = __import__('json').loads(os.environ['AMG_KW']) if 'AMG_KW' in os.environ else [dict(eig_steps=int(k)) for k in os.environ['AMG_EIG_STEPS'].split(',')] if 'AMG_EIG_STEPS' in os.environ else [dict()] * 2 if len(sys.argv) < 3 else [dict(strength_threshold=t) for t in (-1, 0.01, 0.02, 0.05, 0.1)]
    # (the fp32-stored and the all-fp64 hierarchy side by side - the last run is repeated with option amg_coarse_fp32 = 0)
    runs = [(kw, 1) for kw in variants] + ([(variants[-1], 0)] if os.environ.get('AMG_FP64_TOO', '1') == '1' else [])
    for kw, fp32 in runs:
        B.set_option("amg_coarse_fp32", fp32)
        t0 = time.perf_counter(); amg = B.AMG(A, nullspace='rigid_body' if os.environ.get('AMG_DEVICE_NS', '1') == '1' else ns.reshape(6, -1), **kw); B.synchronize(); t1 = time.perf_counter()
        st = amg.solve(b, x, rtol=1e-8); t2 = time.perf_counter()
        info = amg.info()
        lv = [(amg.level_info(l)['n_nodes'], amg.level_info(l)['block_size'], amg.level_info(l)['nnz_blocks']) for l in range(info['levels'])]
        print(dims, kw, 'coarse / transfer operators stored as %s' % ('fp32' if fp32 else 'fp64'), 'dofs', V.n_owned, 'levels', lv, 'opc %.3f' % info['operator_complexity'],
              'setup %.1f ms (lib %.1f) solve %.1f ms it %d conv %d true %.2e' % ((t1 - t0) * 1e3, info['setup_ms'], (t2 - t1) * 1e3, st['iterations'], st['converged'], st['true_rel_residual']), flush=True)
        amg.close()
    B.set_option("amg_coarse_fp32", 1)
    u = x.get().reshape(-1, 3); print('   tip deflection', u[:, 2].min(), 'beam theory ~', -7800 * 10 * 1 * 10 ** 4 / (8 * E * (1 / 12)), flush=True)
