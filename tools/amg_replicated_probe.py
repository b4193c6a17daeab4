"""BASELINE configs[2] through LinearElasticitySolver.solve_amg on N ranks (BoxMesh(distributed=True), bar along z): iteration
count and time of the default 'replicated' decomposition of the AMG preconditioner against 'schwarz'.
  python -m fenicssolver_amd.launch --nproc N [--devices 0,0,..] tools/amg_replicated_probe.py [scale] [replicated|schwarz]"""
import copy
import os
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from fenicssolver_amd import backend as B, parallel  # noqa: E402
from fenicssolver_amd.fem import BoxMesh, Point, AutoSubDomain, Constant, near  # noqa: E402
from fenicssolver_amd import SolverBase as SB  # noqa: E402
from fenicssolver_amd.LinearElasticitySolver import LinearElasticitySolver  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
mode = sys.argv[2] if len(sys.argv) > 2 else "replicated"
rank, world = parallel.ensure_comm()
n, nz = max(int(round(59 * scale)), 4), max(int(round(472 * scale)), 16)
mesh = BoxMesh(Point(0, 0, 0), Point(1, 1, 10), n, n, nz, distributed=world > 1)
bcs = OrderedDict()
bcs["fixed"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 0.0)), 'boundary_id': 1,
                'type': 'Dirichlet', 'value': Constant((0, 0, 0))}
s = copy.deepcopy(SB.default_case_settings)
s.update({'solver_name': 'LinearElasticitySolver', 'mesh': mesh, 'fe_degree': 1, 'vector_name': 'displacement',
          'boundary_conditions': bcs, 'body_source': (7800 * 10.0, 0, 0), 'initial_values': {'displacement': (0, 0, 0)},
          'material': {'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800}})
s['solver_settings']['solver_parameters'] = {'krylov_relative_tolerance': 1e-8, 'amg_decomposition': mode}
s['report_settings'] = {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
solver = LinearElasticitySolver(s)
t0 = time.perf_counter()
u = solver.solve()
B.synchronize()
t1 = time.perf_counter()
st = solver.last_solve_stats
tip = parallel.max_over_ranks(float(np.abs(u.vector().get_local().reshape(-1, 3)[:, 0]).max()))
if rank == 0:
    print("parts %d mode %s iterations %d true_rel_residual %.2e tip %.5e solve() %.1f ms (AMG set-up %.1f ms, Krylov %.1f ms)"
          % (world, mode, st["iterations"], st["true_rel_residual"], tip, (t1 - t0) * 1e3, st.get("amg_setup_ms", 0.0), st["solve_ms"]), flush=True)
parallel.barrier()
parallel.finalize()
