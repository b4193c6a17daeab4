"""Crank-Nicolson heat conduction through the solver class at 1 M DOF (unit cube n = 99): where does a time step go?"""
import cProfile
import os
import pstats
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from fenicssolver_amd import backend as B  # noqa: E402
from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near  # noqa: E402
from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B.init(0)
t0 = time.perf_counter()
m = UnitCubeMesh(n, n, n)
Q = FunctionSpace(m, "CG", 1)
bcs = OrderedDict()
bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 1.0)), 'boundary_id': 1, 'values': {
    'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(350)}}}
bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0.0)), 'boundary_id': 2, 'values': {
    'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
     'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
     'material': {'density': 1000.0, 'specific_heat_capacity': 4200.0, 'thermal_conductivity': 20.0},
     'solver_settings': {'transient_settings': {'transient': True, 'starting_time': 0, 'time_step': 100.0,
                                                'ending_time': 100.0 * steps - 1e-6},
                         'reference_values': {'temperature': 300}, 'solver_parameters': {}},
     'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0},
     'scalar_name': 'temperature'}
solver = ScalarTransportSolver(s)
t1 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
T = solver.solve()
B.synchronize()
pr.disable()
t2 = time.perf_counter()
print("n %d: %d DOF, set-up %.2f s, %d steps in %.3f s = %.1f ms per step, last solve %d iterations" % (
    n, Q.dim(), t1 - t0, steps, t2 - t1, (t2 - t1) / steps * 1e3, solver.last_solve_stats["iterations"]))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
