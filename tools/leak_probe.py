import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from collections import OrderedDict
import numpy as np
from fenicssolver_amd import backend as B
from fenicssolver_amd.fem import UnitCubeMesh, FunctionSpace, AutoSubDomain, Constant, near
from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
B.init(0)
def run(steps):
    m = UnitCubeMesh(30, 30, 30); Q = FunctionSpace(m, "CG", 1)
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 1.0)), 'boundary_id': 1, 'values': {'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(350)}}}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[2], 0.0)), 'boundary_id': 2, 'values': {'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 1000.0, 'specific_heat_capacity': 4200.0, 'thermal_conductivity': 20.0},
         'solver_settings': {'transient_settings': {'transient': True, 'starting_time': 0, 'time_step': 100.0, 'ending_time': 100.0 * steps - 1e-6}, 'reference_values': {'temperature': 300}, 'solver_parameters': {}},
         'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    solver = ScalarTransportSolver(s)
    solver.solve()
import gc
for rep in range(6):
    run(30)
    gc.collect()
    print(rep, B.memory_info(), flush=True)
B.trim_memory(); print('trimmed', B.memory_info())
