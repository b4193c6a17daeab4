import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from collections import OrderedDict
import numpy as np
from fenicssolver_amd.fem import UnitSquareMesh, FunctionSpace, AutoSubDomain, Constant, near, Point
from fenicssolver_amd.ScalarTransportSolver import ScalarTransportSolver
mesh = UnitSquareMesh(20, 20); Q = FunctionSpace(mesh, "CG", 1)
def settings(**extra):
    bcs = OrderedDict()
    bcs["hot"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 1)), 'boundary_id': 1, 'values': {'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}}}
    bcs["cold"] = {'boundary': AutoSubDomain(lambda x: near(x[1], 0)), 'boundary_id': 2, 'values': {'temperature': {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)}}}
    s = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None, 'boundary_conditions': bcs, 'body_source': None, 'initial_values': {'temperature': 300},
         'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.6},
         'solver_settings': {'transient_settings': {'transient': False, 'starting_time': 0, 'time_step': 0.1, 'ending_time': 1}, 'reference_values': {'temperature': 300}, 'solver_parameters': {}},
         'report_settings': {"logging_level": 40, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}, 'scalar_name': 'temperature'}
    s.update(extra); return s
a = ScalarTransportSolver(settings()); a.material['conductivity'] = lambda T: 0.6 * (1 + 0.002 * (T - 300)); Ta = a.solve().vector().get_local()
print('k(T) 2-D: newton its', a.newton_iterations, 'range', Ta.min(), Ta.max())
b = ScalarTransportSolver(settings(point_source=[(Point(0.5, 0.5), 100.0)])); Tb = b.solve().vector().get_local()
print('point source 2-D: max', Tb.max())
c = ScalarTransportSolver(settings(material={'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': [[0.6, 0.1], [0.1, 0.9]]})); Tc = c.solve().vector().get_local()
print('tensor k 2-D: ', Tc.min(), Tc.max())
