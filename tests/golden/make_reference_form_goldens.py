#!/usr/bin/env python3
"""Runs the REFERENCE's own solver classes (/root/reference/FenicsSolver, imported unchanged)
against the recording `dolfin` stub of tests/refstub and writes what they hand to
DOLFIN — Dirichlet sets and the integrals of the variational form — as JSON goldens:

    tests/golden/reference_forms.json

This pins the Python-side semantics of the reference ("partial-import
option"): which settings produce which integrals, coefficients, signs and Dirichlet conditions,
quirks included.  tests/test_reference_forms.py checks that fenicssolver_amd's generate_form
recognises exactly the same terms from the same settings.  Only runs in the build container
(needs /root/reference); the GPU box sees the JSON only.

    python tests/golden/make_reference_form_goldens.py
"""
import collections
import copy
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests", "refstub"))     # the stub named `dolfin` / `ufl`
sys.path.insert(0, "/root/reference")
sys.argv = [sys.argv[0]]                                         # FenicsSolver/__init__ runs main(argv) otherwise

import dolfin                                                    # noqa: E402  (the stub)
from dolfin import (Constant, Expression, AutoSubDomain, SubDomain, FunctionSpace, VectorFunctionSpace,  # noqa: E402
                    UnitCubeMesh, BoxMesh, Point)
from FenicsSolver import SolverBase, ScalarTransportSolver, LinearElasticitySolver                     # noqa: E402
from FenicsSolver.main import load_settings                                                            # noqa: E402

QUIET = {"logging_level": 50, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
out = collections.OrderedDict()


def run(name, solver):
    dolfin.RECORD["solves"] = []
    dolfin.RECORD.pop("assembled", None)
    solver.solve()
    rec = {"solves": copy.deepcopy(dolfin.RECORD["solves"])}
    if "assembled" in dolfin.RECORD:
        rec["assembled"] = copy.deepcopy(dolfin.RECORD["assembled"])
    if "nullspace_vectors" in dolfin.RECORD:
        rec["nullspace_vectors"] = dolfin.RECORD.pop("nullspace_vectors")
    out[name] = rec


# --- case 1: data/TestHeatTransfer.json (config 1) -------------------------------------------------
os.chdir("/root/reference/FenicsSolver")      # the JSON's mesh path is relative to the package folder
s = load_settings("../data/TestHeatTransfer.json")
s["report_settings"] = dict(QUIET)
run("config1_json", ScalarTransportSolver.ScalarTransportSolver(s))


def heat_settings(transient=False, **extra):
    mesh = UnitCubeMesh(4, 4, 4)
    Q = FunctionSpace(mesh, "CG", 1)
    top = AutoSubDomain(lambda x: True)
    bottom = AutoSubDomain(lambda x: True)
    left = AutoSubDomain(lambda x: True)
    bcs = collections.OrderedDict()
    bcs["hot"] = {'boundary': top, 'boundary_id': 1, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'heatFlux', 'value': Constant(36.0)}}}
    bcs["cold"] = {'boundary': bottom, 'boundary_id': 2, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'HTC', 'value': Constant(100), 'ambient': Constant(300)}}}
    bcs["left"] = {'boundary': left, 'boundary_id': 3, 'values': {
        'temperature': {'variable': 'temperature', 'type': 'symmetry', 'value': None}}}
    st = {'solver_name': 'ScalarEquationSolver', 'mesh': None, 'function_space': Q, 'periodic_boundary': None,
          'boundary_conditions': bcs, 'body_source': 5.0, 'initial_values': {'temperature': 300},
          'material': {'density': 1000, 'specific_heat_capacity': 4200, 'thermal_conductivity': 0.1},
          'solver_settings': {'transient_settings': {'transient': transient, 'starting_time': 0, 'time_step': 0.1,
                                                     'ending_time': 0.1},
                              'reference_values': {'temperature': 300}, 'solver_parameters': {}},
          'report_settings': dict(QUIET), 'scalar_name': 'temperature'}
    st.update(extra)
    return st


# --- case 2: heatFlux + HTC + symmetry + body source, conductivity patched after construction ---------
sol = ScalarTransportSolver.ScalarTransportSolver(heat_settings())
sol.material['conductivity'] = 0.6
run("heat_flux_htc_source", sol)

# --- case 3: the same, transient (Crank-Nicolson) -------------------------------------------------------
sol = ScalarTransportSolver.ScalarTransportSolver(heat_settings(transient=True))
sol.material['conductivity'] = 0.6
run("heat_transient", sol)

# --- case 4: convective velocity, no stabilisation (examples/test_heat_transfer.py active case) ---------
sol = ScalarTransportSolver.ScalarTransportSolver(heat_settings(convective_velocity=Constant((0.005, -0.005, 0.0))))
sol.material['conductivity'] = 0.6
run("heat_convection", sol)

# --- case 4b: the same with SUPG ("SPUG", method 2: the test function becomes q + tau (v . grad q) everywhere) ----
for name, tr in (("heat_convection_supg", False), ("heat_convection_supg_transient", True)):
    sol = ScalarTransportSolver.ScalarTransportSolver(heat_settings(
        transient=tr, convective_velocity=Constant((0.005, -0.005, 0.0)),
        advection_settings={'stabilization_method': 'SPUG', 'Pe': 10.0}))
    sol.material['conductivity'] = 0.6
    run(name, sol)

# --- case 4c: interior-penalty stabilisation ("IP": an interior-facet integral added to the convection form) ------
sol = ScalarTransportSolver.ScalarTransportSolver(heat_settings(
    convective_velocity=Constant((0.005, -0.005, 0.0)), advection_settings={'stabilization_method': 'IP', 'alpha': 0.1}))
sol.material['conductivity'] = 0.6
run("heat_convection_ip", sol)

# --- case 5: Dirichlet + Neumann(fixedGradient) + Robin --------------------------------------------------
st = heat_settings()
st['body_source'] = None
st['boundary_conditions']["hot"]['values']['temperature'] = {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(360)}
st['boundary_conditions']["cold"]['values']['temperature'] = {'variable': 'temperature', 'type': 'fixedGradient', 'value': Constant(2.0)}
st['boundary_conditions']["left"]['values']['temperature'] = {'variable': 'temperature', 'type': 'Robin', 'value': Constant(310), 'gradient': Constant(1.5)}
run("heat_dirichlet_neumann_robin", ScalarTransportSolver.ScalarTransportSolver(st))


# --- case 6: linear elasticity (examples/test_linear_elasticity.py) ------------------------------------
def elasticity_settings(bcs, **extra):
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 8, 2, 2)
    st = copy.deepcopy(SolverBase.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = None
    st.update(extra)
    return st


class Left(SubDomain):
    pass


class Right(SubDomain):
    pass


bcs = collections.OrderedDict()
bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
bcs["tensile"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'stress', 'value': Constant((1e8, 0, 0))}
run("elasticity_stress_body_thermal", LinearElasticitySolver.LinearElasticitySolver(elasticity_settings(
    bcs, body_source=Expression(("10*rho", "0", "0.0"), rho=7800, omega=100, degree=2),
    temperature_distribution=Expression("343", degree=1))))

bcs = collections.OrderedDict()
bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}
bcs["displ"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant((0, 0, 1e-3))}
run("elasticity_displacement", LinearElasticitySolver.LinearElasticitySolver(elasticity_settings(
    bcs, body_source=Expression(("10*rho", "0", "0.0"), rho=7800, omega=100, degree=2))))

bcs = collections.OrderedDict()
bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}
bcs["bending"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'force', 'value': Constant((0, 1e6, 0))}
run("elasticity_force", LinearElasticitySolver.LinearElasticitySolver(elasticity_settings(bcs)))

# --- case 6a2: a 2-D (plane strain) problem: the reference sends it to solve_linear_problem, not solve_amg
# (LinearElasticitySolver.py:247-253); per-component clamp, pressure on the top edge, body force
def elasticity_2d_settings():
    from dolfin import RectangleMesh
    mesh = RectangleMesh(Point(0, 0), Point(4, 1), 8, 2)
    st = copy.deepcopy(SolverBase.default_case_settings)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 1)
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), Constant(0))}
    bcs["roller"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'displacement', 'value': (Constant(1e-3), None)}
    bcs["load"] = {'boundary': Top(), 'boundary_id': 3, 'type': 'stress', 'value': Constant((0, -5e6))}
    st['boundary_conditions'] = bcs
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = None
    st['body_source'] = Constant((0, -76440.0))
    return st


class Top(SubDomain):
    pass


run("elasticity_2d", LinearElasticitySolver.LinearElasticitySolver(elasticity_2d_settings()))

# --- case 6b: the reference's own elasticity example, as its __main__ runs it (examples/test_linear_elasticity.py:42-129,
# 170: BoxMesh 40x10x10, VectorFunctionSpace(mesh, "Lagrange", 2), left face (0, free, free), right face (0, 0, 1e-3),
# body force, thermal stress at 343 K)
def example_settings():
    mesh = BoxMesh(Point(0, 0, 0), Point(10, 1, 1), 40, 10, 10)
    st = copy.copy(SolverBase.default_case_settings)            # shallow, as the example does (Appendix B-Q12)
    st['material'] = {'name': 'steel', 'elastic_modulus': 2e11, 'poisson_ratio': 0.27, 'density': 7800,
                      'thermal_expansion_coefficient': 2e-6}
    st['function_space'] = VectorFunctionSpace(mesh, "Lagrange", 2)
    bcs = collections.OrderedDict()
    bcs["fixed"] = {'boundary': Left(), 'boundary_id': 1, 'type': 'Dirichlet', 'value': (Constant(0), None, None)}
    bcs["displ"] = {'boundary': Right(), 'boundary_id': 2, 'type': 'Dirichlet', 'value': Constant((0, 0, 1 * 1e-3))}
    st['boundary_conditions'] = bcs
    st['solver_settings'] = copy.deepcopy(st['solver_settings'])
    st['solver_settings']['reference_values'] = {'temperature': 293}
    st['report_settings'] = dict(QUIET)
    st['temperature_distribution'] = Expression("343", degree=2)
    st['body_source'] = Expression(("10*rho", "0", "0.0"), omega=100, rho=7800, degree=2)
    return st


run("elasticity_example_p2", LinearElasticitySolver.LinearElasticitySolver(example_settings()))
out["elasticity_example_p2"]["function_space"] = {"family": "Lagrange", "degree": 2, "mesh": "BoxMesh((0,0,0),(10,1,1),40,10,10)"}

# --- Taylor-Hood Navier-Stokes (CoupledNavierStokesSolver.py) ---------------------------------------
from FenicsSolver import CoupledNavierStokesSolver                                                     # noqa: E402


def ns_settings(transient, body_source=None, nonlinear=True):
    mesh = UnitCubeMesh(4, 4, 4)
    walls = AutoSubDomain(lambda x, on_boundary: on_boundary)
    lid = AutoSubDomain(lambda x, on_boundary: on_boundary)
    bcs = collections.OrderedDict()
    bcs["walls"] = {'boundary': walls, 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': lid, 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    st = copy.deepcopy(SolverBase.default_case_settings)
    st.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG',
               'boundary_conditions': bcs, 'body_source': body_source,
               'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
               'material': {'density': 2.0, 'kinematic_viscosity': 0.01}})
    st['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01,
                                                   'ending_time': 0.01}
    st['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    st['report_settings'] = dict(QUIET)
    return st


run("navier_stokes_steady", CoupledNavierStokesSolver.CoupledNavierStokesSolver(ns_settings(False)))
run("navier_stokes_transient_gravity",
    CoupledNavierStokesSolver.CoupledNavierStokesSolver(ns_settings(True, body_source=Constant((0, 0, -9.8)))))

st = ns_settings(False)
st['boundary_conditions']["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 3,
                                       'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(5.0)}]}
st['boundary_conditions']["far"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 4,
                                    'values': [{'variable': "pressure", 'type': 'farfield', 'value': Constant(0.0)}]}
run("navier_stokes_pressure_boundaries", CoupledNavierStokesSolver.CoupledNavierStokesSolver(st))

# G2 streamline term (advection_settings, CoupledNavierStokesSolver.py:334-363): convection dominated (Re > 1), steady and
# transient, and the Re <= 1 branch
for name, transient, re_ in (("navier_stokes_g2_steady", False, 100), ("navier_stokes_g2_transient", True, 100),
                             ("navier_stokes_g2_low_re", False, 0.5)):
    st = ns_settings(transient)
    st['advection_settings'] = {'stabilization_method': 'G2', 'Re': re_, 'kappa1': 4, 'kappa2': 2}
    try:
        run(name, CoupledNavierStokesSolver.CoupledNavierStokesSolver(st))
    except Exception as e:            # the reference's own failure is what gets pinned then
        out[name] = {"reference_raises": "%s: %s" % (type(e).__name__, e)}

# non-Newtonian material (CoupledNavierStokesSolver.viscosity :194-213, the branch without a temperature), with a pressure outlet
# so that the viscosity of the boundary term (:401) is recorded as well
st = ns_settings(False)
st['material'] = {'density': 2.0, 'kinematic_viscosity': 0.01, 'Newtonian': False}
st['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 1.0e5}
st['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 1.0e5}
st['boundary_conditions']["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 3,
                                       'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(1.0e5)}]}
try:
    run("navier_stokes_non_newtonian", CoupledNavierStokesSolver.CoupledNavierStokesSolver(st))
except Exception as e:
    out["navier_stokes_non_newtonian"] = {"reference_raises": "%s: %s" % (type(e).__name__, e)}

# coupled temperature (solving_temperature, CoupledNavierStokesSolver.py:236-239, 247-286)
st = ns_settings(False)
st['solving_temperature'] = True
st['material'] = {'density': 2.0, 'kinematic_viscosity': 0.01, 'specific_heat_capacity': 3.0, 'thermal_conductivity': 0.1}
st['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 0, 'temperature': 320}
st['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0, 'temperature': 300}
st['boundary_conditions']['walls']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(350)})
st['boundary_conditions']['lid']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)})
try:
    run("navier_stokes_coupled_temperature", CoupledNavierStokesSolver.CoupledNavierStokesSolver(st))
except Exception as e:
    import traceback
    out["navier_stokes_coupled_temperature"] = {"reference_raises": "%s: %s" % (type(e).__name__, e),
                                                "where": traceback.format_exc().strip().splitlines()[-3].strip()}

# 2-D Taylor-Hood: the set-up of the reference's own CFD example (examples/test_cfd_solver.py:83-170, UnitSquareMesh channel:
# no-slip side walls, velocity inlet at the bottom, pressure outlet at the top), steady and transient with gravity
from dolfin import UnitSquareMesh                                                                      # noqa: E402


def ns2d_settings(transient, body_source=None):
    mesh = UnitSquareMesh(4, 6)
    bcs = collections.OrderedDict()
    bcs["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 3,
                     'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(2.0)}]}
    bcs["static"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                     'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0))}]}
    bcs["inlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 2,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 1))}]}
    st = copy.deepcopy(SolverBase.default_case_settings)
    st.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'fe_family': 'CG',
               'boundary_conditions': bcs, 'body_source': body_source,
               'initial_values': {'velocity': (0, 0.2), 'pressure': 0},
               'material': {'density': 1.5, 'kinematic_viscosity': 0.1}})
    st['solver_settings']['transient_settings'] = {'transient': transient, 'starting_time': 0.0, 'time_step': 0.01,
                                                   'ending_time': 0.01}
    st['solver_settings']['reference_values'] = {'velocity': (1, 1), 'pressure': 0}
    st['report_settings'] = dict(QUIET)
    return st


for name, transient, body in (("navier_stokes_2d_steady", False, None), ("navier_stokes_2d_transient_gravity", True, Constant((0, -9.8)))):
    try:
        run(name, CoupledNavierStokesSolver.CoupledNavierStokesSolver(ns2d_settings(transient, body)))
    except Exception as e:
        import traceback
        out[name] = {"reference_raises": "%s: %s" % (type(e).__name__, e), "where": traceback.format_exc().strip().splitlines()[-3].strip()}

# (last, so that the auto-numbered symbols of the cases above keep their names)
# non-Newtonian material WITH the coupled temperature: nu (1 + 0.1 p / p_ref)(1 - 0.2 T / T_ref) (CoupledNavierStokesSolver.py:199-203),
# with a pressure outlet so that the viscosity of the boundary term (:401) is recorded as well
st = ns_settings(False)
st['solving_temperature'] = True
st['material'] = {'density': 2.0, 'kinematic_viscosity': 0.01, 'specific_heat_capacity': 3.0, 'thermal_conductivity': 0.1, 'Newtonian': False}
st['initial_values'] = {'velocity': (0, 0, 0), 'pressure': 1.0e5, 'temperature': 320}
st['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 1.0e5, 'temperature': 300}
st['boundary_conditions']['walls']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(350)})
st['boundary_conditions']['lid']['values'].append({'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(300)})
st['boundary_conditions']["outlet"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 3,
                                       'values': [{'variable': "pressure", 'type': 'Dirichlet', 'value': Constant(1.0e5)},
                                                  # (the reference's thermal form reads bc['type'] of EVERY boundary: one without a
                                                  # temperature entry is a KeyError there, ScalarTransportSolver.py:169)
                                                  {'variable': 'temperature', 'type': 'Dirichlet', 'value': Constant(330)}]}
try:
    run("navier_stokes_non_newtonian_temperature", CoupledNavierStokesSolver.CoupledNavierStokesSolver(st))
except Exception as e:
    import traceback
    out["navier_stokes_non_newtonian_temperature"] = {"reference_raises": "%s: %s" % (type(e).__name__, e),
                                                      "where": traceback.format_exc().strip().splitlines()[-3].strip()}

path = os.path.join(HERE, "reference_forms.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", path)
for k, v in out.items():
    print("==", k)
    if "reference_raises" in v:
        print("   reference raises", v["reference_raises"])
        continue
    for sv in v["solves"][:1]:
        print("  ", sv["kind"])
        for b in sv["bcs"]:
            print("     bc", b)
        for t in sv["terms"]:
            print("     %+d  %s * %s" % (t["sign"], t["integrand"], t["measure"]))
