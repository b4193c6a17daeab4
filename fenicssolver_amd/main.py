"""Case entry point: JSON/dict settings -> solver object -> solve().

Counterpart of FenicsSolver/main.py:65-95 of the reference (``load_settings``
and ``main`` keep their names, arguments and error behaviour: a dict is passed
through, a path is parsed as JSON, anything else raises TypeError; an unknown
``solver_name`` raises NameError).  Solver modules are imported lazily, as the
reference does, so ``load_settings`` works without a GPU.
"""
from __future__ import annotations

import json
import os


def load_settings(case_input):
    if isinstance(case_input, dict):
        return case_input
    if isinstance(case_input, (str, bytes, os.PathLike)) and os.path.exists(case_input):
        with open(case_input, "r") as fh:
            settings = json.load(fh)
        # a mesh path in a case file is relative to the case file, which is what the
        # reference relies on by being started from its own folder (main.py:98-100)
        mesh = settings.get("mesh")
        if isinstance(mesh, str) and not os.path.isabs(mesh) and not os.path.exists(mesh):
            candidate = os.path.join(os.path.dirname(os.path.abspath(case_input)), os.path.basename(mesh))
            if os.path.exists(candidate):
                settings["mesh"] = candidate
        return settings
    raise TypeError('{} is not supported as case input, only path string or dict'.format(type(case_input)))


_SOLVERS = ("CoupledNavierStokesSolver", "ScalarTransportSolver", "LinearElasticitySolver")


def main(case_input):
    settings = load_settings(case_input)
    solver_name = settings['solver_name']
    if solver_name == "ScalarTransportSolver":
        from .ScalarTransportSolver import ScalarTransportSolver as cls
    elif solver_name == "LinearElasticitySolver":
        from .LinearElasticitySolver import LinearElasticitySolver as cls
    elif solver_name == "CoupledNavierStokesSolver":
        from .CoupledNavierStokesSolver import CoupledNavierStokesSolver as cls
    else:
        raise NameError('Solver name : {} is not supported, choose one of {}'.format(solver_name, _SOLVERS))
    solver = cls(settings)
    solver.solve()
    solver.plot()
    return solver
