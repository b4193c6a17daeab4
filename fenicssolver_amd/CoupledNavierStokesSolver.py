"""Drop-in counterpart of FenicsSolver/CoupledNavierStokesSolver.py (incompressible, laminar, Taylor-Hood),
MI355X back end: the coupled velocity-pressure system is assembled and solved on the device
(fs_assemble_navier_stokes / fs_saddle_solve) instead of DOLFIN + PETSc LU.

Same settings dict as the reference: ``fe_degree`` = pressure degree (velocity is one
higher, CoupledNavierStokesSolver.py:84-88), ``material`` with ``density`` and ``kinematic_viscosity``,
``boundary_conditions`` whose ``values`` are a list/dict of {'variable': 'velocity'|'pressure', 'type': ..., 'value': ...},
``initial_values`` {'velocity': (..), 'pressure': ..}, ``body_source`` (acceleration, e.g. gravity), transient_settings
(backward Euler, :367-381).  ``solver.using_nonlinear_solver`` (default True) selects Newton, False the Picard loop
with under-relaxation 0.7 of :492-528.

Built: velocity Dirichlet conditions (constants, tuples, C-string Expressions; per-time-step lists are not),
pressure Dirichlet (inlet / outlet, with the reference's boundary integrals p n.v ds - nu ((grad u + grad u^T) n).v ds)
and pressure 'farfield' boundaries, body force, steady and backward-Euler transient, Newton and Picard.
ALE reference frames with a constant ``mesh_velocity`` (:321-329), ``viscous_stress`` (the consistent L2 projection of
nu (grad u + grad u^T) - p I onto CG1, nine mass-matrix solves on the device), ``boundary_traction`` and
``calc_drag_and_lift`` (the reference's versions reference an undefined ``self.ds`` and index symbols, Appendix B-Q13;
the evident intent is built: facet integrals of the projected stress over ``boundary_facets``).
G2 stabilisation (``advection_settings = {'stabilization_method': 'G2', 'Re':, 'kappa1':, 'kappa2':}``, :334-363):
F -= delta1 (a.grad u).(a.grad v) dx with the reference's sign and its delta1 (kappa1 h^2 for Re <= 1, kappa1/2 h/|a|
otherwise; the transient branch, where the reference reads an undefined ``time_iter_`` and raises NameError, uses the
step's dt in its formula).  The term enters the Jacobian with the advecting velocity frozen at the iterate; the system is
written for the new iterate, so the Newton residual is the exact one and the fixed point is the reference's.
Non-Newtonian material (``material['Newtonian'] = False``, ``viscosity`` :194-213 without a temperature): nu (p / p_ref)^0.1
with the pressure of the current iterate, evaluated at the quadrature points on the device (``viscosity_law``).
``solving_temperature`` (:236-239, 247-286): the transport equation of the temperature on the pressure space, IP-stabilised,
convected by the velocity iterate - solved after the flow of every step (block-triangular for a Newtonian fluid); ``split``
then returns (u, p, T) as the reference's MixedElement([V, Q, Q]) does.
2-D (triangles; the reference's own CFD example runs on UnitSquareMesh(40, 100), examples/test_cfd_solver.py:83): the same
block layout with a dummy third velocity slot (mixed.py), 6-node element kernel with Radon's 7-point rule, edge integrals of
the pressure boundaries, the same FGMRES / block preconditioner; G2, ALE, the non-Newtonian law and solving_temperature
and the stress post-processing (viscous_stress, boundary_traction, drag / lift) included.
With solving_temperature the non-Newtonian law is nu (1 + 0.1 p/p_ref)(1 - 0.2 T/T_ref) (:199-203): the temperature enters the
momentum equation, and the step is solved as the fixed point of flow(T) -> T(u) (temperature_law, _solve_coupled_step).
Raise: velocity 'symmetry' / 'farfield' (the reference's own forms for them are not valid UFL), non-constant mesh velocities.
"""
from __future__ import annotations

import numbers

import numpy as np

from .SolverBase import SolverBase, SolverError
from .fem import Measure, Function, Constant, Expression, DirichletBC, FunctionSpace
from .mixed import TaylorHoodSpace, split
from . import forms


class CoupledNavierStokesSolver(SolverBase):
    """incompressible and laminar flow"""

    def __init__(self, case_input):
        self.solving_temperature = bool(case_input.get('solving_temperature', False)) if isinstance(case_input, dict) else False
        self._Tsolver = None
        SolverBase.__init__(self, case_input)
        self.solving_temperature = self.solving_temperature or bool(self.settings.get('solving_temperature', False))
        self.compressible = False
        self.using_nonlinear_solver = True
        self.settings['mixed_variable'] = ('velocity', 'pressure')

    # ------------------------------------------------------------------ space
    def generate_function_space(self, periodic_boundary):
        self.vel_degree = self.settings['fe_degree'] + 1
        self.pressure_degree = self.settings['fe_degree']
        self.is_mixed_function_space = True
        self._update_function_space(periodic_boundary)

    def _update_function_space(self, periodic_boundary=None):
        self.function_space = TaylorHoodSpace(self.mesh, self.settings['fe_family'], self.pressure_degree,
                                              constrained_domain=periodic_boundary)
        self.velocity_subfunction_space = self.function_space.sub(0)

    def get_variable_name(self):
        return "velocity_pressure"

    # ------------------------------------------------------------------ values
    def get_body_source(self):
        """(CoupledNavierStokesSolver.py:114-123) default gravity along -z in 3D, along -y in 2D."""
        bs = self.settings.get('body_source')
        if bs:
            return self._vector3(bs, 'body_source')
        return np.array([0.0, 0.0, -9.8]) if self.dimension == 3 else np.array([0.0, -9.8, 0.0])

    def _vector3(self, value, what):
        """A constant vector of the mesh dimension, padded to the three slots the device forms carry."""
        if isinstance(value, Constant):
            value = value.values()
        d = getattr(self, 'dimension', 3)
        if isinstance(value, (tuple, list, np.ndarray)) and len(value) == d and \
                all(isinstance(x, numbers.Number) for x in value):
            return np.concatenate([np.asarray(value, dtype=np.float64), np.zeros(3 - d)])
        raise SolverError("{} must be {} numbers (a Constant or a tuple) on this back end".format(what, d))

    def get_initial_field(self):
        W = self.function_space
        iv = self.initial_values
        up0 = Function(W)
        if isinstance(iv, Function):
            if iv.vector().size() != up0.vector().size():
                raise SolverError("initial_values Function lives on a different space")
            up0.assign(iv)
            return up0
        a = up0.vector().array().reshape(-1, 4)
        d = self.dimension
        vel = iv.get('velocity', d * (0.0,)) if isinstance(iv, dict) else d * (0.0,)
        pre = iv.get('pressure', 0.0) if isinstance(iv, dict) else 0.0
        if len(vel) != d:
            raise SolverError("initial_values['velocity'] must have {} components".format(d))
        co = W.node_coordinates()
        expr = Expression(tuple(str(v) for v in list(vel) + [pre]), degree=self.settings['fe_degree'])
        vals = expr.eval_points(co)
        a[:, :d] = vals[:, :d]
        a[:self.mesh.num_vertices(), 3] = vals[:self.mesh.num_vertices(), d]
        return up0

    def viscosity_law(self):
        """None (Newtonian) or (p_ref, exponent) of the reference's non-Newtonian law (:194-213, the branch without a
        temperature): nu(p) = nu * pow(p / reference_values['pressure'], 0.1), evaluated on the current iterate as the
        reference does (F_static :306 takes up_0, the boundary terms :401 w_current)."""
        if 'Newtonian' in self.material and (not self.material['Newtonian']):
            if self.solving_temperature:
                return None          # nu(p, T): attached to the device space together with the temperature (temperature_law)
            pref = (getattr(self, 'reference_values', None) or {}).get('pressure')
            if pref is None or not float(pref) > 0.0:
                raise SolverError("non-Newtonian viscosity needs a positive reference_values['pressure']")
            return (float(pref), 0.1)
        return None

    def temperature_law(self):
        """None, or ('pT', p_ref, 0.1, T_ref, 0.2): the reference's non-Newtonian law with solving_temperature (:199-203),
        nu (1 + (p/p_ref) 0.1) (1 - (T/T_ref) 0.2) on the current iterate (u, p, T).  The temperature then enters the momentum
        equation: the monolithic (u, p, T) system of the reference is solved here as the fixed point of  flow with T frozen ->
        temperature with the new velocity  (solve_current_step); at convergence both residuals of the reference's form vanish."""
        if not (self.solving_temperature and 'Newtonian' in self.material and not self.material['Newtonian']):
            return None
        rv = getattr(self, 'reference_values', None) or {}
        pref, tref = rv.get('pressure'), rv.get('temperature')
        if pref is None or not float(pref) > 0.0 or tref is None or float(tref) == 0.0:
            raise SolverError("non-Newtonian viscosity with solving_temperature needs reference_values['pressure'] > 0 and "
                              "a non-zero reference_values['temperature']")
        return ('pT', float(pref), 0.1, float(tref), 0.2)

    def _attach_temperature_law(self):
        """Hand the law and the CURRENT temperature (vertex values in the device space's local numbering) to the Taylor-Hood
        device space: every routine that evaluates nu on it - cell terms, pressure-boundary traction, stress projection - uses it."""
        from . import backend
        law = self.temperature_law()
        if law is None:
            # the law lives on the (shared) device space: a solver without one - another solver on the same function_space, or this
            # one after its material went back to Newtonian / pressure-only - must not assemble with what was attached before
            root = self.function_space.root() if hasattr(self.function_space, 'root') else self.function_space
            dW = getattr(root, '_device', None)           # (no device space yet: nothing can be attached to it)
            if dW is not None and (getattr(dW, '_law_temperature', None) is not None or self.__dict__.get('_law_T') is not None):
                backend.set_viscosity_law(dW, None)
                self._law_T = None
            return
        dW = self.function_space.device()
        Ts = self._temperature_solver()
        vals = Ts.w_current.vector()._values()
        loc = self.function_space.localizer()
        # one value per LOCAL NODE of the flow space (read at the vertex nodes): on one GPU the vertices are the first nodes; a
        # decomposed CG2 space orders [owned vertices | owned edges | ghost vertices | ghost edges]
        if loc is None:
            arr = np.zeros(dW.n_local // 4)
            arr[:len(vals)] = vals
        elif hasattr(loc, 'l2h'):                     # distributed box: host vectors are local, host nodes = vertices, then edges
            arr = np.zeros(len(loc.l2h))
            m = loc.l2h < len(vals)
            arr[m] = vals[loc.l2h[m]]
        else:                                         # replicated host mesh: global node ids, the vertices first
            arr = np.zeros(len(loc.l2g))
            m = loc.l2g < loc.n_global_vertices
            arr[m] = vals[loc.l2g[m]]
        vals = arr
        vec = self.__dict__.get('_law_T')
        if vec is None or vec.n != len(vals):
            vec = self._law_T = backend.DeviceVector(len(vals))
        vec.set(vals)
        backend.set_viscosity_law(dW, law, vec)

    def viscosity(self, current_w=None):
        """The constant kinematic viscosity nu0; a non-Newtonian material multiplies it by (p / p_ref)^0.1 inside the device
        kernels (viscosity_law)."""
        nu = self.material['kinematic_viscosity']
        if isinstance(nu, Constant):
            nu = float(nu)
        if not isinstance(nu, numbers.Number):
            raise SolverError("kinematic_viscosity must be a number on this back end")
        return float(nu)

    # ------------------------------------------------------------------ form
    def generate_form(self, time_iter_, trial_function, test_function, up_current, up_prev):
        self._attach_temperature_law()      # nu(p, T) belongs to the form: attached (or detached) where the form is made, not only in the coupled step
        F = self._cell_form(time_iter_, up_current, up_prev)
        bcs, F.pressure_boundaries = self.update_boundary_conditions(time_iter_, trial_function, test_function,
                                                                                  Measure("ds", subdomain_data=self.boundary_facets))
        self.J = F if self.using_nonlinear_solver else None
        return F, bcs

    def _cell_form(self, time_iter_, up_current, up_prev):
        F = forms.NavierStokesForm(self.function_space)
        F.nu = self.viscosity()
        F.viscosity_law = self.viscosity_law()
        F.rho = float(self.material['density'])
        F.w_current = up_current
        F.w_prev = up_prev
        F.newton = bool(self.using_nonlinear_solver)
        if self.settings.get('body_source'):          # "just gravity, without * rho" (:315-316)
            F.body_force = self.get_body_source()
        rfs = self.settings.get('reference_frame_settings')
        if rfs:
            if rfs.get('type') != 'ALE':
                raise SolverError('reference_frame_settings type `{}` is not supported'.format(rfs.get('type')))
            mv = rfs.get('mesh_velocity')
            if callable(mv) and self.transient_settings['transient']:
                mv = mv(self.get_current_time())
            F.mesh_velocity = self._vector3(mv, "reference_frame_settings['mesh_velocity'] (a constant vector)")
        ads = self.settings.get('advection_settings') or {}
        if ads.get('stabilization_method') == 'G2':
            # F -= delta1 inner(dot(a, grad(u)), dot(a, grad(v))) dx (:334-363); delta2 / kappa2 only enter the density
            # term the reference leaves commented out
            for key in ('Re', 'kappa1'):
                if key not in ads:
                    raise SolverError("advection_settings for G2 need '{}' (the reference reads Re, kappa1, kappa2)".format(key))
            F.g2 = (1 if float(ads['Re']) <= 1 else 2, float(ads['kappa1']))
        elif ads.get('stabilization_method'):
            raise SolverError("advection stabilisation '{}' is not built".format(ads['stabilization_method']))
        if self.transient_settings['transient']:
            F.inv_dt = 1.0 / self.get_time_step(time_iter_)      # backward Euler (:367-381)
        return F

    def _boundary_value(self, value):
        """The time-dependent forms translate_value accepts (SolverBase.py:365-366, 376-377): one entry per time step (a
        sequence longer than the dimension, as examples/test_cfd_solver.py:127-129 passes for its inlet) or a callable
        of the time; everything else goes to DirichletBC as it is."""
        if not self.transient_settings['transient']:
            return value
        if isinstance(value, (list, tuple)) and len(value) > self.dimension:
            if self.current_step >= len(value):
                raise SolverError("boundary value list has {} entries, time step {} asked".format(len(value), self.current_step))
            return value[self.current_step]
        if callable(value) and not isinstance(value, (Constant, Function)) and not hasattr(value, 'eval_points'):
            return value(self.get_current_time())
        return value

    # The reference splits its form into F_static / F_transient (:288-365, 367-381), which subclasses and the FSI solver call;
    # here both return the form description generate_form builds (without / with the backward-Euler term).
    def F_static(self, trial_function, test_function, up_0):
        saved = self.transient_settings
        try:
            self.transient_settings = dict(saved, transient=False)
            return self._cell_form(0, up_0, None)
        finally:
            self.transient_settings = saved

    def F_transient(self, time_iter_, trial_function, test_function, up_current, up_prev):
        F = self.F_static(trial_function, test_function, up_current)
        F.w_prev = up_prev
        F.inv_dt = 1.0 / self.get_time_step(time_iter_)      # backward Euler (:381)
        return F

    def update_solver_function_space(self, periodic_boundary=None):
        """After a mesh update (FSI, :104-117): rebuild the mixed space and carry the iterates over (same topology)."""
        old_c, old_p = self.w_current.vector().get_local(), self.w_prev.vector().get_local()
        self._update_function_space(periodic_boundary)
        self.trial_function = self.test_function = None
        self.w_current, self.w_prev = Function(self.function_space), Function(self.function_space)
        self.w_current.vector().set_local(old_c)
        self.w_prev.vector().set_local(old_p)
        self._ns_ctx = None

    # ---- the coupled temperature equation (solving_temperature, :236-239 and :247-286) ---------------------------------
    # The reference adds the form of a ScalarTransportSolver on W.sub(2) - the flow case's own settings with scalar_name
    # 'temperature', IP stabilisation alpha = 0.1, convective velocity = the CURRENT velocity iterate - to the flow form
    # and solves (u, p, T) monolithically.  For a Newtonian fluid nothing of T enters the momentum or continuity equations
    # (the viscous heating stays commented out, :281-285), so the converged (u, p) is the flow solution and T solves the
    # transport equation with that velocity: the block-triangular system is solved here in that order, step by step.
    def generate_thermal_form(self, time_iter_, trial_function, test_function, up_current, up_prev):
        Ts = self._temperature_solver()
        Ts.convective_velocity = split(up_current)[0]
        return Ts.generate_form(time_iter_, None, None, Ts.w_current, Ts.w_prev)

    def _temperature_solver(self):
        if self._Tsolver is None:
            import copy
            from collections import OrderedDict
            from .ScalarTransportSolver import ScalarTransportSolver
            from . import case
            ts = copy.copy(self.settings)                    # "Tsettings = copy.copy(self.settings)" (:255)
            ts['scalar_name'] = 'temperature'
            ts['mesh'] = None
            from . import parallel
            # MixedElement([V, Q, Q]): T lives on Q (:94-95).  Several GPUs: a space of its own - its interior-penalty term cuts a
            # two-layer part with its own device mesh, the flow's pressure space stays on the mesh the stress projections share
            ts['function_space'] = FunctionSpace(self.mesh, "CG", 1) if parallel.active() else self.function_space.pressure_space()
            ts['advection_settings'] = {'stabilization_method': 'IP', 'alpha': 0.1}     # (:262)
            ts['convective_velocity'] = None
            ts['solving_temperature'] = False
            if ts.get('body_source'):
                # the copied settings would hand the flow's body force (a vector) to the scalar equation; the reference
                # leaves "Tsettings['body_source'] = # incomplate form?" open (:263)
                self.logger.warning("solving_temperature: the flow's body_source is not a heat source; the temperature "
                                    "equation runs without one")
                ts['body_source'] = None
            keep = OrderedDict()
            for name, bc in (self.settings.get('boundary_conditions') or {}).items():
                sub = case.boundary_variable(bc, 'temperature')
                if sub is not bc or bc.get('variable') == 'temperature':
                    keep[name] = bc                          # boundaries without a temperature entry stay natural (zero flux)
            ts['boundary_conditions'] = keep
            self._Tsolver = ScalarTransportSolver(ts)
            self._Tsolver.init_solver()
        return self._Tsolver

    def solve_current_step(self):
        if self.solving_temperature and self.temperature_law() is not None:
            return self._solve_coupled_step()
        SolverBase.solve_current_step(self)
        if self.solving_temperature:
            Ts = self._temperature_solver()
            Ts.current_step, Ts.current_time = self.current_step, getattr(self, 'current_time', 0.0)
            Ts.convective_velocity = split(self.w_current)[0]       # the P2 velocity just solved for
            Ts.solve_current_step()
            self.w_current._temperature = Ts.w_current
            self.result = self.w_current

    def _solve_coupled_step(self):
        """nu(p, T) (temperature_law): the step's (u, p, T) as the fixed point of  flow(T) -> T(u).  The history rotates once, in
        the first pass (SolverBase.solve_current_step); later passes re-generate the forms from the current iterates and the same
        previous step.  Stops when the temperature changes by less than coupling_relative_tolerance (default 1e-10) of its range."""
        sp = self.solver_settings.get('solver_parameters', {}) or {}
        tol = float(sp.get('coupling_relative_tolerance', 1e-10))
        max_it = int(sp.get('coupling_maximum_iterations', 50))
        Ts = self._temperature_solver()
        Ts.current_step, Ts.current_time = self.current_step, getattr(self, 'current_time', 0.0)
        self.coupling_iterations = 0
        for k in range(max_it):
            self._attach_temperature_law()
            if k == 0:
                SolverBase.solve_current_step(self)
            else:
                F, bcs = self.generate_form(self.current_step, self.trial_function, self.test_function, self.w_current, self.w_prev)
                self.w_current = self.solve_form(F, self.w_current, bcs)
            T_old = Ts.w_current.vector()._values().copy()
            Ts.convective_velocity = split(self.w_current)[0]
            if k == 0:
                Ts.solve_current_step()
            else:
                Ft, bct = Ts.generate_form(Ts.current_step, Ts.trial_function, Ts.test_function, Ts.w_current, Ts.w_prev)
                Ts.w_current = Ts.solve_form(Ft, Ts.w_current, bct)
            T_new = Ts.w_current.vector()._values()
            self.coupling_iterations = k + 1
            change = np.abs(T_new - T_old).max() / max(np.abs(T_new).max(), 1e-300)
            if k > 0 and change <= tol:
                break
        else:
            raise SolverError("flow / temperature coupling did not converge in {} passes (last change {:.3e})".format(max_it, change))
        self._attach_temperature_law()           # post-processing (viscous_stress, drag / lift) sees the converged temperature
        self.w_current._temperature = Ts.w_current
        self.result = self.w_current

    def temperature(self):
        return self._Tsolver.w_current if self._Tsolver is not None else None

    def viscous_heat(self, u, p):
        raise SolverError("viscous_heat: the reference projects a scalar, inner(sigma, grad(u)), onto the VECTOR velocity space "
                          "(:187-192, 'not tested code'): there is no such projection; sigma is available from viscous_stress()")

    def update_boundary_conditions(self, time_iter_, trial_function, test_function, ds):
        """-> (Dirichlet conditions, pressure-boundary integrals) (CoupledNavierStokesSolver.py:383-490)."""
        W = self.function_space
        bcs, pressure_terms = [], []
        for key, boundary in self.boundary_conditions.items():
            if boundary.get('coupling') == 'FSI' and 'values' not in boundary:
                boundary['values'] = [{'variable': "velocity", 'type': 'Dirichlet', 'value': self.dimension * (0.0,)}]
            values = boundary.get('values')
            if values is None:
                raise SolverError("boundary '{}' has no 'values'".format(key))
            bc_values = values if isinstance(values, list) else list(values.values())
            bid = boundary['boundary_id']
            for bc in bc_values:
                var, typ = bc.get('variable'), bc.get('type')
                if var == 'velocity':
                    if typ == 'Dirichlet':
                        value = self._boundary_value(bc['value'])
                        if isinstance(value, (tuple, list, np.ndarray)):
                            value = Constant(tuple(float(x) for x in value))
                        bcs.append(DirichletBC(W.sub(0), value, self.boundary_facets, bid))
                    elif typ == 'Neumann':
                        # the reference builds a NotImplementedError without raising it (:431): a no-op
                        self.logger.warning("velocity boundary type `Neumann` is a no-op, as in the reference")
                    elif typ in ('symmetry', 'farfield'):
                        raise SolverError("velocity boundary type `{}`: the reference's form for it is not valid UFL "
                                          "(inner of a scalar and a vector, :434 / product of two vectors, :437); "
                                          "not built".format(typ))
                    else:
                        self.logger.warning('velocity boundary type`%s` is not supported', typ)
                elif var == 'pressure':
                    if typ == 'Dirichlet':      # pressure inlet or outlet (:445-453)
                        value = self._boundary_value(bc['value'])
                        if isinstance(value, numbers.Number):
                            value = Constant(float(value))
                        bcs.append(DirichletBC(W.sub(1), value, self.boundary_facets, bid))
                        pressure_terms.append((bid, value))
                    elif typ == 'farfield':     # no viscous stress (:459-460)
                        pressure_terms.append((bid, None))
                    elif typ in ('symmetry',):
                        pass
                    elif typ == 'Neumann':
                        self.logger.warning("pressure boundary type `Neumann` is a no-op, as in the reference")
                    else:
                        self.logger.warning('pressure boundary type`%s` is not supported thus ignored', typ)
                elif var == 'temperature':
                    continue      # "boundary setup is done in scalar transport for incompressible flow" (:483)
                else:
                    self.logger.warning('boundary variable `%s` is not handled by the incompressible flow solver', var)
        return bcs, pressure_terms

    def solve_form(self, F, up_, Dirichlet_bcs_up):
        if self.using_nonlinear_solver:
            return self.solve_nonlinear_problem(F, up_, Dirichlet_bcs_up, self.J)
        # Picard iteration with under-relaxation (:497-528)
        F.newton = False
        iter_, max_iter, eps, tol, under_relax_ratio = 0, 50, 1.0, 1e-4, 0.7
        while iter_ < max_iter and eps > tol:
            old = up_.vector().get_local()
            up_ = self.solve_linear_problem(F, up_, Dirichlet_bcs_up)
            diff = up_.vector().get_local() - old
            eps = float(np.linalg.norm(diff, ord=np.inf))
            self.logger.info("iter = %d; eps_up = %e", iter_, eps)
            up_.vector().set_local(old + diff * under_relax_ratio)
            iter_ += 1
        self.picard_iterations = iter_
        return up_

    def save(self, result_filename):
        """PVD collection; each frame is a VTU with the velocity (vertex values) and the pressure."""
        import os
        from .SolverBase import write_vtu
        assert result_filename[-4:] == '.pvd'
        root = result_filename[:-4]
        if not hasattr(self, '_saved_frames'):
            self._saved_frames = []
        parts = split(self.w_current)
        u, p = parts[0], parts[1]
        vtu = "%s%06d.vtu" % (root, len(self._saved_frames))
        write_vtu(vtu, self.mesh, u, "velocity", extra=[(p, "pressure")] + ([(parts[2], "temperature")] if len(parts) > 2 else []))
        self._saved_frames.append((getattr(self, 'current_time', 0.0), os.path.basename(vtu)))
        with open(result_filename, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n  <Collection>\n')
            for t, f in self._saved_frames:
                fh.write('    <DataSet timestep="%g" part="0" file="%s" />\n' % (t, f))
            fh.write('  </Collection>\n</VTKFile>\n')

    def plot(self):
        self.logger.info("plot(): use save() and ParaView for velocity-pressure fields")

    plot_result = plot

    # ------------------------------------------------------------------ post-processing
    def split(self, w=None):
        return split(w if w is not None else self.w_current)

    def viscous_stress(self, up, T_space=None):
        """project(nu (grad u + grad u^T) - p I, TensorFunctionSpace(mesh, 'CG', 1)) (:149-155).  Right-hand sides on the
        device (fs_assemble_viscous_stress), then one CG1 mass-matrix solve per tensor component (Jacobi-CG, 1e-12).
        Returns a Function on TensorFunctionSpace(mesh, 'CG', 1): node_values() is [num_vertices, d * d], row-major
        (d = 3 on tetrahedra, 2 on triangles)."""
        from . import backend
        from .fem import FunctionSpace, TensorFunctionSpace
        W = up.function_space()
        if W is self.function_space:
            self._attach_temperature_law()      # this solver's law (or none), whatever was assembled on the space last
        d = self.dimension
        nt = d * d
        if T_space is None:
            T_space = TensorFunctionSpace(self.mesh, 'CG', 1)
        elif T_space.degree() != 1 or T_space._ncomp != nt:
            raise SolverError('viscous_stress: T_space must be TensorFunctionSpace(mesh, "CG", 1)')
        P = W.pressure_space()
        dW, dP = W.device(), P.device()
        nv = self.mesh.num_vertices()
        loc, ploc = W.localizer(), P.localizer()      # several GPUs: this rank's rows of the mass-matrix solves, gathered at the end
        wh = up.vector()._values()
        wd = backend.DeviceVector(dW.n_local, wh if loc is None else loc.nodes(wh))
        bt = backend.DeviceVector(nt * dP.n_owned)
        backend.assemble_viscous_stress(dW, wd, self.viscosity(), dP, bt, viscosity_law=self.viscosity_law())
        rhs = bt.get().reshape(dP.n_owned, nt)
        M = backend.DeviceMatrix(dP)
        M.assemble(mass=1.0)
        b, x = backend.DeviceVector(dP.n_owned), backend.DeviceVector(dP.n_local)
        out = np.zeros((nv, nt))
        for k in range(nt):
            i, j = divmod(k, d)
            if i > j:                     # sigma is symmetric: the lower triangle copies the upper one
                out[:, k] = out[:, j * d + i]
                continue
            b.set(rhs[:, k])
            st = backend.krylov_solve(M, b, x, rtol=1e-12, max_iter=2000, precond="jacobi", norm="preconditioned")
            if st['converged'] != 1:
                raise SolverError('viscous_stress: the mass-matrix solve did not converge')
            if ploc is not None and getattr(ploc, 'is_local_view', False):
                from . import parallel          # distributed mesh: this rank's vertices, ghosts refreshed
                if parallel.world()[1] > 1:
                    backend.halo_exchange(dP, x)
                xo = x.get()[:nv]
            else:
                xo = x.get()[:dP.n_owned]
                if ploc is not None:
                    from . import parallel
                    xo = parallel.gather_owned(xo, ploc.owned_gids(), ploc.n_global, 1)
            out[:, k] = xo
        sigma = Function(T_space)
        sigma.vector().set_local(out.reshape(-1))
        return sigma

    def _facet_geometry(self, sel):
        """(vertices [nf, d] of the boundary facets with indices sel, outward normal * area (length in 2-D) [nf, d])."""
        mesh = self.mesh
        d = self.dimension
        fv = mesh.facets()[sel].astype(np.int64)
        co = mesh.coordinates()[:, :d]
        X = co[fv]
        if d == 3:
            nrm = 0.5 * np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0])
        else:
            t = X[:, 1] - X[:, 0]
            nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)
        owner = np.full(mesh.num_facets(), -1, dtype=np.int64)
        owner[mesh.cell_facets().ravel()] = np.repeat(np.arange(mesh.num_cells()), d + 1)
        inward = np.einsum("fi,fi->f", nrm, X.mean(axis=1) - co[mesh.cells().astype(np.int64)[owner[sel]]].mean(axis=1)) < 0
        nrm[inward] *= -1.0
        return fv, nrm

    def boundary_traction(self, up, target_space=None):
        """sigma . n on the boundary, as a CG1 vector Function (interior vertices 0): at a boundary vertex the area-weighted
        mean of sigma(vertex) . n over its exterior facets.  (The reference's version, :157-170, calls viscous_stress
        without its second argument and uses undefined index symbols; this is its stated intent: traction = dot(sigma, n).)"""
        from .fem import VectorFunctionSpace
        d = self.dimension
        V = target_space or VectorFunctionSpace(self.mesh, 'CG', 1)
        if V.degree() != 1 or V._ncomp != d:
            raise SolverError('boundary_traction: target_space must be VectorFunctionSpace(mesh, "CG", 1)')
        sig = self.viscous_stress(up).node_values().reshape(-1, d, d)
        fv, nrm = self._facet_geometry(np.nonzero(self.mesh.exterior_facets())[0])
        nv = self.mesh.num_vertices()
        num, den = np.zeros((nv, d)), np.zeros(nv)
        area = np.linalg.norm(nrm, axis=1)
        for k in range(d):
            np.add.at(num, fv[:, k], np.einsum("fij,fj->fi", sig[fv[:, k]], nrm))
            np.add.at(den, fv[:, k], area)
        vals = np.zeros_like(num)
        on = den > 0
        vals[on] = num[on] / den[on, None]
        t = Function(V)
        t.vector().set_local(vals.reshape(-1))
        return t

    def calc_drag_and_lift(self, up, drag_axis_index, lift_axis_index, boundary_index_list):
        """(drag, lift) = -int T[axis, j] n_j ds over the listed boundaries (:172-192), T = viscous_stress(up), n outward.
        With CG1 T the facet integral is area * mean of the facet's vertex tensors (exact)."""
        if not (boundary_index_list and len(boundary_index_list)):
            raise SolverError('Error: boundary_index_list must be specified to calc drag and lift forces')
        d = self.dimension
        T = self.viscous_stress(up).node_values().reshape(-1, d, d)
        fv, nrm = self._facet_geometry(np.concatenate([self.boundary_facets.where(i) for i in boundary_index_list]))
        ploc = up.function_space().pressure_space().localizer()
        if ploc is not None and getattr(ploc, 'is_local_view', False):
            # distributed mesh: the cells between an owned and a ghost vertex plane exist on two ranks - a facet counts on the rank
            # that owns its vertex of smallest global id - and the force is summed over the ranks
            from . import backend, parallel
            g = np.asarray(ploc.l2g)[fv]
            first = fv[np.arange(len(fv)), np.argmin(g, axis=1)]
            mine = first < ploc.n_owned
            force = -np.einsum("fij,fj->i", T[fv[mine]].mean(axis=1), nrm[mine]) if mine.any() else np.zeros(d)
            if parallel.world()[1] > 1:
                force = np.asarray(backend.comm_allreduce_sum(force), dtype=np.float64)
            return float(force[drag_axis_index]), float(force[lift_axis_index])
        force = -np.einsum("fij,fj->i", T[fv].mean(axis=1), nrm)
        return float(force[drag_axis_index]), float(force[lift_axis_index])
