"""ScalarTransportSolver — diffusion of a scalar (heat, electric potential, species),
GPU back end.

Counterpart of FenicsSolver/ScalarTransportSolver.py: same class name, settings
keys, material look-ups (:73-129), boundary types (:142-211), body source (:213-226)
and time scheme (Crank-Nicolson, :287-293).  ``generate_form`` returns a
``forms.ScalarForm`` (which integrals, which coefficients) instead of a UFL form.

Built: steady and Crank-Nicolson transient diffusion on P1 and P2; Dirichlet / Neumann / Robin / flux / HTC boundaries;
body and point sources; convective velocity (non-symmetric operator -> BiCGStab) with SUPG ('SPUG') and interior-penalty
('IP') stabilisation (:244-276, 305-328; P1, IP on one GPU); surface radiation and temperature-dependent conductivity
through Newton (:338-357, 361-376; P1).  Raises SolverError (never silently ignored): ``surface_source`` (undefined
symbols in the reference as well, Appendix B-Q6), advection / nonlinear terms on P2, periodic boundaries.
Reference quirks kept on purpose: Neumann ('fixedGradient') terms are scaled by the
capacity rho*cp, not the conductivity (B-Q8).
"""
from __future__ import annotations

import math
import numbers
import os

import numpy as np

from .fem import Measure, Constant, Expression, Function, DirichletBC, PointSource, Point, nodal_values, is_constant_value
from .SolverBase import SolverBase, SolverError
from . import forms

supported_scalars = {'temperature', 'electric_potential', 'species_concentration'}
VACUUM_PERMITTIVITY = 8.854187817e-12
electric_permittivity_in_vacumm = VACUUM_PERMITTIVITY      # the reference's (misspelt) module-level name, kept for importers


_A1, _B1, _A2, _B2, _A3, _B3 = 0.0673422422100982, 0.3108859192633006, 0.7217942490673264, 0.0927352503108912, \
    0.0455037041256496, 0.4544962958743504
# the 14 points of the degree-5 rule on the tetrahedron, in the order of the device table FS_TET14_QP
_TET14_POINTS = np.array([[_A1, _B1, _B1, _B1], [_B1, _A1, _B1, _B1], [_B1, _B1, _A1, _B1], [_B1, _B1, _B1, _A1],
                          [_A2, _B2, _B2, _B2], [_B2, _A2, _B2, _B2], [_B2, _B2, _A2, _B2], [_B2, _B2, _B2, _A2],
                          [_A3, _A3, _B3, _B3], [_A3, _B3, _A3, _B3], [_A3, _B3, _B3, _A3], [_B3, _A3, _A3, _B3],
                          [_B3, _A3, _B3, _A3], [_B3, _B3, _A3, _A3]])


_TRI6_POINTS = np.array([[0.108103018168070, 0.445948490915965, 0.445948490915965], [0.445948490915965, 0.108103018168070, 0.445948490915965],
                         [0.445948490915965, 0.445948490915965, 0.108103018168070], [0.816847572980459, 0.091576213509771, 0.091576213509771],
                         [0.091576213509771, 0.816847572980459, 0.091576213509771], [0.091576213509771, 0.091576213509771, 0.816847572980459]])


def _p2_shape_at(pts):
    """[nq, 10] P2 basis on the tetrahedron at barycentric points: vertices, then the UFC edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1)."""
    cols = [pts[:, i] * (2.0 * pts[:, i] - 1.0) for i in range(4)]
    for i, j in ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)):
        cols.append(4.0 * pts[:, i] * pts[:, j])
    return np.stack(cols, axis=1)


class ScalarTransportSolver(SolverBase):
    """general scalar transportation (diffusion) solver, exampled by heat transfer"""

    def __init__(self, s):
        SolverBase.__init__(self, s)
        if 'scalar_name' in self.settings:
            self.scalar_name = self.settings['scalar_name'].lower()
        else:
            self.scalar_name = "temperature"
        self.using_diffusion_form = False
        self.nonlinear = False
        self.nonlinear_material = False
        for v in self.material.values():
            if callable(v) and not isinstance(v, (Function, Constant, Expression)):
                self.nonlinear = True   # (re-checked in generate_form: users patch the material afterwards)
        if self.scalar_name == "electric_potential":
            assert self.settings['solver_settings']['transient_settings']['transient'] is False

    # ------------------------------------------------------------------ material
    def _finish_material(self, c, T):
        """Material value; a python function of T (ScalarTransportSolver.py:88-91, 106-109, 125-128) makes the
        problem nonlinear and is evaluated cell-wise at the mean of the current vertex values."""
        from inspect import isfunction
        if isfunction(c):
            self.nonlinear_material = True
            if isinstance(T, Function):
                Tbar = T.vertex_values()[self.mesh.cells().astype(np.int64)].mean(axis=1)
                return forms.VolumeCoefficient("cell", np.broadcast_to(np.asarray(c(Tbar), dtype=np.float64), Tbar.shape).copy())
            if T is None:
                raise SolverError('a temperature-dependent material property needs the current field')
            return c(T)
        return self.get_material_value(c)

    # Which material entry (or combination) a transport coefficient comes from, per scalar: an explicit generic key wins,
    # then the scalar's own physical properties (ScalarTransportSolver.py:73-129).  Entries are functions of the material dict.
    _COEFFICIENT_RULES = {
        'capacity': {
            'temperature': lambda m, self: m['density'] * m['specific_heat_capacity'],
            'electric_potential': lambda m, self: VACUUM_PERMITTIVITY,
            'species_concentration': lambda m, self: 1,
        },
        'diffusivity': {
            'temperature': lambda m, self: m['thermal_conductivity'] / self.capacity(),
            'electric_potential': lambda m, self: m['relative_electric_permittivity'],
        },
        'conductivity': {
            'temperature': lambda m, self: m['thermal_conductivity'],
            'electric_potential': lambda m, self: m['relative_electric_permittivity'] * VACUUM_PERMITTIVITY,
            'species_concentration': lambda m, self: m['diffusivity'],
        },
    }

    def _coefficient(self, kind, T):
        m = self.material
        if kind in m:
            c = m[kind]
        elif self.scalar_name in self._COEFFICIENT_RULES[kind]:
            c = self._COEFFICIENT_RULES[kind][self.scalar_name](m, self)
        elif kind == 'conductivity':
            # conductivity = diffusivity * capacity (ScalarTransportSolver.py:119-121).  Either factor may be a python function of
            # T (:88-91, :106-109): the product then is one - the reference calls both factors WITHOUT the field here and fails on a
            # function that uses its argument; evaluated at the current field instead
            from inspect import isfunction
            d_raw, c_raw = self._raw_coefficient('diffusivity'), self._raw_coefficient('capacity')
            if isfunction(d_raw) or isfunction(c_raw):
                d_fn = d_raw if isfunction(d_raw) else (lambda Tv, v=self.get_material_value(d_raw): v)
                c_fn = c_raw if isfunction(c_raw) else (lambda Tv, v=self.get_material_value(c_raw): v)
                self._derived_conductivity_fn = lambda Tv: d_fn(Tv) * c_fn(Tv)
                return self._finish_material(self._derived_conductivity_fn, T)
            c = self.diffusivity() * self.capacity()
        else:
            raise SolverError('material {} property is not found for {}'.format(kind, self.scalar_name))
        return self._finish_material(c, T)

    def _raw_coefficient(self, kind):
        """The material entry a coefficient comes from, before any evaluation (a number, a Constant, a field, a function of T)."""
        m = self.material
        if kind in m:
            return m[kind]
        if self.scalar_name in self._COEFFICIENT_RULES[kind]:
            return self._COEFFICIENT_RULES[kind][self.scalar_name](m, self)
        raise SolverError('material {} property is not found for {}'.format(kind, self.scalar_name))

    def capacity(self, T=None):
        return self._coefficient('capacity', T)

    def diffusivity(self, T=None):
        return self._coefficient('diffusivity', T)

    def conductivity(self, T=None):
        return self._coefficient('conductivity', T)

    # ------------------------------------------------------------------ coefficients
    def _volume_coefficient(self, value, what):
        """number | Constant | 3x3 | per-subdomain | Expression/Function -> VolumeCoefficient.
        A spatially varying P1 coefficient enters cell-wise by the mean of its vertex values,
        which is exactly what one-point quadrature of k*grad.grad gives."""
        if isinstance(value, forms.VolumeCoefficient):
            return value
        if isinstance(value, numbers.Number):
            return forms.VolumeCoefficient("const", float(value))
        if isinstance(value, Constant):
            v = value.values()
            if v.size == 1:
                return forms.VolumeCoefficient("const", float(v[0]))
            if v.size == 9:
                return forms.VolumeCoefficient("tensor", v.reshape(3, 3))
            if v.size == 4 and self.dimension == 2:
                return forms.VolumeCoefficient("tensor", self._embed_2x2(v.reshape(2, 2)))
            raise SolverError('{}: Constant of size {} is not a scalar or a dim x dim tensor'.format(what, v.size))
        if isinstance(value, np.ndarray) and value.shape == (3, 3):
            return forms.VolumeCoefficient("tensor", value)
        if isinstance(value, np.ndarray) and value.shape == (2, 2) and self.dimension == 2:
            return forms.VolumeCoefficient("tensor", self._embed_2x2(value))
        if isinstance(value, (Expression, Function)):
            if isinstance(value, Expression) and value.value_size() != 1:
                # K_anisotropic = Expression((('exp(x[0])','sin(x[1])'), ('sin(x[0])','tan(x[1])')), degree=0)
                # (examples/test_heat_transfer.py:90): a tensor per cell, evaluated at the cell mid-points as DOLFIN
                # interpolates a degree-0 Expression into DG0
                d = self.dimension
                if value.ufl_shape() != (d, d) or int(value.degree) != 0:
                    raise SolverError('{}: a tensor-valued Expression must be {}x{} and of degree 0'.format(what, d, d))
                if what != 'conductivity':
                    raise SolverError('{} cannot be a tensor'.format(what))
                co, cells = self.mesh.coordinates(), self.mesh.cells().astype(np.int64)
                vals = value.eval_points(co[cells].mean(axis=1)).reshape(-1, d, d)
                full = np.zeros((len(cells), 3, 3))
                full[:, :d, :d] = vals
                return forms.VolumeCoefficient("cell_tensor", full)
            nod = nodal_values(value, self.function_space)
            cells = self.mesh.cells().astype(np.int64)      # vertex nodes come first in P1 and P2 alike
            return forms.VolumeCoefficient("cell", nod[cells].mean(axis=1))
        raise SolverError('{}: value of type {} is not supported'.format(what, type(value)))

    @staticmethod
    def _embed_2x2(k):
        """2-D anisotropic tensor (examples/test_heat_transfer.py: K_anisotropic) in the leading block of the 3x3 the
        device kernels take."""
        out = np.zeros((3, 3))
        out[:2, :2] = np.asarray(k, dtype=np.float64)
        out[2, 2] = 1.0
        return out

    def _source_coefficient(self, value):
        if isinstance(value, numbers.Number) or (isinstance(value, Constant) and value.value_size() == 1):
            return forms.VolumeCoefficient("const", float(value))
        if isinstance(value, (Expression, Function)):
            return forms.VolumeCoefficient("nodal", nodal_values(value, self.function_space))
        raise SolverError('body source of type {} is not supported'.format(type(value)))

    def _facet_value(self, value, marker_id, what):
        """Constant over the boundary; a varying value (Expression / Function) by its values at the vertices of every facet,
        [n_facets, d]: int g q ds is then integrated exactly for the P1 interpolant of g (SolverBase._facet_nodal_loads).
        P2 spaces take the per-facet mean."""
        if is_constant_value(value):
            return float(value)
        if isinstance(value, (Expression, Function)):
            nod = nodal_values(value, self.function_space)
            tri = self._facets_of(marker_id).astype(np.int64)
            if self.function_space.degree() == 1:
                return nod[tri]
            return nod[tri].mean(axis=1)
        raise SolverError('{}: boundary value of type {} is not supported'.format(what, type(value)))

    # int phi^P2_n phi^P1_a dx on a tetrahedron of unit volume (n: 4 vertices, 6 UFC edges (2,3)(1,3)(1,2)(0,3)(0,2)(0,1))
    _P2_P1_MASS = np.array([[0.0 if n == a else -1.0 / 60.0 for a in range(4)] for n in range(4)] +
                           [[1.0 / 15.0 if a in e else 1.0 / 30.0 for a in range(4)]
                            for e in ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))])

    def get_convective_velocity_function(self, convective_velocity, per_test_function=True):
        """-> constant 3-vector, or the velocities of a field (ScalarTransportSolver.py:131-140, 305-311).
        per_test_function: [n_cells, d+1, 3] with V_a = (d+1)/|K| int_K u phi_a dx, the weights that integrate
        inner(u, grad(T)) * q * dx exactly for a P1 field (Expression / Function sampled at the vertices) or a P2 vector
        Function (its vertex and edge values); False (SUPG, whose tau needs one |u| per cell): the cell mean [n_cells, 3]."""
        v = convective_velocity
        if isinstance(v, Constant):
            vals = v.values()
            if vals.size != self.dimension:
                raise SolverError('convective_velocity must have {} components'.format(self.dimension))
            return self._pad3(np.asarray(vals, dtype=np.float64))
        if isinstance(v, (tuple, list, np.ndarray)) and len(v) == self.dimension and \
                all(isinstance(c, numbers.Number) for c in v):
            return self._pad3(np.asarray(v, dtype=np.float64))
        if isinstance(v, (tuple, list)) and len(v) == self.dimension and all(isinstance(c, str) for c in v):
            v = Expression(tuple(v), degree=self.settings['fe_degree'])
        if isinstance(v, Expression):
            nod = v.eval_points(self.mesh.coordinates())
        elif isinstance(v, Function):
            nod = v.vertex_values()
        else:
            raise SolverError('convective_velocity of type {} is not supported'.format(type(v)))
        nod = np.asarray(nod, dtype=np.float64)
        if nod.ndim != 2 or nod.shape[1] != self.dimension:
            raise SolverError('convective_velocity must be a {}-vector field'.format(self.dimension))
        cells = self.mesh.cells().astype(np.int64)
        if not per_test_function:
            return self._pad3(nod[cells].mean(axis=1))
        if isinstance(v, Function) and v.function_space().degree() == 2 and self.dimension == 3:
            Uc = v.node_values()[v.function_space().cell_nodes().astype(np.int64)]            # [nc,10,3]
            return self._pad3(4.0 * np.einsum("na,cni->cai", self._P2_P1_MASS, Uc))
        Uc = nod[cells]                                                                          # [nc,d+1,dim]
        return self._pad3((Uc.sum(axis=1, keepdims=True) + Uc) / (cells.shape[1] + 1.0))

    @staticmethod
    def _pad3(v):
        """Velocities travel to the device as 3-vectors (2-D problems: v_z = 0)."""
        v = np.asarray(v, dtype=np.float64)
        if v.shape[-1] == 3:
            return v
        return np.concatenate([v, np.zeros(v.shape[:-1] + (3 - v.shape[-1],))], axis=-1)

    # ------------------------------------------------------------------ boundary conditions
    def update_boundary_conditions(self, time_iter_, T, Tq, ds):
        """-> (Dirichlet bcs, list of FacetLoad / FacetRobin)  (ScalarTransportSolver.py:142-211)"""
        # (a capacity that is a python function of T is evaluated on the current field - generate_form leaves it in _material_field; the
        # boundary terms that scale with the capacity, Neumann / Robin gradients, then refuse it: _scalar_capacity)
        capacity = self.capacity(T if isinstance(T, Function) else getattr(self, '_material_field', None))
        bcs = []
        integrals_N = []
        self._point_sources = []
        if 'point_source' in self.settings and self.settings['point_source']:
            ps = self.settings['point_source']      # a PointSource, or a list of (point, magnitude) (:148-155)
            if isinstance(ps, PointSource):
                self._point_sources = [ps]
            else:
                self._point_sources = [p if isinstance(p, PointSource) else
                                       PointSource(self.function_space, Point(*np.ravel(p[0])) if not isinstance(p[0], Point) else p[0], p[1])
                                       for p in ps]
        if 'surface_source' in self.settings and self.settings['surface_source']:
            raise SolverError('surface_source is not supported (undefined in the reference as well)')

        for name, bc_settings in self.boundary_conditions.items():
            i = bc_settings['boundary_id']
            bc = self.get_boundary_variable(bc_settings)
            btype = bc['type']
            if btype == 'Dirichlet' or btype == 'fixedValue':
                if not isinstance(bc['value'], DirichletBC):
                    T_bc = self.translate_value(bc['value'])
                    bcs.append(DirichletBC(self.function_space, T_bc, self.boundary_facets, i))
                else:
                    bcs.append(bc['value'])
            elif btype == 'Neumann' or btype == 'fixedGradient':
                g = self._facet_value(self.translate_value(bc['value']), i, name)
                scale = 1.0 if self.using_diffusion_form else self._scalar_capacity(capacity)
                integrals_N.append(forms.FacetLoad(i, scale * g, 'Neumann(capacity*g)'))
            elif btype == 'symmetry':
                pass
            elif btype == 'mixed' or btype == 'Robin':
                T_bc = self.translate_value(bc['value'])
                g = self._facet_value(self.translate_value(bc['gradient']), i, name)
                scale = 1.0 if self.using_diffusion_form else self._scalar_capacity(capacity)
                integrals_N.append(forms.FacetLoad(i, scale * g, 'Robin(capacity*g)'))
                bcs.append(DirichletBC(self.function_space, T_bc, self.boundary_facets, i))
            elif btype.lower().find('flux') >= 0 or btype == 'electric_current':
                g = self._facet_value(self.translate_value(bc['value']), i, name)
                if self.using_diffusion_form:
                    g = g / self._scalar_capacity(capacity)
                integrals_N.append(forms.FacetLoad(i, g, 'flux'))
            elif btype == 'HTC':
                Ta = self.translate_value(bc['ambient'])
                htc = self.translate_value(bc['value'])
                if not is_constant_value(htc):
                    raise SolverError("boundary '{}': the HTC value must be a constant".format(name))
                h = float(htc) / (self._scalar_capacity(capacity) if self.using_diffusion_form else 1.0)
                # an ambient temperature that varies over the boundary enters the load h * Ta * q * ds by its vertex values
                integrals_N.append(forms.FacetRobin(i, h, float(Ta) if is_constant_value(Ta) else self._facet_value(Ta, i, name)))
            else:
                raise SolverError('boundary type`{}` is not supported'.format(btype))
        return bcs, integrals_N

    @staticmethod
    def _scalar_capacity(capacity):
        if isinstance(capacity, forms.VolumeCoefficient):
            if capacity.kind != "const":
                raise SolverError('a spatially varying capacity cannot scale a boundary term')
            return float(capacity.value)
        if is_constant_value(capacity):
            return float(capacity)
        raise SolverError('a spatially varying capacity cannot scale a boundary term')

    def get_body_source_items(self, time_iter_, T, Tq, dx):
        bs = self.get_body_source()
        if bs and isinstance(bs, dict):
            cells = self.subdomains.array()
            S = []
            for k, v in bs.items():
                if not is_constant_value(v['value']):
                    raise SolverError("body source '{}': per-subdomain values must be constants".format(k))
                S.append(forms.VolumeCoefficient("cell", np.where(cells == v['subdomain_id'], float(v['value']), 0.0)))
            return S
        if bs:
            return [self._source_coefficient(bs)]
        return None

    # ------------------------------------------------------------------ the form
    def generate_form(self, time_iter_, T, T_test, T_current, T_prev):
        conductivity = self.conductivity(T_current)
        capacity = self.capacity(T_current)

        if not hasattr(self, 'convective_velocity'):
            if 'convective_velocity' in self.settings and self.settings['convective_velocity']:
                self.convective_velocity = self.settings['convective_velocity']
            else:
                self.convective_velocity = None
        velocity = None
        if self.convective_velocity:
            ads = self.settings.get('advection_settings') or {'stabilization_method': None}
            method = ads.get('stabilization_method')
            supg_pe, ip_coef = 0.0, 0.0
            if method == 'IP':        # interior penalty on the jump of the normal gradient (:312-315), P1 and P2
                cap = self.capacity()
                if not isinstance(cap, numbers.Number):
                    raise SolverError("IP stabilisation needs a constant capacity")
                ip_coef = float(ads['alpha']) * float(cap)
            elif method == 'SPUG':      # the reference's spelling; its SPUG_method == 2 variant (:259-270), P1 and P2
                supg_pe = float(ads['Pe'])
                if not supg_pe > 0.0:
                    raise SolverError("advection_settings['Pe'] must be positive")
            elif method:
                raise SolverError("advection stabilization '{}' is not known ('SPUG', 'IP' or None)".format(method))
            # P2 spaces: a constant velocity is exact (degree-3 rule); a field enters by its cell mean
            velocity = self.get_convective_velocity_function(
                self.convective_velocity, per_test_function=(not supg_pe > 0.0) and self.function_space.degree() == 1)
        F = forms.ScalarForm(self.function_space)
        F.supg_pe = supg_pe if self.convective_velocity else 0.0
        F.ip_coefficient = ip_coef if self.convective_velocity else 0.0
        F.conductivity = self._volume_coefficient(conductivity, 'conductivity')
        from inspect import isfunction
        kraw = self.material.get('conductivity', self.material.get('thermal_conductivity'))
        if isfunction(kraw):
            F.conductivity_fn = kraw
        elif getattr(self, '_derived_conductivity_fn', None) is not None and 'conductivity' not in self.material and \
                self.scalar_name not in self._COEFFICIENT_RULES['conductivity']:
            F.conductivity_fn = self._derived_conductivity_fn      # diffusivity(T) * capacity(T)
        # a capacity that depends on T (a python function, ScalarTransportSolver.py:88-91): the transient term
        # (1/dt) (T - T_prev) c(T) q dx is re-evaluated at every Newton iterate, cell by cell at the mean of the vertex values, like k(T)
        craw = self.material.get('capacity')
        if isfunction(craw):
            F.capacity_fn = craw

        if self.transient_settings['transient']:
            F.transient = True
            F.dt = float(self.get_time_step(time_iter_))
            F.theta = 0.5   # Crank-Nicolson
            F.capacity = self._volume_coefficient(capacity, 'capacity')
            if F.capacity.kind == "tensor":
                raise SolverError('capacity cannot be a tensor')
            # keep the object, not a copy: solve_current_step assigns w_current to w_prev AFTER the form
            # is generated and BEFORE it is assembled (SolverBase.py:486-489), as UFL's late binding does
            F.T_prev = T_prev

        if velocity is not None:
            # F += inner(velocity, grad(T))*Tq*capacity*dx  (:311) - not theta-weighted in the reference either
            F.advection = (velocity, self._scalar_capacity(self._volume_coefficient(capacity, 'capacity')))
            F.symmetric = False

        self._material_field = T_current
        bcs, integrals_N = self.update_boundary_conditions(time_iter_, T, T_test, Measure("ds", subdomain_data=self.boundary_facets))
        for item in integrals_N:
            (F.robin if isinstance(item, forms.FacetRobin) else F.facet_loads).append(item)
        F.point_sources = list(self._point_sources)
        bs_items = self.get_body_source_items(time_iter_, T, T_test, None)
        if bs_items:
            F.sources.extend(bs_items)

        # surface radiation to the ambient (heat only): settings key, or an attribute the user set after construction
        rs = self.settings.get('radiation_settings') or getattr(self, 'radiation_settings', None)
        self.has_radiation = self.scalar_name == "temperature" and bool(rs)
        if self.has_radiation:
            self.radiation_settings = rs
            self.nonlinear = True
            F.radiation = self.radiation_coefficients()
        if self.nonlinear_material:
            self.nonlinear = True
        F.nonlinear = bool(self.nonlinear)
        return F, bcs

    def radiation_coefficients(self):
        """(emissivity * Stefan-Boltzmann, ambient temperature) of  m (Ta^4 - T^4)  (:361-376)."""
        sigma_sb = 5.670367e-8                                   # W / (m^2 K^4)
        rs = self.radiation_settings
        emissivity = self.material.get('emissivity', rs.get('emissivity', 1.0))          # the material wins over the settings
        T_amb = rs.get('ambient_temperature', self.reference_values.get('temperature'))
        if T_amb is None:
            raise SolverError("radiation needs 'ambient_temperature' or a reference temperature")
        return (float(emissivity) * sigma_sb, float(T_amb))

    def radiation_flux(self, T):
        """m (Ta^4 - T^4) evaluated on numbers / arrays / a Function's nodal values (:361-376)."""
        m, T_amb = self.radiation_coefficients()
        vals = T.vector()._values() if isinstance(T, Function) else np.asarray(T, dtype=np.float64)
        return m * (T_amb ** 4 - vals ** 4)

    def refresh_nonlinear_form(self, F, T):
        """Re-evaluate the temperature-dependent coefficients of F at the Newton iterate T."""
        self._refresh_capacity(F, T)
        if F.conductivity_fn is not None and self.function_space.degree() == 2:
            # P2 iterate: k(T_h) at the 14 points of the degree-5 rule, where the integrand k grad T . grad q is evaluated
            # (exact for a conductivity that is linear in T, e.g. examples/test_heat_transfer.py:53)
            Tc = T.vector()._values()[self.function_space.cell_nodes().astype(np.int64)]          # [nc,10] or [nc,6]
            if self.dimension == 3:
                Tq = Tc @ _p2_shape_at(_TET14_POINTS).T                                            # [nc,14]
                kq = np.asarray(F.conductivity_fn(Tq), dtype=np.float64) * np.ones_like(Tq)
            else:       # triangles: the 6 points of the degree-4 rule, padded to the 14 columns of the device layout
                p = _TRI6_POINTS
                shape = np.stack([p[:, i] * (2.0 * p[:, i] - 1.0) for i in range(3)] +
                                 [4.0 * p[:, i] * p[:, j] for i, j in ((1, 2), (0, 2), (0, 1))], axis=1)       # UFC edges of a triangle
                Tq = Tc @ shape.T
                kq = np.zeros((len(Tc), 14))
                kq[:, :6] = np.asarray(F.conductivity_fn(Tq), dtype=np.float64) * np.ones_like(Tq)
            F.conductivity = forms.VolumeCoefficient("cell_qp", kq)
            return
        if F.conductivity_fn is not None:
            Tbar = T.vertex_values()[self.mesh.cells().astype(np.int64)].mean(axis=1)
            F.conductivity = forms.VolumeCoefficient(
                "cell", np.broadcast_to(np.asarray(F.conductivity_fn(Tbar), dtype=np.float64), Tbar.shape).copy())

    def _refresh_capacity(self, F, T):
        if getattr(F, 'capacity_fn', None) is not None and F.transient:
            Tbar = T.vertex_values()[self.mesh.cells().astype(np.int64)].mean(axis=1)
            F.capacity = forms.VolumeCoefficient(
                "cell", np.broadcast_to(np.asarray(F.capacity_fn(Tbar), dtype=np.float64), Tbar.shape).copy())

    def solve_form(self, F, T_current, bcs):
        if self.nonlinear:
            return self.solve_nonlinear_problem(F, T_current, bcs, None)
        return self.solve_linear_problem(F, T_current, bcs)

    # ------------------------------------------------------------------ public API
    def export(self):
        result_filename = self.settings['case_folder'] + os.path.sep + self.get_variable_name() + "_time0" + ".vtk"
        return result_filename

    def boundary_flux(self, marker_id, conductivity=None):
        """assemble(k*dot(grad(T), n)*ds(id)) of the current result — the check the reference's
        example scripts print (examples/test_heat_transfer.py:189-190)."""
        k = self.conductivity() if conductivity is None else conductivity
        if not isinstance(k, numbers.Number):
            raise SolverError('boundary_flux needs a constant conductivity')
        mesh = self.mesh
        co, cells = mesh.coordinates(), mesh.cells().astype(np.int64)
        d = co.shape[1]                                   # 3: tetrahedra, 2: triangles
        sel = self.boundary_facets.where(marker_id)
        cf = mesh.cell_facets()
        c_idx, lf = np.nonzero(np.isin(cf, sel))          # boundary facets belong to exactly one cell
        T = self.result.vector()._values()
        X = co[cells[c_idx]]                              # [nf, d+1, d]
        J = np.stack([X[:, k + 1] - X[:, 0] for k in range(d)], axis=2)
        ginv = np.linalg.inv(J)                           # rows: grad lambda_1..d
        g = np.concatenate([-ginv.sum(axis=1, keepdims=True), ginv], axis=1)       # [nf, d+1, d] grad lambda_a
        vol = np.abs(np.linalg.det(J)) / math.factorial(d)
        gradT = np.einsum("fa,fad->fd", T[cells[c_idx]], g)
        # outward normal times facet measure = - d * |cell| * grad lambda_(opposite vertex)
        nmeas = -d * vol[:, None] * g[np.arange(len(c_idx)), lf]
        return float(k) * float(np.einsum("fd,fd->", gradT, nmeas))
