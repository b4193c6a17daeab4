"""LinearElasticitySolver — small-strain isotropic elasticity on vector P1 / P2, GPU back end.

Counterpart of FenicsSolver/LinearElasticitySolver.py: same class/constructor (forces
vector_name='displacement', :55-60), sigma(u) = 2 mu sym(grad u) + lambda div(u) I
(:62-69), boundary types displacement/Dirichlet (per-component with None = free,
:122-131), force / pressure / stress (:165-196), body force (:227-228), thermal stress
(:78-85, 231-238); 3D problems go through solve_amg (CG), as in the reference (:247-253).

Reference quirk kept by default (Appendix B-Q3): body forces and tractions are ADDED to
F (:227-228, 242-243), so they act with reversed sign; set
``solver.reference_load_sign = False`` for the physical convention.  The thermal term has
the conventional sign in both.  Modal analysis (:270-312, SLEPc) is out of scope.

Vector P2 (the reference's own example, examples/test_linear_elasticity.py:105-106) runs on one GPU; its solve_amg uses
Jacobi-CG (the aggregation hierarchy is built on P1 node patterns).  ``von_Mises`` is the consistent L2 projection onto P1,
assembled and solved on the device (fs_assemble_von_mises + P1 mass matrix + CG).  Elastodynamics
(``solving_dynamics = True``, :216-220) subtracts rho * acceleration with the reference's own finite-difference
acceleration (SolverBase.py:477-482, including its division by 1/dt).
"""
from __future__ import annotations

import numbers

import numpy as np

from .fem import Measure, Constant, Expression, Function, DirichletBC, nodal_values, is_constant_value
from .SolverBase import SolverBase, SolverError
from . import forms


class LinearElasticitySolver(SolverBase):
    def __init__(self, case_settings):
        case_settings['vector_name'] = 'displacement'
        SolverBase.__init__(self, case_settings)
        self.solving_modal = False
        self.solving_dynamics = False
        self.reference_load_sign = True

    def lame_parameters(self):
        elasticity = self.material['elastic_modulus']
        nu = self.material['poisson_ratio']
        if not (isinstance(elasticity, numbers.Number) and isinstance(nu, numbers.Number)):
            raise SolverError('elastic_modulus and poisson_ratio must be numbers (homogeneous material)')
        mu = elasticity / (2.0 * (1.0 + nu))
        lmbda = elasticity * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
        return mu, lmbda

    def _cell_gradients(self, u):
        """grad u per cell [nc,3,3] (component i, derivative j); P2 displacements: at the cell centroid."""
        co = self.mesh.coordinates()
        ce = self.mesh.cells().astype(np.int64)
        X = co[ce]
        d = ce.shape[1] - 1                 # 3: tetrahedra, 2: triangles ([nc,2,2] gradients)
        J = np.stack([X[:, k] - X[:, 0] for k in range(1, d + 1)], axis=2)
        Ji = np.linalg.inv(J)
        g = np.zeros((len(ce), d + 1, d))
        g[:, 1:, :] = Ji
        g[:, 0, :] = -Ji.sum(axis=1)
        V = u.function_space()
        if V.degree() == 1:
            return np.einsum("cai,caj->cij", u.vertex_values()[ce], g)
        cd = V.cell_nodes()
        if d == 2:
            # triangles at the centroid (lambda = 1/3): vertex i (4/3 - 1) g_i, edge (i,j): 4/3 (g_i + g_j)
            ei, ej = [1, 0, 0], [2, 2, 1]
            gv = g / 3.0
            ge = (4.0 / 3.0) * (g[:, ei, :] + g[:, ej, :])
            return np.einsum("cai,caj->cij", u.node_values()[cd], np.concatenate([gv, ge], axis=1))
        # tetrahedra at the centroid (lambda = 1/4): vertex functions have zero gradient there, edge (i,j): g_i + g_j
        ei, ej = [2, 1, 1, 0, 0, 0], [3, 3, 2, 3, 2, 1]
        U = u.node_values()[cd[:, 4:]]                  # [nc,6,3]
        ge = g[:, ei, :] + g[:, ej, :]
        return np.einsum("cai,caj->cij", U, ge)

    def sigma(self, u):
        """Cell-wise (DG0) stress tensor [nc,3,3] of a displacement Function (post-processing helper on the host; the
        reference returns the UFL expression, LinearElasticitySolver.py:62-69)."""
        mu, lmbda = self.lame_parameters()
        G = self._cell_gradients(u)
        eps = 0.5 * (G + np.transpose(G, (0, 2, 1)))
        tr = np.trace(G, axis1=1, axis2=2)
        return 2.0 * mu * eps + lmbda * tr[:, None, None] * np.eye(G.shape[1])[None]

    def von_Mises(self, u):
        """project(sqrt(3/2 s:s), FunctionSpace(mesh, 'P', 1)) (:71-76): the consistent L2 projection, on the device -
        right-hand side int vm phi_a dx (fs_assemble_von_mises), P1 mass matrix, Jacobi-CG to 1e-12."""
        from .fem import FunctionSpace
        from . import backend
        V = u.function_space()
        P = FunctionSpace(self.mesh, 'P', 1)
        dV, dP = V.device(), P.device()
        mu, lmbda = self.lame_parameters()
        loc, ploc = V.localizer(), P.localizer()
        uh = u.vector()._values()
        if loc is not None and not getattr(loc, 'is_identity', False):
            uh = loc.nodes(uh)                       # host field -> this rank's owned + ghost entries in device order
        ud = backend.DeviceVector(dV.n_local, uh)
        b = backend.DeviceVector(dP.n_owned)
        backend.assemble_von_mises(dV, ud, mu, lmbda, dP, b)
        M = backend.DeviceMatrix(dP)
        M.assemble(mass=1.0)
        x = backend.DeviceVector(dP.n_local)
        st = backend.krylov_solve(M, b, x, rtol=1e-12, max_iter=2000, precond="jacobi", norm="preconditioned")
        if st['converged'] != 1:
            raise SolverError('von_Mises: the mass-matrix solve did not converge')
        f = Function(P)
        if ploc is None:
            f.vector().set_local(x.get()[:dP.n_owned])
        elif getattr(ploc, 'is_local_view', False):
            from . import parallel
            if parallel.world()[1] > 1:
                backend.halo_exchange(dP, x)
            f.vector()._adopt_device(x)
        else:
            from . import parallel
            f.vector().set_local(parallel.gather_owned(x.get()[:dP.n_owned], ploc.owned_gids(), ploc.n_global, 1))
        return f

    def thermal_stress_coefficient(self):
        elasticity = self.material['elastic_modulus']
        nu = self.material['poisson_ratio']
        tec = self.material['thermal_expansion_coefficient']
        return elasticity / (1.0 - 2.0 * nu) * tec

    def thermal_stress(self, T):
        """The isotropic thermal stress  E/(1-2nu) * alpha * (T - T_ref)  (the multiplier of Identity(dim),
        LinearElasticitySolver.py:78-85) for a number, an array of nodal temperatures or a Function."""
        vals = T.vector()._values() if isinstance(T, Function) else np.asarray(T, dtype=np.float64)
        return self.thermal_stress_coefficient() * (vals - float(self.reference_values['temperature']))

    def strain_energy(self, u):
        raise SolverError("strain_energy: the reference's expression (LinearElasticitySolver.py:87-93) uses an undefined "
                          "symbol and '^' on UFL objects; nothing to reproduce")

    def solve_modal_form(self, F, bcs):
        raise SolverError("modal analysis (SLEPc eigen-solver, LinearElasticitySolver.py:283-310) is not built")

    def get_flux(self, u, mag_vector):
        return mag_vector

    # ------------------------------------------------------------------ boundary conditions
    def _facet_normals(self, marker_id):
        """Outward unit normals and areas (lengths in 2-D) of the facets carrying marker_id."""
        mesh = self.mesh
        sel = self.boundary_facets.where(marker_id)
        tri = mesh.facets()[sel].astype(np.int64)
        co = mesh.coordinates()
        p = co[tri]
        if tri.shape[1] == 2:          # boundary edges of a triangular mesh: the tangent turned by -90 degrees
            t = p[:, 1] - p[:, 0]
            n = np.stack([t[:, 1], -t[:, 0]], axis=1)
        else:
            n = 0.5 * np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        area = np.linalg.norm(n, axis=1)
        # orient away from the interior: use the centroid of the (single) adjacent cell
        cf = mesh.cell_facets()
        cells = mesh.cells().astype(np.int64)
        owner = np.full(mesh.num_facets(), -1, dtype=np.int64)
        owner[cf.ravel()] = np.repeat(np.arange(len(cells)), cells.shape[1])
        cc = co[cells[owner[sel]]].mean(axis=1)
        flip = np.einsum("fi,fi->f", n, p.mean(axis=1) - cc) < 0
        n[flip] *= -1.0
        return tri, n / area[:, None], area

    # int phi_node lambda_b ds / |F| on a boundary facet: P2 nodes (vertices, then edge mid-points) against the P1 weights
    _W_TRI_P2 = np.array([[1 / 30 if b == v else -1 / 60 for b in range(3)] for v in range(3)] +
                         [[2 / 15 if b in e else 1 / 15 for b in range(3)] for e in ((0, 1), (0, 2), (1, 2))])
    _W_SEG_P2 = np.array([[1 / 6, 0.0], [0.0, 1 / 6], [1 / 3, 1 / 3]])

    def _varying_pressure_load(self, marker_id, pval, direction, name):
        from .fem import nodal_values, FunctionSpace
        V = self.function_space
        tri, nrm, area = self._facet_normals(marker_id)
        if direction:
            nrm = np.tile(self._vector_of(direction, name), (len(tri), 1))
        pv = nodal_values(pval, FunctionSpace(self.mesh, 'P', 1))[tri]              # [nf, d]: p at the facet's vertices
        d = tri.shape[1]
        nv = self.mesh.num_vertices()
        if V.degree() == 1:
            nodes = tri
            w = (pv.sum(axis=1, keepdims=True) + pv) / (d * (d + 1.0))              # [nf, d]
        else:
            ed = V.edge_nodes().astype(np.int64)
            ekey = ed[:, 0] * nv + ed[:, 1]
            sorter = np.argsort(ekey)
            pairs = ((0, 1), (0, 2), (1, 2)) if d == 3 else ((0, 1),)
            enodes = []
            for a, b in pairs:
                lo, hi = np.minimum(tri[:, a], tri[:, b]), np.maximum(tri[:, a], tri[:, b])
                enodes.append(nv + sorter[np.searchsorted(ekey[sorter], lo * nv + hi)])
            nodes = np.concatenate([tri, np.stack(enodes, axis=1)], axis=1)         # [nf, 6] or [nf, 3]
            w = pv @ (self._W_TRI_P2 if d == 3 else self._W_SEG_P2).T               # [nf, n_nodes]
        loads = self.load_sign_for_tractions() * area[:, None, None] * w[:, :, None] * nrm[:, None, :]     # [nf, nn, dim]
        dofs = nodes[:, :, None] * self.dimension + np.arange(self.dimension)[None, None, :]
        return forms.NodalLoad(dofs, loads, 'pressure(field) ds(%d)' % marker_id)

    def load_sign_for_tractions(self):
        return -1.0 if self.reference_load_sign else 1.0

    def _total_area(self, tri, area):
        """assemble(Constant(1)*ds(id)): on a distributed mesh every facet is counted once - by the rank owning its vertex
        of smallest global id - and the ranks' sums are added."""
        mesh = self.mesh
        if not (hasattr(mesh, 'is_distributed') and mesh.is_distributed()):
            return float(area.sum())
        from . import backend
        gid = mesh.global_vertex_ids()[tri]
        first = np.take_along_axis(tri, np.argmin(gid, axis=1)[:, None], axis=1)[:, 0]
        mine = first < mesh.num_owned_vertices()
        return float(backend.comm_allreduce_sum([float(area[mine].sum())])[0])

    def _vector_of(self, value, what):
        if isinstance(value, Constant):
            v = value.values()
        elif isinstance(value, (tuple, list, np.ndarray)):
            v = np.asarray(value, dtype=np.float64)
        else:
            raise SolverError('{}: a constant vector is required, got {}'.format(what, type(value)))
        if v.size != self.dimension:
            raise SolverError('{}: vector of size {} in a {}D problem'.format(what, v.size, self.dimension))
        return v

    def update_boundary_conditions(self, time_iter_, u, v, ds):
        V = self.function_space
        bcs = []
        integrals_N = []
        if 'point_source' in self.settings and self.settings['point_source']:
            # The reference reads the WRONG key here - settings['surface_source'] (LinearElasticitySolver.py:105-108) - and appends that
            # dict to the Dirichlet list, which assemble_system (:643-650 of SolverBase.py) then rejects: no elasticity case with a
            # point_source runs upstream.  Kept as an error ("refusals"); a point load is a 'force' boundary
            # condition on a small marked facet patch, or a PointSource of the scalar solver for scalar problems.
            raise SolverError("point_source is not supported by LinearElasticitySolver (the reference's own branch reads "
                              "settings['surface_source'] and fails in assemble_system); use a 'force' boundary condition")
        if 'surface_source' in self.settings and self.settings['surface_source']:
            raise SolverError('surface_source is not supported')

        for name, bc_settings in self.boundary_conditions.items():
            i = bc_settings['boundary_id']
            bc = self.get_boundary_variable(bc_settings)
            btype = bc['type']
            if btype == 'Dirichlet' or btype == 'displacement':
                bv = bc['value']
                if isinstance(bv, (tuple, list)) and len(bv) == self.dimension and \
                        any(c is None or isinstance(c, (Constant, Expression, str)) for c in bv):
                    for axis_i, disp in enumerate(bv):
                        if disp is not None:   # None = free; zero is a constraint
                            val = self.translate_value(disp)
                            bcs.append(DirichletBC(V.sub(axis_i), val, self.boundary_facets, i))
                else:
                    bcs.append(DirichletBC(V, self.translate_value(bv), self.boundary_facets, i))
            elif btype == 'force':
                val = bc['value']
                if isinstance(val, (tuple, list)) and len(val) == self.dimension:
                    # the reference applies a tuple-valued 'force' as a traction density (:166-167)
                    g = self._vector_of(val, name)
                    integrals_N.append(forms.FacetLoad(i, g, 'force(vector density)'))
                elif isinstance(val, Constant) and val.value_size() == self.dimension:
                    # Constant((Fx,Fy,Fz)) as in examples/test_linear_elasticity.py:86: the reference forms
                    # n * (F / area), a vector-vector product UFL rejects; the evident intent - the total
                    # force vector spread over the face - is what is applied here: traction = F / area
                    tri, nrm, area = self._facet_normals(i)
                    bc_area = self._total_area(tri, area)
                    self.logger.info('boundary area (m2) for force boundary is %g', bc_area)
                    integrals_N.append(forms.FacetLoad(i, val.values() / bc_area, 'force(vector / area)'))
                else:
                    bc_force = self.translate_value(val)
                    if not is_constant_value(bc_force):
                        raise SolverError("boundary '{}': force magnitude must be a constant".format(name))
                    tri, nrm, area = self._facet_normals(i)
                    bc_area = self._total_area(tri, area)     # assemble(Constant(1)*ds(id)) (:171)
                    self.logger.info('boundary area (m2) for force boundary is %g', bc_area)
                    gmag = float(bc_force) / bc_area
                    if 'direction' in bc and bc['direction']:
                        integrals_N.append(forms.FacetLoad(i, self._vector_of(bc['direction'], name) * gmag,
                                                           'force(direction)'))
                    else:
                        integrals_N.append(forms.FacetLoad(i, nrm * gmag, 'force(normal)'))
            elif btype == 'pressure':
                pval = self.translate_value(bc['value'])
                if not is_constant_value(pval):
                    # a pressure that varies over the boundary (hydrostatic load on a wall): n * p with p through its P1
                    # interpolant on every facet, integrated exactly against the P1 / P2 test functions
                    integrals_N.append(self._varying_pressure_load(i, pval, bc.get('direction'), name))
                    continue
                if 'direction' in bc and bc['direction']:
                    integrals_N.append(forms.FacetLoad(i, self._vector_of(bc['direction'], name) * float(pval),
                                                       'pressure(direction)'))
                else:
                    tri, nrm, area = self._facet_normals(i)
                    integrals_N.append(forms.FacetLoad(i, nrm * float(pval), 'pressure(normal)'))
            elif btype == 'stress':
                g = self.translate_value(bc['value'])
                if isinstance(g, Constant) and g.value_size() == self.dimension:
                    integrals_N.append(forms.FacetLoad(i, g.values(), 'stress(vector)'))
                elif isinstance(g, Constant) and g.value_size() == self.dimension ** 2:
                    tri, nrm, area = self._facet_normals(i)
                    integrals_N.append(forms.FacetLoad(i, nrm @ g.values().reshape(self.dimension, self.dimension).T, 'stress(tensor.n)'))
                else:
                    raise SolverError("boundary '{}': stress must be a constant vector or tensor".format(name))
            elif btype == 'Neumann':
                raise SolverError('Neumann boundary type`{}` is not supported'.format(btype))
            elif btype == 'symmetry':
                raise SolverError('symmetry boundary type`{}` is not supported'.format(btype))
            else:
                raise SolverError('boundary type`{}` is not supported'.format(btype))
        return bcs, integrals_N

    @staticmethod
    def _try_values(val):
        if isinstance(val, Constant):
            return val.values()
        try:
            return np.asarray(val, dtype=np.float64)
        except (TypeError, ValueError):
            return np.zeros(0)

    # ------------------------------------------------------------------ the form
    def generate_form(self, time_iter_, u, v, u_current, u_prev):
        F = forms.ElasticityForm(self.function_space)
        if self.transient_settings['transient'] and self.solving_dynamics and time_iter_ >= 1:
            # F -= density * inner(accel, v) * dx (:216-220) with the explicit acceleration of SolverBase.get_acceleration
            F.inertia = (float(self.material['density']), self.get_acceleration(time_iter_))
        F.mu, F.lmbda = self.lame_parameters()
        F.load_sign = -1.0 if self.reference_load_sign else 1.0

        bcs, integrals_F = self.update_boundary_conditions(time_iter_, u, v, Measure("ds", subdomain_data=self.boundary_facets))
        F.tractions.extend(integrals_F)

        if self.body_source:
            bs = self.body_source
            if isinstance(bs, Expression):
                vals = bs.eval_points(self.mesh.coordinates()[:1])   # constant body force expected
                allv = bs.eval_points(self.mesh.coordinates())
                if np.abs(allv - vals).max() > 1e-12 * max(1.0, np.abs(allv).max()):
                    # a field (e.g. a centrifugal load): its interpolant in the displacement space, integrated with the
                    # consistent mass matrix - what FFC's quadrature gives for an Expression of the element's degree
                    F.body_force_nodal = bs.eval_points(self.function_space.node_coordinates())[:, :self.dimension]
                    F.body_force = None
                else:
                    F.body_force = tuple(float(x) for x in vals[0])
            else:
                F.body_force = tuple(float(x) for x in self._vector_of(bs, 'body_source'))

        if not hasattr(self, 'temperature_distribution'):
            if 'temperature_distribution' in self.settings and self.settings['temperature_distribution']:
                self.temperature_distribution = self.translate_value(self.settings['temperature_distribution'])
        if hasattr(self, 'temperature_distribution') and self.temperature_distribution:
            T = self.temperature_distribution
            T_ref = float(self.reference_values['temperature'])
            if is_constant_value(T):
                Tval = float(T)
            else:
                from .fem import FunctionSpace
                nod = nodal_values(T, FunctionSpace(self.mesh, 'P', 1))
                Tval = float(nod[0]) if np.ptp(nod) == 0.0 else nod
            F.thermal = (self.thermal_stress_coefficient(), Tval, T_ref)
        return F, bcs

    def solve_form(self, F, u_, bcs):
        if self.dimension == 3:
            u_ = self.solve_amg(F, u_, bcs)
        else:
            u_ = self.solve_linear_problem(F, u_, bcs)
        return u_

    def displacement(self):
        if self.is_mixed_function_space:
            raise SolverError('subclass with mixed_function_space must override this function')
        return self.w_current

    def velocity(self):
        dt = self.get_time_step(self.current_step)
        out = Function(self.function_space)
        out.vector().set_local((self.w_current.vector()._values() - self.w_prev.vector()._values()) / dt)
        return out

    def solve_modal(self):
        raise SolverError('modal analysis (SLEPc) is out of scope of the GPU back end')
