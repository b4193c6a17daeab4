"""Operator specifications: what ``generate_form`` returns in place of a UFL form.

The reference builds a UFL expression tree and lets FFC turn it into
tabulate_tensor code (ScalarTransportSolver.py:228-359,
LinearElasticitySolver.py:206-245).  Here the solver classes *recognise* which
integrals the settings ask for and record them, with their coefficients, in one
of the small classes below; ``SolverBase.solve_linear_problem`` / ``solve_amg``
hand them to the HIP kernels.  The classes are plain data — printable and
comparable in tests — and contain no arithmetic.
"""
from __future__ import annotations

import numpy as np


class VolumeCoefficient:
    """kind: 'const' (value float), 'cell' (value array[n_cells]), 'tensor' (value 3x3),
    'nodal' (value array[n_vertices], linear forms only)."""

    def __init__(self, kind, value):
        self.kind = kind
        self.value = value

    def spec(self, scale=1.0):
        """Argument for fenicssolver_amd.backend (None | number | (kind, array))."""
        if self.kind == "const":
            return float(self.value) * scale
        return (self.kind, np.asarray(self.value, dtype=np.float64) * scale)

    def describe(self):
        if self.kind == "const":
            return ("const", float(self.value))
        a = np.asarray(self.value, dtype=np.float64)
        return (self.kind, a.shape, float(a.min()), float(a.max()))

    def __repr__(self):
        return "VolumeCoefficient%r" % (self.describe(),)


class FacetLoad:
    """int g * q ds(marker_id) added to the load vector; g scalar or [ncomp]."""

    def __init__(self, marker_id, g, origin=""):
        self.marker_id = int(marker_id)
        self.g = g
        self.origin = origin

    def __repr__(self):
        return "FacetLoad(ds(%d), g=%r, %s)" % (self.marker_id, self.g, self.origin)


class NodalLoad:
    """b[dofs] += values: load contributions worked out per node on the host (boundary terms with a varying magnitude,
    integrated exactly for its P1 interpolant)."""

    def __init__(self, dofs, values, origin=""):
        self.dofs = np.asarray(dofs, dtype=np.int64).ravel()
        self.values = np.asarray(values, dtype=np.float64).ravel()
        self.origin = origin

    def __repr__(self):
        return "NodalLoad(%d entries, %s)" % (len(self.dofs), self.origin)


class FacetRobin:
    """htc*(Ta - T)*q*ds(i): +h int T q ds on the matrix, +h*Ta int q ds on the load."""

    def __init__(self, marker_id, h, ambient):
        self.marker_id = int(marker_id)
        self.h = float(h)
        self.ambient = float(ambient) if np.ndim(ambient) == 0 else np.asarray(ambient, dtype=np.float64)   # [nf, d]: vertex values

    def __repr__(self):
        return "FacetRobin(ds(%d), h=%g, Ta=%s)" % (self.marker_id, self.h, _plain(self.ambient))


class ScalarForm:
    """F = (1/dt) c (T-T_prev) q dx + theta a_k(T,q) + (1-theta) a_k(T_prev,q) - loads
    with a_k = int k grad T . grad q dx  (ScalarTransportSolver.py:284-303)."""

    def __init__(self, space):
        self.space = space
        self.conductivity = None      # VolumeCoefficient
        self.capacity = None          # VolumeCoefficient (transient only)
        self.transient = False
        self.dt = None
        self.theta = 1.0
        self.T_prev = None            # Function
        self.sources = []             # [VolumeCoefficient]  int S q dx
        self.facet_loads = []         # [FacetLoad]
        self.robin = []               # [FacetRobin]
        self.point_sources = []       # [fem.PointSource]: b[dofs] += weights, before the Dirichlet rows
        self.supg_pe = 0.0            # > 0: every test function is q + tau (v . grad q) ("SPUG", :259-270)
        self.ip_coefficient = 0.0     # alpha * capacity of + alpha avg(h)^2 jump(grad T,n) jump(grad q,n) capacity dS ("IP", :312-315)
        self.advection = None         # (velocity: 3-vector or array[n_cells,3], scale = capacity) -> non-symmetric
        self.symmetric = True
        # nonlinear terms (Newton): radiation  - m (Ta^4 - T^4) q ds over the whole boundary, m = emissivity*sigma
        # (ScalarTransportSolver.py:338-350, 361-376) and material callables re-evaluated every iteration
        self.radiation = None         # (m, T_ambient)
        self.conductivity_fn = None   # callable(T array) -> k
        self.capacity_fn = None       # callable(T array) -> volumetric capacity (transient term; re-evaluated per Newton iterate)
        self.nonlinear = False

    def describe(self):
        """Canonical, order-stable description used by the golden-term tests."""
        return {
            "type": "scalar",
            "conductivity": self.conductivity.describe() if self.conductivity else None,
            "capacity": self.capacity.describe() if self.capacity else None,
            "transient": self.transient, "dt": self.dt, "theta": self.theta,
            "supg_pe": self.supg_pe,
            "ip_coefficient": self.ip_coefficient,
            "sources": [s.describe() for s in self.sources],
            "facet_loads": [(f.marker_id, _plain(f.g), f.origin) for f in self.facet_loads],
            "robin": [(r.marker_id, r.h, _plain(r.ambient)) for r in self.robin],
            "advection": None if self.advection is None else (_plain(self.advection[0]), float(self.advection[1])),
            "radiation": self.radiation, "nonlinear": self.nonlinear,
        }


class ElasticityForm:
    """F = int sigma(u):grad v dx  +/- loads  (LinearElasticitySolver.py:206-245)."""

    def __init__(self, space):
        self.space = space
        self.mu = None
        self.lmbda = None
        self.body_force = None        # (fx, fy, fz) or None
        self.body_force_nodal = None  # [n_nodes, dim]: a body force FIELD by its nodal values (consistent-mass load)
        self.tractions = []           # [FacetLoad] with vector g
        self.thermal = None           # (coefficient E*alpha/(1-2nu), T nodal array or float, T_ref)
        self.load_sign = -1.0         # reference adds the load terms to F => rhs = -loads (Appendix B-Q3)
        self.inertia = None           # (density, acceleration dof array): F -= rho a . v dx => rhs += rho M a (:216-220)

    def describe(self):
        return {
            "type": "elasticity", "mu": self.mu, "lambda": self.lmbda,
            "body_force": None if self.body_force is None else tuple(float(x) for x in self.body_force),
            "tractions": [(t.marker_id, _plain(t.g), t.origin) if isinstance(t, FacetLoad) else ("nodal", len(t.dofs), t.origin)
                          for t in self.tractions],
            "thermal": None if self.thermal is None else (self.thermal[0], _plain(self.thermal[1]), self.thermal[2]),
            "load_sign": self.load_sign,
        }


def _plain(v):
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 0:
        return float(a)
    if a.size <= 4:
        return tuple(float(x) for x in a.ravel())
    return ("array", a.shape, float(a.min()), float(a.max()))


class NavierStokesForm:
    """F = 2 nu eps(u):eps(v) - (p/rho) div v + (q/rho) div u - f.v + (grad(u) u0).v [+ (1/dt)(u - u_prev).v]
    with u0 = the velocity of ``w_current`` (CoupledNavierStokesSolver.py:288-381).  ``newton``: the reference's
    action(F, w_current) + derivative (Newton); otherwise the Picard linearisation with u0 frozen."""

    def __init__(self, space):
        self.space = space
        self.nu = None
        self.rho = None
        self.inv_dt = 0.0
        self.body_force = None        # 3 numbers or None
        self.w_current = None         # Function (late bound, like UFL coefficients)
        self.w_prev = None
        self.newton = True
        self.symmetric = False
        self.nonlinear = True
        # pressure boundaries: [(marker_id, value | None)]: + inner(value*n, v)*ds(id) (value given) and
        # - nu*inner((grad(u) + grad(u).T)*n, v)*ds(id)   (CoupledNavierStokesSolver.py:449-453, 459-460)
        self.pressure_boundaries = []
        # ALE frame (reference_frame_settings {'type': 'ALE', 'mesh_velocity': ..}, :321-329): advecting velocity u0 - w
        self.mesh_velocity = None     # 3 numbers or None
        # G2 stabilisation (advection_settings {'stabilization_method': 'G2', 'Re':, 'kappa1':, 'kappa2':}, :334-363):
        # (mode, kappa1) with mode 1 for Re <= 1 (delta1 = kappa1 h^2), 2 otherwise; None = off
        self.g2 = None
        # non-Newtonian law (material 'Newtonian': False, :194-213): (p_ref, exponent) -> nu (p / p_ref)^exponent; None = off
        self.viscosity_law = None

    @staticmethod
    def _value_name(v):
        try:
            return "Constant(%g)" % float(v)
        except (TypeError, ValueError):
            return type(v).__name__

    def describe(self):
        d = self.space.velocity_dim() if hasattr(self.space, "velocity_dim") else 3      # vectors are stored padded to 3 slots
        return {"type": "navier_stokes", "nu": self.nu, "rho": self.rho, "inv_dt": self.inv_dt,
                "pressure_boundaries": [(int(m), None if v is None else self._value_name(v)) for m, v in self.pressure_boundaries],
                "body_force": None if self.body_force is None else [float(x) for x in self.body_force][:d],
                "mesh_velocity": None if self.mesh_velocity is None else [float(x) for x in self.mesh_velocity][:d],
                "g2": None if self.g2 is None else [int(self.g2[0]), float(self.g2[1])],
                "viscosity_law": None if self.viscosity_law is None else [float(self.viscosity_law[0]), float(self.viscosity_law[1])],
                "newton": bool(self.newton)}
