"""Taylor-Hood mixed space P2 (vector) x P1 for CoupledNavierStokesSolver.

The reference builds it as ``FunctionSpace(mesh, VectorElement(family, cell, fe_degree+1) *
FiniteElement(family, cell, fe_degree))`` (CoupledNavierStokesSolver.py:84-102).  DOLFIN's mixed dof
numbering is not reproducible outside DOLFIN; the layout fixed here - and used by libfsamd.so - is one block
of four unknowns (u_x, u_y, u_z, p) per P2 node (vertices first, then edge mid-points); the pressure is P1,
so only vertex nodes carry one: the pressure slot of an edge node is a dummy unknown that stays 0.
On triangles (the reference's own CFD example is 2-D, examples/test_cfd_solver.py:83) the block stays four wide -
(u_x, u_y, -, p) with the third slot one more dummy unknown - so that one operator layout and one solver serve both
dimensions; ``split`` hands back a 2-vector velocity.
"""
from __future__ import annotations

import numpy as np

from .fem import FunctionSpace, Function, SolverError, DirichletBC, _Element, periodic_vertex_pairs   # noqa: F401


class TaylorHoodSpace(FunctionSpace):
    def __init__(self, mesh, family="CG", pressure_degree=1, constrained_domain=None):
        if family not in ("CG", "P", "Lagrange"):
            raise SolverError("fe_family '{}' is not supported (CG/P/Lagrange only)".format(family))
        if int(pressure_degree) != 1:
            raise SolverError("Taylor-Hood is built for fe_degree 1 (P2 velocity / P1 pressure) only")
        if mesh.geometry().dim() not in (2, 3):
            raise SolverError("the Navier-Stokes path is built for triangular and tetrahedral meshes")
        self._gdim = mesh.geometry().dim()
        self._mesh = mesh
        self._degree = 2
        self._ufl_element = _Element("Mixed(P2^3 x P1)", 2, 4)
        self._ncomp = 4
        self._component = None
        self._parent = None
        self._device = None
        # periodic_boundary (CoupledNavierStokesSolver.py:97-100): vertex and edge nodes of the slave side are tied to the
        # master side, all four unknowns of a node together; the pressure space carries the same constraint
        self._constrained_domain = constrained_domain
        self._periodic = None if constrained_domain is None else periodic_vertex_pairs(mesh, constrained_domain)
        FunctionSpace._next_serial += 1
        self._serial = FunctionSpace._next_serial

    def velocity_dim(self):
        return self._gdim

    def num_sub_spaces(self):
        return 2

    def sub(self, i):
        if int(i) not in (0, 1):
            raise SolverError("the velocity-pressure space has sub spaces 0 (velocity) and 1 (pressure)")
        return TaylorHoodSub(self, int(i))

    def pressure_dofs(self):
        return np.arange(self._mesh.num_vertices(), dtype=np.int64) * 4 + 3

    def dummy_dofs(self):
        """Unknowns that are not unknowns: the pressure slot of the edge nodes and, on triangles, the third velocity slot."""
        edge_p = np.arange(self._mesh.num_vertices(), self.num_nodes(), dtype=np.int64) * 4 + 3
        if self._gdim == 3:
            return edge_p
        return np.concatenate([np.arange(self.num_nodes(), dtype=np.int64) * 4 + 2, edge_p])

    def pressure_space(self):
        if getattr(self, "_q", None) is None:
            self._q = FunctionSpace(self._mesh, "CG", 1, constrained_domain=self._constrained_domain)
        return self._q

    def velocity_space(self):
        if getattr(self, "_v", None) is None:
            self._v = FunctionSpace(self._mesh, "CG", 2, _ncomp=self._gdim, _holder=True)
        return self._v


class TaylorHoodSub:
    """W.sub(0) (velocity, optionally one component of it) or W.sub(1) (pressure): what DirichletBC needs."""

    def __init__(self, W, index, component=None):
        self._W, self._index, self._comp = W, index, component

    def mesh(self):
        return self._W.mesh()

    def root(self):
        return self._W

    def sub(self, j):
        if self._index != 0 or self._comp is not None:
            raise SolverError("only the velocity sub space has components")
        return TaylorHoodSub(self._W, 0, int(j))

    def dirichlet_dofs(self, facet_ids, evaluate):
        """(dofs, values) of a DirichletBC on the marked facets; evaluate(points, size) -> [n, size]."""
        W = self._W
        if self._index == 0:
            nodes = W.facet_nodes(facet_ids).astype(np.int64)
            co = W.node_coordinates()[nodes]
            d = W.velocity_dim()
            if self._comp is None:
                vals = evaluate(co, d)
                return (nodes[:, None] * 4 + np.arange(d)[None, :]).ravel().astype(np.int32), vals.reshape(-1)
            if self._comp >= d:
                raise SolverError("the velocity has {} components".format(d))
            return (nodes * 4 + self._comp).astype(np.int32), evaluate(co, 1).reshape(-1)
        verts = np.unique(W.mesh().facets()[facet_ids].astype(np.int64).ravel())
        return (verts * 4 + 3).astype(np.int32), evaluate(W.mesh().coordinates()[verts], 1).reshape(-1)


def split(w):
    """(u, p) copies of a Function of the mixed space (dolfin: w.split(deepcopy=True)); (u, p, T) when the flow solver
    solved the coupled temperature equation (MixedElement([V, Q, Q]), CoupledNavierStokesSolver.py:94-95)."""
    W = w.function_space()
    if not isinstance(W, TaylorHoodSpace):
        raise SolverError("split(): not a velocity-pressure function")
    a = w.vector()._values().reshape(-1, 4)
    u = Function(W.velocity_space())
    u.vector().set_local(np.ascontiguousarray(a[:, :W.velocity_dim()]).reshape(-1))
    p = Function(W.pressure_space())
    p.vector().set_local(a[:W.mesh().num_vertices(), 3])
    T = getattr(w, "_temperature", None)
    if T is not None:
        return u, p, T
    return u, p
