"""Host-side data model the solver API exposes: the small subset of the dolfin
namespace that FenicsSolver's settings dicts and example scripts use
 — Mesh, BoxMesh, UnitCubeMesh, MeshFunction,
SubDomain/AutoSubDomain, FunctionSpace/VectorFunctionSpace, Function, Constant,
Expression, DirichletBC, interpolate, near, Point.

These objects only describe the problem (topology, markers, coefficients,
Dirichlet sets).  They do no assembly and no linear algebra: that is
libfsamd.so's job (fenicssolver_amd.backend).  Semantics follow DOLFIN's as the
reference relies on them:
  * Mesh(xml) sorts each cell's vertices ascending; facets and edges are
    numbered lexicographically by their sorted vertex tuples (Appendix C1);
  * SubDomain.mark marks a facet when all its vertices and its midpoint are
    inside (D-4); DirichletBC is topological: all dofs on marked facets (D-3);
  * VectorFunctionSpace dofs are node-interleaved (D-7).
"""
from __future__ import annotations

import math
import numbers
import os
import re

import zlib

import numpy as np

DOLFIN_EPS = 3.0e-16


class SolverError(Exception):
    pass


def near(a, b, eps=DOLFIN_EPS):
    """dolfin.near: |a-b| < eps.  Works on scalars and arrays."""
    return np.abs(np.asarray(a) - b) < eps if isinstance(a, np.ndarray) else abs(a - b) < eps


class Point:
    def __init__(self, *xyz):
        if len(xyz) == 1 and hasattr(xyz[0], "__len__"):
            xyz = tuple(xyz[0])
        self._x = np.zeros(3)
        self._x[: len(xyz)] = xyz

    def x(self):
        return self._x[0]

    def y(self):
        return self._x[1]

    def z(self):
        return self._x[2]

    def array(self):
        return self._x.copy()

    def __getitem__(self, i):
        return self._x[i]


# --------------------------------------------------------------------------------------------
# mesh
# --------------------------------------------------------------------------------------------
_VERT_RE = re.compile(r'<vertex\s+index="(\d+)"\s+x="([^"]+)"\s+y="([^"]+)"(?:\s+z="([^"]+)")?')
_TET_RE = re.compile(r'<tetrahedron\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"\s+v3="(\d+)"')
_TRI_RE = re.compile(r'<triangle\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"')
_ENT_RE = re.compile(r'<entity\s+index="(\d+)"\s+value="(-?\d+)"')
_MF_RE = re.compile(r'<mesh_function\s+type="(\w+)"\s+dim="(\d+)"\s+size="(\d+)"')
_MVC_RE = re.compile(r'<value\s+cell_index="(\d+)"\s+local_entity="(\d+)"\s+value="(-?\d+)"')


class _Geometry:
    def __init__(self, mesh):
        self._m = mesh

    def dim(self):
        return self._m._coords.shape[1]


class _Topology:
    def __init__(self, mesh):
        self._m = mesh

    def dim(self):
        return self._m._cells.shape[1] - 1


class Mesh:
    """Tetrahedral (3-D) or triangular (2-D) mesh (dolfin.Mesh; SolverBase.py:203-258).  ``Mesh(path)`` reads
    DOLFIN XML.  Facets are the entities of dimension tdim - 1: triangles in 3-D, edges in 2-D."""

    def __init__(self, filename=None, coords=None, cells=None, _ordered=False):
        if filename is not None:
            coords, cells = self._read_xml(filename)
        if coords is None or cells is None:
            raise SolverError("Mesh needs a DOLFIN-XML file name or (coords, cells) arrays")
        self._coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        # mesh.order(): vertices of a cell ascending by GLOBAL id (_ordered: the caller already did that - the local
        # numbering of a distributed mesh is not monotone in the global one)
        self._cells = cells if _ordered else np.sort(cells, axis=1)
        if (self._coords.shape[1], self._cells.shape[1]) not in ((3, 4), (2, 3)):
            raise SolverError("tetrahedral meshes in 3D and triangular meshes in 2D are supported by fenicssolver_amd "
                              "(got gdim=%d, %d vertices per cell)" % (self._coords.shape[1], self._cells.shape[1]))
        self._topo = None
        self._device = None

    @staticmethod
    def _read_xml(path):
        if not os.path.exists(path):
            raise SolverError('mesh file: {} , does not exist'.format(path))
        text = open(path, "r").read()
        verts = _VERT_RE.findall(text)
        tets = _TET_RE.findall(text)
        tris = _TRI_RE.findall(text) if not tets else []
        if not verts or not (tets or tris):
            raise SolverError("{}: not a DOLFIN-XML tetrahedral or triangular mesh".format(path))
        if tris:
            coords = np.zeros((len(verts), 2))
            for idx, x, y, z in verts:
                coords[int(idx)] = (float(x), float(y))
            cells = np.zeros((len(tris), 3), dtype=np.int32)
            for t in tris:
                cells[int(t[0])] = (int(t[1]), int(t[2]), int(t[3]))
            return coords, cells
        coords = np.zeros((len(verts), 3))
        for idx, x, y, z in verts:
            coords[int(idx)] = (float(x), float(y), float(z) if z != "" else 0.0)
        cells = np.zeros((len(tets), 4), dtype=np.int32)
        for t in tets:
            cells[int(t[0])] = (int(t[1]), int(t[2]), int(t[3]), int(t[4]))
        return coords, cells

    # dolfin API -------------------------------------------------------------
    def geometry(self):
        return _Geometry(self)

    def topology(self):
        return _Topology(self)

    def coordinates(self):
        return self._coords

    def cells(self):
        return self._cells

    def num_vertices(self):
        return self._coords.shape[0]

    def num_cells(self):
        return self._cells.shape[0]

    def num_entities(self, dim):
        tdim = self._cells.shape[1] - 1
        if dim == 0:
            return self.num_vertices()
        if dim == tdim:
            return self.num_cells()
        if dim == tdim - 1:
            return len(self.facets())
        if dim == 1:
            return len(self.edges())
        raise SolverError("bad entity dimension %d" % dim)

    def num_facets(self):
        return len(self.facets())

    def hmin(self):
        c = self._coords[self._cells.astype(np.int64)]
        nvc = self._cells.shape[1]
        e = [np.linalg.norm(c[:, i] - c[:, j], axis=1) for i in range(nvc) for j in range(i + 1, nvc)]
        return float(np.min(e))

    # topology (lexicographic numbering) --------------------------------
    def _build_topology(self):
        if self._topo is not None:
            return self._topo
        cells = self._cells.astype(np.int64)
        nv = self.num_vertices()
        if cells.shape[1] == 3:      # triangles: facets are the edges, facet i opposite local vertex i
            ed = np.stack([cells[:, [1, 2]], cells[:, [0, 2]], cells[:, [0, 1]]], axis=1).reshape(-1, 2)
            ekey = ed[:, 0] * nv + ed[:, 1]
            uekey, einv, cnt = np.unique(ekey, return_inverse=True, return_counts=True)
            edges = np.stack([uekey // nv, uekey % nv], axis=1).astype(np.int32)
            self._topo = dict(facets=edges, cell_facets=einv.reshape(-1, 3).astype(np.int32),
                              facet_count=cnt.astype(np.int32), edges=edges,
                              cell_edges=einv.reshape(-1, 3).astype(np.int32))
            return self._topo
        opp = ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2))  # facet i is opposite local vertex i
        tri = np.stack([cells[:, list(o)] for o in opp], axis=1).reshape(-1, 3)  # already ascending
        key = (tri[:, 0] * nv + tri[:, 1]) * nv + tri[:, 2]
        ukey, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
        facets = np.stack([ukey // (nv * nv), (ukey // nv) % nv, ukey % nv], axis=1).astype(np.int32)
        loc = ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
        ed = np.stack([cells[:, list(e)] for e in loc], axis=1).reshape(-1, 2)
        ekey = ed[:, 0] * nv + ed[:, 1]
        uekey, einv = np.unique(ekey, return_inverse=True)
        edges = np.stack([uekey // nv, uekey % nv], axis=1).astype(np.int32)
        self._topo = dict(facets=facets, cell_facets=inv.reshape(-1, 4).astype(np.int32),
                          facet_count=cnt.astype(np.int32), edges=edges,
                          cell_edges=einv.reshape(-1, 6).astype(np.int32))
        return self._topo

    def facets(self):
        return self._build_topology()["facets"]

    def edges(self):
        return self._build_topology()["edges"]

    def cell_facets(self):
        return self._build_topology()["cell_facets"]

    def exterior_facets(self):
        """bool[num_facets]: facet belongs to exactly one cell."""
        return self._build_topology()["facet_count"] == 1

    def interior_facet_cells(self):
        """(cell pairs [nf,2], opposite-vertex pairs [nf,2]) of the facets shared by two cells (tetrahedra / triangles): what an
        interior-facet (dS) integral runs over and the node couplings it adds to the sparsity pattern."""
        if getattr(self, "_interior", None) is None:
            cf = self.cell_facets().astype(np.int64)                 # [nc,d+1], facet i opposite local vertex i
            nc, nl = cf.shape
            order = np.argsort(cf.ravel(), kind="stable")
            fid = cf.ravel()[order]
            dup = np.nonzero(fid[1:] == fid[:-1])[0]                  # consecutive equal ids = the two cells of a facet
            a, b = order[dup], order[dup + 1]
            ca, la, cb, lb = a // nl, a % nl, b // nl, b % nl
            cells = self._cells.astype(np.int64)
            self._interior = (np.stack([ca, cb], axis=1).astype(np.int32),
                              np.stack([cells[ca, la], cells[cb, lb]], axis=1).astype(np.int32))
            assert nc > 0
        return self._interior

    def device(self):
        """The mesh resident in HBM (created on first use)."""
        if self._device is None:
            from . import backend
            self._device = self._make_device(backend)
        return self._device

    def _make_device(self, backend):
        return backend.DeviceMesh(self._coords, self._cells)


class BoxMesh(Mesh):
    """dolfin.BoxMesh(Point, Point, nx, ny, nz) ordering (Appendix D-8;
    examples/test_linear_elasticity.py:42).  The host copy is built with numpy; the
    device copy is generated by a kernel, not uploaded.

    ``distributed=True`` under several ranks (DOLFIN's BoxMesh is distributed under mpirun): every rank holds ONLY its
    z-slab - the vertex planes it owns plus one ghost plane on either side and the cell layers touching owned planes -
    numbered as libfsamd.so numbers a slab: owned planes first (x fastest), then the lower, then the upper ghost plane.
    Nothing of global size is ever built on the host; global vertex ids are arithmetic (``global_vertex_ids()``).
    Boundary markers, Dirichlet sets, coefficients and the result live on the local mesh (the part of the field a rank can
    see, as with DOLFIN); ``parallel.gather_function(u)`` assembles the global nodal array when one is wanted.
    Built for P1 and P2 spaces (scalar / vector; the P2 node plan comes from the local cells alone,
    partition.build_p2_plan_local); the Taylor-Hood space uses the default replicated mesh."""

    def __init__(self, p0, p1, nx, ny, nz, distributed=False):
        a = p0.array() if isinstance(p0, Point) else np.asarray(p0, dtype=np.float64)
        b = p1.array() if isinstance(p1, Point) else np.asarray(p1, dtype=np.float64)
        nx, ny, nz = int(nx), int(ny), int(nz)
        self._box = (nx, ny, nz, tuple(a), tuple(b))
        self._slab = None
        planes = np.arange(nz + 1)
        layers = np.arange(nz)
        if distributed:
            from . import parallel, partition
            rank, size, _ = parallel.world()
            if size > 1:
                zb, ze = partition.slab_ranges(nz + 1, size)[rank]
                if ze - zb < 1:
                    raise SolverError("BoxMesh(distributed=True): rank {} would own no vertex plane ({} planes, {} ranks)".format(rank, nz + 1, size))
                lay = partition.slab_layout(nx, ny, nz, (zb, ze), rank, size)
                planes = np.asarray(lay["planes"])                          # owned planes, lower ghost, upper ghost
                layers = np.arange(max(zb - 1, 0), min(ze, nz))              # cell layers touching an owned plane
                self._slab = dict(lay, zplanes=(zb, ze), rank=rank, size=size, n_global=(nx + 1) * (ny + 1) * (nz + 1))
        x = a[0] + (np.arange(nx + 1, dtype=np.float64) * (b[0] - a[0])) / float(nx)
        y = a[1] + (np.arange(ny + 1, dtype=np.float64) * (b[1] - a[1])) / float(ny)
        z = a[2] + (planes.astype(np.float64) * (b[2] - a[2])) / float(nz)
        Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        P = (nx + 1) * (ny + 1)
        local_plane = np.full(nz + 1, -1, dtype=np.int64)                    # global z index -> local plane position
        local_plane[planes] = np.arange(len(planes))
        iz, iy, ix = np.meshgrid(layers, np.arange(ny), np.arange(nx), indexing="ij")
        inplane = (iy * (nx + 1) + ix).ravel().astype(np.int64)
        lo, hi = local_plane[iz.ravel()] * P + inplane, local_plane[iz.ravel() + 1] * P + inplane
        sx, sy = 1, nx + 1
        # corners in GLOBAL order v0..v7 (v4..v7 one plane up): a cell's vertices stay in ascending global order
        corner = [lo, lo + sx, lo + sy, lo + sx + sy, hi, hi + sx, hi + sy, hi + sx + sy]
        tets = ((0, 1, 3, 7), (0, 1, 5, 7), (0, 4, 5, 7), (0, 2, 3, 7), (0, 4, 6, 7), (0, 2, 6, 7))
        cells = np.stack([np.stack([corner[i] for i in t], axis=1) for t in tets], axis=1).reshape(-1, 4)
        Mesh.__init__(self, coords=coords, cells=cells, _ordered=self._slab is not None)

    def is_distributed(self):
        return self._slab is not None

    def global_vertex_ids(self):
        """Global id of every local vertex (the mesh's own numbering when it is not distributed)."""
        return self._slab["l2g"] if self._slab is not None else np.arange(self.num_vertices(), dtype=np.int64)

    def num_owned_vertices(self):
        return self._slab["n_owned"] if self._slab is not None else self.num_vertices()

    def exterior_facets(self):
        """Facets of the DOMAIN boundary: on a distributed slab the cut faces through the ghost planes have one cell too,
        but they are interior facets of the global mesh."""
        ext = Mesh.exterior_facets(self)
        if self._slab is None:
            return ext
        nx, ny, nz = self._box[:3]
        P = (nx + 1) * (ny + 1)
        gz = np.asarray(self._slab["planes"])[self.facets().astype(np.int64) // P]      # global z index of the facet's vertices
        flat = (gz[:, 0] == gz[:, 1]) & (gz[:, 1] == gz[:, 2])
        return ext & ~(flat & (gz[:, 0] != 0) & (gz[:, 0] != nz))

    def _make_device(self, backend):
        nx, ny, nz, a, b = self._box
        if self._slab is not None:
            return backend.DeviceMesh.box(nx, ny, nz, a, b, zplanes=self._slab["zplanes"])
        return backend.DeviceMesh.box(nx, ny, nz, a, b)


class UnitCubeMesh(BoxMesh):
    def __init__(self, nx, ny, nz, distributed=False):
        BoxMesh.__init__(self, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), nx, ny, nz, distributed=distributed)


class RectangleMesh(Mesh):
    """dolfin.RectangleMesh(Point, Point, nx, ny[, diagonal]): vertices x-fastest; per square (iy outer, ix inner) the
    two triangles (v0, v1, v3), (v0, v2, v3) of the default "right" diagonal (examples/test_heat_transfer.py:34 uses
    UnitSquareMesh(40, 40))."""

    def __init__(self, p0, p1, nx, ny, diagonal="right"):
        if diagonal != "right":
            raise SolverError("RectangleMesh: only the default 'right' diagonal is built")
        a = (p0.array() if isinstance(p0, Point) else np.asarray(p0, dtype=np.float64))[:2]
        b = (p1.array() if isinstance(p1, Point) else np.asarray(p1, dtype=np.float64))[:2]
        nx, ny = int(nx), int(ny)
        x = a[0] + (np.arange(nx + 1, dtype=np.float64) * (b[0] - a[0])) / float(nx)
        y = a[1] + (np.arange(ny + 1, dtype=np.float64) * (b[1] - a[1])) / float(ny)
        coords = np.stack([np.tile(x, ny + 1), np.repeat(y, nx + 1)], axis=1)
        ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
        v0 = (iy * (nx + 1) + ix).ravel().astype(np.int64)
        v1, v2 = v0 + 1, v0 + (nx + 1)
        v3 = v2 + 1
        cells = np.stack([np.stack([v0, v1, v3], axis=1), np.stack([v0, v2, v3], axis=1)], axis=1).reshape(-1, 3)
        Mesh.__init__(self, coords=coords, cells=cells)


class UnitSquareMesh(RectangleMesh):
    def __init__(self, nx, ny, diagonal="right"):
        RectangleMesh.__init__(self, (0.0, 0.0), (1.0, 1.0), nx, ny, diagonal)


class MeshFunction:
    """dolfin.MeshFunction("size_t", mesh, dim | filename[, value])."""

    def __init__(self, value_type, mesh, dim_or_file, value=None):
        self._mesh = mesh
        self._type = value_type
        dt = np.float64 if value_type == "double" else np.int64
        if isinstance(dim_or_file, str):
            self._dim, self._a = self._read_xml(mesh, dim_or_file, dt)
        else:
            self._dim = int(dim_or_file)
            self._a = np.zeros(mesh.num_entities(self._dim), dtype=dt)
            if value is not None:
                self._a[:] = value

    @staticmethod
    def _read_xml(mesh, path, dt):
        text = open(path, "r").read()
        m = _MF_RE.search(text)
        if m:  # old style: entity index = lexicographic facet/cell index
            dim, size = int(m.group(2)), int(m.group(3))
            if size != mesh.num_entities(dim):
                raise SolverError("{}: {} entities of dim {} but the mesh has {}".format(
                    path, size, dim, mesh.num_entities(dim)))
            a = np.zeros(size, dtype=dt)
            for idx, v in _ENT_RE.findall(text):
                a[int(idx)] = int(v)
            return dim, a
        vals = _MVC_RE.findall(text)
        if vals:  # mesh_value_collection: (cell, local entity) pairs
            dm = re.search(r'<mesh_value_collection[^>]*dim="(\d+)"', text)
            dim = int(dm.group(1))
            a = np.zeros(mesh.num_entities(dim), dtype=dt)
            cf = mesh.cell_facets()
            tdim = mesh.topology().dim()
            for c, le, v in vals:
                if dim == tdim - 1:
                    a[cf[int(c), int(le)]] = int(v)
                elif dim == tdim:
                    a[int(c)] = int(v)
            return dim, a
        raise SolverError("{}: unknown DOLFIN-XML mesh function format".format(path))

    def dim(self):
        return self._dim

    def mesh(self):
        return self._mesh

    def array(self):
        self._where = {}            # the caller may write through the returned array
        return self._a

    def set_all(self, v):
        self._where = {}
        self._a[:] = v

    def where(self, value):
        """Indices of the entities carrying `value` (cached: a time loop rebuilds its boundary conditions every step and
        a scan of 12 M facet markers takes milliseconds).  Every public accessor that can write - array(), set_all(),
        mf[i] = v - drops the cache; write through an array obtained earlier and call array() again before relying on it."""
        w = self.__dict__.setdefault("_where", {})
        if value not in w:
            if len(w) > 64:
                w.clear()
            w[value] = np.nonzero(self._a == value)[0]
        return w[value]

    def size(self):
        return self._a.size

    def __getitem__(self, i):
        return self._a[i]

    def __setitem__(self, i, v):
        self._where = {}
        self._a[i] = v


class SubDomain:
    """dolfin.SubDomain: override ``inside(x, on_boundary)``; ``mark(mf, id)``
    marks facets (or cells) whose vertices AND midpoint are all inside."""

    def inside(self, x, on_boundary):  # pragma: no cover - user supplied
        raise NotImplementedError

    def _inside_points(self, pts, on_boundary):
        """Vectorised where the user's predicate allows it, per-point otherwise."""
        ob = np.broadcast_to(np.asarray(on_boundary, dtype=bool), (pts.shape[0],))
        try:
            r = self.inside(pts.T, ob)
            r = np.asarray(r)
            if r.shape == (pts.shape[0],) and r.dtype == bool:
                return r
        except Exception:
            pass
        return np.fromiter((bool(self.inside(pts[i], bool(ob[i]))) for i in range(pts.shape[0])),
                           dtype=bool, count=pts.shape[0])

    def mark(self, mesh_function, marker_id):
        mesh = mesh_function.mesh()
        dim = mesh_function.dim()
        co = mesh.coordinates()
        tdim = mesh.topology().dim()
        if dim == tdim - 1:
            ent = mesh.facets().astype(np.int64)
            on_b = mesh.exterior_facets()
        elif dim == tdim:
            ent = mesh.cells().astype(np.int64)
            on_b = np.zeros(len(ent), dtype=bool)
        else:
            raise SolverError("SubDomain.mark: only facet and cell functions are supported")
        vert_on_b = np.zeros(mesh.num_vertices(), dtype=bool)
        vert_on_b[mesh.facets()[mesh.exterior_facets()].ravel()] = True
        vin = self._inside_points(co, vert_on_b)
        cand = np.nonzero(vin[ent].all(axis=1))[0]
        if cand.size:
            mid = co[ent[cand]].mean(axis=1)
            ok = self._inside_points(mid, on_b[cand])
            mesh_function.array()[cand[ok]] = marker_id


class AutoSubDomain(SubDomain):
    """dolfin.AutoSubDomain(lambda x[, on_boundary]: ...) (examples/test_heat_transfer.py:42-45)."""

    def __init__(self, fn):
        self._fn = fn
        try:
            self._nargs = fn.__code__.co_argcount
        except AttributeError:
            self._nargs = 2

    def inside(self, x, on_boundary):
        return self._fn(x) if self._nargs == 1 else self._fn(x, on_boundary)


# --------------------------------------------------------------------------------------------
# coefficients
# --------------------------------------------------------------------------------------------
class Constant:
    def __init__(self, value):
        self._v = np.asarray(value, dtype=np.float64)

    def values(self):
        return np.atleast_1d(self._v)

    def ufl_shape(self):
        return self._v.shape

    def value_size(self):
        return int(self._v.size)

    def __float__(self):
        return float(self._v)

    def __call__(self, *x):
        return self._v if self._v.ndim else float(self._v)


def _compile_cexpr(code):
    """C++-syntax scalar expression in x[0..2] (dolfin.Expression string) -> cexpr.CExpr (parsed, never eval'ed)."""
    from .cexpr import CExpr, CExprError
    try:
        return CExpr(code)
    except CExprError as e:
        raise SolverError("Expression '{}' cannot be parsed: {}".format(code, e))


class Expression:
    """dolfin.Expression("C++ string" | tuple of strings, degree=.., **params)
    (SolverBase.py:310-314, 364, 387; examples/test_linear_elasticity.py:68)."""

    def __init__(self, code, degree=1, **params):
        self._code = code
        self.degree = degree
        self.params = dict(params)
        if isinstance(code, (tuple, list)):
            if len(code) and isinstance(code[0], (tuple, list)):
                self._shape = (len(code), len(code[0]))
                flat = [c for row in code for c in row]
            else:
                self._shape = (len(code),)
                flat = list(code)
        else:
            self._shape = ()
            flat = [code]
        self._objs = [_compile_cexpr(c) for c in flat]

    def __setattr__(self, k, v):  # user parameters may be updated like dolfin: expr.t = ...
        if not k.startswith("_") and k not in ("degree", "params") and hasattr(self, "params"):
            self.params[k] = v
        object.__setattr__(self, k, v)

    def ufl_shape(self):
        return self._shape

    def value_size(self):
        return int(np.prod(self._shape)) if self._shape else 1

    def eval_points(self, pts):
        """pts[n,3] (or [n,2] on 2-D meshes) -> values[n] (scalar) or [n, size]."""
        pts = np.asarray(pts, dtype=np.float64)
        if pts.ndim == 1:
            pts = pts.reshape(1, -1)
        if pts.shape[1] < 3:
            pts = np.concatenate([pts, np.zeros((pts.shape[0], 3 - pts.shape[1]))], axis=1)
        from .cexpr import CExprError
        cols = []
        for o in self._objs:
            try:
                cols.append(o(pts, self.params))
            except CExprError as e:
                raise SolverError("Expression {}: {}".format(self._code, e))
        out = np.stack(cols, axis=1)
        return out[:, 0] if not self._shape else out

    def __call__(self, *x):
        p = np.zeros(3)
        xs = x[0] if len(x) == 1 and hasattr(x[0], "__len__") else x
        p[: len(xs)] = xs
        r = self.eval_points(p[None, :])
        return r[0]


class UserExpression:
    """dolfin.UserExpression: a Python subclass overrides ``eval(self, value, x)`` (and ``value_shape()`` for vector values),
    as the reference's CFD example does for its inlet profile (examples/test_cfd_solver.py:118-135).  Evaluated point by
    point on the host (boundary nodes, quadrature-free uses only)."""

    def __init__(self, degree=1, **kwargs):
        self.degree = degree
        for k, v in kwargs.items():
            setattr(self, k, v)

    def value_shape(self):
        return ()

    def ufl_shape(self):
        return tuple(self.value_shape())

    def value_size(self):
        sh = tuple(self.value_shape())
        return int(np.prod(sh)) if sh else 1

    def eval(self, value, x):
        raise SolverError("UserExpression subclasses must override eval(self, value, x)")

    def eval_points(self, pts):
        pts = np.asarray(pts, dtype=np.float64)
        if pts.ndim == 1:
            pts = pts.reshape(1, -1)
        n = self.value_size()
        out = np.zeros((pts.shape[0], n))
        for k in range(pts.shape[0]):
            self.eval(out[k], pts[k])
        return out[:, 0] if not tuple(self.value_shape()) else out

    def __call__(self, *x):
        xs = x[0] if len(x) == 1 and hasattr(x[0], "__len__") else x
        r = self.eval_points(np.asarray(xs, dtype=np.float64)[None, :])
        return r[0]


# --------------------------------------------------------------------------------------------
# function spaces / functions
# --------------------------------------------------------------------------------------------
class _Element:
    def __init__(self, family, degree, ncomp):
        self._f, self._d, self._n = family, degree, ncomp

    def degree(self):
        return self._d

    def family(self):
        return self._f

    def value_size(self):
        return self._n


def periodic_vertex_pairs(mesh, pb):
    """Slave / master vertex pairs of a periodic SubDomain, DOLFIN's rule (PeriodicBoundaryComputation): a boundary vertex
    x with pb.inside(x, True) is a master; any other boundary vertex is a slave when y = pb.map(x, y) lands inside, tied
    to the master vertex at y.  Chains (a master that is itself mapped, corners of doubly periodic domains) are followed
    to their end."""
    from scipy.spatial import cKDTree
    if not hasattr(pb, "map"):
        raise SolverError("periodic_boundary must be a SubDomain with inside(x, on_boundary) and map(x, y)")
    co = mesh.coordinates()
    bverts = np.unique(mesh.facets()[mesh.exterior_facets()].astype(np.int64).ravel())
    inside = pb._inside_points(co[bverts], True)
    masters_all, cand = bverts[inside], bverts[~inside]
    if len(masters_all) == 0:
        raise SolverError("periodic_boundary: inside() selects no boundary vertex")
    y = co[cand].copy()
    for i in range(len(cand)):
        pb.map(co[cand[i]], y[i])
    ok = pb._inside_points(y, True)
    slaves, targets = cand[ok], y[ok]
    if len(slaves) == 0:
        raise SolverError("periodic_boundary: map() sends no boundary vertex onto the master part")
    span = float(np.ptp(co, axis=0).max())
    dist, idx = cKDTree(co[masters_all]).query(targets)
    if dist.max() > 1e-8 * span:
        raise SolverError("periodic_boundary: the meshes of the two sides do not match ({} slave vertices have no master "
                          "within {:g})".format(int(np.count_nonzero(dist > 1e-8 * span)), 1e-8 * span))
    masters = masters_all[idx]
    fold = np.arange(mesh.num_vertices(), dtype=np.int64)
    fold[slaves] = masters
    for _ in range(8):                       # follow chains
        nxt = fold[fold]
        if np.array_equal(nxt, fold):
            break
        fold = nxt
    return slaves.astype(np.int32), fold[slaves].astype(np.int32)


class FunctionSpace:
    """dolfin.FunctionSpace / VectorFunctionSpace(mesh, "CG"|"P"|"Lagrange", degree) (SolverBase.py:260-275).
    Built: 3-D continuous P1 and P2, scalar or 3-vector (P2 nodes = vertices, then edge midpoints); 2-D P1 and P2, scalar or
    2-vector.
    Component i of node n of an ncomp-vector space is dof n*ncomp + i (DOLFIN interleaves the same way).
    Periodic constraints (constrained_domain): P1 and P2, one GPU, slave dofs kept and tied (see periodic_pairs()).
    Not built: degree > 2."""

    def __init__(self, mesh, family="CG", degree=1, constrained_domain=None, _ncomp=1, _component=None,
                 _parent=None, _holder=False):
        if family not in ("CG", "P", "Lagrange"):
            raise SolverError("fe_family '{}' is not supported (CG/P/Lagrange only)".format(family))
        if int(degree) not in (1, 2):
            raise SolverError("fe_degree {} is not built in fenicssolver_amd (P1 and P2 only)".format(degree))
        self._periodic = None
        if constrained_domain is not None:
            if _holder:
                raise SolverError("periodic_boundary (constrained_domain) is not meaningful for a container space")
            self._periodic = periodic_vertex_pairs(mesh, constrained_domain)      # vertices; periodic_pairs() adds P2 edge nodes
        if mesh.topology().dim() == 2 and not _holder and _ncomp not in (1, 2) and not (_ncomp == 4 and int(degree) == 2):
            raise SolverError("2-D (triangular) meshes carry scalar and 2-vector spaces (P1 or P2) in fenicssolver_amd")
        self._mesh = mesh
        self._degree = int(degree)
        self._ufl_element = _Element("Lagrange", int(degree), _ncomp)
        self._ncomp = _ncomp
        self._component = _component
        self._parent = _parent
        self._device = None
        FunctionSpace._next_serial += 1
        self._serial = FunctionSpace._next_serial

    _next_serial = 0

    def serial(self):
        """A number no other FunctionSpace of this process ever had (id() is recycled after garbage collection: caches keyed
        on a space use this)."""
        return self._serial

    def mesh(self):
        return self._mesh

    def periodic_pairs(self):
        """(slave vertices, master vertices) of the constrained_domain the space was built with, or None.  DOLFIN removes
        the slave dofs from the space; here they stay (dim() is unchanged), the assembled system is folded onto the masters
        on the device (fs_matrix_tie_nodes) and the slaves receive their masters' values after the solve."""
        root = self.root()
        if root._periodic is None or root._degree == 1:
            return root._periodic
        if getattr(root, "_periodic_nodes", None) is None:
            # P2: an edge whose two end points fold onto the end points of another edge is tied to that edge
            sl, ma = root._periodic
            nv = root._mesh.num_vertices()
            fold = np.arange(nv, dtype=np.int64)
            fold[sl] = ma
            ed = root.edge_nodes().astype(np.int64)
            fe = np.sort(fold[ed], axis=1)
            moved = (fe != ed).any(axis=1) & (fe[:, 0] != fe[:, 1])
            key = ed[:, 0] * nv + ed[:, 1]
            sorter = np.argsort(key)
            fk = fe[moved, 0] * nv + fe[moved, 1]
            pos = np.searchsorted(key[sorter], fk)
            pos[pos >= len(key)] = 0
            hit = key[sorter][pos] == fk
            slave_e = np.nonzero(moved)[0][hit]
            master_e = sorter[pos[hit]]
            root._periodic_nodes = (np.concatenate([sl, nv + slave_e]).astype(np.int32),
                                    np.concatenate([ma, nv + master_e]).astype(np.int32))
        return root._periodic_nodes

    def _periodic_couplings(self):
        """Node pairs the sparsity pattern needs for the folded system: (master, j) and (master, fold(j)) for every node
        j sharing a cell with a slave of that master, the slave itself included."""
        sl, ma = self.periodic_pairs()
        nv = self.num_nodes()
        fold = np.arange(nv, dtype=np.int64)
        fold[sl] = ma
        ce = self.cell_nodes().astype(np.int64)
        is_slave = np.zeros(nv, dtype=bool)
        is_slave[sl] = True
        touched = ce[is_slave[ce].any(axis=1)]
        k = touched.shape[1]
        a = np.repeat(touched, k, axis=1).ravel()
        b = np.tile(touched, (1, k)).ravel()
        keep = is_slave[a]                       # b == a too: the slave's diagonal moves to (slave, master) first
        m = fold[a[keep]]
        pairs = np.concatenate([np.stack([m, b[keep]], axis=1), np.stack([m, fold[b[keep]]], axis=1)])
        pairs = pairs[pairs[:, 0] != pairs[:, 1]]
        return np.unique(pairs, axis=0).astype(np.int32)

    def _interior_facet_node_pairs(self):
        """CG2: the node couplings an interior-facet integral adds to the sparsity pattern - every node of K+ that K- does not
        share (its vertex opposite the facet and the edges from it) with every such node of K-."""
        ca, cb = self._mesh.interior_facet_cells()[0].astype(np.int64).T
        cn = self.cell_nodes().astype(np.int64)
        A, B = cn[ca], cn[cb]
        eq = A[:, :, None] == B[:, None, :]
        only_a, only_b = ~eq.any(axis=2), ~eq.any(axis=1)
        k = int(only_a[0].sum()) if len(A) else 0
        if len(A) and not (np.all(only_a.sum(axis=1) == k) and np.all(only_b.sum(axis=1) == k)):
            raise SolverError("internal error: interior facets do not share a full facet's nodes")
        pa, pb = A[only_a].reshape(-1, k), B[only_b].reshape(-1, k)
        return np.stack([np.repeat(pa, k, axis=1).ravel(), np.tile(pb, (1, k)).ravel()], axis=1).astype(np.int32)

    def ufl_element(self):
        return self._ufl_element

    def degree(self):
        return self._degree

    P2_MAX_EDGE_CLASSES = 16

    def edge_nodes(self):
        """P2: the edge table [ne,2] in edge-node order (edge node e is dof num_vertices + e).
        DOLFIN's dof numbering is not reproducible, so the build fixes its own rule, the same as
        libfsamd.so applies on the device: lexicographic (v0, v1) in general; meshes with at most 16
        distinct index differences v1 - v0 (structured meshes) group their edges by that difference
        first, so that consecutive matrix rows share their column offsets (DIA slices)."""
        root = self.root()
        if getattr(root, "_edge_nodes", None) is None:
            ed = root._mesh.edges()
            delta = ed[:, 1].astype(np.int64) - ed[:, 0]
            if len(np.unique(delta)) <= self.P2_MAX_EDGE_CLASSES:
                ed = ed[np.lexsort((ed[:, 0], delta))]
            root._edge_nodes = ed
        return root._edge_nodes

    def cell_nodes(self):
        """[num_cells, 4 | 10] node ids of every cell: its vertices, then (P2) the nodes of its 6 UFC edges
        e0=(v2,v3) e1=(v1,v3) e2=(v1,v2) e3=(v0,v3) e4=(v0,v2) e5=(v0,v1); triangles: [num_cells, 3 | 6] with the UFC
        edges e0=(v1,v2) e1=(v0,v2) e2=(v0,v1)."""
        ce = self._mesh.cells().astype(np.int64)
        if self._degree == 1:
            return ce
        root = self.root()
        if getattr(root, "_cell_nodes", None) is None:
            nv = self._mesh.num_vertices()
            ed = self.edge_nodes().astype(np.int64)
            ekey = ed[:, 0] * nv + ed[:, 1]
            sorter = np.argsort(ekey)
            cols = [ce]
            local_edges = ((1, 2), (0, 2), (0, 1)) if ce.shape[1] == 3 else ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
            for i, j in local_edges:
                a, b = np.minimum(ce[:, i], ce[:, j]), np.maximum(ce[:, i], ce[:, j])
                cols.append((nv + sorter[np.searchsorted(ekey[sorter], a * nv + b)])[:, None])
            root._cell_nodes = np.concatenate(cols, axis=1)
        return root._cell_nodes

    def num_nodes(self):
        """P1: vertices; P2: vertices + edge nodes."""
        n = self._mesh.num_vertices()
        return n + len(self.edge_nodes()) if self._degree == 2 else n

    def dim(self):
        return self.num_nodes() * self._ncomp

    def node_coordinates(self):
        co = self._mesh.coordinates()
        if self._degree == 1:
            return co
        root = self.root()
        if getattr(root, "_node_co", None) is None:       # time loops rebuild their DirichletBCs every step
            ed = self.edge_nodes().astype(np.int64)
            root._node_co = np.concatenate([co, 0.5 * (co[ed[:, 0]] + co[ed[:, 1]])], axis=0)
            root._node_co.setflags(write=False)
        return root._node_co

    def facet_nodes(self, facet_ids):
        """Nodes in the closure of the given facets: their vertices (+ their edges for P2), ascending."""
        mesh = self._mesh
        root = self.root()
        cache = root.__dict__.setdefault("_facet_node_cache", {})    # a time loop rebuilds its DirichletBCs every step
        ids = np.ascontiguousarray(facet_ids)
        key = (ids.size, zlib.crc32(ids.tobytes()))
        if key in cache:
            return cache[key]
        if len(cache) > 64:
            cache.clear()
        f = mesh.facets()[facet_ids].astype(np.int64)
        verts = np.unique(f.ravel())
        out = verts if self._degree == 1 else self._facet_nodes_p2(f, verts)
        out.setflags(write=False)
        cache[key] = out
        return out

    def facet_node_table(self, facet_vertices):
        """[n_facets, n] nodes of every facet in a fixed local order: its vertices, then (P2) the nodes of its edges
        (0,1), (0,2), (1,2) - or the one edge node of a boundary edge in 2-D."""
        f = np.asarray(facet_vertices, dtype=np.int64)
        if self._degree == 1:
            return f
        nv = self._mesh.num_vertices()
        ed = self.edge_nodes().astype(np.int64)
        ekey = ed[:, 0] * nv + ed[:, 1]
        sorter = np.argsort(ekey)
        pairs = ((0, 1), (0, 2), (1, 2)) if f.shape[1] == 3 else ((0, 1),)
        cols = []
        for a, b in pairs:
            lo, hi = np.minimum(f[:, a], f[:, b]), np.maximum(f[:, a], f[:, b])
            cols.append(nv + sorter[np.searchsorted(ekey[sorter], lo * nv + hi)])
        return np.concatenate([f, np.stack(cols, axis=1)], axis=1)

    def _facet_nodes_p2(self, f, verts):
        mesh = self._mesh
        nv = mesh.num_vertices()
        ed = self.edge_nodes().astype(np.int64)
        ekey = ed[:, 0] * nv + ed[:, 1]
        sorter = np.argsort(ekey)
        if f.shape[1] == 2:            # 2-D: a facet is an edge, its closure the two vertices and the edge's own node
            fe = f
        else:
            fe = np.concatenate([f[:, [0, 1]], f[:, [0, 2]], f[:, [1, 2]]], axis=0)   # facet vertices are ascending
        fkey = np.unique(fe[:, 0] * nv + fe[:, 1])
        eid = sorter[np.searchsorted(ekey[sorter], fkey)]
        return np.concatenate([verts, np.sort(nv + eid)])

    def num_sub_spaces(self):
        return self._ncomp if self._ncomp > 1 else 0

    def sub(self, i):
        if self._ncomp == 1:
            raise SolverError("sub(): not a vector space")
        return FunctionSpace(self._mesh, "CG", self._degree, _ncomp=self._ncomp, _component=int(i), _parent=self)

    def component(self):
        return self._component

    def root(self):
        return self._parent if self._parent is not None else self

    def tabulate_dof_coordinates(self):
        return np.repeat(self.node_coordinates(), self._ncomp, axis=0)

    def device(self, facet_coupling=False):
        """Device space (sparsity + SELL slot table), built once per space.  facet_coupling: the pattern also holds
        the couplings of interior-facet integrals (the vertices opposite every interior facet); asking for it
        rebuilds a space that was created without."""
        root = self.root()
        if root._device is not None and facet_coupling and not getattr(root._device, "facet_coupled", False):
            root._device = None
        if root._device is None:
            from . import backend, parallel
            if not parallel.active() and not facet_coupling and root._periodic is None and self._wants_renumbering(root):
                # a mesh in FILE order on one GPU: the same machinery with one part whose numbering is the locality order
                root._device = self._make_parallel_device(root, backend, parallel, renumber=True)
            elif parallel.active():
                if root._periodic is not None and (facet_coupling or getattr(root._mesh, "_slab", None) is not None or root._ncomp == 4):
                    raise SolverError("periodic_boundary (constrained_domain) under domain decomposition is built for P1 / P2 spaces on a "
                                      "replicated host mesh (no interior-facet terms); Taylor-Hood spaces: one GPU; "
                                      "BoxMesh(distributed=True): one GPU or the replicated box")
                root._device = self._make_parallel_device(root, backend, parallel, facet_coupling=facet_coupling)
            else:
                pairs = None
                if facet_coupling:
                    if root._ncomp != 1:
                        raise SolverError("interior-facet (IP) terms are built for scalar spaces")
                    pairs = root._mesh.interior_facet_cells()[1] if root._degree == 1 else root._interior_facet_node_pairs()
                if root._periodic is not None:
                    extra = root._periodic_couplings()
                    pairs = extra if pairs is None else np.concatenate([np.asarray(pairs, dtype=np.int32).reshape(-1, 2), extra])
                root._device = backend.DeviceSpace(root._mesh.device(), root._ncomp, root._degree, coupled_pairs=pairs)
                root._localizer = None      # (a renumbered upload built earlier for this space had one: this device space is in file order)
        return root._device

    RENUMBER_MIN_VERTICES = 50000

    @staticmethod
    def _wants_renumbering(root):
        """DOLFIN renumbers the dofs of every FunctionSpace for locality (reorder_dofs_serial behind FunctionSpace(...),
        SolverBase.py:260-275); a mesh FILE's vertex order says nothing about which vertices are neighbours, and the SELL
        slices of the operator (64 consecutive rows = one wavefront) then gather x from all over the vector: on the 10 M-DOF
        cube with shuffled vertices the CG product runs at 0.76 instead of 4.9 TB/s, the assembly takes 19 instead of 3.3 ms
        (bench.py --mesh shuffled / renumbered).  So meshes that do not come from a generator are uploaded in the locality
        order of fs_mesh_locality_order (Morton curve of the coordinates, computed on the device) - through the Localizer
        that also serves domain decomposition, with one part: host arrays, dof numbers and results keep the file numbering.
        FS_RENUMBER = 0 / 1 pins the choice; automatic for tetrahedral file meshes of RENUMBER_MIN_VERTICES vertices or more."""
        mesh = root._mesh
        env = os.environ.get("FS_RENUMBER", "")
        if env == "0" or mesh.topology().dim() != 3 or getattr(mesh, "_box", None) is not None:
            return False
        if root._degree == 2 and root._ncomp not in (1, 3, 4):
            return False
        return env == "1" or mesh.num_vertices() >= FunctionSpace.RENUMBER_MIN_VERTICES

    @staticmethod
    def _make_parallel_device(root, backend, parallel, renumber=False, facet_coupling=False):
        """This rank's share of the space: owner-computes vertex slabs along the longest axis,
        one ghost-cell layer, halo plan on the device space (fenicssolver_amd/partition.py).
        facet_coupling (scalar P1 spaces): interior-facet (dS) integrals - the part takes a SECOND cell layer (the cell across a
        facet of a cell around an owned vertex), its own device mesh, and the pattern the couplings across the local facets."""
        from . import partition
        if facet_coupling:
            return FunctionSpace._make_parallel_facet_device(root, backend, parallel, partition)
        if root._degree == 2 and (root._ncomp not in (1, 3, 4) or mesh_dim(root) != 3):
            raise SolverError("multi-GPU decomposition is built for P1 spaces and, on tetrahedra, scalar / vector P2 spaces and the "
                              "Taylor-Hood space")
        rank, size = (0, 1) if renumber else parallel.ensure_comm()
        mesh = root._mesh
        if getattr(mesh, "_slab", None) is not None:
            # distributed box: the host mesh already IS this rank's part, numbered as the device numbers a slab
            lay = mesh._slab
            if root._degree == 2:      # (scalar, vector and - round 4 - the four-unknown Taylor-Hood node blocks)
                return FunctionSpace._make_distributed_p2_device(root, backend, parallel, partition, mesh, lay, rank)
            ds = backend.DeviceSpace(mesh.device(), root._ncomp, 1)
            if ds.n_owned != lay["n_owned"] * root._ncomp or ds.n_local != lay["n_local"] * root._ncomp:
                raise SolverError("internal error: host and device disagree on the slab layout")
            nc_ = root._ncomp
            sends = lay["send_lists"] if nc_ == 1 else [
                (np.asarray(l, dtype=np.int64)[:, None] * nc_ + np.arange(nc_)[None, :]).reshape(-1).astype(np.int32) for l in lay["send_lists"]]
            ds.set_halo(lay["neighbors"], sends, [c * nc_ for c in lay["recv_counts"]])
            root._localizer = parallel.LocalView(lay["n_owned"], lay["n_local"], mesh.num_cells(), lay["l2g"], lay["n_global"], nc_)
            return ds
        co, ce = mesh.coordinates(), mesh.cells()
        # one partition and ONE device mesh per host mesh: spaces on the same mesh share it (the stress projections assemble a
        # load on the P1 space from a field of another space and need both on the same device mesh)
        cache = mesh.__dict__.setdefault("_parallel_parts", {})
        if root._periodic is not None:
            return FunctionSpace._make_parallel_periodic_device(root, backend, parallel, partition, rank, size, cache)
        if (rank, size) not in cache:
            # an unconstrained space on a mesh that already carries a periodic space (the P1 space a stress is projected onto): the
            # same part and device mesh - its extra ghost vertices are columns nobody couples with here
            pk = next((k for k in cache if len(k) == 4 and k[:3] == (rank, size, "periodic")), None)
            if pk is not None:
                cache[(rank, size)] = cache[pk]
            pk2 = next((k for k in cache if len(k) == 4 and k[:3] == (rank, size, "periodic_p2")), None)
            if pk is None and pk2 is not None:       # (the part of a periodic CG2 space: its device mesh carries ORDER ids, its part extra cells)
                o2, p2, oid2, dm2 = cache[pk2]
                cache[(rank, size)] = (o2, p2, dm2)
                cache[("p2_plan_args", rank, size)] = (oid2, pk2)
        if (rank, size) not in cache and size == 1 and FunctionSpace._wants_renumbering(root):
            # one part in locality order (a mesh FILE on one GPU): ordered, re-indexed and built on the device in one upload
            # (fs_mesh_create_renumbered; the numpy re-indexing of the general path below took 6.7 s at 10 M vertices)
            dm, vo, cord = backend.DeviceMesh.renumbered(co, ce)
            part = partition.LocalPart(0, vo.astype(np.int64), len(vo), None, cord.astype(np.int64), [], [], [])
            cache[(rank, size)] = (np.zeros(len(vo), dtype=np.int32), part, dm)
        if (rank, size) not in cache:
            axis = int(np.argmax(co.max(axis=0) - co.min(axis=0)))
            owner = partition.slab_owner(co, size, axis=axis)
            vrank = crank = None
            if FunctionSpace._wants_renumbering(root):          # (file meshes on several ranks: the owned part in locality order too)
                vo, cord = backend.locality_order(co, ce)
                vrank, crank = np.empty(len(vo), dtype=np.int64), np.empty(len(cord), dtype=np.int64)
                vrank[vo], crank[cord] = np.arange(len(vo)), np.arange(len(cord))
            part = partition.build_local_part(ce, owner, rank, vrank, crank)
            cache[(rank, size)] = (owner, part, backend.DeviceMesh(co[part.l2g], part.cells, n_owned=part.n_owned, global_ids=part.l2g))
        owner, part, dm = cache[(rank, size)]
        ds = backend.DeviceSpace(dm, root._ncomp, root._degree)
        if root._degree == 1:
            if size > 1:
                ds.set_halo(part.neighbors, part.dof_send_lists(root._ncomp), [c * root._ncomp for c in part.recv_counts])
            root._localizer = parallel.Localizer(part, mesh.num_vertices(), root._ncomp)
        else:
            extra = cache.get(("p2_plan_args", rank, size))
            if extra is None:
                plan = partition.build_p2_plan(ce, owner, rank, part, ds.edges(), root.edge_nodes())
            else:       # the part belongs to a periodic CG2 space of this mesh (see _make_parallel_periodic_p2_device)
                sl_, ma_ = (np.asarray(a, dtype=np.int64) for a in cache[("p2_tied", rank, size)])
                mo_ = np.full(mesh.num_vertices(), -1, dtype=np.int64)
                mo_[sl_] = ma_
                ce64_ = np.asarray(ce, dtype=np.int64)
                plan = partition.build_p2_plan(ce, owner, rank, part, ds.edges(), root.edge_nodes(), order_id=extra[0],
                                               part_cells_of=lambda q: partition._local_cell_mask(ce64_, owner, q, None, mo_))
            nc_ = root._ncomp
            if plan.n_owned_nodes * nc_ != ds.n_owned:
                raise SolverError("internal error: host and device disagree on the owned P2 nodes")
            if size > 1:
                def dof_lists(lists):      # node lists -> dof lists (block of ncomp unknowns per node)
                    if nc_ == 1:
                        return lists
                    return [(np.asarray(l, dtype=np.int64)[:, None] * nc_ + np.arange(nc_)[None, :]).reshape(-1).astype(np.int32)
                            for l in lists]
                ds.set_halo(plan.neighbors, dof_lists(plan.send_lists), [c * nc_ for c in plan.recv_counts],
                            recv_lists=dof_lists(plan.recv_lists))
            root._localizer = parallel.Localizer(part, mesh.num_vertices(), nc_, p2_plan=plan, n_global_nodes=root.num_nodes())
        return ds

    @staticmethod
    def _make_parallel_periodic_device(root, backend, parallel, partition, rank, size, cache):
        """A P1 space with a periodic constraint on several ranks (SolverBase.py:260-275 under mpirun).  DOLFIN's dofmap has no
        slave dofs: a cell at the slave side couples its vertices with the MASTER, wherever that lives.  Here the slaves stay
        nodes and the assembled system is folded (fs_matrix_tie_nodes), so
          * a slave is owned by its master's rank (the row fold slave -> master is then rank-local),
          * a part holds the master of every slave among its vertices, as an extra ghost without cells where need be (the column
            fold (i, slave) -> (i, master) of a row i at the far side of the domain),
          * the pattern holds the (master, neighbour-of-slave) couplings that touch a local row.
        The part gets its own device mesh (its ghost list differs from the unconstrained part of the same host mesh)."""
        mesh = root._mesh
        co, ce = mesh.coordinates(), mesh.cells()
        sl, ma = (np.asarray(a, dtype=np.int64) for a in root._periodic)
        if root._degree == 2:
            return FunctionSpace._make_parallel_periodic_p2_device(root, backend, parallel, partition, rank, size, cache, co, ce, sl, ma)
        key = (rank, size, "periodic", hash(sl.tobytes()) ^ hash(ma.tobytes()))
        if key not in cache:
            axis = int(np.argmax(co.max(axis=0) - co.min(axis=0)))
            owner = np.array(partition.slab_owner(co, size, axis=axis))
            owner[sl] = owner[ma]
            part = partition.build_local_part(ce, owner, rank, tied=(sl, ma))
            cache[key] = (owner, part, backend.DeviceMesh(co[part.l2g], part.cells, n_owned=part.n_owned, global_ids=part.l2g))
        owner, part, dm = cache[key]
        g2l = part.g2l(mesh.num_vertices())
        pairs = g2l[root._periodic_couplings().astype(np.int64)]
        pairs = pairs[(pairs >= 0).all(axis=1) & (pairs < part.n_owned).any(axis=1)]
        ds = backend.DeviceSpace(dm, root._ncomp, 1, coupled_pairs=pairs.astype(np.int32))
        if size > 1:
            ds.set_halo(part.neighbors, part.dof_send_lists(root._ncomp), [c * root._ncomp for c in part.recv_counts])
        root._localizer = parallel.Localizer(part, mesh.num_vertices(), root._ncomp)
        return ds

    @staticmethod
    def _make_parallel_periodic_p2_device(root, backend, parallel, partition, rank, size, cache, co, ce, sl, ma):
        """A CG2 space with a periodic constraint on several ranks (round 5).  As for P1 (above), and for the EDGE nodes:
          * the master of a slave edge node is an edge node, and an edge exists on the device only inside a cell: a part also takes
            the cells around the masters of its slave vertices (partition.build_local_part(tied_cells=True));
          * an edge belongs to the owner of its end point of smaller id.  The device mesh is handed ORDER ids - a slave vertex right
            behind its master, 2 fold(v) + is_slave(v) - instead of the vertex numbers, so that a slave edge and its master edge are
            owned through corresponding end points, by one rank (a slave vertex already lives with its master);
          * the pattern's extra couplings are given in node numbers, which the device space itself defines: it is built twice,
            first without them (for its edge table), then with."""
        mesh = root._mesh
        nvg = mesh.num_vertices()
        key = (rank, size, "periodic_p2", hash(sl.tobytes()) ^ hash(ma.tobytes()))
        if key not in cache:
            axis = int(np.argmax(co.max(axis=0) - co.min(axis=0)))
            owner = np.array(partition.slab_owner(co, size, axis=axis))
            owner[sl] = owner[ma]
            part = partition.build_local_part(ce, owner, rank, tied=(sl, ma), tied_cells=True)
            fold = np.arange(nvg, dtype=np.int64)
            fold[sl] = ma
            order_id = 2 * fold
            order_id[sl] += 1
            cache[key] = (owner, part, order_id, backend.DeviceMesh(co[part.l2g], part.cells, n_owned=part.n_owned, global_ids=order_id[part.l2g]))
        owner, part, order_id, dm = cache[key]
        cache[("p2_tied", rank, size)] = (sl, ma)
        master_of = np.full(nvg, -1, dtype=np.int64)
        master_of[sl] = ma
        ce64 = np.asarray(ce, dtype=np.int64)
        cells_of = lambda q: partition._local_cell_mask(ce64, owner, q, None, master_of)
        nc_ = root._ncomp
        probe = backend.DeviceSpace(dm, nc_, 2)                  # (edge table and node numbering: the same with the extra couplings)
        plan = partition.build_p2_plan(ce, owner, rank, part, probe.edges(), root.edge_nodes(), order_id=order_id, part_cells_of=cells_of)
        if plan.n_owned_nodes * nc_ != probe.n_owned:
            raise SolverError("internal error: host and device disagree on the owned P2 nodes of a periodic space")
        loc = parallel.Localizer(part, nvg, nc_, p2_plan=plan, n_global_nodes=root.num_nodes())
        pairs = loc.g2l[root._periodic_couplings().astype(np.int64)]
        pairs = pairs[(pairs >= 0).all(axis=1) & (pairs < plan.n_owned_nodes).any(axis=1)]
        probe_edges = np.asarray(probe.edges()).copy()
        probe.close()
        ds = backend.DeviceSpace(dm, nc_, 2, coupled_pairs=pairs.astype(np.int32))
        if not np.array_equal(np.asarray(ds.edges()), probe_edges):
            raise SolverError("internal error: the extra couplings of a periodic space changed the device's edge numbering")
        if size > 1:
            def dof_lists(lists):
                if nc_ == 1:
                    return lists
                return [(np.asarray(l, dtype=np.int64)[:, None] * nc_ + np.arange(nc_)[None, :]).reshape(-1).astype(np.int32) for l in lists]
            ds.set_halo(plan.neighbors, dof_lists(plan.send_lists), [c * nc_ for c in plan.recv_counts], recv_lists=dof_lists(plan.recv_lists))
        root._localizer = loc
        return ds

    @staticmethod
    def _make_parallel_facet_device(root, backend, parallel, partition):
        mesh = root._mesh
        if root._degree != 1 or root._ncomp != 1:
            raise SolverError("interior-facet (IP) terms under domain decomposition are built for scalar P1 spaces")
        if getattr(mesh, "_slab", None) is not None:
            raise SolverError("interior-facet (IP) terms need two ghost-cell layers: BoxMesh(distributed=True) carries one - use the "
                              "replicated BoxMesh (every rank then cuts its own two-layer part)")
        rank, size = parallel.ensure_comm()
        co, ce = mesh.coordinates(), mesh.cells()
        cache = mesh.__dict__.setdefault("_parallel_parts", {})
        key = (rank, size, "facets")
        if key not in cache:
            axis = int(np.argmax(co.max(axis=0) - co.min(axis=0)))
            owner = partition.slab_owner(co, size, axis=axis)
            fcells, fopp = mesh.interior_facet_cells()
            part = partition.build_local_part(ce, owner, rank, face_pairs=fcells)
            # the interior facets both of whose cells are local, in local cell / vertex numbers
            cg2l = np.full(len(ce), -1, dtype=np.int64)
            cg2l[part.cell_gids] = np.arange(len(part.cell_gids))
            lf = cg2l[fcells.astype(np.int64)]
            sel = (lf >= 0).all(axis=1)
            g2l = part.g2l(mesh.num_vertices())
            cache[key] = (owner, part, backend.DeviceMesh(co[part.l2g], part.cells, n_owned=part.n_owned, global_ids=part.l2g),
                          lf[sel].astype(np.int32), g2l[fopp.astype(np.int64)[sel]].astype(np.int32))
        owner, part, dm, lfacets, lpairs = cache[key]
        ds = backend.DeviceSpace(dm, 1, 1, coupled_pairs=lpairs)
        if size > 1:
            ds.set_halo(part.neighbors, part.dof_send_lists(1), list(part.recv_counts))
        root._localizer = parallel.Localizer(part, mesh.num_vertices(), 1)
        root._localizer.interior_facet_cells = lfacets
        return ds

    @staticmethod
    def _make_distributed_p2_device(root, backend, parallel, partition, mesh, lay, rank):
        """CG2 space on a distributed box: the node plan comes from this rank's cells alone (partition.build_p2_plan_local), the
        host keeps its own node order and talks to the device through a permutation of LOCAL data (parallel.LocalNodeView)."""
        nc_ = root._ncomp
        ds = backend.DeviceSpace(mesh.device(), nc_, 2)
        nv = mesh.num_vertices()
        P = lay["plane_size"]
        owner = np.full(nv, rank, dtype=np.int32)
        off = lay["n_owned"]
        for q in lay["neighbors"]:                     # ghost planes follow the owned ones: lower neighbour first
            owner[off:off + P] = q
            off += P
        dev_edges = ds.edges().astype(np.int64)
        plan = partition.build_p2_plan_local(mesh.cells(), lay["l2g"], owner, rank, lay["neighbors"], dev_edges, lay["n_global"])
        if plan.n_owned_nodes * nc_ != ds.n_owned:
            raise SolverError("internal error: host and device disagree on the owned P2 nodes")
        # device node -> host node: vertices keep their (local) vertex id, an edge is looked up by its end points
        he = root.edge_nodes().astype(np.int64)
        hkey = np.minimum(he[:, 0], he[:, 1]) * nv + np.maximum(he[:, 0], he[:, 1])
        hsort = np.argsort(hkey)
        dkey = np.minimum(dev_edges[:, 0], dev_edges[:, 1]) * nv + np.maximum(dev_edges[:, 0], dev_edges[:, 1])
        pos = hsort[np.searchsorted(hkey[hsort], dkey)]
        if len(he) != len(dev_edges) or not np.array_equal(hkey[pos], dkey):
            raise SolverError("internal error: host and device disagree on the edges of the local mesh")
        l2h = np.empty(nv + len(dev_edges), dtype=np.int64)
        l2h[plan.node_of_vertex] = np.arange(nv)
        l2h[plan.node_of_edge] = nv + pos

        def dof_lists(lists):
            if nc_ == 1:
                return lists
            return [(np.asarray(l, dtype=np.int64)[:, None] * nc_ + np.arange(nc_)[None, :]).reshape(-1).astype(np.int32) for l in lists]
        ds.set_halo(plan.neighbors, dof_lists(plan.send_lists), [c * nc_ for c in plan.recv_counts], recv_lists=dof_lists(plan.recv_lists))
        root._localizer = parallel.LocalNodeView(plan.n_owned_nodes, l2h, mesh.num_cells(), lay["n_owned"], nc_, lay["l2g"],
                                                 lay["n_global"], plan.edge_gid_pairs)
        return ds

    def localizer(self):
        """None on one GPU; the global->local mapper of this rank's part otherwise (after device())."""
        return getattr(self.root(), "_localizer", None)


def mesh_dim(space):
    return space._mesh.geometry().dim()


def VectorFunctionSpace(mesh, family="CG", degree=1, dim=None, constrained_domain=None):
    return FunctionSpace(mesh, family, degree, constrained_domain, _ncomp=dim or mesh.geometry().dim())


def TensorFunctionSpace(mesh, family="CG", degree=1, shape=None):
    """Host-side container of a nodal tensor field (dim x dim values per node, row-major): what viscous_stress /
    project(sigma, T) return (CoupledNavierStokesSolver.py:149-155).  Never assembled on: no device space behind it."""
    d = mesh.geometry().dim()
    n = int(np.prod(shape)) if shape else d * d
    return FunctionSpace(mesh, family, degree, _ncomp=n, _holder=True)


class _Vector:
    """Minimal GenericVector: numpy storage with the calls the reference's users make.

    The values may also live in a device vector (the solution a solve just produced stays in HBM: the next time step reads
    its previous field from there, no PCIe round trip).  Exactly one side may be stale: ``array()`` (a writable view) and
    ``set_local`` make the host side the truth, ``_adopt_device`` the device side; ``get_local`` / ``_values`` only read."""

    def __init__(self, n):
        self._a = np.zeros(n)
        self._dev = None          # backend.DeviceVector holding (at least) the first len(_a) entries
        self._host_ok = True
        self._dev_ok = False

    def _sync_host(self):
        if not self._host_ok:
            self._a[:] = self._dev.get(self._a.size)
            self._host_ok = True

    def _values(self):
        """Read-only access for the library itself: the device copy, if any, stays valid."""
        self._sync_host()
        return self._a

    def _adopt_device(self, dev):
        """dev (length >= len(self)) now IS this vector; the host copy is refreshed on first use."""
        self._dev, self._dev_ok, self._host_ok = dev, True, False

    def _device(self, n=None):
        """A device vector with these values (n entries, default len(self); the rest - ghost slots - zero)."""
        from . import backend
        n = self._a.size if n is None else int(n)
        if self._dev_ok and self._dev is not None and self._dev.n >= n:
            return self._dev
        self._sync_host()
        host = self._a if n == self._a.size else np.concatenate([self._a, np.zeros(n - self._a.size)])
        if self._dev is None or self._dev.n != n:
            self._dev = backend.DeviceVector(n)
        self._dev.set(host)
        self._dev_ok = True
        return self._dev

    def get_local(self):
        return self._values().copy()

    def set_local(self, v):
        self._a[:] = v
        self._host_ok, self._dev_ok = True, False

    def array(self):
        self._sync_host()
        self._dev_ok = False      # the caller may write through the view
        return self._a

    def assign_from(self, other):
        """Copy of another vector of the same size: device to device when the source lives there."""
        if other._dev_ok and other._dev is not None and not other._host_ok:
            from . import backend
            if self._dev is None or self._dev.n != other._dev.n:
                self._dev = backend.DeviceVector(other._dev.n)
            self._dev.copy_from(other._dev)
            self._dev_ok, self._host_ok = True, False
        else:
            self.set_local(other._values())

    def copy(self):
        v = _Vector(self._a.size)
        v.assign_from(self)
        return v

    def size(self):
        return self._a.size

    def norm(self, kind="l2"):
        return float(np.linalg.norm(self._values(), {"l2": 2, "linf": np.inf, "l1": 1}[kind]))

    def apply(self, mode):
        pass

    def __len__(self):
        return self._a.size

    def __getitem__(self, i):
        return self._values()[i]

    def __setitem__(self, i, v):
        self.array()[i] = v


def locate_point(mesh, p):
    """(cell index, barycentric coordinates) of the cell holding point p (brute force; post-processing sizes)."""
    co, ce = mesh.coordinates(), mesh.cells().astype(np.int64)
    gdim = co.shape[1]
    c = co[ce]
    T = np.stack([c[:, k + 1] - c[:, 0] for k in range(gdim)], axis=2)
    lam = np.linalg.solve(T, np.broadcast_to(np.asarray(p, dtype=np.float64)[:gdim] - c[:, 0], (len(ce), gdim))[:, :, None])[:, :, 0]
    bary = np.concatenate([1.0 - lam.sum(axis=1, keepdims=True), lam], axis=1)
    i = int(np.argmax(bary.min(axis=1)))
    if bary[i].min() < -1e-10:
        raise SolverError("point {} is outside the mesh".format(p))
    return i, bary[i]


class Function:
    """dolfin.Function(V) (SolverBase.py:472-475)."""

    def __init__(self, V, other=None):
        if isinstance(V, Function):  # copy constructor Function(u)
            other, V = V, V.function_space()
        self._V = V
        self._vec = _Vector(V.dim())
        self._name = "f"
        if isinstance(other, Function):
            self._vec.assign_from(other._vec)

    def function_space(self):
        return self._V

    def vector(self):
        return self._vec

    def assign(self, other):
        if isinstance(other, Function):
            self._vec.assign_from(other._vec)
        elif isinstance(other, Constant):
            self._vec.array().reshape(-1, self._V._ncomp)[:] = other.values()
        else:
            raise SolverError("Function.assign: unsupported source {}".format(type(other)))

    def rename(self, name, label):
        self._name = name

    def name(self):
        return self._name

    def copy(self, deepcopy=True):
        return Function(self._V, self)

    def compute_vertex_values(self, mesh=None):
        """dolfin layout: component-major [ncomp * num_vertices]."""
        v = self.vertex_values()
        return v.copy() if v.ndim == 1 else v.T.ravel().copy()

    def vertex_values(self):
        """[num_vertices] (scalar) or [num_vertices, ncomp]: the dofs that sit on mesh vertices."""
        n = self._V._ncomp
        nv = self._V.mesh().num_vertices()
        return self._vec._values()[:nv] if n == 1 else self._vec._values().reshape(-1, n)[:nv]

    def node_values(self):
        """All nodal dofs: [num_nodes] or [num_nodes, ncomp] (P2: vertices then edge midpoints)."""
        n = self._V._ncomp
        return self._vec._values() if n == 1 else self._vec._values().reshape(-1, n)

    def ufl_shape(self):
        return () if self._V._ncomp == 1 else (self._V._ncomp,)

    def __call__(self, *x):
        """Point evaluation by brute-force cell search (small meshes / post-processing only)."""
        p = np.zeros(3)
        xs = x[0] if len(x) == 1 and hasattr(x[0], "__len__") else x
        if isinstance(xs, Point):
            xs = xs.array()
        p[: len(xs)] = xs
        mesh = self._V.mesh()
        i, bary = locate_point(mesh, p)
        vals = self.vertex_values()[mesh.cells().astype(np.int64)[i]]
        return bary @ vals


def interpolate(v, V):
    """dolfin.interpolate(Expression|Constant|Function, V) for P1: nodal evaluation."""
    f = Function(V)
    co = V.node_coordinates()
    n = V._ncomp
    if isinstance(v, Expression):
        vals = v.eval_points(co)
    elif isinstance(v, Constant):
        vals = np.broadcast_to(v.values() if n > 1 else float(v), (co.shape[0], n) if n > 1 else (co.shape[0],))
    elif isinstance(v, Function):
        vals = v.node_values()
        if len(vals) != co.shape[0]:
            raise SolverError("interpolate: source and target function spaces differ")
    elif isinstance(v, numbers.Number):
        vals = np.full(co.shape[0], float(v))
    else:
        raise SolverError("interpolate: unsupported source {}".format(type(v)))
    vals = np.asarray(vals, dtype=np.float64)
    if n > 1 and vals.shape != (co.shape[0], n):
        raise SolverError("interpolate: value shape {} does not match a {}-vector space".format(vals.shape, n))
    f.vector().set_local(vals.reshape(-1))
    return f


def project(v, V):
    """For the P1 sources used here projection == interpolation of nodal data."""
    return interpolate(v, V)


def nodal_values(value, V):
    """Values of a coefficient at the vertices: number/Constant/Expression/Function -> array."""
    co = V.node_coordinates()
    if isinstance(value, numbers.Number):
        return np.full(co.shape[0], float(value))
    if isinstance(value, Constant):
        v = value.values()
        return np.full(co.shape[0], float(v[0])) if v.size == 1 else np.broadcast_to(v, (co.shape[0], v.size)).copy()
    if isinstance(value, Expression):
        return value.eval_points(co)
    if isinstance(value, Function):
        vals = value.node_values()
        return vals if len(vals) == co.shape[0] else value.vertex_values()
    raise SolverError("cannot evaluate {} at the nodes of the space".format(type(value)))


class Measure:
    """dolfin.Measure("ds" | "dx" | "dS", subdomain_data=markers[, domain=mesh]): the integration measure the reference
    builds for its boundary terms (ScalarTransportSolver.py:262, LinearElasticitySolver.py:210) and hands to
    update_boundary_conditions.  ``ds(3)`` names the part marked 3: a (kind, marker_id, markers) record - what the
    term descriptions of forms.py carry instead of UFL integrals; ``ds(3).facets()`` lists the marked entities."""

    def __init__(self, kind, subdomain_data=None, domain=None, subdomain_id=None):
        if kind not in ("dx", "ds", "dS"):
            raise SolverError("Measure: integral type '{}' is not one of dx, ds, dS".format(kind))
        self.kind, self.subdomain_data, self.domain, self.subdomain_id = kind, subdomain_data, domain, subdomain_id

    def __call__(self, subdomain_id=None, domain=None, subdomain_data=None):
        return Measure(self.kind, subdomain_data if subdomain_data is not None else self.subdomain_data,
                       domain if domain is not None else self.domain, subdomain_id)

    def integral_type(self):
        return {"dx": "cell", "ds": "exterior_facet", "dS": "interior_facet"}[self.kind]

    def facets(self):
        """Indices of the marked entities (all of them for an unrestricted measure without markers)."""
        if self.subdomain_data is None or self.subdomain_id is None:
            raise SolverError("Measure.facets(): needs subdomain_data and a subdomain id")
        return self.subdomain_data.where(self.subdomain_id)

    def __repr__(self):
        return "%s(%s)" % (self.kind, "" if self.subdomain_id is None else self.subdomain_id)


class PointSource:
    """dolfin.PointSource(V, Point, magnitude): a Dirac load, b[dof] += magnitude * phi_dof(point) on the cell that
    holds the point (examples/test_electrostatics.py:52-53; ScalarTransportSolver.py:148-155)."""

    def __init__(self, V, point, magnitude=1.0):
        self.function_space, self.magnitude = V, float(magnitude)
        p = np.zeros(3)
        xs = point.array() if isinstance(point, Point) else np.asarray(point, dtype=np.float64).ravel()
        p[: len(xs)] = xs
        self.point = p
        if V._degree != 1 or V._ncomp != 1:
            raise SolverError("PointSource is built for scalar P1 spaces")
        mesh = V.mesh()
        i, bary = locate_point(mesh, p)
        self.dofs = mesh.cells()[i].astype(np.int32)
        self.weights = self.magnitude * bary


def is_constant_value(value):
    return isinstance(value, (numbers.Number, Constant))


class DirichletBC:
    """dolfin.DirichletBC(V | V.sub(i), value, facet_markers, id) — topological
    (ScalarTransportSolver.py:169-175; LinearElasticitySolver.py:125-133)."""

    def __init__(self, V, value, markers, marker_id):
        self.function_space = V
        self.value = value
        self.marker_id = marker_id
        mesh = V.mesh()
        if isinstance(markers, MeshFunction):
            if markers.dim() != mesh.topology().dim() - 1:
                raise SolverError("DirichletBC needs a facet MeshFunction")
            sel = markers.where(marker_id)
        else:
            raise SolverError("DirichletBC: markers must be a MeshFunction")
        if hasattr(V, "dirichlet_dofs"):                # sub space of the velocity-pressure space (mixed.py)
            self.dofs, self.values = V.dirichlet_dofs(sel, lambda pts, size: self._eval(value, pts, size))
            return
        verts = V.facet_nodes(sel).astype(np.int64)     # P2: vertices and edge nodes of the marked facets
        n = V._ncomp
        comp = V.component()
        co = V.node_coordinates()[verts]
        if isinstance(value, Function):
            # translate_value() turns strings / tuples of strings / file names into Functions (SolverBase.py:349-393);
            # DOLFIN's DirichletBC takes their values at the constrained dofs
            vals = value.node_values()
            if vals.shape[0] != V.num_nodes():
                raise SolverError("DirichletBC: the value Function lives on another space ({} nodes, the BC space has {})".format(
                    vals.shape[0], V.num_nodes()))
            vals = vals.reshape(V.num_nodes(), -1)[verts]
            if comp is not None and vals.shape[1] == n:
                vals = vals[:, [comp]]                  # vector Function on V.sub(i): its i-th component
            size = 1 if (n == 1 or comp is not None) else n
            if vals.shape[1] != size:
                raise SolverError("DirichletBC value has {} components, the (sub)space has {}".format(vals.shape[1], size))
            ev = lambda pts, sz: vals                   # noqa: E731
        else:
            ev = lambda pts, sz: self._eval(value, pts, sz)     # noqa: E731
        if n == 1:
            self.dofs = verts.astype(np.int32)
            self.values = ev(co, 1).reshape(-1).astype(np.float64)
        elif comp is not None:
            self.dofs = (verts * n + comp).astype(np.int32)
            self.values = ev(co, 1).reshape(-1).astype(np.float64)
        else:
            self.dofs = (verts[:, None] * n + np.arange(n)[None, :]).ravel().astype(np.int32)
            self.values = ev(co, n).reshape(-1).astype(np.float64)

    @staticmethod
    def _eval(value, pts, size):
        if isinstance(value, numbers.Number):
            v = np.full((pts.shape[0], 1), float(value))
        elif isinstance(value, Constant):
            v = np.broadcast_to(value.values().reshape(1, -1), (pts.shape[0], value.value_size())).copy()
        elif isinstance(value, (Expression, UserExpression)):
            v = value.eval_points(pts).reshape(pts.shape[0], -1)
        elif isinstance(value, (tuple, list, np.ndarray)):
            v = np.broadcast_to(np.asarray(value, dtype=np.float64).reshape(1, -1), (pts.shape[0], len(value))).copy()
        else:
            raise SolverError("DirichletBC value of type {} is not supported".format(type(value)))
        if v.shape[1] != size:
            raise SolverError("DirichletBC value has {} components, the (sub)space has {}".format(v.shape[1], size))
        return v

    def get_boundary_values(self):
        return dict(zip(self.dofs.tolist(), self.values.tolist()))
