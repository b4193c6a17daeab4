"""Device objects over the C-ABI: meshes, function spaces, matrices, vectors
and the Krylov solve, one GPU per process.

These play the role of dolfin's Mesh/FunctionSpace/PETScMatrix/PETScVector/
PETScKrylovSolver for the hot path of FenicsSolver/SolverBase.py:592-672.
Everything here runs on the MI355X through libfsamd.so; nothing is computed
on the host and nothing falls back to it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import BackendError  # noqa: F401  (re-export)


def init(device_id=0):
    L.check(L.load().fs_init(int(device_id)), "fs_init")


def device_count():
    n = C.c_int(0)
    L.load().fs_device_count(C.byref(n))
    return n.value


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    L.check(L.load().fs_device_info(name, 256, C.byref(cus), C.byref(mem)), "fs_device_info")
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}


def synchronize():
    L.check(L.load().fs_device_synchronize(), "fs_device_synchronize")


def profile_marker(phase):
    """An empty, named launch (k_profile_marker) between the legs of a traced command: the n-th one opens phase n of
    tools/summarize_profiles.py."""
    L.check(L.load().fs_profile_marker(int(phase)), "fs_profile_marker")


def memory_info():
    """Bytes of device memory the library holds: in use by live objects / idle in its block cache."""
    live, cached = C.c_int64(0), C.c_int64(0)
    L.check(L.load().fs_memory_info(C.byref(live), C.byref(cached)), "fs_memory_info")
    return {"live_bytes": live.value, "cached_bytes": cached.value}


def trim_memory():
    """Return the idle blocks of the cache to the driver."""
    L.check(L.load().fs_memory_trim(), "fs_memory_trim")


class _Handle:
    _destroy = None
    _serials = [0]

    def __init__(self):
        self.h = C.c_void_p()
        _Handle._serials[0] += 1
        self.serial = _Handle._serials[0]      # never reused (id() is): what caches keyed on a device object compare

    def close(self):
        if self.h is not None and self.h.value:
            getattr(L.load(), self._destroy)(self.h)
            self.h = C.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


class DeviceMesh(_Handle):
    """Tetrahedral mesh resident in HBM (dolfin.Mesh, SolverBase.py:203-258)."""
    _destroy = "fs_mesh_destroy"

    def __init__(self, coords=None, cells=None, n_owned=None, global_ids=None):
        super().__init__()
        if coords is None:
            return
        coords = L.f64(coords)
        cells = L.i32(cells)
        if coords.ndim != 2 or cells.ndim != 2:
            raise BackendError("DeviceMesh: coords must be [nv,gdim], cells [nc,verts]")
        nv = coords.shape[0]
        if n_owned is None:
            n_owned = nv
        self._keep = (coords, cells)
        L.check(L.load().fs_mesh_create(coords.shape[1], nv, L.p_f64(coords), cells.shape[0], L.p_i32(cells),
                                        cells.shape[1], int(n_owned), C.byref(self.h)), "fs_mesh_create")
        self._keep = None
        if global_ids is not None:
            g = L.i64(global_ids)
            if g.shape != (nv,):
                raise BackendError("DeviceMesh: global_ids must have one entry per local vertex")
            L.check(L.load().fs_mesh_set_global_ids(self.h, L.p_i64(g)), "fs_mesh_set_global_ids")

    @classmethod
    def box(cls, nx, ny, nz, p0=(0.0, 0.0, 0.0), p1=(1.0, 1.0, 1.0), zplanes=None):
        """Slab [zplanes[0], zplanes[1]) of BoxMesh(p0,p1,nx,ny,nz) generated on the device."""
        m = cls()
        if zplanes is None:
            zplanes = (0, nz + 1)
        a = L.f64(p0)
        b = L.f64(p1)
        L.check(L.load().fs_mesh_create_box(int(nx), int(ny), int(nz), L.p_f64(a), L.p_f64(b), int(zplanes[0]),
                                            int(zplanes[1]), C.byref(m.h)), "fs_mesh_create_box")
        return m

    @classmethod
    def renumbered(cls, coords, cells):
        """A tetrahedral mesh that arrives in FILE order, uploaded once and built on the device in the locality order of
        fs_mesh_locality_order (global vertex ids = the file's numbers).  Returns (mesh, vertex_order, cell_order):
        vertex_order[k] / cell_order[c] = file number of device vertex k / device cell c."""
        co, ce = L.f64(coords), L.i32(cells)
        if co.ndim != 2 or co.shape[1] != 3 or ce.ndim != 2 or ce.shape[1] != 4:
            raise BackendError("DeviceMesh.renumbered: coords [nv,3] and cells [nc,4]")
        m = cls()
        vo = np.empty(co.shape[0], dtype=np.int32)
        cord = np.empty(ce.shape[0], dtype=np.int32)
        L.check(L.load().fs_mesh_create_renumbered(co.shape[0], L.p_f64(co), ce.shape[0], L.p_i32(ce), L.p_i32(vo), L.p_i32(cord),
                                                   C.byref(m.h)), "fs_mesh_create_renumbered")
        return m, vo, cord

    def info(self):
        nv, nc, no = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.load().fs_mesh_info(self.h, C.byref(nv), C.byref(nc), C.byref(no)), "fs_mesh_info")
        return nv.value, nc.value, no.value

    def get(self, want_coords=True, want_cells=True, want_gids=True):
        nv, nc, _ = self.info()
        xyz = np.empty((nv, 3)) if want_coords else None
        cells = np.empty((nc, 4), dtype=np.int32) if want_cells else None
        gid = np.empty(nv, dtype=np.int64) if want_gids else None
        L.check(L.load().fs_mesh_get(self.h, L.p_f64(xyz), L.p_i32(cells), L.p_i64(gid)), "fs_mesh_get")
        return xyz, cells, gid


def locality_order(coords, cells):
    """(vertex_order, cell_order): the order in which to upload the vertices and cells of a mesh that arrives in file order
    (Morton curve of the coordinates, computed on the device; fs_mesh_locality_order).  vertex_order[k] = old id of new
    vertex k."""
    co, ce = L.f64(coords), L.i32(cells)
    vo = np.empty(co.shape[0], dtype=np.int32)
    cord = np.empty(ce.shape[0], dtype=np.int32)
    L.check(L.load().fs_mesh_locality_order(co.shape[1], co.shape[0], L.p_f64(co), ce.shape[0], L.p_i32(ce), ce.shape[1],
                                            L.p_i32(vo), L.p_i32(cord)), "fs_mesh_locality_order")
    return vo, cord


class DeviceSpace(_Handle):
    """CG1 (scalar or 3-vector) space + sparsity (FunctionSpace, SolverBase.py:260-275)."""
    _destroy = "fs_space_destroy"

    def __init__(self, mesh, ncomp=1, degree=1, coupled_pairs=None):
        """coupled_pairs [n,2]: extra node couplings of the sparsity pattern (interior-facet integrals)."""
        super().__init__()
        self.mesh = mesh
        self.ncomp = int(ncomp)
        if coupled_pairs is None:
            L.check(L.load().fs_space_create(mesh.h, 0, int(degree), int(ncomp), C.byref(self.h)), "fs_space_create")
        else:
            pairs = L.i32(coupled_pairs).reshape(-1, 2)
            L.check(L.load().fs_space_create_coupled(mesh.h, 0, int(degree), int(ncomp), pairs.shape[0], L.p_i32(pairs),
                                                     C.byref(self.h)), "fs_space_create_coupled")
        self.facet_coupled = coupled_pairs is not None
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.load().fs_space_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "fs_space_info")
        self.n_local, self.n_owned, self.nnz, self.sell_entries = a.value, b.value, c.value, d.value
        L.check(L.load().fs_space_format_info(self.h, C.byref(a), C.byref(b), C.byref(c)), "fs_space_format_info")
        self.n_slices, self.n_dia_slices, self.spmv_matrix_bytes = a.value, b.value, c.value
        self.degree = int(degree)

    def edges(self):
        """CG2: [n_edges,2] vertex pairs of the edge nodes (dof = n_vertices + row index)."""
        ne = C.c_int64(0)
        L.check(L.load().fs_space_get_edges(self.h, C.byref(ne), None), "fs_space_get_edges")
        out = np.empty((ne.value, 2), dtype=np.int32)
        if ne.value:
            L.check(L.load().fs_space_get_edges(self.h, C.byref(ne), L.p_i32(out)), "fs_space_get_edges")
        return out

    def set_halo(self, neighbors, send_lists, recv_counts, recv_lists=None):
        """neighbors: ranks; send_lists: per neighbour array of owned local dofs; recv_counts: ghosts per neighbour;
        recv_lists: per neighbour the local ghost dof of every received value (None: ghosts grouped by neighbour)."""
        if getattr(self, "_p2p", False):
            self.enable_p2p_halo(False)
        nb = L.i32(neighbors)
        sc = L.i64([len(s) for s in send_lists])
        si = L.i32(np.concatenate([np.asarray(s, dtype=np.int32) for s in send_lists]) if len(send_lists) else [])
        rc = L.i64(recv_counts)
        if recv_lists is None:
            L.check(L.load().fs_space_set_halo(self.h, len(nb), L.p_i32(nb), L.p_i64(sc), L.p_i32(si), L.p_i64(rc)),
                    "fs_space_set_halo")
        else:
            ri = L.i32(np.concatenate([np.asarray(r, dtype=np.int32) for r in recv_lists]) if len(recv_lists) else [])
            L.check(L.load().fs_space_set_halo_indexed(self.h, len(nb), L.p_i32(nb), L.p_i64(sc), L.p_i32(si), L.p_i64(rc),
                                                       L.p_i32(ri)), "fs_space_set_halo_indexed")
        # The peer-to-peer exchange (29 instead of 55 us per CG iteration at 1 M rows per rank with the rank its own neighbour,
        # profiles/*_p2p_self_halo_timings.txt) is OPT-IN until it has crossed a device boundary on hardware (round 5):
        # FS_HALO_P2P=auto tries it - the set-up ends with a self-test whose verdict all ranks agree on, and a node where the
        # mappings cannot be made falls back to RCCL on EVERY rank, here; FS_HALO_P2P=1: a failed set-up is an error instead of a
        # fall-back; unset / 0: RCCL send / recv.  bench.py tries every transport explicitly and times the fastest that reproduces
        # the RCCL field.
        mode = os.environ.get("FS_HALO_P2P", "0")
        if _comm_up and mode != "0" and comm_info()[0] > 1:
            try:
                self.enable_p2p_halo(True)
            except L.BackendError as e:
                if mode == "1":
                    raise
                import logging
                logging.getLogger("fenicssolver_amd").info("peer-to-peer halo exchange not available (%s): RCCL send / recv", str(e)[:160])

    def enable_p2p_halo(self, on=True):
        """Ghost refresh by direct stores into the neighbours' memory (hipIpc mappings, one node) instead of RCCL send / recv.
        COLLECTIVE: every rank of the communicator calls it for its space, in the same order (also to turn it off).
        Environment FS_HALO_P2P=auto / 1 turns it on for every halo plan set while a communicator of more than one rank is up."""
        self._p2p = False
        L.check(L.load().fs_space_enable_p2p_halo(self.h, 1 if on else 0), "fs_space_enable_p2p_halo")
        self._p2p = bool(on)


class DeviceVector(_Handle):
    _destroy = "fs_vector_destroy"

    def __init__(self, n, values=None):
        super().__init__()
        self.n = int(n)
        L.check(L.load().fs_vector_create(self.n, C.byref(self.h)), "fs_vector_create")
        if values is not None:
            self.set(values)

    def set(self, values):
        v = L.f64(values).ravel()
        if v.size != self.n:
            raise BackendError("DeviceVector.set: %d values for a vector of %d" % (v.size, self.n))
        L.check(L.load().fs_vector_set(self.h, L.p_f64(v), self.n), "fs_vector_set")

    def get(self, n=None):
        n = self.n if n is None else int(n)
        out = np.empty(n)
        L.check(L.load().fs_vector_get(self.h, L.p_f64(out), n), "fs_vector_get")
        return out

    def fill(self, a):
        L.check(L.load().fs_vector_fill(self.h, float(a)), "fs_vector_fill")

    def copy_from(self, src, n=None):
        n = min(self.n, src.n) if n is None else int(n)
        L.check(L.load().fs_vector_copy(self.h, src.h, n), "fs_vector_copy")

    def axpy(self, a, x):
        L.check(L.load().fs_vector_axpy(self.h, float(a), x.h), "fs_vector_axpy")

    def add_entries(self, idx, vals):
        i, v = L.i32(idx).ravel(), L.f64(vals).ravel()
        L.check(L.load().fs_vector_add_entries(self.h, i.size, L.p_i32(i), L.p_f64(v)), "fs_vector_add_entries")

    def assign_entries(self, dst_nodes, src_nodes, block=1):
        """self[dst*block + c] = self[src*block + c] (the slaves of a periodic constraint take their masters' values)."""
        d, sr = L.i32(dst_nodes).ravel(), L.i32(src_nodes).ravel()
        L.check(L.load().fs_vector_assign_entries(self.h, d.size, L.p_i32(d), L.p_i32(sr), int(block)), "fs_vector_assign_entries")

    def dot(self, y):
        r = C.c_double(0.0)
        L.check(L.load().fs_vector_dot(self.h, y.h, C.byref(r)), "fs_vector_dot")
        return r.value


def _coef(spec, keep):
    """spec: None | number | ('cell', array) | ('nodal', array) | ('tensor', 3x3) | ('cell_tensor', [n_cells,3,3])."""
    c = L.fs_coef()
    if spec is None:
        c.mode = L.FS_COEF_NONE
    elif isinstance(spec, tuple):
        kind, arr = spec
        if kind == "tensor":
            c.mode = L.FS_COEF_TENSOR
            t = L.f64(arr).reshape(9)
            for i in range(9):
                c.tensor[i] = t[i]
        elif kind == "cell_qp":              # [n_cells, 14]: the coefficient at the points of the degree-5 rule (CG2 spaces)
            a = L.f64(np.ascontiguousarray(arr).reshape(-1, 14))
            keep.append(a)
            c.mode = L.FS_COEF_CELL_QP
            c.data = L.p_f64(a)
        elif kind == "cell_tensor":          # [n_cells, 3, 3]
            a = L.f64(np.ascontiguousarray(arr).reshape(-1, 9))
            keep.append(a)
            c.mode = L.FS_COEF_CELL_TENSOR
            c.data = L.p_f64(a)
        else:
            a = L.f64(arr).ravel()
            keep.append(a)
            c.mode = L.FS_COEF_CELL if kind == "cell" else L.FS_COEF_NODAL
            c.data = L.p_f64(a)
    else:
        c.mode = L.FS_COEF_CONST
        c.value = float(spec)
    return c


def _bilinear_form(keep, stiffness=None, mass=None, lame=None, advection=None, advection_scale=1.0, supg_pe=0.0):
    f = L.fs_bilinear_form()
    f.stiffness = _coef(stiffness, keep)
    f.mass = _coef(mass, keep)
    if advection is not None:
        v = L.f64(advection)
        if v.size == 3:
            f.advection.mode = L.FS_COEF_CONST
            for i in range(3):
                f.advection.tensor[i] = float(v.ravel()[i])
        elif v.ndim == 3:           # [n_cells, d+1, 3]: one velocity per cell and test function (fem.row_velocities)
            v = L.f64(np.ascontiguousarray(v).reshape(-1, 3))
            keep.append(v)
            f.advection.mode = L.FS_COEF_CELL_ROW
            f.advection.data = L.p_f64(v)
        else:
            v = L.f64(v.reshape(-1, 3))
            keep.append(v)
            f.advection.mode = L.FS_COEF_CELL
            f.advection.data = L.p_f64(v)
        f.advection_scale = float(advection_scale)
        f.supg_pe = float(supg_pe)
    if lame is not None:
        f.lame_mu, f.lame_lambda = float(lame[0]), float(lame[1])
    return f


def apply_operator(space, x, y, stiffness=None, mass=None, advection=None, advection_scale=1.0, supg_pe=0.0, reps=0):
    """Matrix-free y = K(form) x on a scalar CG1 or CG2 space over tetrahedra (no matrix is formed, no Dirichlet rows; CG2: constant
    or per-cell scalar coefficients, no advection).
    reps > 1: also returns the mean milliseconds of reps products timed with HIP events."""
    keep = []
    f = _bilinear_form(keep, stiffness, mass, None, advection, advection_scale, supg_pe)
    ms = C.c_double(0.0)
    L.check(L.load().fs_operator_apply(space.h, C.byref(f), x.h, y.h, int(reps), C.byref(ms)), "fs_operator_apply")
    return ms.value if reps > 1 else None


class DeviceMatrix(_Handle):
    """SELL-64 matrix on a space's pattern (PETSc AIJ behind dolfin.assemble)."""
    _destroy = "fs_matrix_destroy"

    def __init__(self, space):
        super().__init__()
        self.space = space
        L.check(L.load().fs_matrix_create(space.h, C.byref(self.h)), "fs_matrix_create")

    def assemble(self, stiffness=None, mass=None, lame=None, advection=None, advection_scale=1.0, add=False, supg_pe=0.0):
        """advection: constant velocity (3 numbers) or per-cell array [n_cells,3]; supg_pe > 0: SUPG test function
        q + tau (v . grad q) on the advection and mass terms."""
        keep = []
        f = _bilinear_form(keep, stiffness, mass, lame, advection, advection_scale, supg_pe)
        L.check(L.load().fs_assemble_matrix(self.h, C.byref(f), 1 if add else 0), "fs_assemble_matrix")

    def add_facet_mass(self, tri, h):
        tri = L.i32(tri)
        tri = tri.reshape(-1, 3) if tri.ndim == 1 else tri       # [nf,3] triangles (3-D) or [nf,2] edges (2-D)
        h = L.f64(np.broadcast_to(h, (tri.shape[0],)))
        L.check(L.load().fs_assemble_facet_matrix(self.h, tri.shape[0], L.p_i32(tri), L.p_f64(h)),
                "fs_assemble_facet_matrix")

    def add_interior_penalty(self, facet_cells, coefficient):
        """+= coefficient * avg(h)^2 jump(grad u, n) jump(grad v, n) dS over the interior facets [nf,2] (cell pairs)."""
        fc = L.i32(facet_cells).reshape(-1, 2)
        L.check(L.load().fs_assemble_interior_penalty(self.h, fc.shape[0], L.p_i32(fc), float(coefficient)),
                "fs_assemble_interior_penalty")

    def axpy(self, a, X):
        L.check(L.load().fs_matrix_axpy(self.h, float(a), X.h), "fs_matrix_axpy")

    def tie_nodes(self, b, slaves, masters):
        """Fold the periodic constraint u[slaves] = u[masters] into the assembled system (and b, which may be None)."""
        sl, ma = L.i32(slaves).ravel(), L.i32(masters).ravel()
        L.check(L.load().fs_matrix_tie_nodes(self.h, b.h if b is not None else None, sl.size, L.p_i32(sl), L.p_i32(ma)),
                "fs_matrix_tie_nodes")

    def copy_from(self, src):
        """self = src (same space), on the library's stream."""
        L.check(L.load().fs_matrix_copy(self.h, src.h), "fs_matrix_copy")

    def zero(self):
        L.check(L.load().fs_matrix_zero(self.h), "fs_matrix_zero")

    def apply_dirichlet(self, b, dofs, vals, symmetric=True):
        dofs = L.i32(dofs).ravel()
        vals = L.f64(np.broadcast_to(vals, dofs.shape))
        L.check(L.load().fs_apply_dirichlet(self.h, b.h if b is not None else None, dofs.size, L.p_i32(dofs),
                                            L.p_f64(vals), 1 if symmetric else 0), "fs_apply_dirichlet")

    def to_csr(self):
        """(rowptr, colidx, vals) sorted-column CSR copy on the host."""
        nr, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.load().fs_matrix_info(self.h, C.byref(nr), C.byref(ncol), C.byref(nnz)), "fs_matrix_info")
        rp = np.empty(nr.value + 1, dtype=np.int32)
        ci = np.empty(nnz.value, dtype=np.int32)
        va = np.empty(nnz.value)
        L.check(L.load().fs_matrix_get_csr(self.h, L.p_i32(rp), L.p_i32(ci), L.p_f64(va)), "fs_matrix_get_csr")
        return rp, ci, va, (nr.value, ncol.value)

    def spmv(self, x, y):
        L.check(L.load().fs_spmv(self.h, x.h, y.h), "fs_spmv")

    def spmv_dictionary(self, x, y):
        """y = A x through the row-dictionary product where the rows repeat; returns the number of distinct rows used (0: streaming)."""
        nc = C.c_int(0)
        L.check(L.load().fs_spmv_dictionary(self.h, x.h, y.h, C.byref(nc)), "fs_spmv_dictionary")
        return nc.value

    def spmv_benchmark(self, x, y, reps=20):
        ms = C.c_double(0.0)
        L.check(L.load().fs_spmv_benchmark(self.h, x.h, y.h, int(reps), C.byref(ms)), "fs_spmv_benchmark")
        return ms.value


def _velocity_coef(v, keep):
    """Constant velocity (3 numbers) or per-cell array [n_cells,3] -> fs_coef."""
    c = L.fs_coef()
    v = L.f64(v)
    if v.size == 3:
        c.mode = L.FS_COEF_CONST
        for i in range(3):
            c.tensor[i] = float(v.ravel()[i])
    else:
        v = L.f64(v.reshape(-1, 3))
        keep.append(v)
        c.mode = L.FS_COEF_CELL
        c.data = L.p_f64(v)
    return c


def assemble_facet_supg(space, A, b, facet_cell, facet_opposite, velocity, pe, g=None, h=None):
    """SUPG part of ds(i) terms: b_a += g area w_a, A_ab += h (area/3) w_a (b on the facet), w_a = tau (v . grad phi_a)."""
    fc = np.ascontiguousarray(facet_cell, dtype=np.int32)
    fo_ = np.ascontiguousarray(facet_opposite, dtype=np.int32)
    keep = []
    vel = _velocity_coef(velocity, keep)
    gg = None if g is None else np.ascontiguousarray(np.broadcast_to(np.asarray(g, dtype=np.float64), fc.shape))
    hh = None if h is None else np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), fc.shape))
    L.check(L.load().fs_assemble_facet_supg(space.h, A.h if A is not None else None, b.h if b is not None else None, len(fc),
                                            L.p_i32(fc), L.p_i32(fo_), L.p_f64(gg), L.p_f64(hh), C.byref(vel), float(pe)),
            "fs_assemble_facet_supg")


def assemble_vector(space, b, source=None, vector_value=None, div_coef=None, add=False, supg=None):
    """b (+)= int source q dx [+ int f.v dx + int div_coef div v dx on vector spaces]; supg = (velocity, Pe) adds
    int source tau (v . grad q) dx."""
    keep = []
    f = L.fs_linear_form()
    f.source = _coef(source, keep)
    f.div_coef = _coef(div_coef, keep)
    if supg is not None:
        f.supg_velocity = _velocity_coef(supg[0], keep)
        f.supg_pe = float(supg[1])
    if vector_value is not None:
        for i, x in enumerate(vector_value):         # 3 components (2 on triangular meshes)
            f.vector_value[i] = float(x)
    L.check(L.load().fs_assemble_vector(space.h, C.byref(f), b.h, 1 if add else 0), "fs_assemble_vector")


def assemble_facet_vector(space, b, tri, g):
    """b_a += int g phi_a ds over the facets tri[nf,3]; g scalar, [ncomp], [nf] or [nf,ncomp]."""
    tri = L.i32(tri)
    tri = tri.reshape(-1, 3) if tri.ndim == 1 else tri           # [nf,3] triangles (3-D) or [nf,2] edges (2-D)
    nf, nc = tri.shape[0], space.ncomp
    g = np.asarray(g, dtype=np.float64)
    if g.ndim == 0:
        g = np.full((nf, nc), float(g))
    elif g.ndim == 1 and nc > 1 and g.size == nc:
        g = np.broadcast_to(g[None, :], (nf, nc))
    elif g.ndim == 1:
        g = g.reshape(nf, 1) if nc == 1 else np.broadcast_to(g[:, None], (nf, nc))
    g = L.f64(g)
    if g.shape != (nf, nc):
        raise BackendError("assemble_facet_vector: g has shape %s, expected (%d,%d)" % (g.shape, nf, nc))
    L.check(L.load().fs_assemble_facet_vector(space.h, nf, L.p_i32(tri), L.p_f64(g), b.h),
            "fs_assemble_facet_vector")


def _same_device_mesh(field_space, p1_space, what):
    """A projection reads a field of one space cell by cell and loads another: both on ONE device mesh.  On several ranks a periodic
    space cuts its own part (its masters are extra ghosts), and an unconstrained space made BEFORE it on the same host mesh sits on
    another one (fem.FunctionSpace._make_parallel_device shares the periodic part only with spaces made after it)."""
    if field_space.mesh is not p1_space.mesh:
        raise BackendError("%s: the two spaces live on different device meshes (parts) of the same host mesh - on several ranks create "
                           "the periodic function space BEFORE the unconstrained spaces that share its mesh" % what)


def assemble_von_mises(disp_space, u, mu, lmbda, p1_space, b):
    """b_a = int sqrt(3/2 s:s) phi_a dx, s the deviator of sigma(u), on the scalar CG1 space of the same mesh."""
    _same_device_mesh(disp_space, p1_space, "assemble_von_mises")
    L.check(L.load().fs_assemble_von_mises(disp_space.h, u.h, float(mu), float(lmbda), p1_space.h, b.h), "fs_assemble_von_mises")


def assemble_viscous_stress(th_space, w, nu, p1_space, b, viscosity_law=None):
    """b[vertex*9 + 3i + j] = int (nu (grad u + grad u^T) - p I)_ij phi_vertex dx for a Taylor-Hood iterate w.
    viscosity_law = (p_ref, exponent): nu (p / p_ref)^exponent."""
    pref, ex = (0.0, 0.0) if viscosity_law is None else (float(viscosity_law[0]), float(viscosity_law[1]))
    _same_device_mesh(th_space, p1_space, "assemble_viscous_stress")
    L.check(L.load().fs_assemble_viscous_stress_nn(th_space.h, w.h, float(nu), p1_space.h, b.h, pref, ex), "fs_assemble_viscous_stress")


def set_dirichlet_values(b, dofs, vals):
    dofs = L.i32(dofs).ravel()
    vals = L.f64(np.broadcast_to(vals, dofs.shape))
    L.check(L.load().fs_apply_dirichlet(None, b.h, dofs.size, L.p_i32(dofs), L.p_f64(vals), 0), "fs_apply_dirichlet")


def krylov_solve(A, b, x, rtol=1e-8, atol=0.0, max_iter=10000, precond="jacobi", batch=0, nonzero_guess=False,
                 method="cg", diagonal_scale=True, norm="unpreconditioned", pipelined=None):
    """CG (SPD) or BiCGStab (non-symmetric) on the device.  Returns a stats dict.
    pipelined: True selects the Ghysels-Vanroose recurrence (the all-reduce of the sums hidden under the product, 112 instead
    of 72 B/DOF of vector traffic); None / False (default) the single-reduction recurrence - in every measurement so far the
    pipelined one was slower, so it is opt-in."""
    o = L.fs_krylov_opts()
    o.method = {"cg": L.FS_KSP_CG, "bicgstab": L.FS_KSP_BICGSTAB}[method]
    o.precond = {"none": L.FS_PC_NONE, None: L.FS_PC_NONE, "jacobi": L.FS_PC_JACOBI}[precond]
    o.rtol, o.atol, o.max_iter, o.batch = float(rtol), float(atol), int(max_iter), int(batch)
    o.nonzero_guess = 1 if nonzero_guess else 0
    o.diagonal_scale = 1 if diagonal_scale else 0
    o.pipelined = -1 if pipelined is None else (1 if pipelined else 0)
    if o.pipelined < 0 and not (method == "cg" and o.precond == L.FS_PC_JACOBI and diagonal_scale):
        o.pipelined = 0
    o.norm_type = {"unpreconditioned": L.FS_NORM_UNPRECONDITIONED, "preconditioned": L.FS_NORM_PRECONDITIONED}[norm]
    st = L.fs_krylov_stats()

    def solve():
        L.check(L.load().fs_krylov_solve(A.h, b.h, x.h, C.byref(o), C.byref(st)), "fs_krylov_solve")
    _with_p2p_fallback(A.space, solve, x if nonzero_guess else None)
    return {k: getattr(st, k) for k, _ in L.fs_krylov_stats._fields_}


def _with_p2p_fallback(spaces, solve, x_guess=None):
    """Run a solve over space(s) whose ghost refresh is the peer-to-peer exchange.  A wait of that transport that times out (peer
    process gone, stores over the mappings not visible on this system) fails the solve with FS_ERR_P2P_TIMEOUT on the rank that
    saw it.  The ranks then compare notes over RCCL proper (never over the transport in doubt) on THAT code only, and if any of
    them saw it ALL of them turn the exchange of the space(s) off and solve again over RCCL send / recv - also a rank whose own
    solve failed with another code (stale ghosts produce a breakdown as easily as a time-out; a rank that raised here would leave
    the others alone in the collectives of the second solve).  When NO rank timed out, any failure - a Krylov breakdown, a zero
    diagonal, a bad argument - is not the transport's: it is raised as it is (after the agreement, so that the ranks stay
    paired), nothing is turned off and nothing is repeated.  One small RCCL all-gather per solve, only while the
    exchange is on: a rank can finish its last wait while a neighbour times out on its own, so success on this rank says nothing
    about the others."""
    if not isinstance(spaces, (list, tuple)):
        spaces = [spaces]
    live = []
    for sp in spaces:
        if sp is not None and getattr(sp, "_p2p", False) and not any(sp is q for q in live):
            live.append(sp)
    if not live or not _comm_up:
        return solve()
    keep = None
    if x_guess is not None:
        keep = DeviceVector(x_guess.n)
        keep.copy_from(x_guess)
    err = None
    try:
        solve()
    except L.BackendError as e:
        err = e
    timed_out = err is not None and err.rc == L.FS_ERR_P2P_TIMEOUT
    if float(np.sum(comm_allgather([1.0 if timed_out else 0.0], 1))) == 0.0:
        if err is not None:
            raise err
        return None
    import logging
    logging.getLogger("fenicssolver_amd").warning("peer-to-peer halo exchange timed out during a solve (%s): turned off, solving again over RCCL",
                                                  "on this rank" if timed_out else "on another rank")
    for sp in live:
        sp.enable_p2p_halo(False)
    # Once ANY rank timed out, every local error of this solve is taken as a consequence of it (a rank whose neighbour's stores never
    # arrived consumes stale ghosts or garbage sums and may fail with a breakdown instead of the time-out code): ALL ranks solve
    # again over RCCL, so that nobody is left alone in a collective.  A genuine non-transport failure shows again in the
    # second solve, on every rank alike, and is raised there.
    if err is not None and not timed_out:
        logging.getLogger("fenicssolver_amd").warning("this rank's solve failed with %r while another rank's exchange timed out: "
                                                      "treated as a consequence, solving again", err)
    if keep is not None:
        x_guess.copy_from(keep)
    return solve()


class AMG(_Handle):
    """Smoothed-aggregation hierarchy of an assembled SPD matrix (PETSc GAMG behind
    PETScPreconditioner("petsc_amg"), SolverBase.py:643-672).  nullspace: [nb, n_dofs] near-null-space
    vectors, "rigid_body" = the six rigid-body modes built on the device from the node coordinates (3-vector CG1
    / CG2 spaces: vertices and edge mid-points), or None = constants per component."""
    _destroy = "fs_amg_destroy"

    def __init__(self, A, nullspace=None, strength_threshold=0.0, max_levels=0, coarse_size=0, smoother_steps=0,
                 eig_steps=0):
        super().__init__()
        self.A = A                      # keeps the matrix alive
        o = L.fs_amg_opts()
        o.strength_threshold, o.max_levels, o.coarse_size = float(strength_threshold), int(max_levels), int(coarse_size)
        o.smoother_steps, o.eig_steps = int(smoother_steps), int(eig_steps)
        ns, nb = None, 0
        if isinstance(nullspace, str):
            if nullspace != "rigid_body":
                raise ValueError("nullspace must be an array, None or 'rigid_body'")
            o.rigid_body_modes = 1
        elif nullspace is not None:
            ns = np.ascontiguousarray(np.asarray(nullspace, dtype=np.float64).reshape(len(nullspace), -1))
            nb = ns.shape[0]
            if ns.shape[1] != A.space.n_owned:
                raise ValueError("near-null space has %d entries per vector, the matrix has %d rows" % (ns.shape[1], A.space.n_owned))
        L.check(L.load().fs_amg_setup(A.h, nb, L.p_f64(ns), C.byref(o), C.byref(self.h)), "fs_amg_setup")

    def attach_distributed_fine(self, A_local, owned_global_nodes):
        """The hierarchy was built on the undecomposed operator: from now on level 0 works on A_local, this rank's rows of the
        decomposed operator (owned_global_nodes[i] = undecomposed node of local owned node i); levels >= 1 stay replicated."""
        g = L.i32(owned_global_nodes)
        L.check(L.load().fs_amg_attach_distributed_fine(self.h, A_local.h, len(g), L.p_i32(g)), "fs_amg_attach_distributed_fine")
        self.A_local = A_local          # keeps the decomposed operator alive

    def info(self):
        nl, oc, gc, ms = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        L.check(L.load().fs_amg_info(self.h, C.byref(nl), C.byref(oc), C.byref(gc), C.byref(ms)), "fs_amg_info")
        return {"levels": nl.value, "operator_complexity": oc.value, "grid_complexity": gc.value, "setup_ms": ms.value}

    def level_info(self, level):
        nn, pn, bs, pc, nz, lm = C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_int64(), C.c_double()
        L.check(L.load().fs_amg_level_info(self.h, int(level), C.byref(nn), C.byref(bs), C.byref(nz), C.byref(pn),
                                           C.byref(pc), C.byref(lm)), "fs_amg_level_info")
        return {"n_nodes": nn.value, "block_size": bs.value, "nnz_blocks": nz.value, "p_nnz_blocks": pn.value,
                "p_block_cols": pc.value, "lambda_max": lm.value}

    def level_matrix(self, level, which="A"):
        """scipy BSR copy of a level operator ('A') or of the prolongator from level+1 ('P')."""
        import scipy.sparse as sp
        li = self.level_info(level)
        if which == "A":
            nnz, br, bc, nrows = li["nnz_blocks"], li["block_size"], li["block_size"], li["n_nodes"]
            ncols = nrows
        else:
            nnz, br, bc, nrows = li["p_nnz_blocks"], li["block_size"], li["p_block_cols"], li["n_nodes"]
            ncols = self.level_info(level + 1)["n_nodes"]
        rp = np.empty(nrows + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        va = np.empty(nnz * br * bc)
        L.check(L.load().fs_amg_level_get(self.h, int(level), 0 if which == "A" else 1, L.p_i32(rp), L.p_i32(ci),
                                          L.p_f64(va)), "fs_amg_level_get")
        return sp.bsr_matrix((va.reshape(nnz, br, bc), ci, rp), shape=(nrows * br, ncols * bc)).tocsr()

    def level_nullspace(self, level, nb):
        li = self.level_info(level)
        out = np.empty(li["n_nodes"] * li["block_size"] * nb)
        L.check(L.load().fs_amg_level_get(self.h, int(level), 2, None, None, L.p_f64(out)), "fs_amg_level_get")
        return out.reshape(-1, nb)

    def apply(self, r, z):
        L.check(L.load().fs_amg_apply(self.h, r.h, z.h), "fs_amg_apply")

    def solve(self, b, x, rtol=1e-8, atol=0.0, max_iter=500, nonzero_guess=False, norm="unpreconditioned"):
        o = L.fs_krylov_opts()
        o.rtol, o.atol, o.max_iter = float(rtol), float(atol), int(max_iter)
        o.nonzero_guess = 1 if nonzero_guess else 0
        o.norm_type = {"unpreconditioned": L.FS_NORM_UNPRECONDITIONED, "preconditioned": L.FS_NORM_PRECONDITIONED}[norm]
        st = L.fs_krylov_stats()

        def solve():
            L.check(L.load().fs_amg_solve(self.h, b.h, x.h, C.byref(o), C.byref(st)), "fs_amg_solve")
        fine = getattr(self, "A_local", None) or self.A
        _with_p2p_fallback(fine.space, solve, x if nonzero_guess else None)
        return {k: getattr(st, k) for k, _ in L.fs_krylov_stats._fields_}


def assemble_navier_stokes(J, g, w0, w_prev=None, nu=1.0, rho=1.0, inv_dt=0.0, body_force=(0.0, 0.0, 0.0),
                           convection=True, newton=True, mesh_velocity=(0.0, 0.0, 0.0), g2=None, viscosity_law=None):
    """Linearised Taylor-Hood system at the state w0 (J w_new = g), J on a DeviceSpace(mesh, ncomp=4, degree=2).
    g2 = (mode, kappa1): the G2 streamline term (mode 1: Re <= 1, 2: convection dominated).
    viscosity_law = (p_ref, exponent): nu (p0 / p_ref)^exponent with the pressure of w0 (the reference's non-Newtonian law)."""
    f = L.fs_ns_form()
    f.kinematic_viscosity, f.density, f.inv_dt = float(nu), float(rho), float(inv_dt)
    for i in range(3):
        f.body_force[i] = float(body_force[i])
        f.mesh_velocity[i] = float(mesh_velocity[i])
    f.convection, f.newton = (1 if convection else 0), (1 if newton else 0)
    if g2 is not None:
        f.g2_mode, f.g2_kappa1 = int(g2[0]), float(g2[1])
    if viscosity_law is not None:
        f.viscosity_pressure_ref, f.viscosity_pressure_exponent = float(viscosity_law[0]), float(viscosity_law[1])
    L.check(L.load().fs_assemble_navier_stokes(J.h, g.h, w0.h if w0 is not None else None,
                                               w_prev.h if w_prev is not None else None, C.byref(f)),
            "fs_assemble_navier_stokes")


def set_viscosity_law(th_space, law=None, temperature=None):
    """Attach a non-Newtonian law to a Taylor-Hood space (fs_space_set_viscosity_law): law = None detaches; (p_ref, exponent):
    nu (p / p_ref)^exponent; ('pT', p_ref, c_p, T_ref, c_T) with temperature = DeviceVector of the CG1 vertex values (local
    numbering; read at every assembly, so the caller keeps it alive and current): nu (1 + c_p p/p_ref)(1 - c_T T/T_ref)."""
    f = L.fs_viscosity_law()
    if law is not None and law[0] == 'pT':
        if temperature is None:
            raise BackendError("set_viscosity_law: the 'pT' law needs the temperature vector")
        f.kind, f.pressure_ref, f.pressure_coef, f.temperature_ref, f.temperature_coef = 2, float(law[1]), float(law[2]), float(law[3]), float(law[4])
        f.temperature = temperature.h.value
        th_space._law_temperature = temperature
    elif law is not None:
        f.kind, f.pressure_ref, f.pressure_exponent = 1, float(law[0]), float(law[1])
    L.check(L.load().fs_space_set_viscosity_law(th_space.h, C.byref(f) if law is not None else None), "fs_space_set_viscosity_law")
    if law is None or law[0] != 'pT':
        th_space._law_temperature = None


def assemble_ns_pressure_boundary(J, g, facet_cell, facet_opposite, nu, facet_value=None, viscosity_law=None, w0=None):
    """J, g += p_b n.v ds - nu ((grad u + grad u^T) n).v ds on the listed boundary facets (value None: traction term only;
    a number, one value per facet, or [n_facets, 3] values at the facet's vertices in the cell's local order).
    viscosity_law = (p_ref, exponent) with the state w0: nu (p0 / p_ref)^exponent."""
    fc = np.ascontiguousarray(facet_cell, dtype=np.int32)
    fo_ = np.ascontiguousarray(facet_opposite, dtype=np.int32)
    fv, per = None, 1
    if facet_value is not None:
        a = np.asarray(facet_value, dtype=np.float64)
        if a.ndim == 2 and a.shape[0] == len(fc) and a.shape[1] in (2, 3):       # values at the vertices of every facet (edge: 2)
            fv, per = np.ascontiguousarray(a), a.shape[1]
        else:
            fv = np.ascontiguousarray(np.broadcast_to(a, fc.shape))
    pref, ex = (0.0, 0.0) if viscosity_law is None else (float(viscosity_law[0]), float(viscosity_law[1]))
    L.check(L.load().fs_assemble_ns_pressure_boundary_nn(J.h, g.h, len(fc), L.p_i32(fc), L.p_i32(fo_), L.p_f64(fv), float(nu),
                                                         w0.h if w0 is not None else None, pref, ex, per),
            "fs_assemble_ns_pressure_boundary")


def saddle_solve(J, Kp, Mp, b, x, nu, rho=1.0, inv_dt=0.0, rtol=1e-8, atol=0.0, max_iter=0, restart=0,
                 velocity_sweeps=0, inner_rtol=0.0, nonzero_guess=False, Kp_amg=None):
    """FGMRES with the block-triangular Cahouet-Chabard preconditioner; Kp may be None for steady problems."""
    o = L.fs_saddle_opts()
    o.rtol, o.atol, o.max_iter, o.restart = float(rtol), float(atol), int(max_iter), int(restart)
    o.kinematic_viscosity, o.density, o.inv_dt = float(nu), float(rho), float(inv_dt)
    o.velocity_sweeps, o.inner_rtol, o.nonzero_guess = int(velocity_sweeps), float(inner_rtol), 1 if nonzero_guess else 0
    st = L.fs_krylov_stats()

    def solve():
        L.check(L.load().fs_saddle_solve(J.h, Kp.h if Kp is not None else None, Kp_amg.h if Kp_amg is not None else None,
                                         Mp.h, b.h, x.h, C.byref(o), C.byref(st)),
                "fs_saddle_solve")
    _with_p2p_fallback([M.space for M in (J, Kp, Mp) if M is not None], solve, x if nonzero_guess else None)
    return {k: getattr(st, k) for k, _ in L.fs_krylov_stats._fields_}


def krylov_history():
    n = C.c_int(0)
    L.load().fs_krylov_history(None, 0, C.byref(n))
    out = np.empty(max(n.value, 1))
    L.load().fs_krylov_history(L.p_f64(out), n.value, C.byref(n))
    return out[: n.value]


# ---- multi-GPU ------------------------------------------------------------------------------
def comm_unique_id():
    buf = C.create_string_buffer(L.FS_UNIQUE_ID_BYTES)
    L.check(L.load().fs_comm_get_unique_id(buf), "fs_comm_get_unique_id")
    return bytes(buf.raw)


_comm_up = False


def comm_init(n_ranks, rank, uid):
    global _comm_up
    L.check(L.load().fs_comm_init(int(n_ranks), int(rank), C.c_char_p(uid)), "fs_comm_init")
    _comm_up = True


def comm_info():
    """(n_ranks, rank) of the communicator that is up ((1, 0) without one)."""
    n, r = C.c_int(1), C.c_int(0)
    L.check(L.load().fs_comm_info(C.byref(n), C.byref(r)), "fs_comm_info")
    return n.value, r.value


def comm_finalize():
    global _comm_up
    L.check(L.load().fs_comm_finalize(), "fs_comm_finalize")
    _comm_up = False


def comm_allreduce_sum(values):
    v = L.f64(values).ravel().copy()
    L.check(L.load().fs_comm_allreduce_sum(L.p_f64(v), v.size), "fs_comm_allreduce_sum")
    return v


def comm_allgather(values, n_max):
    """[n_ranks, n_max]: every rank's values (padded to n_max), one ncclAllGather."""
    v = L.f64(values).ravel()
    n_ranks = comm_info()[0]
    out = np.empty((n_ranks, int(n_max)))
    L.check(L.load().fs_comm_allgather(L.p_f64(v), v.size, int(n_max), L.p_f64(out)), "fs_comm_allgather")
    return out


def halo_exchange(space, v):
    L.check(L.load().fs_halo_exchange(space.h, v.h), "fs_halo_exchange")


def comm_benchmark(space, reps=200):
    """(allreduce_ms, halo_ms): mean latency of the 3-double all-reduce and of this space's ghost refresh, as a CG
    iteration issues them.  Collective (every rank calls it); zeros on one rank."""
    a, h = C.c_double(), C.c_double()
    L.check(L.load().fs_comm_benchmark(space.h, int(reps), C.byref(a), C.byref(h)), "fs_comm_benchmark")
    return a.value, h.value


def last_product_kind():
    """Kernel family of the last product this process launched (fs_last_product_kind): 0 streaming, 1 row-dictionary work items,
    2 lattice tiles (CG2 box), 3 marching windows (P1 box), 4 block-row dictionary, 5 marching windows (CG2 box in lattice order)."""
    return int(L.load().fs_last_product_kind())


def set_option(name, value):
    L.check(L.load().fs_set_option(name.encode(), float(value)), "fs_set_option")
