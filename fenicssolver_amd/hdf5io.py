"""HDF5 datasets through the HDF5 C library (libhdf5 via ctypes; h5py is not part of this environment).

What the reference reads from HDF5 (SolverBase.py:203-221: ``HDF5File(comm, file, 'r').read(mesh, '/mesh', False)``,
``/subdomains``, ``/boundaries``) and what an XDMF file with ``Format="HDF"`` DataItems points to are plain numeric
datasets; this module opens a file, lists / reads datasets as numpy arrays and - for the tests and for ``save`` users -
writes them.  The layout DOLFIN's HDF5File uses for a mesh and for mesh functions is restated in read_dolfin_mesh /
write_dolfin_mesh:

    /mesh/coordinates [nv, gdim] float64        /mesh/topology [nc, nvc] integer (attribute "celltype")
    /<name>/topology  [ne, nve]  integer (the vertices of every marked entity)   /<name>/values [ne] integer

The library is looked up as libhdf5.so in the loader path and in the usual prefixes (/opt/conda/lib in this image);
when none is found every call raises SolverError naming the file it could not read.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

from .fem import SolverError

_LIB = None
_hid = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0


def _candidates():
    env = os.environ.get("FS_HDF5_LIBRARY")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for prefix in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib64", "/usr/local/lib"):
        for f in sorted(glob.glob(os.path.join(prefix, "libhdf5.so*")), key=len):
            yield f
        for f in sorted(glob.glob(os.path.join(prefix, "libhdf5_serial.so*")), key=len):
            yield f


def library():
    """The loaded libhdf5 (ctypes.CDLL) with argument types set, or SolverError."""
    global _LIB
    if _LIB is not None:
        return _LIB
    lib, tried = None, []
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            break
        except OSError:
            tried.append(cand)
    if lib is None:
        raise SolverError("no HDF5 library (libhdf5.so) could be loaded{}; set FS_HDF5_LIBRARY, or convert the file to DOLFIN XML "
                          "/ ASCII XDMF".format(" (tried " + ", ".join(tried) + ")" if tried else ""))
    sig = {
        "H5open": (C.c_int, []),
        "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Oclose": (C.c_int, [_hid]),
        "H5Iget_type": (C.c_int, [_hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dclose": (C.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Sclose": (C.c_int, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]),
        "H5Tget_size": (C.c_size_t, [_hid]),
        "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tclose": (C.c_int, [_hid]),
        "H5Pcreate": (_hid, [_hid]),
        "H5Pset_create_intermediate_group": (C.c_int, [_hid, C.c_uint]),
        "H5Pclose": (C.c_int, [_hid]),
        "H5Gget_num_objs": (C.c_int, [_hid, C.POINTER(C.c_uint64)]),
        "H5Gget_objname_by_idx": (C.c_ssize_t, [_hid, C.c_uint64, C.c_char_p, C.c_size_t]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Gclose": (C.c_int, [_hid]),
    }
    optional = ("H5Gget_num_objs", "H5Gget_objname_by_idx")     # deprecated API: absent from a libhdf5 built without it
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if name in optional:
                continue
            raise SolverError("libhdf5 lacks {}: cannot read HDF5 / HDF5-backed XDMF meshes".format(name))
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise SolverError("H5open() failed")
    lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes, not printed stacks
    _LIB = lib
    return lib


def _native(lib, name):
    return _hid.in_dll(lib, name).value


def _memory_type(lib, cls, size, signed):
    """(HDF5 native memory type id, numpy dtype) for a dataset's element class."""
    if cls == H5T_FLOAT:
        return (_native(lib, "H5T_NATIVE_DOUBLE_g"), np.float64)            # float32 files are converted by the library
    if cls == H5T_INTEGER:
        if signed:
            return (_native(lib, "H5T_NATIVE_INT64_g"), np.int64)
        return (_native(lib, "H5T_NATIVE_UINT64_g"), np.uint64)
    raise SolverError("HDF5 dataset of type class {} (not integer / float) is not supported".format(cls))


class H5File:
    """``with H5File(path) as f: a = f.read('/mesh/coordinates')``; mode 'r' or 'w' (truncate)."""

    def __init__(self, path, mode="r"):
        self.lib = library()
        self.path = path
        p = os.fsencode(path)
        if mode == "r":
            if not os.path.exists(path):
                raise SolverError("{}: no such file".format(path))
            self.id = self.lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self.id = self.lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise SolverError("H5File mode must be 'r' or 'w'")
        if self.id < 0:
            raise SolverError("{}: not an HDF5 file (or it cannot be opened {})".format(path, "for reading" if mode == "r" else "for writing"))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if getattr(self, "id", -1) >= 0:
            self.lib.H5Fclose(self.id)
            self.id = -1

    def has(self, name):
        """True if every component of the path exists (group or dataset)."""
        parts = [p for p in name.split("/") if p]
        cur = ""
        for p in parts:
            cur += "/" + p
            if self.lib.H5Lexists(self.id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def keys(self, group="/"):
        g = self.lib.H5Gopen2(self.id, group.encode(), H5P_DEFAULT)
        if g < 0:
            raise SolverError("{}: no group '{}'".format(self.path, group))
        n = C.c_uint64(0)
        if not hasattr(self.lib, "H5Gget_num_objs") or not hasattr(self.lib, "H5Gget_objname_by_idx"):
            self.lib.H5Gclose(g)
            raise SolverError("{}: this libhdf5 was built without the group-listing calls (H5Gget_num_objs)".format(self.path))
        self.lib.H5Gget_num_objs(g, C.byref(n))
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(1024)
            self.lib.H5Gget_objname_by_idx(g, i, buf, 1024)
            out.append(buf.value.decode())
        self.lib.H5Gclose(g)
        return out

    def read(self, name):
        """The whole dataset as a numpy array (float64 for floating point data, int64 / uint64 for integers)."""
        lib = self.lib
        d = lib.H5Dopen2(self.id, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise SolverError("{}: no dataset '{}'".format(self.path, name))
        try:
            sp, tp = lib.H5Dget_space(d), lib.H5Dget_type(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            mem, dtype = _memory_type(lib, lib.H5Tget_class(tp), lib.H5Tget_size(tp), lib.H5Tget_sign(tp) != H5T_SGN_NONE)
            out = np.empty(shape, dtype=dtype)
            rc = 0
            if out.size:
                rc = lib.H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
            lib.H5Tclose(tp)
            lib.H5Sclose(sp)
            if rc < 0:
                raise SolverError("{}: reading dataset '{}' failed".format(self.path, name))
            return out
        finally:
            lib.H5Dclose(d)

    def write(self, name, array):
        """Create dataset `name` (intermediate groups included) holding `array` (float64 or int64 on disk)."""
        lib = self.lib
        a = np.ascontiguousarray(array)
        if a.dtype.kind == "f":
            a, ftype = a.astype(np.float64), _native(lib, "H5T_NATIVE_DOUBLE_g")
        elif a.dtype.kind in "iub":
            a, ftype = a.astype(np.int64), _native(lib, "H5T_NATIVE_INT64_g")
        else:
            raise SolverError("H5File.write: arrays of dtype {} are not supported".format(a.dtype))
        dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
        sp = lib.H5Screate_simple(a.ndim, dims, None)
        lcpl = lib.H5Pcreate(_native(lib, "H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        d = lib.H5Dcreate2(self.id, name.encode(), ftype, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            raise SolverError("{}: cannot create dataset '{}'".format(self.path, name))
        rc = lib.H5Dwrite(d, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) if a.size else 0
        lib.H5Dclose(d)
        lib.H5Pclose(lcpl)
        lib.H5Sclose(sp)
        if rc < 0:
            raise SolverError("{}: writing dataset '{}' failed".format(self.path, name))


def read_dataset(spec, relative_to=None):
    """``file.h5:/path/to/dataset`` (the text of an XDMF DataItem with Format="HDF")."""
    if ":" not in spec:
        raise SolverError("HDF5 reference '{}' is not of the form file.h5:/dataset".format(spec))
    fname, dset = spec.strip().rsplit(":", 1)
    if relative_to is not None and not os.path.isabs(fname):
        fname = os.path.join(relative_to, fname)
    with H5File(fname) as f:
        return f.read(dset)


def _entity_values(f, group, entities, what):
    """Values of a DOLFIN mesh function stored as (topology = vertices of every entity, values): mapped onto `entities`
    ([n, k] vertex tuples of the mesh, any vertex order) by their sorted vertex tuples; entities the file does not list get 0."""
    topo = f.read(group + "/topology").astype(np.int64)
    vals = f.read(group + "/values").astype(np.int64).ravel()
    if topo.ndim != 2 or topo.shape[0] != len(vals) or topo.shape[1] != entities.shape[1]:
        raise SolverError("{}: {}/topology has shape {} for {} values; expected [n, {}] ({})".format(
            f.path, group, topo.shape, len(vals), entities.shape[1], what))
    ne = len(entities)
    both = np.concatenate([np.sort(entities.astype(np.int64), axis=1), np.sort(topo, axis=1)])
    _, inv = np.unique(both, axis=0, return_inverse=True)
    inv = np.asarray(inv).ravel()
    entity_of = np.full(int(inv.max()) + 1, -1, dtype=np.int64)
    entity_of[inv[:ne]] = np.arange(ne)
    where = entity_of[inv[ne:]]
    if np.any(where < 0):
        raise SolverError("{}: {} of the {} in {} are not {} of the mesh".format(f.path, int(np.count_nonzero(where < 0)), what, group, what))
    out = np.zeros(ne, dtype=np.int64)
    out[where] = vals
    return out


def read_dolfin_mesh(path):
    """(coords, cells, facet_values_fn, cell_values_fn): the mesh of a DOLFIN HDF5 file (/mesh) and readers for its
    optional /boundaries and /subdomains mesh functions (each takes the mesh's entity-vertex table and returns values, or is
    None when the file has no such group).  Follows SolverBase._read_hdf5_mesh (:203-221)."""
    f = H5File(path)
    try:
        if not f.has("/mesh/coordinates") or not f.has("/mesh/topology"):
            raise SolverError("{}: no /mesh/coordinates + /mesh/topology (DOLFIN HDF5File layout)".format(path))
        coords = f.read("/mesh/coordinates").astype(np.float64)
        cells = f.read("/mesh/topology").astype(np.int64)
    finally:
        f.close()
    if coords.ndim != 2 or cells.ndim != 2 or cells.shape[1] not in (3, 4):
        raise SolverError("{}: /mesh holds coordinates {} and topology {}; triangles or tetrahedra expected".format(path, coords.shape, cells.shape))
    if cells.shape[1] == 3 and coords.shape[1] == 3 and np.ptp(coords[:, 2]) == 0.0:
        coords = coords[:, :2]

    def reader(group, what):
        with H5File(path) as g:
            present = g.has(group + "/values") and g.has(group + "/topology")
        if not present:
            return None

        def values(entities):
            with H5File(path) as g:
                return _entity_values(g, group, np.asarray(entities, dtype=np.int64), what)
        return values
    return coords, cells, reader("/boundaries", "facets"), reader("/subdomains", "cells")


def write_dolfin_mesh(path, mesh, boundaries=None, subdomains=None):
    """The file SolverBase._read_hdf5_mesh expects: /mesh, and /boundaries, /subdomains for the given MeshFunctions
    (entities with value 0 are left out, as DOLFIN's writers of sparse markers do)."""
    with H5File(path, "w") as f:
        f.write("/mesh/coordinates", mesh.coordinates())
        f.write("/mesh/topology", mesh.cells().astype(np.int64))
        for group, mf, ents in (("/boundaries", boundaries, None if boundaries is None else mesh.facets()),
                                ("/subdomains", subdomains, None if subdomains is None else mesh.cells())):
            if mf is None:
                continue
            a = np.asarray(mf.array())
            sel = np.nonzero(a != 0)[0]
            f.write(group + "/topology", np.asarray(ents)[sel].astype(np.int64))
            f.write(group + "/values", a[sel].astype(np.int64))
