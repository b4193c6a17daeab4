"""ctypes wrapper of oracle/fem_oracle_c.c (TEST INFRASTRUCTURE ONLY: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "fem_oracle_c.c")):
        # -march=native must match the machine that runs it: always rebuild where it is used
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB


def load():
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_LIB):
                raise
        _lib = C.CDLL(_LIB)
        _lib.orc_heat_box_solve.restype = C.c_int
        _lib.orc_heat_box_solve.argtypes = [C.c_int64, C.c_int64, C.c_int64, _f64p, C.c_double, C.c_int, C.c_double,
                                            C.c_double, C.c_double, C.c_int, _f64p, _f64p, C.POINTER(C.c_int64)]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_num_threads.argtypes = [C.c_int]
        _lib.orc_set_num_threads(usable_cores())
        _lib.orc_csr_pattern.restype = C.c_int64
        _lib.orc_csr_pattern.argtypes = [C.c_int64, C.c_int64, _i32p, _i32p, C.POINTER(_i32p)]
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_box_mesh.argtypes = [C.c_int64, C.c_int64, C.c_int64, _f64p, _f64p, _f64p, _i32p]
        _lib.orc_assemble_p1.argtypes = [C.c_int64, C.c_int64, _f64p, _i32p, C.c_double, _f64p, _i32p, _i32p, _f64p]
        _lib.orc_csr_pattern_generic.restype = C.c_int64
        _lib.orc_csr_pattern_generic.argtypes = [C.c_int64, C.c_int64, C.c_int, _i32p, _i32p, C.POINTER(_i32p)]
        _lib.orc_assemble_p2.argtypes = [C.c_int64, C.c_int64, _f64p, _i32p, _i32p, C.c_double, _i32p, _i32p, _f64p]
        _lib.orc_assemble_p1_elasticity.argtypes = [C.c_int64, C.c_int64, _f64p, _i32p, C.c_double, C.c_double, _i32p, _i32p, _f64p]
        _lib.orc_pcg_jacobi.restype = C.c_int
        _lib.orc_pcg_jacobi.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int, _f64p]
    return _lib


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU
    quota (a 256-thread box with a 16-CPU quota must run 16 OpenMP threads, not 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(-(-int(quota) // int(period)))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, -(-q // p)))
    except Exception:
        pass
    return n


def set_num_threads(n):
    load().orc_set_num_threads(int(n))


def num_threads():
    return load().orc_num_threads()


def box_mesh(nx, ny, nz, p0=(0.0, 0.0, 0.0), p1=(1.0, 1.0, 1.0)):
    n = (nx + 1) * (ny + 1) * (nz + 1)
    xyz = np.empty((n, 3))
    cells = np.empty((6 * nx * ny * nz, 4), dtype=np.int32)
    a = np.asarray(p0, dtype=np.float64)
    b = np.asarray(p1, dtype=np.float64)
    load().orc_box_mesh(nx, ny, nz, a.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p), xyz.ctypes.data_as(_f64p),
                        cells.ctypes.data_as(_i32p))
    return xyz, cells


def csr_pattern(n, cells):
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    rowptr = np.empty(n + 1, dtype=np.int32)
    out = _i32p()
    nnz = load().orc_csr_pattern(n, len(cells), cells.ctypes.data_as(_i32p), rowptr.ctypes.data_as(_i32p), C.byref(out))
    colidx = np.ctypeslib.as_array(out, shape=(nnz,)).copy()
    load().orc_free(out)
    return rowptr, colidx


def csr_pattern_generic(n, cell_dofs):
    """(rowptr, colidx) of a space with cell_dofs [nc, nd] (CG2: 10 nodes; vector CG1: the 12 scalar dofs 3 v + i)."""
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    rowptr = np.empty(n + 1, dtype=np.int32)
    out = _i32p()
    nnz = load().orc_csr_pattern_generic(n, cd.shape[0], cd.shape[1], cd.ctypes.data_as(_i32p), rowptr.ctypes.data_as(_i32p), C.byref(out))
    if nnz < 0:
        raise ValueError("pattern exceeds 32-bit row pointers")
    colidx = np.ctypeslib.as_array(out, shape=(nnz,)).copy()
    load().orc_free(out)
    return rowptr, colidx


def assemble_p2(coords, cells, cell_dofs, k, rowptr, colidx):
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    cd = np.ascontiguousarray(cell_dofs, dtype=np.int32)
    vals = np.empty(len(colidx))
    load().orc_assemble_p2(len(rowptr) - 1, len(cells), coords.ctypes.data_as(_f64p), cells.ctypes.data_as(_i32p), cd.ctypes.data_as(_i32p),
                           float(k), rowptr.ctypes.data_as(_i32p), colidx.ctypes.data_as(_i32p), vals.ctypes.data_as(_f64p))
    return vals


def assemble_p1_elasticity(coords, cells, mu, lmbda, rowptr, colidx):
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    vals = np.empty(len(colidx))
    load().orc_assemble_p1_elasticity(len(rowptr) - 1, len(cells), coords.ctypes.data_as(_f64p), cells.ctypes.data_as(_i32p), float(mu),
                                      float(lmbda), rowptr.ctypes.data_as(_i32p), colidx.ctypes.data_as(_i32p), vals.ctypes.data_as(_f64p))
    return vals


def assemble_p1(coords, cells, k, rowptr, colidx):
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    vals = np.empty(len(colidx))
    kc = np.ascontiguousarray(k, dtype=np.float64) if np.ndim(k) else None
    load().orc_assemble_p1(len(coords), len(cells), coords.ctypes.data_as(_f64p), cells.ctypes.data_as(_i32p),
                           0.0 if kc is not None else float(k), kc.ctypes.data_as(_f64p) if kc is not None else None,
                           rowptr.ctypes.data_as(_i32p), colidx.ctypes.data_as(_i32p), vals.ctypes.data_as(_f64p))
    return vals


def pcg_jacobi(rowptr, colidx, vals, b, rtol=1e-8, maxit=10000):
    n = len(b)
    x = np.empty(n)
    hist = np.zeros(maxit + 2)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    it = load().orc_pcg_jacobi(n, rowptr.ctypes.data_as(_i32p), colidx.ctypes.data_as(_i32p),
                               vals.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p), x.ctypes.data_as(_f64p), rtol,
                               maxit, hist.ctypes.data_as(_f64p))
    return x, it, hist[: it + 1]


def heat_box_solve(nx, ny, nz, p1=(1.0, 1.0, 1.0), k=20.0, axis=2, t_lo=350.0, t_hi=300.0, rtol=1e-8, maxit=20000):
    """Returns dict(x, iterations, t_mesh, t_symbolic, t_assemble, t_solve, nnz, threads)."""
    n = (nx + 1) * (ny + 1) * (nz + 1)
    x = np.empty(n)
    times = np.zeros(4)
    nnz = C.c_int64(0)
    p = np.asarray(p1, dtype=np.float64)
    it = load().orc_heat_box_solve(nx, ny, nz, p.ctypes.data_as(_f64p), k, axis, t_lo, t_hi, rtol, maxit,
                                   x.ctypes.data_as(_f64p), times.ctypes.data_as(_f64p), C.byref(nnz))
    return dict(x=x, iterations=it, t_mesh=times[0], t_symbolic=times[1], t_assemble=times[2], t_solve=times[3],
                nnz=nnz.value, threads=num_threads())
