"""ctypes binding of libfsamd.so (include/fenicssolver_amd.h).

This is the only place that touches the C-ABI.  The reference has no FFI; the
calls bound here stand in for the dolfin/PETSc calls made from
FenicsSolver/SolverBase.py:592-672 (see the header for the per-function
mapping).  There is NO CPU fallback: if the shared library is missing, or no
gfx950 device is visible, every compute entry point raises ``BackendError``.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsamd.so")

FS_OK = 0
FS_ERR_COMM, FS_ERR_NUMERIC, FS_ERR_P2P_TIMEOUT = -5, -6, -7      # include/fenicssolver_amd.h
FS_COEF_NONE, FS_COEF_CONST, FS_COEF_CELL, FS_COEF_TENSOR, FS_COEF_NODAL, FS_COEF_CELL_ROW, FS_COEF_CELL_TENSOR, FS_COEF_CELL_QP = 0, 1, 2, 3, 4, 5, 6, 7
FS_KSP_CG = 0
FS_KSP_BICGSTAB = 1
FS_PC_NONE, FS_PC_JACOBI = 0, 1
FS_NORM_UNPRECONDITIONED, FS_NORM_PRECONDITIONED = 0, 1
FS_UNIQUE_ID_BYTES = 128

c_i64 = C.c_int64
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class BackendError(RuntimeError):
    """libfsamd.so is missing, no GPU is visible, or a call failed (rc = the FS_ERR_* code of the call, None otherwise)."""
    rc = None


class fs_coef(C.Structure):
    _fields_ = [("mode", C.c_int), ("value", C.c_double), ("data", c_f64p), ("tensor", C.c_double * 9)]


class fs_bilinear_form(C.Structure):
    _fields_ = [("stiffness", fs_coef), ("mass", fs_coef), ("lame_mu", C.c_double), ("lame_lambda", C.c_double),
                ("advection", fs_coef), ("advection_scale", C.c_double), ("supg_pe", C.c_double)]


class fs_linear_form(C.Structure):
    _fields_ = [("source", fs_coef), ("vector_value", C.c_double * 3), ("div_coef", fs_coef), ("supg_velocity", fs_coef),
                ("supg_pe", C.c_double)]


class fs_krylov_opts(C.Structure):
    _fields_ = [("method", C.c_int), ("precond", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_iter", C.c_int), ("batch", C.c_int), ("nonzero_guess", C.c_int), ("norm_type", C.c_int),
                ("diagonal_scale", C.c_int), ("pipelined", C.c_int)]


class fs_krylov_stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("bnorm", C.c_double),
                ("rel_residual", C.c_double), ("true_rel_residual", C.c_double), ("solve_ms", C.c_double),
                ("spmv_ms", C.c_double), ("update_ms", C.c_double), ("spmv_bytes", c_i64), ("row_classes", C.c_int),
                ("fused_iteration", C.c_int), ("classes_kept", C.c_int), ("lattice_order", C.c_int), ("launches", C.c_int), ("product_kind", C.c_int)]


class fs_ns_form(C.Structure):
    _fields_ = [("kinematic_viscosity", C.c_double), ("density", C.c_double), ("inv_dt", C.c_double),
                ("body_force", C.c_double * 3), ("convection", C.c_int), ("newton", C.c_int), ("mesh_velocity", C.c_double * 3),
                ("g2_mode", C.c_int), ("g2_kappa1", C.c_double),
                ("viscosity_pressure_ref", C.c_double), ("viscosity_pressure_exponent", C.c_double)]


class fs_viscosity_law(C.Structure):
    _fields_ = [("kind", C.c_int), ("pressure_ref", C.c_double), ("pressure_exponent", C.c_double), ("pressure_coef", C.c_double),
                ("temperature_coef", C.c_double), ("temperature_ref", C.c_double), ("temperature", C.c_void_p)]


class fs_saddle_opts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("max_iter", C.c_int), ("restart", C.c_int),
                ("kinematic_viscosity", C.c_double), ("density", C.c_double), ("inv_dt", C.c_double),
                ("velocity_sweeps", C.c_int), ("inner_rtol", C.c_double), ("nonzero_guess", C.c_int)]


class fs_amg_opts(C.Structure):
    _fields_ = [("strength_threshold", C.c_double), ("max_levels", C.c_int), ("coarse_size", C.c_int),
                ("smoother_steps", C.c_int), ("eig_steps", C.c_int), ("rigid_body_modes", C.c_int)]


# name -> (restype, argtypes); every symbol declared in include/fenicssolver_amd.h
_H = C.c_void_p
SIGNATURES = {
    "fs_init": (C.c_int, [C.c_int]),
    "fs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "fs_device_synchronize": (C.c_int, []),
    "fs_profile_marker": (C.c_int, [C.c_int]),
    "fs_memory_info": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "fs_memory_trim": (C.c_int, []),
    "fs_last_error": (C.c_char_p, []),
    "fs_version": (C.c_char_p, []),
    "fs_set_option": (C.c_int, [C.c_char_p, C.c_double]),
    "fs_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), c_i64p]),
    "fs_mesh_create": (C.c_int, [C.c_int, c_i64, c_f64p, c_i64, c_i32p, C.c_int, c_i64, C.POINTER(_H)]),
    "fs_mesh_create_box": (C.c_int, [c_i64, c_i64, c_i64, c_f64p, c_f64p, c_i64, c_i64, C.POINTER(_H)]),
    "fs_mesh_info": (C.c_int, [_H, c_i64p, c_i64p, c_i64p]),
    "fs_mesh_get": (C.c_int, [_H, c_f64p, c_i32p, c_i64p]),
    "fs_mesh_destroy": (C.c_int, [_H]),
    "fs_space_create": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "fs_space_create_coupled": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, c_i64, c_i32p, C.POINTER(_H)]),
    "fs_space_info": (C.c_int, [_H, c_i64p, c_i64p, c_i64p, c_i64p]),
    "fs_space_format_info": (C.c_int, [_H, c_i64p, c_i64p, c_i64p]),
    "fs_space_get_edges": (C.c_int, [_H, c_i64p, c_i32p]),
    "fs_space_destroy": (C.c_int, [_H]),
    "fs_vector_create": (C.c_int, [c_i64, C.POINTER(_H)]),
    "fs_vector_size": (C.c_int, [_H, c_i64p]),
    "fs_vector_set": (C.c_int, [_H, c_f64p, c_i64]),
    "fs_vector_get": (C.c_int, [_H, c_f64p, c_i64]),
    "fs_vector_fill": (C.c_int, [_H, C.c_double]),
    "fs_vector_add_entries": (C.c_int, [_H, C.c_int64, c_i32p, c_f64p]),
    "fs_vector_copy": (C.c_int, [_H, _H, c_i64]),
    "fs_vector_axpy": (C.c_int, [_H, C.c_double, _H]),
    "fs_vector_dot": (C.c_int, [_H, _H, c_f64p]),
    "fs_vector_destroy": (C.c_int, [_H]),
    "fs_matrix_create": (C.c_int, [_H, C.POINTER(_H)]),
    "fs_matrix_info": (C.c_int, [_H, c_i64p, c_i64p, c_i64p]),
    "fs_matrix_zero": (C.c_int, [_H]),
    "fs_matrix_axpy": (C.c_int, [_H, C.c_double, _H]),
    "fs_matrix_copy": (C.c_int, [_H, _H]),
    "fs_matrix_tie_nodes": (C.c_int, [_H, _H, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "fs_vector_assign_entries": (C.c_int, [_H, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "fs_matrix_get_csr": (C.c_int, [_H, c_i32p, c_i32p, c_f64p]),
    "fs_matrix_destroy": (C.c_int, [_H]),
    "fs_assemble_matrix": (C.c_int, [_H, C.POINTER(fs_bilinear_form), C.c_int]),
    "fs_operator_apply": (C.c_int, [_H, C.POINTER(fs_bilinear_form), _H, _H, C.c_int, C.POINTER(C.c_double)]),
    "fs_assemble_vector": (C.c_int, [_H, C.POINTER(fs_linear_form), _H, C.c_int]),
    "fs_assemble_facet_vector": (C.c_int, [_H, c_i64, c_i32p, c_f64p, _H]),
    "fs_assemble_facet_matrix": (C.c_int, [_H, c_i64, c_i32p, c_f64p]),
    "fs_assemble_interior_penalty": (C.c_int, [_H, c_i64, c_i32p, C.c_double]),
    "fs_apply_dirichlet": (C.c_int, [_H, _H, c_i64, c_i32p, c_f64p, C.c_int]),
    "fs_spmv": (C.c_int, [_H, _H, _H]),
    "fs_krylov_solve": (C.c_int, [_H, _H, _H, C.POINTER(fs_krylov_opts), C.POINTER(fs_krylov_stats)]),
    "fs_krylov_history": (C.c_int, [c_f64p, C.c_int, C.POINTER(C.c_int)]),
    "fs_spmv_benchmark": (C.c_int, [_H, _H, _H, C.c_int, c_f64p]),
    "fs_spmv_dictionary": (C.c_int, [_H, _H, _H, C.POINTER(C.c_int)]),
    "fs_last_product_kind": (C.c_int, []),
    "fs_amg_setup": (C.c_int, [_H, C.c_int, c_f64p, C.POINTER(fs_amg_opts), C.POINTER(_H)]),
    "fs_amg_attach_distributed_fine": (C.c_int, [_H, _H, c_i64, c_i32p]),
    "fs_amg_destroy": (C.c_int, [_H]),
    "fs_amg_info": (C.c_int, [_H, C.POINTER(C.c_int), c_f64p, c_f64p, c_f64p]),
    "fs_amg_level_info": (C.c_int, [_H, C.c_int, c_i64p, C.POINTER(C.c_int), c_i64p, c_i64p, C.POINTER(C.c_int), c_f64p]),
    "fs_amg_level_get": (C.c_int, [_H, C.c_int, C.c_int, c_i32p, c_i32p, c_f64p]),
    "fs_amg_apply": (C.c_int, [_H, _H, _H]),
    "fs_amg_solve": (C.c_int, [_H, _H, _H, C.POINTER(fs_krylov_opts), C.POINTER(fs_krylov_stats)]),
    "fs_assemble_facet_supg": (C.c_int, [_H, _H, _H, C.c_int64, c_i32p, c_i32p, c_f64p, c_f64p, C.POINTER(fs_coef), C.c_double]),
    "fs_assemble_navier_stokes": (C.c_int, [_H, _H, _H, _H, C.POINTER(fs_ns_form)]),
    "fs_space_set_viscosity_law": (C.c_int, [_H, C.POINTER(fs_viscosity_law)]),
    "fs_assemble_ns_pressure_boundary": (C.c_int, [_H, _H, C.c_int64, c_i32p, c_i32p, c_f64p, C.c_double]),
    "fs_assemble_ns_pressure_boundary_nn": (C.c_int, [_H, _H, C.c_int64, c_i32p, c_i32p, c_f64p, C.c_double, _H, C.c_double, C.c_double,
                                                    C.c_int]),
    "fs_saddle_solve": (C.c_int, [_H, _H, _H, _H, _H, _H, C.POINTER(fs_saddle_opts), C.POINTER(fs_krylov_stats)]),
    "fs_comm_get_unique_id": (C.c_int, [C.c_char_p]),
    "fs_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
    "fs_assemble_von_mises": (C.c_int, [_H, _H, C.c_double, C.c_double, _H, _H]),
    "fs_assemble_viscous_stress": (C.c_int, [_H, _H, C.c_double, _H, _H]),
    "fs_assemble_viscous_stress_nn": (C.c_int, [_H, _H, C.c_double, _H, _H, C.c_double, C.c_double]),
    "fs_comm_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fs_comm_allreduce_sum": (C.c_int, [c_f64p, C.c_int]),
    "fs_comm_allgather": (C.c_int, [c_f64p, c_i64, c_i64, c_f64p]),
    "fs_comm_finalize": (C.c_int, []),
    "fs_mesh_set_global_ids": (C.c_int, [_H, c_i64p]),
    "fs_space_set_halo_indexed": (C.c_int, [_H, C.c_int, c_i32p, c_i64p, c_i32p, c_i64p, c_i32p]),
    "fs_space_set_halo": (C.c_int, [_H, C.c_int, c_i32p, c_i64p, c_i32p, c_i64p]),
    "fs_halo_exchange": (C.c_int, [_H, _H]),
    "fs_mesh_locality_order": (C.c_int, [C.c_int, c_i64, c_f64p, c_i64, c_i32p, C.c_int, c_i32p, c_i32p]),
    "fs_mesh_create_renumbered": (C.c_int, [c_i64, c_f64p, c_i64, c_i32p, c_i32p, c_i32p, C.POINTER(_H)]),
    "fs_space_enable_p2p_halo": (C.c_int, [_H, C.c_int]),
    "fs_comm_benchmark": (C.c_int, [_H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """dlopen libfsamd.so and type every entry point.  Needs no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). fenicssolver_amd has no CPU fallback." % LIB_PATH)
    # RCCL between the processes of a node needs dmabuf IPC on hosts whose driver has no legacy IPC (hipIpcGetMemHandle
    # fails otherwise); the HSA runtime reads this when the first HIP call initialises it, i.e. after this dlopen
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise BackendError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != FS_OK:
        msg = load().fs_last_error()
        err = BackendError("%s failed (code %d): %s" % (what or "libfsamd call", rc, (msg or b"").decode()))
        err.rc = int(rc)
        raise err


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def p_f64(a):
    return a.ctypes.data_as(c_f64p) if a is not None else None


def p_i32(a):
    return a.ctypes.data_as(c_i32p) if a is not None else None


def p_i64(a):
    return a.ctypes.data_as(c_i64p) if a is not None else None
