"""Edge cases of the C-ABI on the GPU: degenerate sizes, empty inputs, zero data, bad input must fail loudly."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def _csr(A):
    rp, ci, va, shape = A.to_csr()
    return sp.csr_matrix((va, ci, rp), shape=shape)


def test_single_tetrahedron_all_element_types(gpu):
    co = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    ce = np.array([[0, 1, 2, 3]], dtype=np.int32)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=2.0, mass=3.0)
    ref = fo.assemble_p1_scalar(co, ce, 2.0) + fo.assemble_matrix(4, ce, fo.p1_mass_local(co, ce, 3.0))
    assert abs(_csr(A) - ref).max() <= 1e-14
    V3 = gpu.DeviceSpace(mesh, ncomp=3)
    A3 = gpu.DeviceMatrix(V3)
    A3.assemble(lame=(1.0, 2.0))
    assert _csr(A3).shape == (12, 12) and abs(_csr(A3) - _csr(A3).T).max() <= 1e-14
    V2 = gpu.DeviceSpace(mesh, degree=2)
    A2 = gpu.DeviceMatrix(V2)
    A2.assemble(stiffness=1.0)
    K2 = fo.assemble_generic(10, fo.p2_cell_dofs(4, ce)[0], fo.p2_stiffness_local(co, ce, 1.0))
    assert V2.n_owned == 10 and abs(_csr(A2) - K2).max() <= 1e-13
    assert np.abs(_csr(A2) @ np.ones(10)).max() <= 1e-13          # constants are in the kernel of the stiffness matrix


def test_zero_right_hand_side_and_empty_dirichlet_list(gpu):
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), 3, 3, 3)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0, mass=1.0)
    b = gpu.DeviceVector(V.n_owned)
    A.apply_dirichlet(b, np.zeros(0, dtype=np.int32), np.zeros(0), symmetric=True)     # no-op
    x = gpu.DeviceVector(V.n_local, np.ones(V.n_local))
    st = gpu.krylov_solve(A, b, x, rtol=1e-10)
    assert st["converged"] == 1 and st["iterations"] == 0 and np.all(x.get() == 0.0)   # b = 0 -> x = 0
    for method in ("bicgstab",):
        st = gpu.krylov_solve(A, b, x, rtol=1e-10, method=method)
        assert st["converged"] == 1 and np.all(x.get() == 0.0)
    amg = gpu.AMG(A)
    st = amg.solve(b, x)
    assert st["converged"] == 1 and st["iterations"] == 0 and np.all(x.get() == 0.0)


def test_duplicate_dirichlet_entries_later_wins_and_out_of_range_is_rejected(gpu):
    from fenicssolver_amd._lib import BackendError
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), 2, 2, 2)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    b = gpu.DeviceVector(V.n_owned)
    dofs = np.array([0, 5, 0, 26, 5], dtype=np.int32)
    vals = np.array([1.0, 2.0, 3.0, 4.0, 7.0])
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    bb = b.get()
    assert bb[0] == 3.0 and bb[5] == 7.0 and bb[26] == 4.0
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-12)
    assert st["converged"] == 1 and abs(x.get()[0] - 3.0) < 1e-10 and abs(x.get()[5] - 7.0) < 1e-10
    with pytest.raises(BackendError):
        A.apply_dirichlet(b, np.array([27], dtype=np.int32), np.array([1.0]), symmetric=True)
    with pytest.raises(BackendError):
        A.apply_dirichlet(b, np.array([-1], dtype=np.int32), np.array([1.0]), symmetric=True)


def test_iteration_limit_and_singular_operator_are_reported(gpu):
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), 6, 6, 6)
    mesh = gpu.DeviceMesh(co, ce)
    V = gpu.DeviceSpace(mesh)
    A = gpu.DeviceMatrix(V)
    A.assemble(stiffness=1.0)
    b = gpu.DeviceVector(V.n_owned)
    gpu.assemble_vector(V, b, source=1.0)
    bot = np.nonzero(co[:, 2] == 0)[0].astype(np.int32)
    A.apply_dirichlet(b, bot, np.zeros(len(bot)), symmetric=True)
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-12, max_iter=3)
    assert st["converged"] == 0 and st["iterations"] == 3 and st["true_rel_residual"] > 1e-6
    # the solver API turns that into a SolverError instead of returning a wrong field
    from fenicssolver_amd.fem import SolverError
    from fenicssolver_amd import SolverBase as SB
    assert issubclass(SolverError, Exception) and SB.KRYLOV_RTOL_CAP <= 1e-8
    # pure Neumann problem with an incompatible load: CG cannot converge, and says so
    N = gpu.DeviceMatrix(V)
    N.assemble(stiffness=1.0)
    gpu.assemble_vector(V, b, source=1.0)
    from fenicssolver_amd._lib import BackendError
    try:
        st = gpu.krylov_solve(N, b, x, rtol=1e-10, max_iter=200)
        assert st["converged"] != 1
    except BackendError as e:       # or the recurrence breaks down on the singular operator: loud as well
        assert "breakdown" in str(e)


def test_bad_meshes_are_rejected(gpu):
    from fenicssolver_amd._lib import BackendError
    co = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    with pytest.raises(BackendError):
        gpu.DeviceMesh(co, np.array([[0, 1, 2, 4]], dtype=np.int32))        # vertex index out of range
    with pytest.raises((BackendError, ValueError)):
        gpu.DeviceMesh(co, np.zeros((0, 4), dtype=np.int32))                 # no cells
    with pytest.raises(BackendError):
        gpu.DeviceSpace(gpu.DeviceMesh(co, np.array([[0, 1, 2, 3]], dtype=np.int32)), ncomp=2)   # 2-vectors are not built


def test_released_blocks_are_reused_and_can_be_trimmed(gpu):
    """The block cache of the library (include/fenicssolver_amd.h, fs_memory_info): a released vector's block serves
    the next request of its size, the results of work in re-used blocks are those of fresh ones, trim empties it."""
    import os
    if os.environ.get("FS_POOL_MAX_MB") == "0":
        pytest.skip("the block cache is switched off (FS_POOL_MAX_MB=0)")
    gpu.trim_memory()
    base = gpu.memory_info()
    n = 1 << 20
    a = gpu.DeviceVector(n)
    a.set(np.arange(n, dtype=np.float64))
    assert gpu.memory_info()["live_bytes"] - base["live_bytes"] >= 8 * n
    a.close()
    after_close = gpu.memory_info()
    assert after_close["cached_bytes"] - base["cached_bytes"] >= 8 * n
    b = gpu.DeviceVector(n)                       # same size: comes out of the cache
    assert gpu.memory_info()["cached_bytes"] <= after_close["cached_bytes"] - 8 * n
    b.fill(2.5)
    assert np.all(b.get() == 2.5)
    # a solve whose temporaries live in re-used blocks equals the first one bit for bit
    co, ce = fo.box_mesh((0, 0, 0), (1, 1, 1), 6, 5, 4)
    sols = []
    for _ in range(3):
        mesh = gpu.DeviceMesh(co, ce)
        V = gpu.DeviceSpace(mesh)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=1.0, mass=1.0)
        rhs = gpu.DeviceVector(V.n_owned)
        gpu.assemble_vector(V, rhs, source=1.0)
        x = gpu.DeviceVector(V.n_owned)
        gpu.krylov_solve(A, rhs, x, rtol=1e-12)
        sols.append(x.get())
        for h in (x, rhs, A, V, mesh):
            h.close()
    assert np.array_equal(sols[0], sols[1]) and np.array_equal(sols[0], sols[2])
    b.close()
    gpu.trim_memory()
    assert gpu.memory_info()["cached_bytes"] == 0
