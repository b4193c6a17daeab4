"""GPU tests of the RCCL layer with the one communicator a 1-GPU box allows (n_ranks = 1):
library loading, in-stream all-reduce inside the CG loop, and the grouped send/recv halo
exchange with the rank as its own neighbour (a periodic wrap).  Real N>1 decompositions are
covered on CPU by tests/test_distributed_cpu.py (same plan code, gloo)."""
import numpy as np
import pytest

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_allreduce_and_self_halo(gpu):
    uid = gpu.comm_unique_id()
    assert len(uid) == 128
    gpu.comm_init(1, 0, uid)
    try:
        out = gpu.comm_allreduce_sum([1.5, -2.0, 3.25])
        assert np.array_equal(out, [1.5, -2.0, 3.25])
        # CG with the communicator up: the 3-double all-reduce runs in-stream every iteration
        n = 10
        P = fo.heat_box_problem(n)
        mesh = gpu.DeviceMesh.box(n, n, n)
        V = gpu.DeviceSpace(mesh, 1)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=20.0)
        b = gpu.DeviceVector(V.n_owned)
        A.apply_dirichlet(b, P["dofs"], P["vals"], symmetric=True)
        x = gpu.DeviceVector(V.n_owned)
        st = gpu.krylov_solve(A, b, x, rtol=1e-8, max_iter=2000)
        xo, ito, _ = fo.pcg_jacobi_single_reduction(P["A"], P["b"], rtol=1e-8)
        assert st["converged"] == 1 and abs(st["iterations"] - ito) <= 1
        assert np.abs(x.get() - xo).max() <= 1e-7 * 350
        # halo exchange: slab [0,3) of a 5-layer box has one upper ghost plane; make the rank its own
        # neighbour: the ghost plane must receive the values of the owned dofs listed for sending
        nx, ny, nz = 3, 2, 5
        pl = (nx + 1) * (ny + 1)
        slab = gpu.DeviceMesh.box(nx, ny, nz, zplanes=(0, 3))
        Vs = gpu.DeviceSpace(slab, 1)
        assert Vs.n_local == 4 * pl and Vs.n_owned == 3 * pl
        rng = np.random.default_rng(0)
        send = rng.permutation(Vs.n_owned)[:pl].astype(np.int32)        # non-contiguous -> pack kernel
        Vs.set_halo([0], [send], [pl])
        vals = rng.standard_normal(Vs.n_local)
        v = gpu.DeviceVector(Vs.n_local, vals)
        gpu.halo_exchange(Vs, v)
        got = v.get()
        assert np.array_equal(got[:Vs.n_owned], vals[:Vs.n_owned])
        assert np.array_equal(got[Vs.n_owned:], vals[send])
        send2 = np.arange(pl, 2 * pl, dtype=np.int32)                    # contiguous -> direct send
        Vs.set_halo([0], [send2], [pl])
        gpu.halo_exchange(Vs, v)
        assert np.array_equal(v.get()[Vs.n_owned:], vals[send2])
        # spmv refreshes ghosts first when a plan is attached
        As = gpu.DeviceMatrix(Vs)
        As.assemble(stiffness=1.0)
        y = gpu.DeviceVector(Vs.n_owned)
        v.set(vals)
        As.spmv(v, y)
        rp, ci, va, shape = As.to_csr()
        import scipy.sparse as sp
        M = sp.csr_matrix((va, ci, rp), shape=shape)
        xin = vals.copy()
        xin[Vs.n_owned:] = vals[send2]
        assert np.abs(y.get() - M @ xin).max() <= 1e-13 * (abs(M) @ np.abs(xin)).max()
        Vs.set_halo([], [], [])
    finally:
        gpu.comm_finalize()
    with pytest.raises(gpu.BackendError):
        Vs.set_halo([0], [send2], [pl])
        gpu.halo_exchange(Vs, v)          # plan without communicator: loud
    Vs.set_halo([], [], [])
