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


@pytest.mark.parametrize("layout", ["two_neighbours_contiguous", "one_neighbour_packed", "two_neighbours_p2p", "one_neighbour_p2p"])
def test_rccl_single_rank_pipelined_cg_with_self_halo(gpu, layout):
    """REAL RCCL (the 1-rank communicator a 1-GPU box allows) under the pipelined CG: the all-reduce of the sums is enqueued on
    the communication stream behind the grouped send / recv of the halo, the compute stream waits for both through events -
    the stream / event interplay production runs, with the rank as its own neighbour.  A z-periodic slab: the lower ghost plane
    receives the top owned plane, the upper ghost plane the bottom owned plane, so the operator (stiffness + mass) is the SPD
    operator of the periodic problem.  Both recurrences must agree, with the early start of the exchange (contiguous prefix /
    suffix sends) and with packed sends."""
    uid = gpu.comm_unique_id()
    gpu.comm_init(1, 0, uid)
    try:
        nx, ny, nz = 6, 5, 9
        pl = (nx + 1) * (ny + 1)
        slab = gpu.DeviceMesh.box(nx, ny, nz, zplanes=(1, nz))          # owned planes 1 .. nz-1, ghosts: plane 0 then plane nz
        V = gpu.DeviceSpace(slab, 1)
        n_own = (nz - 1) * pl
        assert V.n_owned == n_own and V.n_local == n_own + 2 * pl
        top = np.arange(n_own - pl, n_own, dtype=np.int32)               # feeds the LOWER ghost plane
        bottom = np.arange(0, pl, dtype=np.int32)                        # feeds the UPPER ghost plane
        if layout.startswith("two_neighbours"):
            V.set_halo([0, 0], [top, bottom], [pl, pl])
        else:
            V.set_halo([0], [np.concatenate([top, bottom])], [2 * pl])
        if layout.endswith("p2p"):          # the kernels of the peer-to-peer exchange, the rank writing into its own buffer
            V.enable_p2p_halo(True)
        A = gpu.DeviceMatrix(V)
        A.assemble(stiffness=3.0, mass=5.0)
        rng = np.random.default_rng(2)
        b = gpu.DeviceVector(V.n_owned, rng.standard_normal(V.n_owned))
        sols = []
        for pipelined in (False, True):
            x = gpu.DeviceVector(V.n_local)
            st = gpu.krylov_solve(A, b, x, rtol=1e-11, max_iter=2000, pipelined=pipelined)
            assert st["converged"] == 1 and st["true_rel_residual"] <= 5e-11, (pipelined, st)
            sols.append((x.get()[:V.n_owned].copy(), st["iterations"]))
        (x0, it0), (x1, it1) = sols
        assert -1 <= it1 - it0 <= 2
        assert np.abs(x1 - x0).max() <= 1e-9 * np.abs(x0).max()
        # and the answer is the periodic problem's: rows of the local matrix with the ghost columns folded onto the owned planes
        rp, ci, va, shape = A.to_csr()
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        M = sp.csr_matrix((va, ci, rp), shape=shape).tocoo()
        col = M.col.copy()
        lower = (col >= n_own) & (col < n_own + pl)
        upper = col >= n_own + pl
        col[lower] = top[col[lower] - n_own]
        col[upper] = bottom[col[upper] - n_own - pl]
        Mp = sp.coo_matrix((M.data, (M.row, col)), shape=(n_own, n_own)).tocsr()
        assert abs(Mp - Mp.T).max() <= 1e-12 * abs(Mp).max()
        ref = spl.spsolve(Mp.tocsc(), b.get())
        assert np.abs(x1 - ref).max() <= 1e-8 * np.abs(ref).max()
        if layout.endswith("p2p"):
            from fenicssolver_amd import _lib as L
            assert L.load().fs_space_set_halo(V.h, 0, None, None, None, None) != 0       # plan replaced with the exchange on: refused
            V.enable_p2p_halo(False)
            assert L.load().fs_space_set_halo(V.h, 0, None, None, None, None) == 0
    finally:
        gpu.comm_finalize()
