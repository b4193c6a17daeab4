"""Several ranks on ONE GPU: the complete distributed algorithm (owner-computes partition, ghost layout, halo packing,
interior/boundary overlap, fused dots + all-reduce, restarts, result gather) with real device kernels in every rank.
RCCL refuses two ranks on one device, so FS_RCCL_PATH points libfsamd.so at tests/shim/libfakerccl.so - a stand-in
that exports the nccl* entry points and moves the data through host shared memory.  The library runs exactly the
ncclGroupStart/Send/Recv/GroupEnd, ncclAllReduce and ncclAllGather call sites it runs in production (there is no
other transport in it); real RCCL is covered by test_gpu_comm.py with the 1-rank communicator a 1-GPU box allows."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import fem_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT = [29610]


def _run(world, case, tmp_path, **extra_env):
    out = str(tmp_path / ("%s_%d.npz" % (case, world)))
    shim = os.path.join(ROOT, "tests", "shim", "libfakerccl.so")
    assert os.path.exists(shim), "build tests/shim first (make -C tests/shim; __graft_entry__.build() does it)"
    env = dict(os.environ, FS_RCCL_PATH=shim, **extra_env)
    PORT[0] += 1
    cmd = [sys.executable, "-m", "fenicssolver_amd.launch", "--nproc", str(world), "--devices", ",".join(["0"] * world),
           "--master-port", str(PORT[0]), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, case]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    return np.load(out)


@pytest.mark.parametrize("world,recurrence", [(2, "pipelined"), (3, "pipelined"), (2, "single_reduction"), (3, "single_reduction"),
                                              (3, "pipelined_no_early_halo"), (2, "pipelined_p2p"), (3, "pipelined_p2p"),
                                              (3, "single_reduction_p2p"), (3, "p2p_unfused"), (2, "p2p_send_recv_kernels"),
                                              (2, "default"), (3, "default"), (2, "p2p_breaks_mid_run"), (3, "p2p_breaks_mid_run")])
def test_box_slabs_over_ranks_equal_the_single_gpu_solve(gpu, tmp_path, world, recurrence):
    """Both CG recurrences and both transports on several ranks: same solution as one GPU <= 1e-9, iteration count within +2.
    Default: the single-reduction recurrence over RCCL (round 5: the peer-to-peer exchange is opt-in, FS_HALO_P2P=auto / 1, set up
    and self-tested when the halo plan is set); a transport that breaks in the middle of a run - every wait from the sixth receive on times out - is left by
    all ranks together and the solve repeated over RCCL (backend._with_p2p_fallback): same field."""
    nx, ny, nz, axis = 9, 7, 23, 0
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 2.0), nx, ny, nz)
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0))
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    b = gpu.DeviceVector(V.n_owned)
    x = gpu.DeviceVector(V.n_local)
    A.assemble(stiffness=20.0)
    gpu.assemble_vector(V, b, source=3.0)
    lo, hi = np.nonzero(co[:, axis] == 0)[0], np.nonzero(co[:, axis] == co[:, axis].max())[0]
    A.apply_dirichlet(b, np.concatenate([lo, hi]).astype(np.int32),
                      np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), symmetric=True)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    env = {"single_reduction": dict(FS_CG_PIPELINED="0", FS_HALO_P2P="0"), "pipelined": dict(FS_CG_PIPELINED="1", FS_HALO_P2P="0"),
           "pipelined_no_early_halo": dict(FS_CG_PIPELINED="1", FS_HALO_P2P="0", FS_HALO_EARLY="0"),
           "default": {}, "p2p_breaks_mid_run": dict(FS_HALO_P2P="auto", FS_P2P_TEST="late:6", FS_P2P_TIMEOUT_MS="100"),
           # ghost refresh by stores into the neighbour PROCESS's memory (hipIpc) instead of send / recv: real on one GPU too
           "pipelined_p2p": dict(FS_HALO_P2P="1", FS_CG_PIPELINED="1"), "single_reduction_p2p": dict(FS_HALO_P2P="1"),
           # FS_P2P_FUSE: 0 = the separate send / receive / all-reduce kernels around a split product, 6 = the fused iteration with
           # its send and receive still kernels of their own
           "p2p_unfused": dict(FS_HALO_P2P="1", FS_P2P_FUSE="0"), "p2p_send_recv_kernels": dict(FS_HALO_P2P="1", FS_P2P_FUSE="6")}[recurrence]
    r = _run(world, "box", tmp_path, **env)
    assert int(r["converged"]) == 1 and float(r["true_res"]) <= 2e-10
    assert abs(int(r["iterations"]) - st["iterations"]) <= 2          # reduction order differs, the recurrence does not
    assert np.abs(r["x"] - x.get()).max() <= 1e-9 * np.abs(x.get()).max()


@pytest.mark.parametrize("world,mode", [(2, "default"), (3, "default"), (2, "two_launch_kernels"), (3, "breaks_mid_run")])
def test_one_launch_iteration_on_a_decomposed_space(gpu, tmp_path, world, mode):
    """The one-launch CG iteration under decomposition (k_cg_p2p_exchange<true> + k_dict_cg_iter<3, true>: two launches per iteration
    instead of three).  A box whose mesh lines are wide enough for the row dictionary (72 vertices per line): every rank reports the
    one-launch iteration, the field is the one-GPU field, a second solve from a perturbed iterate (restart, captured batches re-used)
    as well; FS_CG_FUSED_P2P=0 keeps the three-launch iteration - same field; a transport that breaks in the middle of the run is
    left by all ranks together and the solve repeated over RCCL."""
    nx, ny, nz, axis = 71, 9, 23, 0
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 2.0), nx, ny, nz)
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0))
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    b = gpu.DeviceVector(V.n_owned)
    x = gpu.DeviceVector(V.n_local)
    A.assemble(stiffness=20.0)
    gpu.assemble_vector(V, b, source=3.0)
    lo, hi = np.nonzero(co[:, axis] == 0)[0], np.nonzero(co[:, axis] == co[:, axis].max())[0]
    A.apply_dirichlet(b, np.concatenate([lo, hi]).astype(np.int32),
                      np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)]), symmetric=True)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    assert st["row_classes"] > 0 and st["iterations"] > 64
    env = {"default": dict(FS_HALO_P2P="auto"), "two_launch_kernels": dict(FS_HALO_P2P="auto", FS_CG_FUSED_P2P="0"),
           "breaks_mid_run": dict(FS_HALO_P2P="auto", FS_P2P_TEST="late:6", FS_P2P_TIMEOUT_MS="100")}[mode]
    r = _run(world, "box", tmp_path, FS_TEST_BOX="%d,%d,%d,%d" % (nx, ny, nz, axis), **env)
    assert int(r["converged"]) == 1 and float(r["true_res"]) <= 2e-10 and float(r["true_res2"]) <= 2e-10
    assert abs(int(r["iterations"]) - st["iterations"]) <= 2
    xs = x.get()
    # (the second solve takes another Krylov path to the same tolerance: its error is the tolerance times the condition number)
    assert np.abs(r["x"] - xs).max() <= 1e-9 * np.abs(xs).max() and np.abs(r["x2"] - xs).max() <= 2e-8 * np.abs(xs).max()
    assert np.all(r["dictionary"] == 1)
    if mode == "default":
        assert np.all(r["fused"] == 2) and np.all(r["fused2"] == 2)      # (2: exchange kernel + iteration kernel)
    if mode == "two_launch_kernels":
        assert np.all(r["fused"] == 0)


@pytest.mark.parametrize("world,mode", [(3, "p2p"), (2, "p2p"), (3, "rccl")])
def test_forty_solves_back_to_back_over_ranks(gpu, tmp_path, world, mode):
    """40 solves in a row on one decomposed space (operator, load and tolerance change from solve to solve): hundreds of exchanges
    through the two slots, sequence numbers that advance only with executed exchanges, captured batches re-used - every solve
    converges to its tolerance, and the last field is the one-GPU field."""
    nx, ny, nz, axis = 9, 7, 23, 0
    co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.8, 2.0), nx, ny, nz)
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0))
    V = gpu.DeviceSpace(mesh, 1)
    A = gpu.DeviceMatrix(V)
    b = gpu.DeviceVector(V.n_owned)
    x = gpu.DeviceVector(V.n_local)
    lo, hi = np.nonzero(co[:, axis] == 0)[0], np.nonzero(co[:, axis] == co[:, axis].max())[0]
    dofs = np.concatenate([lo, hi]).astype(np.int32)
    vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
    k = 39
    A.assemble(stiffness=20.0, mass=0.1 * (k % 3))
    gpu.assemble_vector(V, b, source=3.0 + k)
    A.apply_dirichlet(b, dofs, vals * (1.0 + 0.01 * k), symmetric=True)
    gpu.krylov_solve(A, b, x, rtol=10.0 ** -(6 + k % 6), max_iter=5000)
    r = _run(world, "box_stress", tmp_path, **(dict(FS_HALO_P2P="1", FS_P2P_TIMEOUT_MS="3000") if mode == "p2p" else {}))
    assert len(r["iterations"]) == 40
    assert all(float(t) <= 1.5 * 10.0 ** -(6 + i % 6) for i, t in enumerate(r["true_res"]))
    assert np.abs(r["x"] - x.get()[:V.n_owned]).max() <= 1e-8 * np.abs(x.get()).max()


@pytest.mark.parametrize("world,mode", [(2, "rccl"), (3, "p2p"), (2, "p2p_unfused")])
def test_vector_space_slabs_over_ranks(gpu, tmp_path, world, mode):
    """Three dofs per node (elasticity + mass operator), Jacobi-CG: the dof-level halo of a vector space over RCCL, over the fused
    peer-to-peer iteration and over the separate peer-to-peer kernels - same solution as one GPU."""
    nx, ny, nz = 6, 5, 17
    mesh = gpu.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0))
    V = gpu.DeviceSpace(mesh, 3)
    A = gpu.DeviceMatrix(V)
    A.assemble(lame=(1.0, 1.5), mass=4.0)
    g = np.arange(V.n_owned)
    b = gpu.DeviceVector(V.n_owned, np.sin(0.37 * g) + 0.2)
    x = gpu.DeviceVector(V.n_local)
    st = gpu.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    env = {"rccl": {}, "p2p": dict(FS_HALO_P2P="1"), "p2p_unfused": dict(FS_HALO_P2P="1", FS_P2P_FUSE="0")}[mode]
    r = _run(world, "box3", tmp_path, **env)
    assert int(r["converged"]) == 1 and float(r["true_res"]) <= 2e-10
    assert abs(int(r["iterations"]) - st["iterations"]) <= 2
    assert np.abs(r["x"] - x.get()[:V.n_owned]).max() <= 1e-9 * np.abs(x.get()).max()


@pytest.mark.parametrize("case,world", [("heat", 2), ("heat_cn", 2), ("elasticity", 2), ("heat_p2", 2), ("heat_p2", 3), ("heat_supg", 2),
                                        ("heat_supg_field", 2), ("heat_p2_supg_field", 2), ("heat_ip", 2), ("heat_ip", 3), ("elasticity_pfield", 2), ("elasticity_p2_pfield", 2),
                                        ("heat_periodic_y", 2), ("heat_periodic_z", 2), ("heat_periodic_z", 3), ("heat_periodic_z_cn", 2),
                                        ("elasticity_periodic", 2), ("elasticity_periodic", 3),
                                        ("heat_p2_periodic_y", 2), ("heat_p2_periodic_z", 2), ("heat_p2_periodic_z", 3), ("elasticity_p2_periodic", 2)])
def test_solver_classes_under_several_ranks(gpu, tmp_path, case, world):
    _solver_classes_case(gpu, tmp_path, case, world)


@pytest.mark.parametrize("case,world", [("heat", 3), ("heat_cn", 2), ("elasticity", 2), ("heat_p2", 3), ("heat_supg", 2)])
def test_solver_classes_with_the_peer_to_peer_halo(gpu, tmp_path, case, world):
    """FS_HALO_P2P=1: every halo plan of the run (P1, the 3-component one, the indexed CG2 one) refreshes its ghosts by stores
    into the neighbour's hipIpc-mapped buffer."""
    _solver_classes_case(gpu, tmp_path, case, world, FS_HALO_P2P="1")


def _solver_classes_case(gpu, tmp_path, case, world, **env):
    """`python -m fenicssolver_amd.launch --nproc N script.py` with the reference-style solver classes:
    same field as the single-process run, gathered on every rank.  heat_p2: CG2 nodes decomposed as
    [owned vertices | owned edges | ghost vertices | ghost edges] with the indexed halo."""
    import test_gpu_parallel_api as T
    single = T.CASES[case]().solve().vector().get_local()
    r = _run(world, case, tmp_path, **env)
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()


@pytest.mark.parametrize("case,world,renumber", [("heat_file", 2, "1"), ("heat_file", 3, "0"), ("heat_file_p2", 2, "1")])
def test_file_mesh_under_several_ranks(gpu, tmp_path, monkeypatch, case, world, renumber):
    """An unstructured mesh FILE (data/mesh.xml) decomposed over the ranks, each part numbered in locality order (FS_RENUMBER=1)
    or in file order: T = 350 - 2.5 z as on one GPU."""
    import test_gpu_parallel_api as T
    monkeypatch.setenv("FS_RENUMBER", "0")
    one = T.CASES[case]()
    single = one.solve().vector().get_local()
    r = _run(world, case, tmp_path, FS_RENUMBER=renumber)
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()
    X = one.function_space.node_coordinates()
    assert np.abs(r["x"] - (350.0 - 2.5 * X[:, 2])).max() <= 1e-7


@pytest.mark.parametrize("case,world", [("heat_dist", 2), ("heat_dist", 3), ("heat_cn_dist", 2), ("elasticity_dist", 2), ("elasticity_dist", 3),
                                        ("elasticity_pfield_dist", 2)])
def test_distributed_box_mesh_no_global_host_mesh(gpu, tmp_path, case, world):
    """BoxMesh(distributed=True): every rank builds only its slab on the host (vertex planes it owns + one ghost plane each
    side), marks boundaries, evaluates coefficients and Dirichlet sets on it and keeps the local part of the result; the
    global field gathered from the ranks equals the single-process solve on the full mesh."""
    import test_gpu_parallel_api as T
    single = T.DIST_CASES[case]().solve().vector().get_local()       # one process: the slab is the whole box
    r = _run(world, case, tmp_path)
    assert int(r["n_local"]) < len(single) // (3 if "elasticity" in case else 1)
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()


@pytest.mark.parametrize("case,world", [("heat_p2_dist", 2), ("heat_p2_dist", 3), ("heat_p2_cn_dist", 2), ("elasticity_p2_dist", 2),
                                        ("elasticity_p2_dist", 3)])
def test_distributed_box_mesh_with_p2_spaces(gpu, tmp_path, case, world):
    """CG2 spaces on BoxMesh(distributed=True): the node plan is derived from this rank's cells alone
    (partition.build_p2_plan_local), the host keeps only its slab - its node count stays below 1.3 / world of the global one plus
    the ghost layers - and the field gathered by global keys (vertex id, end points of an edge) equals the one-process solve."""
    import test_gpu_parallel_api as T
    one = T.DIST_CASES[case]()
    single = one.solve().vector().get_local()
    V = one.function_space
    nc = V._ncomp
    nv = one.mesh.num_vertices()
    ed = V.edge_nodes().astype(np.int64)
    r = _run(world, case, tmp_path)
    S = single.reshape(-1, nc)
    scale = np.abs(S).max()
    assert len(r["vertex_gids"]) == nv and len(r["edge_keys"]) == len(ed)            # every node owned exactly once
    assert np.abs(r["vertex_values"] - S[r["vertex_gids"]]).max() <= 1e-8 * scale
    key = {(int(a), int(b)): k for k, (a, b) in enumerate(ed)}
    idx = np.array([key[(int(a), int(b))] for a, b in r["edge_keys"]])
    assert len(np.unique(idx)) == len(ed)
    assert np.abs(r["edge_values"] - S[nv + idx]).max() <= 1e-8 * scale
    # nothing of global size on a rank: owned slab + two ghost planes' worth of nodes
    nx, ny, nz = one.mesh._box[:3]
    ghost_layers = 2.0 * (V.num_nodes() / (nz + 1.0)) * 2.0
    assert int(r["n_local"]) <= V.num_nodes() / world * 1.3 + ghost_layers


@pytest.mark.parametrize("case,world", [("cavity_dist", 2), ("cavity_dist", 3), ("channel_dist", 2), ("channel_dist", 3)])
def test_navier_stokes_on_the_distributed_box_mesh(gpu, tmp_path, case, world):
    """Taylor-Hood on BoxMesh(distributed=True) (BASELINE configs[4] is a 4-GPU configuration): every rank
    builds only its slab on the host; the Newton loop, the boundary lists, the pressure pin, the pressure hierarchy (the device
    generates the whole box for the replicated pressure Laplacian, the pressure conditions are agreed over the ranks), the
    stress projection and the boundary force work on that slab; the Function keeps this rank's nodes.  Same field as the
    one-process solve, node by node through global keys; nothing of global size on a rank."""
    import test_gpu_parallel_api as T
    one = T.DIST_CASES[case]()
    single = one.solve().vector().get_local()
    W = one.function_space
    nv = one.mesh.num_vertices()
    ed = W.edge_nodes().astype(np.int64)
    r = _run(world, case, tmp_path)
    S = single.reshape(-1, 4)
    scale = np.abs(S).max()
    assert len(r["vertex_gids"]) == nv and len(r["edge_keys"]) == len(ed)            # every node owned exactly once
    assert np.abs(r["vertex_values"] - S[r["vertex_gids"]]).max() <= 1e-6 * scale
    key = {(int(a), int(b)): k for k, (a, b) in enumerate(ed)}
    idx = np.array([key[(int(a), int(b))] for a, b in r["edge_keys"]])
    assert len(np.unique(idx)) == len(ed)
    assert np.abs(r["edge_values"] - S[nv + idx]).max() <= 1e-6 * scale
    nx, ny, nz = one.mesh._box[:3]
    ghost_layers = 2.0 * (W.num_nodes() / (nz + 1.0)) * 2.0
    assert int(r["n_local"]) <= W.num_nodes() / world * 1.3 + ghost_layers
    if case == "channel_dist":
        sig = one.viscous_stress(one.w_current).node_values().reshape(-1)
        assert np.abs(r["sigma"] - sig).max() <= 1e-5 * np.abs(sig).max()
        force = np.array(one.calc_drag_and_lift(one.w_current, 2, 0, [1]))
        assert np.abs(r["force"] - force).max() <= 1e-6 * np.abs(force).max()


@pytest.mark.parametrize("case,world,p2p", [("cavity", 2, False), ("cavity", 3, False), ("channel", 2, False), ("radiation", 2, False),
                                            ("cavity", 3, True), ("channel", 2, True), ("cavity_thermal", 2, False), ("radiation_p2", 2, False)])
def test_navier_stokes_under_several_ranks(gpu, tmp_path, case, world, p2p):
    """Taylor-Hood on several ranks: block-4 matrix on the decomposed CG2 nodes, two-pass assembly of the owned rows,
    FGMRES with reduced multi-dots, halo exchange of the iterate inside the preconditioner, Schur-complement solves on
    the replicated pressure space (every rank applies the V-cycle of the global pressure Laplacian), Newton residual norm
    reduced over the ranks.  radiation: the scalar Newton loop (facet radiation terms + k(T)) on two ranks."""
    import test_gpu_parallel_api as T
    one = T.NS_CASES[case]()
    single = one.solve().vector().get_local()
    # p2p: every halo (block-4 indexed, pressure space, mass-matrix solves) and every all-reduce of up to 8 doubles through the
    # peer-to-peer kernels; the longer multi-dot reductions of FGMRES stay on ncclAllReduce
    r = _run(world, case, tmp_path, **(dict(FS_HALO_P2P="1") if p2p else {}))
    assert np.abs(r["x"] - single).max() <= 1e-6 * np.abs(single).max()
    if case == "channel":                 # viscous_stress on several ranks: nine decomposed mass-matrix solves
        sig = one.viscous_stress(one.w_current).vector().get_local()
        assert np.abs(r["sigma"] - sig).max() <= 1e-5 * np.abs(sig).max()


@pytest.mark.parametrize("world,mode", [(2, "distributed"), (3, "distributed"), (2, "replicated"), (3, "replicated"), (2, "schwarz")])
def test_amg_pcg_under_several_ranks(gpu, tmp_path, world, mode):
    """solve_amg on several ranks.  Default since round 4 ('distributed'): every rank builds the hierarchy of the UNDECOMPOSED
    operator, then attaches its own rows of the decomposed operator as the fine level (fs_amg_attach_distributed_fine): smoothing,
    residual, restriction and prolongation of level 0 on this rank's rows - ghost refresh before every fine product, the coarse
    right-hand side summed over the ranks -, levels >= 1 replicated; CG runs on the decomposed operator.  The V-cycle is the
    one-GPU V-cycle: the iteration count of one GPU whatever the number of parts, with the fine-level work divided by it
    (rank-local hierarchies have no coarse space coupling the parts: tools/amg_schwarz_probe.py counts 24 / 166 / 321 / 495
    iterations at 1 / 2 / 4 / 8 slabs of BASELINE configs[2]).  'replicated' (round 3): every rank solves the whole gathered
    system.  'schwarz': rank-local hierarchies.  Same displacement field as the single-GPU AMG-PCG every way."""
    import test_gpu_parallel_api as T
    one = T.CASES["elasticity"]()
    single = one.solve().vector().get_local()
    its = one.last_solve_stats["iterations"]
    r = _run(world, "elasticity", tmp_path, **({} if mode == "distributed" else {"FS_TEST_AMG_DECOMPOSITION": mode}))
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()
    if mode in ("replicated", "distributed"):
        assert abs(int(r["iterations"]) - its) <= 1
    else:
        assert its <= int(r["iterations"]) < 200
    vm = one.von_Mises(one.w_current).vector().get_local()
    assert np.abs(r["von_mises"] - vm).max() <= 1e-7 * np.abs(vm).max()


@pytest.mark.parametrize("world", [2, 3])
def test_vector_p2_elasticity_under_several_ranks(gpu, tmp_path, world):
    """The reference's elasticity example space (vector P2) decomposed: block-3 matrix on the decomposed CG2 nodes, indexed halo
    of vertex and edge nodes, additive-Schwarz AMG-CG; same displacement field as one GPU."""
    import test_gpu_parallel_api as T
    one = T.CASES["elasticity_p2"]()
    single = one.solve().vector().get_local()
    r = _run(world, "elasticity_p2", tmp_path)
    assert np.abs(r["x"] - single).max() <= 1e-7 * np.abs(single).max()
    vm = one.von_Mises(one.w_current).vector().get_local()                       # the projection runs decomposed as well
    assert np.abs(r["von_mises"] - vm).max() <= 1e-6 * np.abs(vm).max()


@pytest.mark.parametrize("p2p", ["works", "openfail", "lost", "late:12"])
def test_bench_under_the_drivers_launcher(gpu, tmp_path, p2p):
    """bench.py exactly as the driver starts it for N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench.py
    --gpus N): env rendezvous of the RCCL id through fenicssolver_amd/rendezvous.py (no torch import in bench.py), barrier
    and max-over-ranks over the communicator, one JSON line from rank 0.  One GPU here: tests/shim/on_device0.py pins every
    rank to device 0 and libfakerccl.so stands in for RCCL.  The warm-up tries the three variants (two recurrences over RCCL,
    the peer-to-peer exchange); when the peer-to-peer exchange cannot be set up (openfail) or its data never arrives (lost: every
    wait times out) the line still comes, from an RCCL variant, with the reason recorded."""
    import json
    shim = os.path.join(ROOT, "tests", "shim", "libfakerccl.so")
    PORT[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(PORT[0]), os.path.join(ROOT, "tests", "shim", "on_device0.py"), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "23", "--extra", "strong", "--strong-n", "23"]
    env = dict(os.environ, FS_RCCL_PATH=shim)
    if p2p != "works":
        env.update(FS_P2P_TEST=p2p, FS_P2P_TIMEOUT_MS="100")
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "DOF/s"
    assert d["config"]["n_dof"] == 24 * 24 * 48 and d["config"]["true_rel_residual"] <= 1.1e-8
    assert "roofline" in d and d["value"] > 0
    trial = d["config"]["recurrence_trial_ms_per_step"]
    assert set(trial) - {"single_reduction+p2p (timed steps)"} == {"single_reduction", "pipelined", "single_reduction+p2p"}
    assert all(isinstance(trial[k], float) for k in ("single_reduction", "pipelined"))
    if p2p == "works":
        assert isinstance(trial["single_reduction+p2p"], float)
        assert "dof_per_s" in d["strong"]["single_reduction+p2p"] and d["strong"]["single_reduction+p2p"]["true_rel_residual"] <= 1.1e-8
    elif p2p.startswith("late"):         # it won (or not) the trial, then broke: the steps were timed again over RCCL
        assert isinstance(trial["single_reduction+p2p"], float)
        if "single_reduction+p2p (timed steps)" in trial:
            assert trial["single_reduction+p2p (timed steps)"].startswith("failed")
        assert d["config"]["recurrence"] in ("single_reduction", "pipelined")
    else:
        # (openfail: a mapping is refused; lost: the mappings open but nothing arrives - caught by the self-test of
        # fs_space_enable_p2p_halo, so both end as "unavailable" before any solve runs on the transport)
        assert trial["single_reduction+p2p"].startswith("unavailable"), trial
        assert d["config"]["recurrence"] in ("single_reduction", "pipelined")
        assert "unavailable" in d["strong"]["single_reduction+p2p"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "from torch" not in src


def test_bench_taylor_hood_leg_runs_on_the_distributed_mesh(gpu, tmp_path):
    """`bench.py --gpus N` adds BASELINE configs[4] (`configs4_th`, automatic at N = 4): since round 4 on BoxMesh(distributed=True) -
    rank 0's host holds its slab of the nodes only.  Two ranks on one GPU through the stand-in, a small cube."""
    import json
    shim = os.path.join(ROOT, "tests", "shim", "libfakerccl.so")
    PORT[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(PORT[0]), os.path.join(ROOT, "tests", "shim", "on_device0.py"), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--cells", "15", "--extra", "th", "--th-n", "6", "--recurrence", "single_reduction"]
    p = subprocess.run(cmd, env=dict(os.environ, FS_RCCL_PATH=shim), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert "extra_legs_error" not in d, d.get("extra_legs_error")
    th = d["configs4_th"]
    n_nodes = 13 ** 3
    assert th["time_steps"] == 10 and th["n_dof"] == 3 * n_nodes + 7 ** 3
    assert th["host_nodes_rank0"] < 0.75 * n_nodes and "distributed box mesh" in th["workload"]
    assert th["newton_residuals_last_step"][-1] <= 1e-9 * max(th["newton_residuals_last_step"][0], 1e-300) or th["newton_residuals_last_step"][-1] <= 1e-10
    assert 0.0 < th["max_speed"] <= 1.0 + 1e-12


@pytest.mark.parametrize("case,world", [("elasticity_fine", 2), ("elasticity_fine", 3), ("elasticity_fine_dist", 2), ("elasticity_fine_dist", 4)])
def test_amg_with_a_distributed_fine_level(gpu, tmp_path, case, world):
    """solve_amg on several ranks with a hierarchy of SEVERAL levels (37 x 7 x 7 nodes) - the fine level
    works on each rank's rows (fs_amg_attach_distributed_fine), the coarse levels are replicated; replicated host mesh and
    BoxMesh(distributed=True).  The V-cycle is the one-GPU V-cycle: same iteration count (+-1), same displacement."""
    import test_gpu_parallel_api as T
    one = (T.DIST_CASES if case.endswith("_dist") else T.CASES)[case]()
    single = one.solve().vector().get_local()
    its = one.last_solve_stats["iterations"]
    assert one.last_solve_stats["amg_levels"] >= 2
    r = _run(world, case, tmp_path)
    assert str(r["amg_decomposition"]) == "distributed" and int(r["amg_levels"]) >= 2
    assert abs(int(r["iterations"]) - its) <= 1
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()


@pytest.mark.parametrize("case,world", [("elasticity_wide_dist", 2), ("elasticity_wide_dist", 3)])
def test_block_row_dictionary_on_a_decomposed_fine_level(gpu, tmp_path, case, world):
    """The 3 x 3 block-row dictionary (k_dict_spmv3) on the rank-local fine level of the distributed AMG: the rows next to a slab's
    cut reach into the ghost planes behind the owned nodes - other offsets, other plans, same classes machinery.  The size limit
    of the block form is lifted (FS_DICT3_MIN_NODES=0) so that this small cantilever takes it: same iteration count and
    displacement as one GPU on the streaming product."""
    import test_gpu_parallel_api as T
    one = (T.DIST_CASES if case.endswith("_dist") else T.CASES)[case]()
    single = one.solve().vector().get_local()
    its = one.last_solve_stats["iterations"]
    assert one.last_solve_stats.get("row_classes", 0) == 0
    r = _run(world, case, tmp_path, FS_DICT3_MIN_NODES="0")
    assert str(r["amg_decomposition"]) == "distributed" and int(r["row_classes"]) > 0
    assert abs(int(r["iterations"]) - its) <= 1
    assert np.abs(r["x"] - single).max() <= 1e-8 * np.abs(single).max()
