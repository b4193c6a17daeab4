"""Worker of tests/test_gpu_multirank.py: one rank of a multi-rank run whose ranks SHARE ONE GPU (launched with
--devices 0,0,..; FS_RCCL_PATH = tests/shim/libfakerccl.so).  Started by fenicssolver_amd.launch.
Writes what rank 0 gathered to the .npz named on the command line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fenicssolver_amd import backend as B, partition, parallel    # noqa: E402

out_path, case = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
result = {}

if case == "box":
    # the bench path: device box-slab generator, closed-form halo plan, Jacobi-CG
    nx, ny, nz, axis = [int(v) for v in os.environ.get("FS_TEST_BOX", "9,7,23,0").split(",")]
    parallel.ensure_comm()
    zr = partition.slab_ranges(nz + 1, world)[rank]
    mesh = B.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0), zplanes=zr)
    V = B.DeviceSpace(mesh, 1)
    lay = partition.slab_layout(nx, ny, nz, zr, rank, world)
    dofs, vals = partition.slab_dirichlet(nx, ny, nz, lay, axis)
    V.set_halo(lay["neighbors"], lay["send_lists"], lay["recv_counts"])
    A = B.DeviceMatrix(V)
    b = B.DeviceVector(V.n_owned)
    x = B.DeviceVector(V.n_local)
    A.assemble(stiffness=20.0)
    B.assemble_vector(V, b, source=3.0)
    A.apply_dirichlet(b, dofs, vals, symmetric=True)
    st = B.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    x1 = x.get()[:lay["n_owned"]].copy()
    # a second solve on the same space from a perturbed iterate (restart path, captured batches re-used)
    xg = x.get().copy()
    xg[:lay["n_owned"]] *= 1.0 + 1e-3 * np.cos(np.asarray(lay["l2g"][:lay["n_owned"]], dtype=float))
    x.set(xg)
    st2 = B.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000, nonzero_guess=True)
    n_glob = (nx + 1) * (ny + 1) * (nz + 1)
    full = parallel.gather_owned(x1, lay["l2g"][:lay["n_owned"]], n_glob)
    full2 = parallel.gather_owned(x.get()[:lay["n_owned"]], lay["l2g"][:lay["n_owned"]], n_glob)
    flags = parallel.allgather_values(np.array([float(st["fused_iteration"]), float(st["row_classes"] > 0), float(st2["fused_iteration"])]))
    if rank == 0:
        result = dict(x=full, iterations=st["iterations"], converged=st["converged"], true_res=st["true_rel_residual"],
                      x2=full2, iterations2=st2["iterations"], true_res2=st2["true_rel_residual"],
                      fused=np.array([f[0] for f in flags]), dictionary=np.array([f[1] for f in flags]), fused2=np.array([f[2] for f in flags]))
    parallel.barrier()
    parallel.finalize()
elif case == "box_stress":
    # many solves back to back on one decomposed space: the sequence numbers, slots and captured batches of the peer-to-peer
    # iteration over hundreds of exchanges, with a changing right-hand side (iteration counts differ from solve to solve)
    nx, ny, nz, axis = 9, 7, 23, 0
    parallel.ensure_comm()
    zr = partition.slab_ranges(nz + 1, world)[rank]
    mesh = B.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0), zplanes=zr)
    V = B.DeviceSpace(mesh, 1)
    lay = partition.slab_layout(nx, ny, nz, zr, rank, world)
    dofs, vals = partition.slab_dirichlet(nx, ny, nz, lay, axis)
    V.set_halo(lay["neighbors"], lay["send_lists"], lay["recv_counts"])
    A = B.DeviceMatrix(V)
    b = B.DeviceVector(V.n_owned)
    x = B.DeviceVector(V.n_local)
    its, res = [], []
    for k in range(40):
        A.assemble(stiffness=20.0, mass=0.1 * (k % 3))
        B.assemble_vector(V, b, source=3.0 + k)
        A.apply_dirichlet(b, dofs, vals * (1.0 + 0.01 * k), symmetric=True)
        st = B.krylov_solve(A, b, x, rtol=10.0 ** -(6 + k % 6), max_iter=5000)
        assert st["converged"] == 1, st
        its.append(st["iterations"])
        res.append(st["true_rel_residual"])
    full = parallel.gather_owned(x.get()[:lay["n_owned"]], lay["l2g"][:lay["n_owned"]], (nx + 1) * (ny + 1) * (nz + 1))
    if rank == 0:
        result = dict(x=full, iterations=np.array(its), true_res=np.array(res))
    parallel.barrier()
    parallel.finalize()
elif case == "box3":
    # three components per node (3x3-block operator of linear elasticity + mass), Jacobi-CG: the dof-level halo of a vector space
    nx, ny, nz = 6, 5, 17
    parallel.ensure_comm()
    zr = partition.slab_ranges(nz + 1, world)[rank]
    mesh = B.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), (1.0, 0.8, 2.0), zplanes=zr)
    V = B.DeviceSpace(mesh, 3)
    lay = partition.slab_layout(nx, ny, nz, zr, rank, world)
    sends = [(3 * np.asarray(sl, dtype=np.int64)[:, None] + np.arange(3)).ravel().astype(np.int32) for sl in lay["send_lists"]]
    V.set_halo(lay["neighbors"], sends, [3 * c for c in lay["recv_counts"]])
    A = B.DeviceMatrix(V)
    A.assemble(lame=(1.0, 1.5), mass=4.0)
    g = np.repeat(lay["l2g"][:lay["n_owned"]], 3) * 3 + np.tile(np.arange(3), lay["n_owned"])
    rhs = np.sin(0.37 * g) + 0.2
    b = B.DeviceVector(V.n_owned, rhs)
    x = B.DeviceVector(V.n_local)
    st = B.krylov_solve(A, b, x, rtol=1e-10, max_iter=5000)
    full = parallel.gather_owned(x.get()[:V.n_owned], g, 3 * (nx + 1) * (ny + 1) * (nz + 1))
    if rank == 0:
        result = dict(x=full, iterations=st["iterations"], converged=st["converged"], true_res=st["true_rel_residual"])
    parallel.barrier()
    parallel.finalize()
else:
    # the solver API on several ranks (parallel.py)
    import test_gpu_parallel_api as T
    solver = (T.CASES.get(case) or T.DIST_CASES.get(case) or T.NS_CASES[case])()
    if os.environ.get("FS_TEST_AMG_DECOMPOSITION"):
        sp = solver.solver_settings.setdefault('solver_parameters', {}) or {}
        sp['amg_decomposition'] = os.environ["FS_TEST_AMG_DECOMPOSITION"]
        solver.solver_settings['solver_parameters'] = sp
    u = solver.solve()
    assert solver.function_space.localizer() is not None and parallel.world()[1] == world
    if case in T.DIST_CASES:
        mesh = solver.mesh
        assert mesh.is_distributed() and mesh.num_vertices() < (mesh._box[0] + 1) * (mesh._box[1] + 1) * (mesh._box[2] + 1)
        # the host mesh of this rank is exactly what the device generated for its slab
        xyz, cells, gid = mesh.device().get(True, True, True)
        assert np.array_equal(xyz, mesh.coordinates()) and np.array_equal(cells, mesh.cells()) and np.array_equal(gid, mesh.global_vertex_ids())
        if solver.function_space.degree() == 2:
            # CG2: nodes are named by global keys (vertex id / the two global end points of an edge); no global numbering exists
            vg, vv, ek, ev = parallel.gather_nodes(u)
            extra = {}
            if case == "channel_dist":        # the stress projection and the boundary force on the distributed pressure space
                ploc = solver.function_space.pressure_space().localizer()
                sig = solver.viscous_stress(u).node_values().reshape(-1, 9)
                extra["sigma"] = parallel.gather_owned(sig[:ploc.n_owned].reshape(-1), ploc.owned_gids(), ploc.n_global, 9)
                extra["force"] = np.array(solver.calc_drag_and_lift(u, 2, 0, [1]))
            if rank == 0:
                result = dict(vertex_gids=vg, vertex_values=vv, edge_keys=ek, edge_values=ev, iterations=solver.last_solve_stats["iterations"],
                              n_local=solver.function_space.num_nodes(), **extra)
        else:
            full = parallel.gather_function(u)          # [n_global (, 3)] on every rank
            extra = {}
            if "amg_decomposition" in solver.last_solve_stats:
                extra["amg_decomposition"] = np.array(solver.last_solve_stats["amg_decomposition"])
                extra["amg_levels"] = np.array(solver.last_solve_stats.get("amg_levels", 0))
                extra["row_classes"] = np.array(solver.last_solve_stats.get("row_classes", 0))
            if rank == 0:
                result = dict(x=np.asarray(full).reshape(-1), iterations=solver.last_solve_stats["iterations"], n_local=mesh.num_vertices(), **extra)
    else:
        extra = {}
        if case.startswith("elasticity") and hasattr(solver, "von_Mises"):
            its = solver.last_solve_stats["iterations"]
            extra["von_mises"] = solver.von_Mises(u).vector().get_local()       # the L2 projection on the decomposed P1 space
            solver.last_solve_stats["iterations"] = its
        if case == "channel":                                    # the fluid-stress projection on the decomposed pressure space
            its = solver.last_solve_stats["iterations"]
            extra["sigma"] = solver.viscous_stress(u).vector().get_local()
            solver.last_solve_stats["iterations"] = its
        if "amg_decomposition" in solver.last_solve_stats:
            extra["amg_decomposition"] = np.array(solver.last_solve_stats["amg_decomposition"])
            extra["amg_levels"] = np.array(solver.last_solve_stats.get("amg_levels", 0))
            extra["row_classes"] = np.array(solver.last_solve_stats.get("row_classes", 0))
        if rank == 0:
            result = dict(x=u.vector().get_local(), iterations=solver.last_solve_stats["iterations"], **extra)
    parallel.barrier()
    parallel.finalize()

if rank == 0:
    np.savez(out_path, **result)
