#!/usr/bin/env python3
"""bench.py — DOF/s (assemble + CG solve to 1e-8) on 3D P1 heat conduction.

One "step" = one pass of the hot path over the synthetic mesh already resident in
HBM: numeric assembly of A (5.8 M tets at N=1), Dirichlet elimination, Jacobi-PCG
from x0 = 0 to ||b - A x|| <= 1e-8 ||b||  (what SolverBase.solve_linear_problem does
through DOLFIN/PETSc, FenicsSolver/SolverBase.py:592-613).  Mesh generation and the
sparsity pattern (DOLFIN builds it inside the first assemble) are set-up, timed and
reported separately as `symbolic_ms`.

  python bench.py                       # N=1: BASELINE.json configs[1], 1 M DOF unit cube
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      # one rank per GPU (RCCL); the driver's launcher
  python -m fenicssolver_amd.launch --nproc N bench.py --gpus N      # the package's own launcher, same environment

Multi-GPU: z-slab domain decomposition, halo exchange + one 3-double all-reduce per CG
iteration (fenicssolver_amd/csrc/fs_comm.hip).  Default scaling is WEAK: every GPU
owns 100 vertex planes of 100x100 (1 M DOF), the bar grows along z and the Dirichlet pair
sits on the x-faces so the conditioning does not change with N.  `--scaling strong --cells 215`
splits the 10 M-DOF cube instead.

At N > 1 the SAME command also runs, after the timed weak leg and outside its timing, the legs the north_star
is stated on, and adds them to the one JSON line:
  "comm"    latency of the two collectives of an iteration (3-double all-reduce, ghost refresh) as the solver issues them
  "strong"  the 10 M-DOF cube (n = 215) SPLIT over the N GPUs, with both CG recurrences (single-reduction and
            pipelined); the strong-scaling anchor is the N = 1 line's roofline.dof_per_s (same cube on one GPU)
  "configs3_p2" (N = 8, or --extra p2)  BASELINE configs[3]: P2 heat conduction, unit cube n = 107, 9.94 M DOF, z-slabs
  "configs4_th" (N = 4, or --extra th)  BASELINE configs[4]: Taylor-Hood lid-driven cavity n = 43, 10 backward-Euler steps
`--workload p2 | th` makes one of those the timed leg itself (any N).  The extra legs run under a watchdog: if one of
them does not finish in its time budget the line is printed with what has been measured (`extra_legs_timed_out`).

bench.py imports no torch: the launcher only has to export RANK / WORLD_SIZE / LOCAL_RANK; the RCCL unique id travels
through fenicssolver_amd/rendezvous.py and the timing barrier / max-over-ranks run over the communicator itself.

The JSON line's `roofline` is the dominant kernel (the SpMV fused with the CG dot products) measured on the
HBM-RESIDENT 10 M-DOF problem of the same family; the same kernel on the 1 M-DOF step workload runs out of the
256 MiB Infinity Cache, so its byte rate is reported separately (`dominant_kernel_on_step_workload`) and is not an
HBM fraction.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fenicssolver_amd import backend as B  # noqa: E402
from fenicssolver_amd import partition, parallel  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable


VARIANTS = ("single_reduction", "pipelined", "single_reduction+p2p")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # (--cells under torch.distributed.run: its own parser claims every abbreviation of --nnodes / --nproc-per-node)
    ap.add_argument("--n", "--cells", dest="n", type=int, default=99, help="cells per axis of the (per-GPU) cube")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--bc-axis", type=int, default=None, help="axis of the Dirichlet face pair (default 2 at N=1, 0 at N>1)")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--workload", choices=("p1", "p2", "th"), default="p1",
                    help="p1: BASELINE configs[1] family (default); p2: configs[3] (P2, n=107); th: configs[4] (Taylor-Hood cavity n=43)")
    ap.add_argument("--extra", default="auto", help="extra legs at N>1: auto | none | comma list of strong,p2,th")
    ap.add_argument("--extra-budget", type=float, default=240.0, help="seconds the extra legs may take before the watchdog prints the line")
    ap.add_argument("--recurrence", choices=("auto",) + VARIANTS, default="auto",
                    help="CG recurrence of the timed leg at N>1 (auto: the faster of the two in a warm-up trial, agreed over the ranks)")
    ap.add_argument("--mesh", choices=("structured", "shuffled", "renumbered"), default="structured",
                    help="N=1 only: shuffled = random vertex + cell permutation of the cube uploaded as a file mesh would be, "
                         "renumbering disabled; renumbered = the same upload with the library's locality renumbering")
    ap.add_argument("--strong-n", type=int, default=215, help="cube of the strong extra leg (tests shrink it)")
    ap.add_argument("--p2-n", type=int, default=107, help="cube of the configs[3] extra leg")
    ap.add_argument("--th-n", type=int, default=43, help="cube of the configs[4] extra leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-case", action="store_true", help="skip the extra 10 M-DOF roofline measurement")
    ap.add_argument("--no-cache-free-case", action="store_true",
                    help="skip the 86 M-DOF cube (n = 440) behind the 10 M-DOF roofline measurement: the same kernels where the 256 MiB Infinity Cache cannot help")
    return ap.parse_args()


class Problem:
    """One rank's share of the box heat problem, resident on the device."""

    def __init__(self, nx, ny, nz, p1, zplanes, axis, rank, world):
        P = (nx + 1) * (ny + 1)
        zb, ze = zplanes
        t0 = time.perf_counter()
        self.mesh = B.DeviceMesh.box(nx, ny, nz, (0.0, 0.0, 0.0), p1, zplanes=zplanes)
        B.synchronize()
        t1 = time.perf_counter()
        self.V = B.DeviceSpace(self.mesh, 1)
        B.synchronize()
        t2 = time.perf_counter()
        self.mesh_ms = (t1 - t0) * 1e3
        self.symbolic_ms = (t2 - t1) * 1e3
        # the FIRST pattern build of a process also loads the code object of the sort / scan kernels (12 MB of rocprim
        # instantiations, about 35 ms: tools/probes/symbolic_cold_warm.py); a second build of the same pattern shows the work itself
        if Problem.first_build and world == 1:
            Problem.first_build = False
            t3 = time.perf_counter()
            V2 = B.DeviceSpace(self.mesh, 1)
            B.synchronize()
            self.symbolic_warm_ms = (time.perf_counter() - t3) * 1e3
            del V2
        lay = partition.slab_layout(nx, ny, nz, zplanes, rank, world)
        n_own = lay["n_owned"]
        assert self.V.n_owned == n_own and self.V.n_local == lay["n_local"]
        # Dirichlet dofs in local numbering (owned planes first, then lower, then upper ghost plane)
        self.dofs, self.vals = partition.slab_dirichlet(nx, ny, nz, lay, axis)
        if world > 1:
            self.V.set_halo(lay["neighbors"], lay["send_lists"], lay["recv_counts"])
        self.A = B.DeviceMatrix(self.V)
        self.b = B.DeviceVector(self.V.n_owned)
        self.x = B.DeviceVector(self.V.n_owned)
        self.n_owned = n_own

    pipelined = None     # None: the library's rule (the single-reduction recurrence)
    first_build = True
    symbolic_warm_ms = None

    def step(self, rtol):
        """assemble + Dirichlet + CG.  Returns (stats, t_assemble_ms)."""
        t0 = time.perf_counter()
        self.A.assemble(stiffness=20.0)
        self.b.fill(0.0)
        self.A.apply_dirichlet(self.b, self.dofs, self.vals, symmetric=True)
        t1 = time.perf_counter()
        st = B.krylov_solve(self.A, self.b, self.x, rtol=rtol, max_iter=20000, precond="jacobi", pipelined=self.pipelined)
        if st["converged"] != 1:
            raise RuntimeError("CG did not converge: %r" % (st,))
        return st, (t1 - t0) * 1e3


class ShuffledProblem:
    """The cube of BASELINE configs[1] as a mesh FILE would deliver it: vertices and cells in random order, uploaded through
    fs_mesh_create (no structure for the DIA slices to find, no locality in the numbering).  renumber: the upload goes through
    the library's locality order first (fs_mesh_locality_order: Morton order of the vertices, cells by their first vertex) -
    what fem.Mesh does for file meshes; results are compared in the ORIGINAL numbering either way."""

    pipelined = False

    def __init__(self, n, axis, renumber, seed=0):
        xyz, cells, _ = B.DeviceMesh.box(n, n, n).get()
        nv = len(xyz)
        rng = np.random.default_rng(seed)
        new_of_old = rng.permutation(nv).astype(np.int64)            # the "file" numbering
        co = np.empty_like(xyz)
        co[new_of_old] = xyz
        ce = np.sort(new_of_old[cells], axis=1)[rng.permutation(len(cells))].astype(np.int32)
        del cells
        self.file_of_structured = new_of_old
        t0 = time.perf_counter()
        if renumber:
            self.mesh, vorder, corder = B.DeviceMesh.renumbered(co, ce)      # vorder[new] = file id; ordered and built on the device
            dev_of_file = np.empty(nv, dtype=np.int64)
            dev_of_file[vorder] = np.arange(nv)
            self.dev_of_file = dev_of_file
        else:
            self.dev_of_file = np.arange(nv)
            self.mesh = B.DeviceMesh(co, ce)
        B.synchronize()
        t1 = time.perf_counter()
        self.V = B.DeviceSpace(self.mesh, 1)
        B.synchronize()
        t2 = time.perf_counter()
        self.mesh_ms, self.symbolic_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        c = co[:, axis]
        lo, hi = np.nonzero(c == 0.0)[0], np.nonzero(c == 1.0)[0]
        self.dofs = self.dev_of_file[np.concatenate([lo, hi])].astype(np.int32)
        self.vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
        self.A = B.DeviceMatrix(self.V)
        self.b = B.DeviceVector(self.V.n_owned)
        self.x = B.DeviceVector(self.V.n_owned)
        self.n_owned = self.V.n_owned

    step = Problem.step

    def to_structured(self, x_dev):
        """device numbering -> the lexicographic numbering of the structured cube (what the CPU leg solves in)."""
        return np.asarray(x_dev)[self.dev_of_file[self.file_of_structured]]


class P2Problem:
    """One rank's z-slab of BASELINE configs[3]: P2 heat conduction on the unit cube, CG2 nodes numbered on the device as
    [owned vertices | owned edges | ghost vertices | ghost edges]; the node-level halo plan comes from this rank's cells
    alone (partition.build_p2_plan_local) - nothing of global size is built on any rank."""

    pipelined = None

    def __init__(self, n, zplanes, axis, rank, world):
        t0 = time.perf_counter()
        self.mesh = B.DeviceMesh.box(n, n, n, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), zplanes=zplanes)
        B.synchronize()
        t1 = time.perf_counter()
        self.V = B.DeviceSpace(self.mesh, 1, degree=2)
        B.synchronize()
        t2 = time.perf_counter()
        self.mesh_ms, self.symbolic_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        xyz, cells, gid = self.mesh.get()
        edges = self.V.edges().astype(np.int64)
        nv = len(xyz)
        if world > 1:
            lay = partition.slab_layout(n, n, n, zplanes, rank, world)
            P = lay["plane_size"]
            owner = np.full(nv, rank, dtype=np.int32)
            off = lay["n_owned"]
            for q in lay["neighbors"]:               # ghost planes follow the owned ones: lower neighbour first
                owner[off:off + P] = q
                off += P
            assert np.array_equal(gid, lay["l2g"])
            plan = partition.build_p2_plan_local(cells, gid, owner, rank, lay["neighbors"], edges, (n + 1) ** 3)
            assert plan.n_owned_nodes == self.V.n_owned
            self.V.set_halo(plan.neighbors, plan.send_lists, plan.recv_counts, recv_lists=plan.recv_lists)
            nov, noe = plan.node_of_vertex, plan.node_of_edge
        else:
            nov, noe = np.arange(nv), nv + np.arange(len(edges))
        # Dirichlet nodes (ghosts included): vertices and edge mid-points on the two faces normal to `axis`
        cv = xyz[:, axis]
        ce = 0.5 * (xyz[edges[:, 0], axis] + xyz[edges[:, 1], axis])
        lo = np.concatenate([nov[cv == 0.0], noe[ce == 0.0]])
        hi = np.concatenate([nov[cv == 1.0], noe[ce == 1.0]])
        self.dofs = np.concatenate([lo, hi]).astype(np.int32)
        self.vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
        self.A = B.DeviceMatrix(self.V)
        self.b = B.DeviceVector(self.V.n_owned)
        self.x = B.DeviceVector(self.V.n_owned)
        self.n_owned = self.V.n_owned
        # exact solution at the owned nodes (the linear profile is in the P2 space): the parity check of the leg
        co = np.empty(self.V.n_local)
        co[nov], co[noe] = cv, ce
        self.exact_owned = 350.0 - 50.0 * co[:self.n_owned]

    step = Problem.step


class P2FileOrderProblem:
    """BASELINE configs[3]'s operator on the cube as a mesh FILE would deliver it (one GPU): vertices and cells randomly permuted,
    uploaded through fs_mesh_create (renumber: through the library's locality order first, as fem.Mesh does for file meshes); CG2 space
    on it - nothing for DIA slices or the row dictionary to find: the streaming SELL product every Gmsh / FreeCAD P2 case takes."""

    pipelined = False

    def __init__(self, n, axis, renumber, seed=0):
        xyz, cells, _ = B.DeviceMesh.box(n, n, n).get()
        nv = len(xyz)
        rng = np.random.default_rng(seed)
        new_of_old = rng.permutation(nv).astype(np.int64)
        co = np.empty_like(xyz)
        co[new_of_old] = xyz
        ce = np.sort(new_of_old[cells], axis=1)[rng.permutation(len(cells))].astype(np.int32)
        del cells
        t0 = time.perf_counter()
        if renumber:
            self.mesh, vorder, _ = B.DeviceMesh.renumbered(co, ce)
            co = co[vorder]                      # device vertex order
        else:
            self.mesh = B.DeviceMesh(co, ce)
        B.synchronize()
        t1 = time.perf_counter()
        self.V = B.DeviceSpace(self.mesh, 1, degree=2)
        B.synchronize()
        t2 = time.perf_counter()
        self.mesh_ms, self.symbolic_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        edges = self.V.edges().astype(np.int64)
        cv = co[:, axis]
        cm = 0.5 * (cv[edges[:, 0]] + cv[edges[:, 1]])
        c = np.concatenate([cv, cm])             # node = vertex, then edge in the space's order (one GPU)
        lo, hi = np.nonzero(c == 0.0)[0], np.nonzero(c == 1.0)[0]
        self.dofs = np.concatenate([lo, hi]).astype(np.int32)
        self.vals = np.concatenate([np.full(len(lo), 350.0), np.full(len(hi), 300.0)])
        self.A = B.DeviceMatrix(self.V)
        self.b = B.DeviceVector(self.V.n_owned)
        self.x = B.DeviceVector(self.V.n_owned)
        self.n_owned = self.V.n_owned
        self.exact_owned = 350.0 - 50.0 * c

    step = Problem.step


def p2_global_dofs(n):
    return (n + 1) ** 3 + 3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n ** 3      # vertices + axis, face-diagonal and body-diagonal edges


def timed_steps(prob, rtol, steps, barrier, reduce=True):
    barrier()
    t0 = time.perf_counter()
    asm_ms, stats = 0.0, None
    for _ in range(steps):
        stats, t_asm = prob.step(rtol)
        asm_ms += t_asm
    barrier()
    elapsed = time.perf_counter() - t0
    if reduce:
        elapsed = parallel.max_over_ranks(elapsed)
    return elapsed, asm_ms / steps, stats


def set_variant(prob, name):
    """CG recurrence (+ transport of the ghost refresh / all-reduce) of the next steps.  '+p2p': the peer-to-peer exchange
    (fs_space_enable_p2p_halo: stores into the neighbours' hipIpc-mapped buffers, four kernels per iteration on one stream);
    otherwise RCCL send / recv + ncclAllReduce.  Collective: every rank calls it alike."""
    want = name.endswith("+p2p")
    if want != getattr(prob, "p2p", False):
        prob.V.enable_p2p_halo(want)         # raises on every rank alike when a mapping failed (the ranks agree inside)
        prob.p2p = want
    prob.pipelined = name.startswith("pipelined")


def all_ranks_ok(ok):
    """Agreement over RCCL proper (ncclAllGather), never over the transport under test."""
    flags = B.comm_allgather([0.0 if ok else 1.0], 1)
    return float(np.sum(flags)) == 0.0


def choose_recurrence(prob, rtol, world, requested, barrier, steps=2):
    """At N > 1 the variants are run warm in the warm-up phase and the fastest (max over the ranks, so every rank decides
    alike) carries the timed steps.  A variant that fails on any rank (hipIpc mapping refused, a peer-to-peer wait timed
    out) is dropped on all of them.  Returns (name, {name: ms per step | error})."""
    if world == 1:
        prob.pipelined = False
        return "single_reduction", {}
    if requested != "auto":
        set_variant(prob, requested)
        return requested, {}
    os.environ.setdefault("FS_P2P_TIMEOUT_MS", "4000")
    trial, report, x_ref, first = {}, {}, None, None
    for name in VARIANTS:
        err = None
        try:
            set_variant(prob, name)
        except B.BackendError as e:          # agreed inside the library: every rank is here
            report[name] = "unavailable: " + str(e)[:200]
            prob.p2p = False
            continue
        try:
            prob.step(rtol)
            t = timed_steps(prob, rtol, steps, barrier, reduce=False)[0] / steps
            if name.endswith("+p2p") and not getattr(prob.V, "_p2p", False):
                err, prob.p2p = "a peer-to-peer wait timed out: the library fell back to RCCL mid-solve", False
        except Exception as e:               # this rank only, perhaps: the ranks compare notes below
            err, t = repr(e)[:200], 0.0
        if err is None:                      # every variant must reproduce the first one's field (a transport that delivers stale
            x = prob.x.get()[:prob.n_owned]  # ghost values may still "converge")
            if x_ref is None:
                x_ref = x
            elif not np.abs(x - x_ref).max() <= 1e-5 * np.abs(x_ref).max():
                err = "solution differs from the %s run by %.3g" % (first, float(np.abs(x - x_ref).max()))
        if not all_ranks_ok(err is None):
            report[name] = "failed: " + (err or "on another rank")
            if getattr(prob, "p2p", False):
                set_variant(prob, "single_reduction")
            continue
        first = first or name
        trial[name] = float(np.max(B.comm_allgather([t], 1)))
        report[name] = round(trial[name] * 1e3, 4)
    best = min(trial, key=trial.get)
    set_variant(prob, best)
    return best, report


def timed_steps_with_fallback(prob, rtol, steps, barrier, name, report, world, warmup=0):
    """The timed steps; should the peer-to-peer variant fail AFTER it won the trial (a wait timing out on any rank), every rank
    falls back to the best RCCL variant and the steps are timed again - the line always comes, and says what happened."""
    if world == 1 or not name.endswith("+p2p"):
        for _ in range(warmup):
            prob.step(rtol)
        return timed_steps(prob, rtol, steps, barrier) + (name,)
    err, out = None, None
    try:
        for _ in range(warmup):
            prob.step(rtol)
        out = timed_steps(prob, rtol, steps, barrier, reduce=False)
        if not getattr(prob.V, "_p2p", False):      # backend._with_p2p_fallback: the ranks agreed to leave the transport mid-solve
            err = "a peer-to-peer wait timed out during a solve: the library turned the exchange off and solved again over RCCL"
            prob.p2p = False
    except Exception as e:
        err = repr(e)[:200]
    if all_ranks_ok(err is None):
        return (float(np.max(B.comm_allgather([out[0]], 1))),) + out[1:] + (name,)
    rccl = {k: v for k, v in report.items() if isinstance(v, float) and not k.endswith("+p2p")}
    fallback = min(rccl, key=rccl.get) if rccl else "single_reduction"
    report[name + " (timed steps)"] = "failed: " + (err or "on another rank") + "; timed again with " + fallback
    set_variant(prob, fallback)
    prob.step(rtol)
    return timed_steps(prob, rtol, steps, barrier) + (fallback,)


def strong_leg(n, axis, rank, world, rtol, barrier, steps=3):
    """The n^3 cube SPLIT over the ranks (strong scaling), both recurrences; per-iteration time of the solve."""
    zplanes = partition.slab_ranges(n + 1, world)[rank]
    prob = Problem(n, n, n, (1.0, 1.0, 1.0), zplanes, axis, rank, world)
    res = {"workload": "unit cube n=%d (%d DOF, %d tets) split into %d z-slabs, T=350/300 on the %s-faces" % (n, (n + 1) ** 3, 6 * n ** 3, world, "xyz"[axis]),
           "anchor": "strong-scaling speed-up = dof_per_s / (roofline.dof_per_s of the N=1 line: the same cube on one GPU, which runs the "
                     "row-dictionary product there as the slabs do here; roofline.streaming_kernel.dof_per_s is the same cube with the "
                     "streaming product)"}
    os.environ.setdefault("FS_P2P_TIMEOUT_MS", "4000")
    done = []
    for name in VARIANTS:
        err, st, elapsed, asm_ms = None, None, 0.0, 0.0
        try:
            set_variant(prob, name)
        except B.BackendError as e:
            res[name] = {"unavailable": str(e)[:200]}
            prob.p2p = False
            continue
        try:
            prob.step(rtol)
            elapsed, asm_ms, st = timed_steps(prob, rtol, steps, barrier, reduce=False)
            if name.endswith("+p2p") and not getattr(prob.V, "_p2p", False):
                err, prob.p2p = "a peer-to-peer wait timed out: the library fell back to RCCL mid-solve", False
        except Exception as e:
            err = repr(e)[:200]
        if not all_ranks_ok(err is None):
            res[name] = {"failed": err or "on another rank"}
            if getattr(prob, "p2p", False):
                set_variant(prob, "single_reduction")
            continue
        elapsed = float(np.max(B.comm_allgather([elapsed], 1)))
        ms = elapsed * 1e3 / steps
        res[name] = {"dof_per_s": round((n + 1) ** 3 / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "iterations": st["iterations"],
                     "assemble_ms": round(asm_ms, 4), "ms_per_iteration": round((ms - asm_ms) / max(st["iterations"], 1), 5),
                     "spmv_kernel_ms": round(st["spmv_ms"], 5), "update_kernel_ms": round(st["update_ms"], 5),
                     "true_rel_residual": st["true_rel_residual"]}
        done.append(name)
        last_st = st
        if name.endswith("+p2p"):
            _, h_ms = B.comm_benchmark(prob.V, 200)
            res[name]["halo_ms"] = round(h_ms, 5)
    best = min(done, key=lambda k: res[k]["ms_per_step"])
    res.update({"recurrence": best, "dof_per_s": res[best]["dof_per_s"], "iterations": res[best]["iterations"],
                "ms_per_iteration": res[best]["ms_per_iteration"]})
    set_variant(prob, "single_reduction")
    a_ms, h_ms = B.comm_benchmark(prob.V, 200)
    res.update({"allreduce_ms": round(a_ms, 5), "halo_ms": round(h_ms, 5)})
    k = kernel_rates(last_st, prob.V)
    res["spmv_required_GBps_rank0"] = k["required_GBps"]
    res["spmv_csr_equivalent_GBps_rank0"] = k["csr_equivalent_GBps"]
    return res


def p2_leg(n, axis, rank, world, rtol, barrier, steps=2, warmup=1, recurrence="auto", mesh="structured"):
    """BASELINE configs[3]: P2 heat conduction, unit cube n (107 -> 9 938 375 DOF), z-slabs over the ranks.
    mesh = shuffled / renumbered (one GPU): the same cube in FILE order (P2FileOrderProblem)."""
    zplanes = partition.slab_ranges(n + 1, world)[rank]
    t0 = time.perf_counter()
    if mesh != "structured":
        if world != 1:
            sys.exit("bench.py --workload p2 --mesh %s is a one-GPU measurement" % mesh)
        prob = P2FileOrderProblem(n, axis, renumber=mesh == "renumbered")
    else:
        prob = P2Problem(n, zplanes, axis, rank, world)
    setup_s = time.perf_counter() - t0
    name, trial = choose_recurrence(prob, rtol, world, recurrence, barrier)
    elapsed, asm_ms, st, name = timed_steps_with_fallback(prob, rtol, steps, barrier, name, trial, world, warmup)
    ms = elapsed * 1e3 / steps
    err = float(np.abs(prob.x.get()[:prob.n_owned] - prob.exact_owned).max())
    err = parallel.max_over_ranks(err)
    n_dof = p2_global_dofs(n)
    k = kernel_rates(st, prob.V)
    return {"workload": "BASELINE configs[3]: P2 heat conduction, unit cube n=%d (%d DOF, %d tets), k=20, T=350/300 on the %s-faces, "
                        "Jacobi-PCG rtol %g, %s%s" % (n, n_dof, 6 * n ** 3, "xyz"[axis], rtol, "1 GPU" if world == 1 else "%d z-slabs" % world,
                                                       "" if mesh == "structured" else "; mesh uploaded with RANDOMLY PERMUTED vertices and cells (%s)" % (
                                                           "the library's locality renumbering on" if mesh == "renumbered" else "renumbering off")),
            "mesh_ms": round(prob.mesh_ms, 2),
            "n_dof": n_dof, "dof_per_s": round(n_dof / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "steps": steps,
            "assemble_ms": round(asm_ms, 4), "cg_iterations": st["iterations"], "true_rel_residual": st["true_rel_residual"],
            "ms_per_iteration": round((ms - asm_ms) / max(st["iterations"], 1), 5),
            "max_abs_error_vs_exact_profile": err, "recurrence": name, "recurrence_trial_ms_per_step": trial,
            "symbolic_ms": round(prob.symbolic_ms, 2), "setup_s": round(setup_s, 2),
            "spmv": dict(k, rows_rank0=prob.n_owned, frac_of_8TBps=round(k["required_GBps"] / HBM_PEAK_GBS, 3)),
            "update": update_rates(st, prob.n_owned), "iteration": iteration_rates(st, k, prob.n_owned)}


def th_leg(n, n_steps, rank, world):
    """BASELINE configs[4]: lid-driven cavity, Taylor-Hood P2/P1 on the unit cube n (43 -> 1 975 509 velocity + 85 184 pressure
    dofs), nu = 0.01, rho = 1, dt = 0.01, backward Euler, Newton per step - through the solver class, as a user runs it."""
    import copy
    import logging
    from collections import OrderedDict
    from fenicssolver_amd.fem import UnitCubeMesh, BoxMesh, Point, AutoSubDomain, Constant, near
    from fenicssolver_amd import SolverBase as SB
    from fenicssolver_amd.CoupledNavierStokesSolver import CoupledNavierStokesSolver
    t0 = time.perf_counter()
    # several ranks: the DISTRIBUTED box - every rank holds only its z-slab on the host (round 4: the Taylor-Hood path runs on it)
    mesh = UnitCubeMesh(n, n, n) if world == 1 else BoxMesh(Point(0, 0, 0), Point(1, 1, 1), n, n, n, distributed=True)
    bcs = OrderedDict()
    bcs["walls"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary), 'boundary_id': 1,
                    'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((0, 0, 0))}]}
    bcs["lid"] = {'boundary': AutoSubDomain(lambda x, on_boundary: on_boundary and near(x[2], 1.0)), 'boundary_id': 2,
                  'values': [{'variable': "velocity", 'type': 'Dirichlet', 'value': Constant((1, 0, 0))}]}
    s = copy.deepcopy(SB.default_case_settings)
    s.update({'solver_name': "CoupledNavierStokesSolver", 'mesh': mesh, 'fe_degree': 1, 'boundary_conditions': bcs,
              'body_source': None, 'initial_values': {'velocity': (0, 0, 0), 'pressure': 0},
              'material': {'density': 1.0, 'kinematic_viscosity': 0.01}})
    s['solver_settings']['transient_settings'] = {'transient': True, 'starting_time': 0.0, 'time_step': 0.01,
                                                  'ending_time': 0.01 * n_steps - 1e-9}
    s['solver_settings']['reference_values'] = {'velocity': (1, 1, 1), 'pressure': 0}
    s['report_settings'] = {"logging_level": logging.ERROR, "logging_file": None, "plotting_freq": 0, "saving_freq": 0}
    solver = CoupledNavierStokesSolver(s)
    parallel.barrier()
    t1 = time.perf_counter()
    w = solver.solve()
    parallel.barrier()
    t2 = parallel.max_over_ranks(time.perf_counter() - t1)
    # the dominant kernel of the configuration - the 4 x 4-block product of the FGMRES iterations, k_sell_spmv4_ksplit - timed on the
    # Jacobian of the last Newton step (fs_spmv_benchmark: HIP events around back-to-back launches on the library's stream), against
    # the bytes its storage form has to move: the LIVE value planes (a pressure column / row exists for vertex nodes only: 16 values
    # of a vertex-vertex block, 12 of vertex-edge and edge-vertex, 9 of edge-edge), 4 B of column index per block, x read and
    # y written (4 doubles per node each)
    spmv4 = None
    try:
        ctx = getattr(solver, '_ns_ctx', None)
        if world == 1 and ctx is not None:
            J = ctx['J']
            Vd = J.space
            xb, yb = B.DeviceVector(Vd.n_local * 4), B.DeviceVector(Vd.n_owned * 4)
            xb.fill(1.0)
            ms4 = J.spmv_benchmark(xb, yb, reps=30)
            fsp = solver.function_space
            cn = (fsp.cell_nodes() if hasattr(fsp, 'cell_nodes') else fsp.velocity_space().cell_nodes()).astype(np.int64)
            if cn.shape[1] != 10:
                cn = None
            live = None
            if cn is not None:
                nn = int(cn.max()) + 1
                nvx = mesh.num_vertices()
                keys = np.unique((cn[:, :, None] * nn + cn[:, None, :]).ravel())
                rv, cv = (keys // nn) < nvx, (keys % nn) < nvx
                n_vv, n_ve, n_ev, n_ee = int((rv & cv).sum()), int((rv & ~cv).sum()), int((~rv & cv).sum()), int((~rv & ~cv).sum())
                live = 8 * (16 * n_vv + 12 * (n_ve + n_ev) + 9 * n_ee) + 4 * len(keys) + 64 * nn
            spmv4 = {"kernel": "k_sell_spmv4_ksplit<true,true> (4 x 4-block Taylor-Hood product: the entries of a slice dealt to the four waves)",
                     "avg_launch_ms": round(ms4, 5), "required_bytes_per_launch": live,
                     "stored_bytes": int(Vd.spmv_matrix_bytes)}
    except Exception as e:          # (the measurement must not cost the line)
        spmv4 = {"error": repr(e)[:200]}
    W4 = w.vector().array().reshape(-1, 4)
    nv = mesh.num_vertices()
    n_local_nodes = len(W4)
    # global counts of the unit cube (a distributed mesh holds this rank's slab only)
    n_nodes, nv_g, nc_g = (2 * n + 1) ** 3, (n + 1) ** 3, 6 * n ** 3
    n_dof = 3 * n_nodes + nv_g
    speed = parallel.max_over_ranks(float(np.abs(W4[:, :3]).max()))
    p_lo, p_hi = -parallel.max_over_ranks(-float(W4[:nv, 3].min())), parallel.max_over_ranks(float(W4[:nv, 3].max()))
    return {"workload": "BASELINE configs[4]: lid-driven cavity, Taylor-Hood P2/P1, unit cube n=%d (%d velocity + %d pressure dofs, %d tets), "
                        "nu=0.01, dt=0.01, %d backward-Euler steps, Newton per step, FGMRES + block preconditioner, %s"
                        % (n, 3 * n_nodes, nv_g, nc_g, n_steps, "1 GPU" if world == 1 else
                           "%d z-slabs of the distributed box mesh (%d of %d nodes on rank 0's host)" % (world, n_local_nodes, n_nodes)),
            "n_dof": n_dof, "time_steps": int(solver.current_step), "solve_s": round(t2, 4),
            "dof_per_s": round(n_dof * solver.current_step / t2, 1), "setup_s": round(t1 - t0, 2),
            "newton_residuals_last_step": [float(v) for v in solver.newton_history],
            "krylov_iterations_last_step": int(solver.newton_krylov_iterations),
            "max_speed": speed, "pressure_range": [p_lo, p_hi], "host_nodes_rank0": n_local_nodes, "spmv4": spmv4}


class Watchdog:
    """The extra legs must never cost the line: if they overrun their budget (a collective that never completes would be the
    reason), rank 0 prints what has been measured and every rank leaves."""

    def __init__(self, out, rank, budget_s):
        import threading
        self.out, self.rank, self.done = out, rank, threading.Event()
        self.t = threading.Thread(target=self._run, args=(budget_s,), daemon=True)
        self.t.start()

    def _run(self, budget_s):
        if self.done.wait(budget_s):
            return
        if self.rank == 0 and self.out is not None:
            self.out["extra_legs_timed_out"] = True
            print(json.dumps(self.out), flush=True)
        else:
            time.sleep(2.0)
        os._exit(0)

    def stop(self):
        self.done.set()


def kernel_name(V, st=None):
    if st is not None and st.get("fused_iteration", 0) == 2:     # decomposed space: exchange kernel + iteration kernel
        return ("k_cg_p2p_exchange<true> + k_dict_cg_iter<3,true> (TWO launches per CG iteration on a decomposed space: peer-to-peer "
                "exchange - all-reduce, w to the neighbours, ghost rows of r and s advanced - then update of iteration k + row-dictionary "
                "product of iteration k + 1, %d distinct rows in LDS)" % st["row_classes"])
    if st is not None and st.get("fused_iteration", 0):          # fs_krylov.hip k_dict_cg_iter: one launch per CG iteration
        return ("k_dict_cg_iter<3> (ONE launch per CG iteration: update of iteration k + row-dictionary product of iteration k + 1, "
                "%d distinct rows in LDS; the new residual on the neighbour columns recomputed from the old r, w, s)" % st["row_classes"])
    if st is not None and st.get("lattice_order", 0) and st.get("row_classes", 0) > 0 and st.get("product_kind", 0) == 5:
        # fs_latmarch.h / fs_krylov_lattice.inc k_lat_march (round 6): the lattice-ordered shadow in marching-window form
        return ("k_lat_march<3,%d> (row-dictionary form in the solver's LATTICE order of the half grid, %d distinct rows: windows of x MARCHING "
                "through the lattice planes in an LDS ring filled by a loader wave with global_load_lds_dwordx4 two steps ahead; the loop "
                "structure compile-time - eight parity stencils -, a wave per line with the partial sums of five planes' rows in registers, "
                "the coefficients of a (line, step) one contiguous step list through scalar loads; the rows at the ends of the mesh lines "
                "in column tiles with lanes along Y, in workgroups of their own at the front of the same launch; template arguments: dot "
                "mode, 64-pair pieces of a line)" % (2 if round((V.n_owned) ** (1.0 / 3.0)) + 1 > 128 else 1, st["row_classes"]))
    if st is not None and st.get("lattice_order", 0) and st.get("row_classes", 0) > 0:
        # fs_krylov.hip k_lattice_spmv: the solver's lattice-ordered shadow of a scalar CG2 box operator (fs_lattice.hip)
        return ("k_lattice_spmv<3> (row-dictionary form in the solver's LATTICE order of the half grid: %d distinct rows; tiles of 128 x 4 x 4 rows, "
                "x through LDS windows, a wave per line parity with its class's row broadcast from LDS; interior strips of lines as one long line; the rows at the "
                "ends of the mesh lines in column tiles with lanes along Y, in workgroups of their own)"
                % st["row_classes"])
    if st is not None and st.get("row_classes", 0) > 0 and st.get("product_kind", 0) == 3:
        # fs_box.h k_box_spmv (round 6): a P1 box from 1.5 M rows on - launch shape by the length of a mesh line (fs_krylov.hip box_plan_for)
        nx = int(round(V.n_owned ** (1.0 / 3.0)))
        shape = "6,2,2,2" if nx <= 320 else "8,3,2,2"
        return ("k_box_spmv<3,%s> (row-dictionary form of a P1 box, %d distinct rows: windows of x MARCHING through the mesh planes in an "
                "LDS ring filled by two loader waves with global_load_lds_dwordx4 two steps ahead, dot weights and class numbers through "
                "LDS as well, partial sums of three planes' rows in registers, one resident window per workgroup; template arguments: "
                "dot mode, compute waves, rows per lane, steps ahead, loader waves)" % (shape, st["row_classes"]))
    if st is not None and st.get("row_classes", 0) > 0:       # fs_krylov.hip dict_build(): a few distinct rows, coefficients in LDS
        # template arguments: dot mode, whole dictionary in every workgroup's LDS (<= 32 KB, fs_krylov.hip FS_DICT_WHOLE_LDS_BYTES;
        # class rows are 24 doubles per round of the longest run plan: 1 round on P1, 5 on CG2 Kuhn meshes) / per-item class rows
        whole = st["row_classes"] * (24 if V.degree == 1 else 80) * 8 <= (32 << 10)
        return "k_dict_spmv<3,%s> (row-dictionary form: %d distinct rows, %s)" % (
            "true" if whole else "false", st["row_classes"],
            "whole dictionary in LDS" if whole else "the class rows of each work item copied into its wave's LDS region")
    nt = V.sell_entries * 8 > (192 << 20)          # fs_krylov.hip spmv_nontemporal(): matrix larger than the caches
    one = "k_sell_spmv<1,3,%d,%s>" % (4 if V.n_slices <= 32768 else 16, "true" if nt else "false")
    if V.n_slices > 32768 and V.degree == 1 and V.n_dia_slices > 0:       # spmv_use_pairs(): paired DIA slices, two rows per lane
        return "k_dia_pair_spmv<3,%s> + %s on the unpaired slices (one launch each per product)" % ("true" if nt else "false", one)
    return one


UPDATE_BYTES_PER_DOF = 72      # k_cg_update_scaled, single-reduction scaled CG: reads r, w, p, s, x and writes r, p, s, x
DICT_BYTES_PER_ROW = 26        # k_dict_spmv: z read (its x gathers are the same array), d read, w written, 2-byte class number
FUSED_BYTES_PER_ROW = 90       # k_dict_cg_iter: reads r, w, s, p, x, d + 2-byte class number, writes r, w, s, p, x (one launch per iteration)


def kernel_rates(st, V):
    """Byte rates of the product kernel (the hybrid SELL-64/DIA SpMV or its row-dictionary form, fused with the 3 dot products of the
    diagonally scaled CG).  REQUIRED bytes = what the kernel's own storage form has to move per launch:
    streaming form: stored values + the column indices of SELL slices + z, d reads + w write; row-dictionary form: 26 B per row.
    They are the numerator of every roofline fraction this bench prints.  The CSR-equivalent bytes of SURVEY section 8d
    (nnz*12 + n*20) are reported beside them as a rate a CSR kernel would need to match the duration - not as a fraction of anything:
    a kernel that does not move those bytes can exceed the HBM peak on them.
    Time = mean duration of the live launches sampled with HIP events on the library's stream inside the timed solves."""
    ms = st["spmv_ms"]
    dict_on = st.get("row_classes", 0) > 0
    fused = bool(st.get("fused_iteration", 0))
    required = (FUSED_BYTES_PER_ROW if fused else DICT_BYTES_PER_ROW) * V.n_owned if dict_on else V.spmv_matrix_bytes + 24 * V.n_owned
    rate = lambda nbytes: round(nbytes / ms / 1e6, 1) if ms > 0 else 0.0
    return {"kernel": kernel_name(V, st), "avg_launch_ms": round(ms, 5), "row_classes": st.get("row_classes", 0), "fused_iteration": int(fused),
            "required_bytes_per_launch": required, "required_GBps": rate(required),
            "required_bytes_model": ("90 B/row: the whole iteration in one launch - r, w, s, p, x, d reads + 2-byte class number, r, w, s, p, x writes"
                                     if fused else
                                     "26 B/row: z, d reads + w write + 2-byte class number (the distinct value rows sit in LDS / L2)" if dict_on else
                                     "stored values + column indices of SELL slices + 24 B/row (z, d reads + w write); DIA slices carry no column indices"),
            "csr_equivalent_bytes_per_launch": st["spmv_bytes"], "csr_equivalent_GBps": rate(st["spmv_bytes"]),
            "dia_slices": V.n_dia_slices, "slices": V.n_slices}


def update_rates(st, n_rows):
    """The fused vector update of the CG iteration: 72 B per row, nothing to compress."""
    if st.get("fused_iteration", 0):
        return {"kernel": "none: the update is part of k_dict_cg_iter (one launch per iteration)", "avg_launch_ms": 0.0,
                "required_bytes_per_launch": 0, "required_GBps": 0.0, "frac": None}
    ms = st["update_ms"]
    nbytes = UPDATE_BYTES_PER_DOF * n_rows
    gbps = nbytes / ms / 1e6 if ms > 0 else 0.0
    return {"kernel": "k_cg_update_scaled (x, r, p, s updated; the dot partials of the product summed)", "avg_launch_ms": round(ms, 5),
            "required_bytes_per_launch": nbytes, "required_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 3)}


def iteration_rates(st, k, n_rows):
    """One whole CG iteration (product + update + launch gaps) against the bytes its two kernels have to move."""
    ms = st["solve_ms"] / max(st["iterations"], 1)
    fused = bool(k.get("fused_iteration", 0))
    nbytes = k["required_bytes_per_launch"] + (0 if fused else UPDATE_BYTES_PER_DOF * n_rows)
    gbps = nbytes / ms / 1e6 if ms > 0 else 0.0
    return {"required_bytes": nbytes, "bytes_per_dof": round(nbytes / float(n_rows), 1), "ms": round(ms, 5), "GBps": round(gbps, 1),
            "frac": round(gbps / HBM_PEAK_GBS, 3), "what": "solve_ms / iterations (host clock around the whole solve: products, updates, "
            "launch gaps, convergence polls) against " + ("the 90 B/row of the one-launch iteration" if fused else
                                                          "required bytes of the product + 72 B/row of the update")}


PROFILE_ROUND = "r06"          # the round whose committed PMC passes (profiles/<round>_pmc.json) belong to THIS tree's kernels


def committed_traffic(tag):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS round (profiles/r06_pmc.json;
    collected on the same command in separate --pmc runs by tools/collect_profiles.sh).  Not measured in this run.  A missing file
    or key gives None: an older round's file describes an older kernel and is never used."""
    name = PROFILE_ROUND + "_pmc.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            v = json.load(fh).get(tag)
    except (OSError, ValueError):
        return None, None
    return (v, "profiles/" + name) if v is not None else (None, None)


def committed_kernel_traffic(case, prefixes):
    """HBM bytes per launch of a side workload's product kernel from THIS round's committed counter passes
    (profiles/<round>_<case>_pmc_raw.json, tools/pmc_average.py: mean FETCH_SIZE / WRITE_SIZE in KiB over all launches of the traced
    run, the no-op launches behind convergence included): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 - the read side doubled by the gfx950
    rule the default command's calibration confirms."""
    name = "%s_%s_pmc_raw.json" % (PROFILE_ROUND, case)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, None
    for pre in prefixes:
        for kname, v in d.items():
            if kname.startswith(pre) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                return int((2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024), "profiles/" + name
    return None, None


def make_roofline(k, workload, traffic, traffic_source):
    frac = k["required_GBps"] / HBM_PEAK_GBS
    what = " (product fused with the 3 dot products of the diagonally scaled CG)" if ("k_box_spmv" in k["kernel"] or "k_lat_march" in k["kernel"]) else (
           " (DIA product fused with the 3 dot products of the diagonally scaled CG, values from a dictionary of the distinct rows in "
            "LDS, work items of 126 rows - two per lane -, one 16-byte load per run of consecutive offsets; template arguments: dot mode, "
            "whole dictionary in LDS)") if k.get("row_classes", 0) > 0 else (
        " (hybrid SELL-64/DIA SpMV fused with the 3 dot products of the diagonally scaled CG; template arguments of k_sell_spmv: block "
        "size, dot mode, entries per round, non-temporal matrix loads; of k_dia_pair_spmv: dot mode, non-temporal matrix loads)")
    r = {"kernel": k["kernel"] + what,
         "bound": "hbm", "achieved": k["required_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac, 3),
         "traffic": traffic,
         "traffic_source": ("%s (rocprofv3 --pmc passes of this command, committed; NOT measured in this run)" % traffic_source)
                           if traffic is not None else None,
         "workload": workload,
         "note": "achieved = REQUIRED bytes of the kernel's storage form (required_bytes_model) / avg_launch_ms; avg_launch_ms = mean "
                 "of the LIVE products sampled with HIP events (every 16th iteration of the timed solve; a product = the launches the "
                 "kernel field names) on the library's stream; csr_equivalent_GBps = the SURVEY 8d CSR bytes (nnz*12 + n*20) over the "
                 "same time: a comparison with a CSR kernel, not a fraction of the peak"}
    r.update({kk: k[kk] for kk in ("avg_launch_ms", "required_bytes_per_launch", "required_bytes_model", "csr_equivalent_bytes_per_launch",
                                   "csr_equivalent_GBps", "dia_slices", "slices")})
    return r


def main():
    a = parse()
    # the library turns the peer-to-peer halo exchange on by default; here every transport / recurrence variant is set, tried and
    # timed explicitly (choose_recurrence), so the halo plans start on RCCL
    os.environ["FS_HALO_P2P"] = "0"
    rank, world, _ = parallel.world()
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d needs one rank per GPU: python -m fenicssolver_amd.launch --nproc %d bench.py --gpus %d "
                     "(or torch.distributed.run)" % (a.gpus, a.gpus, a.gpus))
        a.gpus = world
    t_init = time.perf_counter()
    parallel.ensure_comm()      # binds LOCAL_RANK's GPU; N>1: rendezvous of the RCCL id + ncclCommInitRank
    init_ms = (time.perf_counter() - t_init) * 1e3     # fs_init: device context, stream, the library's code objects (+ the communicator)

    def barrier():
        parallel.barrier()      # device sync + (N>1) a 1-double all-reduce over RCCL
        B.synchronize()

    base = {"metric": "DOF/s (assemble+CG solve to 1e-8) on 3D heat transfer", "unit": "DOF/s", "n_gpus": world,
            "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    par = "1 GPU" if world == 1 else "z-slab domain decomposition x%d" % world

    if a.workload == "p2":      # BASELINE configs[3] as the timed leg: the fixed 9.94 M-DOF problem over the ranks (strong)
        n = a.n if a.n != 99 else 107
        axis = a.bc_axis if a.bc_axis is not None else 2
        r = p2_leg(n, axis, rank, world, a.rtol, barrier, steps=a.steps, warmup=a.warmup, recurrence=a.recurrence, mesh=a.mesh)
        if rank == 0:
            out = dict(base, value=r["dof_per_s"], steps=a.steps, warmup=a.warmup, ms_per_step=r["ms_per_step"], scaling="strong",
                       config={"workload": r["workload"], "n_dof": r["n_dof"], "n_cells": 6 * n ** 3, "parallelism": par,
                               "cg_iterations": r["cg_iterations"], "true_rel_residual": r["true_rel_residual"], "recurrence": r["recurrence"]},
                       assemble_ms_per_step=r["assemble_ms"], symbolic_ms=r["symbolic_ms"], mesh_ms=r["mesh_ms"],
                       parity={"max_abs_error_vs_exact_profile": r["max_abs_error_vs_exact_profile"]})
            k = r["spmv"]
            hbm = k["required_bytes_per_launch"] + UPDATE_BYTES_PER_DOF * k["rows_rank0"] > (256 << 20)
            traffic, src = (None, None)
            if world == 1 and a.mesh == "structured" and n == 107:
                traffic, src = committed_kernel_traffic("p2", ("k_lat_march<3", "k_lattice_spmv<3", "k_dict_spmv<3"))
            out["roofline"] = dict(make_roofline(k, "rank 0's part of the step workload", traffic, src),
                                   update_kernel=r["update"], iteration=r["iteration"])
            out["roofline"]["kernel"] = k["kernel"] + " (product of the CG2 operator fused with the 3 CG dot products)"
            if not hbm:
                out["roofline"].update(achieved=None, frac=None, note="this rank's part is cache-resident: no HBM fraction claimed")
            print(json.dumps(out))
        parallel.barrier()
        parallel.finalize()
        return
    if a.workload == "th":      # BASELINE configs[4] as the timed leg (a "step" = one backward-Euler time step with its Newton solves)
        n = a.n if a.n != 99 else 43
        r = th_leg(n, a.steps if a.steps != 20 else 10, rank, world)
        if rank == 0:
            out = dict(base, metric="DOF/s (Taylor-Hood Navier-Stokes time steps, assemble + Newton/FGMRES solve)", value=r["dof_per_s"],
                       steps=r["time_steps"], warmup=0, ms_per_step=round(r["solve_s"] * 1e3 / max(r["time_steps"], 1), 3), scaling="strong",
                       config={"workload": r["workload"], "n_dof": r["n_dof"], "parallelism": par if world == 1 else "node-plan decomposition x%d" % world},
                       detail={k: r[k] for k in ("newton_residuals_last_step", "krylov_iterations_last_step", "max_speed", "pressure_range", "setup_s")},
                       roofline={"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                 "note": "kernel rates of this configuration: profiles/*_ns_kernel_stats.csv (rocprofv3 of tools/prof_ns.sh)"})
            k4 = r.get("spmv4") or {}
            if k4.get("required_bytes_per_launch") and k4.get("avg_launch_ms", 0) > 0:
                gbps = k4["required_bytes_per_launch"] / k4["avg_launch_ms"] / 1e6
                out["roofline"].update(kernel=k4["kernel"], achieved=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBS, 3),
                                       avg_launch_ms=k4["avg_launch_ms"], required_bytes_per_launch=k4["required_bytes_per_launch"],
                                       stored_bytes=k4["stored_bytes"],
                                       required_bytes_model="live value planes of the 4 x 4 blocks (16 / 12 / 9 of 16 by node kinds) + 4 B of column "
                                                            "index per block + 64 B per node (x read, y written)",
                                       note="the dominant kernel of the FGMRES iterations timed by fs_spmv_benchmark (HIP events, 30 back-to-back "
                                            "launches on the Jacobian of the last Newton step) after the timed steps; shares of the other kernels: "
                                            "profiles/*_ns_kernel_stats.csv")
            elif k4:
                out["roofline"]["spmv4"] = k4
            print(json.dumps(out))
        parallel.barrier()
        parallel.finalize()
        return

    n = a.n
    axis = a.bc_axis if a.bc_axis is not None else (2 if world == 1 else 0)
    if a.scaling == "weak":
        nz = world * (n + 1) - 1
        p1 = (1.0, 1.0, nz / float(n))
        zplanes = partition.slab_ranges(nz + 1, world, planes_per_rank=n + 1)[rank]
    else:
        nz = n
        p1 = (1.0, 1.0, 1.0)
        zplanes = partition.slab_ranges(nz + 1, world)[rank]
    if world > 1 and axis == 2 and a.scaling == "weak":
        print("[bench] note: --bc-axis 2 with weak scaling lengthens the bar between the Dirichlet faces; "
              "iteration counts will grow with N", file=sys.stderr)
    B.profile_marker(1)         # phase 1 of a traced run: the step workload (tools/summarize_profiles.py)
    if a.mesh != "structured":
        if world != 1:
            sys.exit("bench.py --mesh %s is a one-GPU measurement" % a.mesh)
        prob = ShuffledProblem(n, axis, renumber=a.mesh == "renumbered")
    else:
        prob = Problem(n, n, nz, p1, zplanes, axis, rank, world)
    n_dof_total = (n + 1) * (n + 1) * (nz + 1)

    recurrence, trial = choose_recurrence(prob, a.rtol, world, a.recurrence, barrier)
    first_step_ms, warmup_left = None, a.warmup
    if world == 1 and a.warmup > 0:      # the COLD step of the process (row classes found from scratch, graphs instantiated): one of the warm-ups
        B.synchronize()
        t0 = time.perf_counter()
        prob.step(a.rtol)
        B.synchronize()
        first_step_ms, warmup_left = (time.perf_counter() - t0) * 1e3, a.warmup - 1
    elapsed, asm_ms_step, stats, recurrence = timed_steps_with_fallback(prob, a.rtol, a.steps, barrier, recurrence, trial, world, warmup_left)
    ms_per_step = elapsed * 1e3 / a.steps

    out = None
    if rank == 0:
        workload = ("P1 Poisson heat conduction, unit cube n=%d (BASELINE.json configs[1]: %d DOF, %d tets), "
                    "k=20, T=350/300 on the %s-faces, Jacobi-PCG rtol %g" %
                    (n, n_dof_total, 6 * n ** 3, "xyz"[axis], a.rtol)) if world == 1 else (
            "P1 Poisson heat conduction, box %dx%dx%d cells (%d DOF), z-slabs over %d GPUs, "
            "T=350/300 on the %s-faces, Jacobi-PCG rtol %g" % (n, n, nz, n_dof_total, world, "xyz"[axis], a.rtol))
        if a.mesh != "structured":
            workload += "; mesh uploaded with RANDOMLY PERMUTED vertices and cells (%s)" % (
                "the library's locality renumbering on" if a.mesh == "renumbered" else "renumbering off: FS_RENUMBER=0")
        out = dict(base, value=round(n_dof_total / (ms_per_step * 1e-3), 1), steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(ms_per_step, 4), scaling=a.scaling,
                   config={"workload": workload, "n_dof": n_dof_total, "n_cells": 6 * n * n * nz, "parallelism": par,
                           "cg_iterations": stats["iterations"], "true_rel_residual": stats["true_rel_residual"],
                           "recurrence": recurrence},
                   assemble_ms_per_step=round(asm_ms_step, 4), solve_ms_per_step=round(ms_per_step - asm_ms_step, 4),
                   symbolic_ms=round(prob.symbolic_ms, 3), mesh_ms=round(prob.mesh_ms, 3),
                   update_kernel_ms=round(stats["update_ms"], 5))
        # key order of the round-1/2 lines: metric, value, unit, ...
        out = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data", "config", "assemble_ms_per_step", "solve_ms_per_step", "symbolic_ms",
                                   "mesh_ms", "update_kernel_ms")}
        if trial:
            out["config"]["recurrence_trial_ms_per_step"] = trial
        # what ONE solve() of the reference API pays on this workload, from the FIRST step of the process (nothing kept, nothing
        # instantiated): mesh + sparsity pattern + first assemble + first solve; cold_one_shot adds fs_init (device context, code objects)
        one = first_step_ms if first_step_ms is not None else ms_per_step
        out["first_step_ms"] = None if first_step_ms is None else round(first_step_ms, 3)
        out["one_shot_dof_per_s"] = round(n_dof_total / (1e-3 * (prob.mesh_ms + prob.symbolic_ms + one)), 1)
        out["cold_one_shot_dof_per_s"] = round(n_dof_total / (1e-3 * (init_ms + prob.mesh_ms + prob.symbolic_ms + one)), 1)
        out["init_ms"] = round(init_ms, 1)
        if getattr(prob, "symbolic_warm_ms", None) is not None:
            out["symbolic_warm_ms"] = round(prob.symbolic_warm_ms, 3)
            out["symbolic_note"] = ("symbolic_ms is the first pattern build of the process, symbolic_warm_ms a second build of the same pattern. "
                                    "one_shot_dof_per_s = n / (mesh_ms + symbolic_ms + first_step_ms): the first step of the process, not the "
                                    "steady one; cold_one_shot_dof_per_s adds init_ms (fs_init: device context + the library's code objects, "
                                    "once per process - what `import dolfin` is to the reference)")
        step_kernel = kernel_rates(stats, prob.V)
        step_kernel["update_kernel"] = update_rates(stats, prob.n_owned)
        step_kernel["iteration"] = iteration_rates(stats, step_kernel, prob.n_owned)
        # (what an iteration touches: the product's bytes + the five vectors of the update)
        hbm_resident = step_kernel["required_bytes_per_launch"] + (0 if step_kernel.get("fused_iteration") else UPDATE_BYTES_PER_DOF * prob.n_owned) > (256 << 20)
        if hbm_resident:
            out["roofline"] = dict(make_roofline(step_kernel, "the step workload itself (rank 0's part)", None, None),
                                   update_kernel=step_kernel["update_kernel"], iteration=step_kernel["iteration"])
        else:
            step_kernel["note"] = ("everything an iteration touches (%s) stays in the "
                                   "256 MiB Infinity Cache between iterations: these are cache rates, not an HBM roofline fraction"
                                   % ("the 90 B/row of the one-launch iteration" if step_kernel.get("fused_iteration") else
                                      "required bytes of the product + 72 B/row of the update"))
            out["dominant_kernel_on_step_workload"] = step_kernel
            if world > 1 or a.no_hbm_case:   # no HBM-resident side measurement: the part of a GPU is cache-resident by construction
                out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                   "note": "1 M DOF per GPU stays in the Infinity Cache: see dominant_kernel_on_step_workload for the "
                                           "cache rates; the HBM roofline of this kernel is measured by the default N=1 run (10 M DOF)"}

    # ---- N > 1: the legs the north_star is stated on, outside the timed region, under a watchdog ----
    if world > 1 and a.extra != "none":
        legs = ["strong"] + (["p2"] if world == 8 else []) + (["th"] if world == 4 else []) if a.extra == "auto" else a.extra.split(",")
        dog = Watchdog(out, rank, a.extra_budget)
        t_extra = time.perf_counter()
        try:
            a_ms, h_ms = B.comm_benchmark(prob.V, 200)
            if rank == 0:
                out["comm"] = {"allreduce_3_doubles_ms": round(a_ms, 5), "halo_exchange_ms": round(h_ms, 5),
                               "halo_bytes_per_neighbour": 8 * (n + 1) * (n + 1),
                               "what": "mean of 200 back-to-back in-stream calls (HIP events), as a CG iteration issues them"}
            del prob
            if "strong" in legs:
                r = strong_leg(a.strong_n, axis, rank, world, a.rtol, barrier)
                if rank == 0:
                    out["strong"] = r
            if "p2" in legs:
                r = p2_leg(a.p2_n, 2, rank, world, a.rtol, barrier)
                if rank == 0:
                    out["configs3_p2"] = r
            if "th" in legs:
                r = th_leg(a.th_n, 10, rank, world)
                if rank == 0:
                    out["configs4_th"] = r
        except Exception as e:          # an extra leg must not cost the line
            if rank == 0:
                out["extra_legs_error"] = repr(e)[:400]
        if rank == 0:
            out["extra_legs_s"] = round(time.perf_counter() - t_extra, 2)
        dog.stop()
        prob = None

    if world == 1:
        x_gpu = prob.x.get()
        if a.mesh != "structured":
            x_gpu = prob.to_structured(x_gpu)
        # --- the dominant kernel on an HBM-resident problem of the same family (10 M DOF): the roofline of the line ---
        if not a.no_hbm_case and "roofline" not in out:
            del prob
            B.profile_marker(2)     # phase 2: the HBM-resident sibling, row-dictionary product
            big = Problem(215, 215, 215, (1.0, 1.0, 1.0), (0, 216), axis, 0, 1)
            B.synchronize()
            t0 = time.perf_counter()
            big.step(a.rtol)
            B.synchronize()
            t_big_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            st_big, asm_big = big.step(a.rtol)
            B.synchronize()
            t_big = time.perf_counter() - t0
            traffic, src = committed_traffic("spmv_fused_n215")
            k_big = kernel_rates(st_big, big.V)
            r = make_roofline(k_big, "same path, unit cube n=215, %d DOF (HBM-resident: %.2f GB required per product, %.2f GB per iteration)"
                              % (big.n_owned, k_big["required_bytes_per_launch"] / 1e9,
                                 (k_big["required_bytes_per_launch"] + UPDATE_BYTES_PER_DOF * big.n_owned) / 1e9), traffic, src)
            r.update({"dof_per_s": round(big.n_owned / t_big, 1), "cg_iterations": st_big["iterations"],
                      "assemble_ms": round(asm_big, 3), "solve_ms": round(st_big["solve_ms"], 3),
                      "update_kernel": update_rates(st_big, big.n_owned), "iteration": iteration_rates(st_big, k_big, big.n_owned),
                      "one_shot": {"mesh_ms": round(big.mesh_ms, 2), "symbolic_ms": round(big.symbolic_ms, 2),
                                   "first_step_ms": round(t_big_first * 1e3, 2),
                                   "dof_per_s": round(big.n_owned / (1e-3 * (big.mesh_ms + big.symbolic_ms) + t_big_first), 1),
                                   "what": "mesh + sparsity pattern + FIRST assemble + solve on this mesh: what ONE solve() of the reference API pays"}})
            if st_big.get("row_classes", 0) > 0:
                # The operator of a uniform box with a constant coefficient has a few dozen distinct rows: the product runs in
                # row-dictionary form and does not stream the matrix at all.  Both are reported: the kernel the path really runs
                # here, and - same problem, option row_dictionary = 0 - the streaming kernels every other operator takes.
                r["note_row_dictionary"] = (
                    "k_dict_spmv reads a 2-byte class number per row and keeps the %d distinct value rows in LDS instead of streaming "
                    "8 B per entry (every row of every solve's matrix verified against its class bit for bit; same offsets, same summation order, same bits as "
                    "the streaming product): its roofline is on the 26 B/row it has to move; csr_equivalent_GBps exceeds the peak because "
                    "those bytes are not moved" % st_big["row_classes"])
                B.profile_marker(3)     # phase 3: the same cube, streaming product
                B.set_option("row_dictionary", 0)
                try:
                    big.step(a.rtol)
                    t0 = time.perf_counter()
                    st_s, asm_s = big.step(a.rtol)
                    B.synchronize()
                    t_s = time.perf_counter() - t0
                    rs = make_roofline(kernel_rates(st_s, big.V), "the same problem with option row_dictionary = 0: the streaming kernels "
                                       "every operator without repeated rows takes", traffic, src)
                    ks = kernel_rates(st_s, big.V)
                    rs.update({"dof_per_s": round(big.n_owned / t_s, 1), "cg_iterations": st_s["iterations"],
                               "update_kernel": update_rates(st_s, big.n_owned), "iteration": iteration_rates(st_s, ks, big.n_owned)})
                    r["traffic"], r["traffic_source"] = committed_traffic("spmv_dict_n215")
                    if r["traffic"] is not None:
                        r["traffic_source"] = "%s (rocprofv3 --pmc passes of this command, committed; NOT measured in this run)" % r["traffic_source"]
                    r["streaming_kernel"] = rs
                finally:
                    B.set_option("row_dictionary", 1)
            out["roofline"] = r
            del big
            if not a.no_cache_free_case:
                # --- the same kernels at 85.8 M rows: 8.4 GB per iteration against 256 MiB of Infinity Cache (at 10 M rows the 988 MB
                # of an iteration are 4 x the cache, and FETCH_SIZE counts its hits: VERDICT r5).  One first step + one timed step.
                B.profile_marker(4)     # phase 4: the cache-free sibling
                try:
                    huge = Problem(440, 440, 440, (1.0, 1.0, 1.0), (0, 441), axis, 0, 1)
                    huge.step(a.rtol)
                    B.synchronize()
                    t0 = time.perf_counter()
                    st_h, asm_h = huge.step(a.rtol)
                    B.synchronize()
                    t_h = time.perf_counter() - t0
                    k_h = kernel_rates(st_h, huge.V)
                    r["cache_free_case"] = {
                        "workload": "same path, unit cube n=440, %d DOF (%.2f GB required per product, %.2f GB per iteration)" % (
                            huge.n_owned, k_h["required_bytes_per_launch"] / 1e9,
                            (k_h["required_bytes_per_launch"] + UPDATE_BYTES_PER_DOF * huge.n_owned) / 1e9),
                        "kernel": k_h["kernel"].split(" (")[0], "avg_launch_ms": k_h["avg_launch_ms"],
                        "required_bytes_per_launch": k_h["required_bytes_per_launch"], "achieved": k_h["required_GBps"],
                        "frac": round(k_h["required_GBps"] / HBM_PEAK_GBS, 3), "dof_per_s": round(huge.n_owned / t_h, 1),
                        "cg_iterations": st_h["iterations"], "assemble_ms": round(asm_h, 3), "solve_ms": round(st_h["solve_ms"], 3),
                        "update_kernel": update_rates(st_h, huge.n_owned), "iteration": iteration_rates(st_h, k_h, huge.n_owned)}
                    del huge
                except Exception as e:          # (a box with less free memory must not cost the line)
                    r["cache_free_case"] = {"error": repr(e)[:300]}
        # --- CPU baseline: the oracle's C restatement of the reference's CPU path, same workload ---
        if not a.no_cpu_baseline:
            from oracle import c_oracle
            c_oracle.set_num_threads(c_oracle.usable_cores())
            # bounded sample: calibrate on n=32 first; fall back to a smaller cube if the full
            # configs[1] pass would take more than ~60 s of host time on this box
            t0 = time.perf_counter()
            c_oracle.heat_box_solve(32, 32, 32, axis=axis, rtol=a.rtol)
            cal = time.perf_counter() - t0
            n_cpu = n if cal * (n / 32.0) ** 4 < 60.0 else 49
            ref = c_oracle.heat_box_solve(n_cpu, n_cpu, n_cpu, axis=axis, rtol=a.rtol)
            # (the host cores of a box are shared and the figure moved by 1.5 x between leases, VERDICT r5: the faster of two passes)
            ref2 = c_oracle.heat_box_solve(n_cpu, n_cpu, n_cpu, axis=axis, rtol=a.rtol)
            if ref2["t_assemble"] + ref2["t_solve"] < ref["t_assemble"] + ref["t_solve"]:
                ref = ref2
            cpu_s = ref["t_assemble"] + ref["t_solve"]
            scale = np.abs(ref["x"]).max()
            out["cpu_baseline"] = {
                "value": round((n_cpu + 1) ** 3 / cpu_s, 1), "unit": "DOF/s", "cores": ref["threads"], "kind": "port",
                "sample": "%s, the faster of two passes (n=%d: assemble %.3f s + %d PCG iterations %.3f s; "
                          "pattern build %.2f s excluded as on the GPU)" % (
                              "the full step workload" if n_cpu == n else "a smaller cube of the same family",
                              n_cpu, ref["t_assemble"], ref["iterations"], ref["t_solve"], ref["t_symbolic"]),
                "what": "oracle/fem_oracle_c.c: C/OpenMP restatement of DOLFIN cell-loop assembly + PETSc KSPCG/PCJACOBI "
                        "(FEniCS itself cannot be installed here)",
                "iterations": ref["iterations"]}
            if n_cpu == n:
                out["parity"] = {"iterations_gpu": stats["iterations"], "iterations_cpu": ref["iterations"],
                                 "max_rel_diff_solution": float(np.abs(x_gpu - ref["x"]).max() / scale)}
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
            # north_star target: >= 10x the host-CPU path on the 10 M-DOF solve.  Timed only when the
            # calibration says the pass stays within ~40 s of host time.
            if out["roofline"].get("dof_per_s") and cpu_s * 10.1 * (451.0 / max(ref["iterations"], 1)) < 40.0:
                big_ref = c_oracle.heat_box_solve(215, 215, 215, axis=axis, rtol=a.rtol)
                t_big_cpu = big_ref["t_assemble"] + big_ref["t_solve"]
                hb = out["roofline"]
                hb["cpu_baseline"] = {"value": round(216 ** 3 / t_big_cpu, 1), "unit": "DOF/s", "cores": big_ref["threads"],
                                      "kind": "port", "iterations": big_ref["iterations"],
                                      "sample": "the 10 M-DOF workload once (assemble %.2f s + PCG %.2f s)" % (
                                          big_ref["t_assemble"], big_ref["t_solve"])}
                hb["speedup_vs_cpu_baseline"] = round(hb["dof_per_s"] / hb["cpu_baseline"]["value"], 2)
    if rank == 0:
        print(json.dumps(out))
    parallel.barrier()
    parallel.finalize()


if __name__ == "__main__":
    main()
