"""fenicssolver_amd — MI355X-native assemble + Krylov-solve pipeline behind the
FenicsSolver Python API (SolverBase / ScalarTransportSolver /
LinearElasticitySolver, JSON case settings).

Mirrors FenicsSolver/__init__.py:9-13 of the reference, except that importing
the package never starts a solve by itself (the reference runs ``main(sys.argv)``
on import when argv has >= 2 entries — SURVEY.md Appendix B-Q1); use
``python -m fenicssolver_amd case.json`` instead.
"""
__version__ = "0.1"

import os as _os


def granted_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CFS quota (cpu.max)."""
    try:
        n = len(_os.sched_getaffinity(0))
    except AttributeError:
        n = _os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _limit_host_thread_pools():
    """numpy's OpenBLAS sizes its pool from the machine (64 threads on a 256-CPU node) and its idle workers spin after
    every BLAS call; under a container CPU quota (16 CPUs on the MI355X boxes) that spinning exhausts the quota and
    the kernel freezes the whole cgroup for the rest of the 100 ms period - the thread that feeds the GPU included.
    Measured on configs[4] (round 1): 62 FGMRES iterations 161 ms with the default pool, 74 ms with the pool limited;
    /sys/fs/cgroup/cpu.stat nr_throttled 25 -> 0.  The host side of this package needs no BLAS parallelism, so the
    pools are capped at half the granted CPUs unless the user has already chosen (OMP/OPENBLAS/MKL_NUM_THREADS)."""
    if any(_os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "FS_KEEP_THREAD_POOLS")):
        return None
    # one process per GPU: the ranks of a node share the quota (the launcher exports LOCAL_WORLD_SIZE)
    ranks = max(1, int(_os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    limit = max(1, granted_cpus() // (2 * ranks))
    _os.environ.setdefault("OPENBLAS_NUM_THREADS", str(limit))     # BLAS libraries loaded from here on (scipy's own copy)
    try:
        import numpy  # noqa: F401  (loads its OpenBLAS, so that the limit below reaches it)
        import threadpoolctl
        return threadpoolctl.threadpool_limits(limits=limit, user_api="blas")      # pools that are already loaded
    except Exception:
        return None


_thread_pool_limit = _limit_host_thread_pools()

from .main import main, load_settings  # noqa: F401
