"""Process rendezvous for one-process-per-GPU runs — no ML framework, no MPI.

The reference becomes parallel under ``mpirun`` (SolverBase.py:102-118); MPI_Init is its rendezvous.
Here the only thing the ranks must agree on before RCCL is up is the 128-byte ``ncclUniqueId`` of rank 0.
Every launcher that matters (``python -m fenicssolver_amd.launch``, PyTorch's elastic launcher, ``mpirun``,
``srun``) exports RANK / WORLD_SIZE / LOCAL_RANK (or the OMPI_/SLURM_ equivalents) and starts the ranks of a
node from ONE parent process, so the id travels through a file in a node-local directory:

    <dir>/fsamd_<key>.hello.<rank>   written by every rank > 0: a random nonce (removed by its writer once it has the id)
    <dir>/fsamd_<key>.id             written by rank 0 once all hellos are there: id + the nonces it saw

A rank accepts an id file only if it carries its own nonce, so a file left behind by a crashed earlier run
with the same key is never mistaken for the current one.  <key> = FS_RDZV_KEY, or MASTER_PORT + the parent
pid; <dir> = FS_RDZV_DIR (a shared filesystem for several nodes), else /dev/shm, else the temp directory.
After the communicator exists everything else (barrier, max, gather of index lists) runs over RCCL.
"""
from __future__ import annotations

import os
import tempfile
import time

_TIMEOUT_S = 300.0


def world():
    """(rank, world_size, local_rank) from the launcher's environment."""
    env = os.environ
    for r, w, l in (("RANK", "WORLD_SIZE", "LOCAL_RANK"),
                    ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"),
                    ("PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"),
                    ("SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID")):
        if r in env and w in env:
            return int(env[r]), int(env[w]), int(env.get(l, env[r]))
    return 0, 1, 0


_published = {}      # rank 0: the id and nonces it published (cleanup() republishes when a hello turns out to have been stale)


def _directory():
    d = os.environ.get("FS_RDZV_DIR")
    if d:
        return d
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()


def _key():
    k = os.environ.get("FS_RDZV_KEY")
    if k:
        return k
    if os.environ.get("FS_RDZV_DIR"):
        # a shared directory is the multi-node set-up: the parent pid differs from node to node, so the key is made of what the
        # launcher gives EVERY rank alike - address, port and the job / run id when there is one
        job = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("SLURM_JOB_ID") or os.environ.get("PMI_JOBID") or "0"
        return "%s_%s_%s" % (os.environ.get("MASTER_ADDR", "local").replace("/", "_"), os.environ.get("MASTER_PORT", "0"), job)
    return "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())


def _write_atomic(path, data: bytes):
    tmp = "%s.tmp.%d" % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    # the directory may be world-writable (/dev/shm): never follow a link somebody planted, never leave the file readable
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path, "rb") as fh:
            return fh.read()
    except OSError:
        return None


def exchange_unique_id(rank: int, size: int, make_id, id_bytes: int = 128) -> bytes:
    """Rank 0 calls ``make_id()`` (-> bytes of length id_bytes) and publishes it; every rank returns it."""
    if size == 1:
        return make_id()
    base = os.path.join(_directory(), "fsamd_" + _key())
    id_path = base + ".id"
    deadline = time.monotonic() + _TIMEOUT_S
    if rank == 0:
        nonces = []
        for r in range(1, size):
            p = "%s.hello.%d" % (base, r)
            while True:
                got = _read(p)
                if got is not None and len(got) == 16:
                    nonces.append(got)
                    break
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: rank %d never announced itself (%s)" % (r, p))
                time.sleep(0.002)
        uid = bytes(make_id())
        assert len(uid) == id_bytes
        _write_atomic(id_path, uid + b"".join(nonces))
        _published.update(uid=uid, nonces=nonces)
        return uid
    nonce = os.urandom(16)
    _write_atomic("%s.hello.%d" % (base, rank), nonce)
    off = id_bytes + 16 * (rank - 1)
    while True:
        got = _read(id_path)
        if got is not None and len(got) == id_bytes + 16 * (size - 1) and got[off:off + 16] == nonce:
            try:
                os.unlink("%s.hello.%d" % (base, rank))      # the acknowledgement rank 0's cleanup waits for
            except OSError:
                pass
            return got[:id_bytes]
        if time.monotonic() > deadline:
            raise TimeoutError("rendezvous: no id from rank 0 (%s)" % id_path)
        time.sleep(0.002)


def cleanup(rank: int, size: int):
    """Rank 0 removes the id file once every other rank has taken it (their hello files are gone)."""
    if size == 1 or rank != 0:
        return
    base = os.path.join(_directory(), "fsamd_" + _key())
    deadline = time.monotonic() + 30.0
    while any(os.path.exists("%s.hello.%d" % (base, r)) for r in range(1, size)) and time.monotonic() < deadline:
        # A hello file left by a crashed run (fixed FS_RDZV_KEY) may have been read before its rank rewrote it: that rank is
        # still waiting for ITS nonce.  Publish the same id again with the nonces as they are now.
        nonces = _published.get("nonces")
        if nonces is not None:
            changed = False
            for r in range(1, size):
                got = _read("%s.hello.%d" % (base, r))
                if got is not None and len(got) == 16 and got != nonces[r - 1]:
                    nonces[r - 1] = got
                    changed = True
            if changed:
                _write_atomic(base + ".id", _published["uid"] + b"".join(nonces))
        time.sleep(0.002)
    _published.clear()
    for p in [base + ".id"] + ["%s.hello.%d" % (base, r) for r in range(1, size)]:
        try:
            os.unlink(p)
        except OSError:
            pass
