"""``python -m fenicssolver_amd.launch --nproc N script.py [args...]`` - one process per GPU of this node.

The counterpart of ``mpirun -n N python script.py`` for the reference (SolverBase.py:102-118): starts N copies of
the script with RANK / WORLD_SIZE / LOCAL_RANK / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment,
which is all fenicssolver_amd.parallel and bench.py read.  LOCAL_RANK is the HIP device the rank binds to;
``--devices 0,1,4,5`` picks the GPUs explicitly (a device may be named more than once).
If one rank fails the others are terminated and its exit code is returned.
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import subprocess
import sys
import time


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m fenicssolver_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc", type=int, required=True, help="number of ranks (= GPUs used)")
    ap.add_argument("--devices", default=None, help="comma-separated HIP device ids, one per rank (default 0..N-1)")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    devices = [int(d) for d in a.devices.split(",")] if a.devices else list(range(a.nproc))
    if len(devices) != a.nproc:
        ap.error("--devices names %d devices for %d ranks" % (len(devices), a.nproc))
    port = a.master_port or _free_port()
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.nproc), LOCAL_RANK=str(devices[r]),
                   LOCAL_WORLD_SIZE=str(a.nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on these hosts
        procs.append(subprocess.Popen([sys.executable, a.script] + a.args, env=env))
    rc = 0
    try:
        live = set(range(a.nproc))
        while live and rc == 0:
            for r in list(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code
            time.sleep(0.02)
    except KeyboardInterrupt:
        rc = 130
    if rc != 0:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        t0 = time.time()
        for p in procs:
            try:
                p.wait(timeout=max(0.1, 10 - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


if __name__ == "__main__":
    sys.exit(main())
