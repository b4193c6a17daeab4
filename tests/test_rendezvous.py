"""Host-side process plumbing of the multi-GPU path, no GPU needed: the package's own launcher, the file rendezvous
of the RCCL unique id (fenicssolver_amd/rendezvous.py) and the librccl stand-in the -m gpu multi-rank tests load."""
import ctypes
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    from fenicssolver_amd import rendezvous as R
    rank, size, local = R.world()
    assert local == int(os.environ["LOCAL_RANK"]) == 0          # --devices 0,0,0: every rank on device 0
    made = []
    uid = R.exchange_unique_id(rank, size, lambda: made.append(1) or bytes(range(128)))
    assert uid == bytes(range(128)) and len(made) == (1 if rank == 0 else 0)
    R.cleanup(rank, size)
    open(os.path.join(sys.argv[1], "done.%%d" %% rank), "w").write("ok")
''' % ROOT)


def test_launcher_and_id_rendezvous_three_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FS_RDZV_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, "-m", "fenicssolver_amd.launch", "--nproc", "3", "--devices", "0,0,0", str(script), str(tmp_path)],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("done.")) == ["done.0", "done.1", "done.2"]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("fsamd_")]      # rendezvous files are gone


def test_stale_id_file_is_not_accepted(tmp_path, monkeypatch):
    """A file left by a crashed earlier run with the same key lacks this rank's nonce."""
    from fenicssolver_amd import rendezvous as R
    monkeypatch.setenv("FS_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("FS_RDZV_KEY", "k")
    (tmp_path / "fsamd_k.id").write_bytes(bytes(128) + bytes(16))
    monkeypatch.setattr(R, "_TIMEOUT_S", 0.3)
    try:
        R.exchange_unique_id(1, 2, lambda: b"")
    except TimeoutError:
        return
    raise AssertionError("stale id accepted")


def test_launcher_propagates_a_failing_rank(tmp_path):
    script = tmp_path / "bad.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(30)\n")
    p = subprocess.run([sys.executable, "-m", "fenicssolver_amd.launch", "--nproc", "2", str(script)], cwd=ROOT, timeout=60)
    assert p.returncode == 3


def test_rccl_stand_in_exports_what_the_library_resolves():
    """tests/shim/libfakerccl.so must provide every nccl* symbol fs_comm.hip's rccl_load() looks up."""
    shim = os.path.join(ROOT, "tests", "shim", "libfakerccl.so")
    if not os.path.exists(shim):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(shim)])
    lib = ctypes.CDLL(shim)
    src = open(os.path.join(ROOT, "fenicssolver_amd", "csrc", "fs_comm.hip")).read()
    import re
    wanted = re.findall(r'LOAD\(\w+, "(nccl\w+)"\)', src)
    assert len(wanted) >= 10
    for sym in wanted:
        assert hasattr(lib, sym), sym


def test_product_has_no_test_transport_and_no_torch():
    pkg = os.path.join(ROOT, "fenicssolver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "torch" not in text, f
                assert "FS_COMM_TRANSPORT" not in text and "shm_open" not in text, f
