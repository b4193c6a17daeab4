"""TEST INFRASTRUCTURE: run a script with LOCAL_RANK forced to 0, so that a launcher which numbers its ranks' devices
(torch.distributed.run) can be exercised on a box with ONE GPU (together with FS_RCCL_PATH = libfakerccl.so).
usage: python -m torch.distributed.run ... tests/shim/on_device0.py bench.py --gpus 2 ..."""
import os
import runpy
import sys

os.environ["LOCAL_RANK"] = "0"
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
