#!/bin/bash
# bench.py under the driver's launcher at N = 2, 4, 8, 2 back to back ON THE SAME PORT, through the RCCL stand-in on one GPU
# (functional check of rendezvous / halo / all-reduce; the timings mean nothing: the ranks share the device)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export FS_RCCL_PATH=$PWD/tests/shim/libfakerccl.so
for N in ${@:-2 4 8 2}; do
  echo "== N=$N"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29655 tests/shim/on_device0.py bench.py --gpus $N --steps 3 --warmup 1 --cells 47 --extra strong,p2,th --strong-n 47 --p2-n 23 --th-n 8 2>&1 | grep -a "^{\|rror\|Traceback" | cut -c1-${CUT:-330}
done
