cd $GRAFT_REPO_ROOT
export FS_RCCL_PATH=$GRAFT_REPO_ROOT/tests/shim/libfakerccl.so
for N in 4 8; do
  echo "== N=$N full-size defaults"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) tests/shim/on_device0.py bench.py --gpus $N --steps 5 --warmup 2 2>&1 | tail -3 | cut -c1-900
done
