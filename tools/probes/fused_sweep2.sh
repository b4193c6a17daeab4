#!/bin/bash
# ON THE GPU BOX: time per iteration (hipGraph batches, no profiler) of the one-launch iteration, variants x launch geometries
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in ${1:-99}; do
for B in 512 1024; do
for V in 0 1 2 4 5; do
  echo "== n=$N FS_DICT_BLOCKS=$B FS_CG_FUSED_VARIANT=$V"
  FS_DICT_BLOCKS=$B FS_CG_FUSED_VARIANT=$V FS_PROBE_MAXIT=${2:-5000} python $R/tools/probes/fused_iter_probe.py $N 2>&1 | grep "fused=" | tail -2
done; done; done
