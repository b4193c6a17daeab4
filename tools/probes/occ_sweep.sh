#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in 215 99; do
for OCC in 0 7 8; do
  echo "== n=$N FS_DICT_OCC=$OCC"
  FS_DICT_OCC=$OCC FS_PROBE_MAXIT=300 python $R/tools/probes/fused_iter_probe.py $N 2>&1 | grep "fused=0" | tail -1
done; done
