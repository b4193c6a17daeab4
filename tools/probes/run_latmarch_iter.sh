#!/bin/bash
# configs[3] solved with the marching-window product and with the tile product: iteration / product / update times of the solve
exec < /dev/null
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/latmarch
O=gpurun_out/latmarch/iter.txt
: > $O
for m in 1 0; do
  echo "== FS_LATTICE_MARCH=$m $EXTRA" >> $O
  env FS_LATTICE_MARCH=$m $EXTRA timeout 300 python tools/probes/p2_lattice_probe.py ${N:-107} 2>&1 | grep "lattice 1:" >> $O
done
cat $O
