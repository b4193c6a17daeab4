#!/bin/bash
# ablations of k_lat_march at configs[3] (FS_LM_DBG bits: 1 no end wave, 2 no line waves, 4 no loads): product alone / with dots
exec < /dev/null
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/latmarch
O=gpurun_out/latmarch/abl.txt
: > $O
for d in ${DBGS:-0 1 2 3 4 7}; do
  echo "== FS_LM_DBG=$d $EXTRA" >> $O
  env FS_LM_DBG=$d $EXTRA FS_LATTICE_DEBUG=2 FS_LATTICE_CHECK=1 timeout 300 python tools/probes/p2_lattice_probe.py ${N:-107} 2>&1 | grep -E "tile product:|tile product, (three|no) dots|lattice 1:|differ" | head -14 >> $O
done
cat $O
