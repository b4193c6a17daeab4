#!/bin/bash
# Runs ON THE GPU BOX: builds tools/probes/box_spmv_probe.hip twice (default / nt cache policy on the read-once streams) and times the variants at m = 216 (10 M rows) and m = 441 (86 M rows).  Output: gpurun_out/box_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
O=$R/gpurun_out/box_probe.txt
: > $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $R/fenicssolver_amd/csrc -o /tmp/box_probe_v $R/tools/probes/box_spmv_probe.hip 2>/dev/null || echo "build v failed" >> $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBOX_DC_AUX=2 -I $R/fenicssolver_amd/csrc -o /tmp/box_probe_p $R/tools/probes/box_spmv_probe.hip 2>/dev/null || echo "build p failed" >> $O
for m in ${1:-216 441}; do
  echo "== default, m = $m" >> $O
  timeout 300 /tmp/box_probe_v $m "${2:-}" >> $O 2>&1
  echo "== nt loads of dot weights and class numbers, m = $m" >> $O
  timeout 300 /tmp/box_probe_p $m "${2:-}" >> $O 2>&1
done
cat $O
