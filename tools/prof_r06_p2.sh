#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the configs[3] evidence of round 6 after k_lat_march - kernel trace + counter passes of
# `bench.py --workload p2` (tools/prof_r06.sh p2), the unprofiled line (its traffic from the counter file just written), the
# ablations of the kernel (FS_LM_DBG: no ends of the lines / no line waves / no loads) and the solve with either product.
# Every command under timeout, stdin closed.  Summaries: gpurun_out/summary_r06/ (copy to profiles/).
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$R/gpurun_out/summary_r06
mkdir -p $S
timeout 1500 bash $R/tools/prof_r06.sh "p2" 2>&1 | tail -3
cp $S/r06_p2_pmc_raw.json $R/profiles/r06_p2_pmc_raw.json 2>/dev/null
timeout 300 python $R/bench.py --workload p2 --steps 3 --warmup 1 > $S/r06_p2_bench_line.json 2>/dev/null
cd $R
DBGS="0 1 3 5 7" timeout 900 bash tools/probes/run_latmarch_abl.sh > /dev/null 2>&1
timeout 600 bash tools/probes/run_latmarch_iter.sh > /dev/null 2>&1
{
  echo "# k_lat_march at BASELINE configs[3] (n = 107, 9 984 600 rows in lattice order), one MI355X; tools/prof_r06_p2.sh"
  echo "# 1. the product alone (FS_LATTICE_DEBUG=2: 10 launches back to back / per-launch events; 'tile product' in the library's"
  echo "#    messages = the product of the lattice-ordered operator, here k_lat_march) with parts of the kernel switched off:"
  echo "#    FS_LM_DBG bit 1 = no ends of the lines, 2 = no line waves, 4 = no loads (0 = the kernel as shipped; 7 = barriers only)"
  grep -a "==\|no dots\|three dots\|differ" gpurun_out/latmarch/abl.txt | awk '!seen[$0]++'
  echo "# 2. the solve (466 iterations) with k_lat_march (FS_LATTICE_MARCH=1) and with the tile product k_lattice_spmv (=0)"
  cat gpurun_out/latmarch/iter.txt
} > $S/r06_p2_latmarch.txt
ls -la $S | tail -12
