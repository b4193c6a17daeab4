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
bash tools/probes/latmarch_report.sh > $S/r06_p2_latmarch.txt
ls -la $S | tail -12
