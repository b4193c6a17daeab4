#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-4 evidence in one call.
#   default bench command: kernel trace + FETCH_SIZE / WRITE_SIZE passes (tools/collect_profiles.sh r04)
#   configs[3] (P2): trace + PMC passes (tools/prof_r04.sh p2)
#   configs[2] (AMG) and configs[4] (Navier-Stokes): kernel traces
# Summaries land in gpurun_out/summary/ and gpurun_out/summary_r04/ (copy to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/collect_profiles.sh r04 2>&1 | tail -4
bash $R/tools/prof_r04.sh p2 2>&1 | tail -3
bash $R/tools/prof_amg.sh 2 > $R/gpurun_out/prof_amg_r04.log 2>&1
python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_amg $R/gpurun_out/summary_r04/r04_amg_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config3_amg_probe.py 2" 2>&1 | tail -1
bash $R/tools/prof_ns.sh > $R/gpurun_out/prof_ns_r04.log 2>&1
python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_ns $R/gpurun_out/summary_r04/r04_ns_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config5_probe.py 43 2" 2>&1 | tail -1
rm -rf $R/gpurun_out/prof_amg $R/gpurun_out/prof_ns $R/gpurun_out/prof
python $R/bench.py > $R/gpurun_out/summary_r04/r04_bench_line.json 2>/dev/null
python $R/bench.py --n 440 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-case > $R/gpurun_out/summary_r04/r04_bench_line_n440_86M_dof.json 2>/dev/null
# the distributed iteration with the one communicator a 1-GPU box allows (the rank its own halo neighbour): plain timings
python $R/tools/probes/rccl_self_halo_probe.py 99 all 2>&1 | grep -a "iteration\|refresh\|rows" > $R/gpurun_out/summary_r04/r04_p2p_self_halo_timings.txt
ls -la $R/gpurun_out/summary_r04 $R/gpurun_out/summary | tail -30
