#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-6 evidence in one call, every file from the tree as it is.  Every command under
# `timeout` with stdin closed (a `head $F` on an empty $F waited for a terminal until gpurun's limit: 40 GPU-minutes of round 6).
#   default bench command: kernel trace + FETCH_SIZE / WRITE_SIZE passes, phases split at bench.py's k_profile_marker launches
#   (tools/collect_profiles.sh r06 -> tools/summarize_profiles.py); the unprofiled line afterwards - with the 86 M-row cache-free case -
#   so that its roofline.traffic names the r06 PMC file just written
#   configs[3] (P2), the 10 M-DOF cube in FILE order after renumbering, configs[3] in file order: trace + PMC passes (tools/prof_r06.sh)
#   configs[2] (AMG) and configs[4] (Navier-Stokes): kernel traces; bench lines of configs[3] / configs[4]
#   the marching-window product against a plain streaming kernel (tools/probes/run_box_probe.sh); the first steps of a process
# Summaries land in gpurun_out/summary/ and gpurun_out/summary_r06/ (copy both to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$R/gpurun_out/summary_r06
mkdir -p $S
exec < /dev/null
timeout 900 bash $R/tools/collect_profiles.sh r06 2>&1 | tail -4
cp $R/gpurun_out/summary/r06_pmc.json $R/profiles/r06_pmc.json          # (the line below reads the file of THIS collection)
timeout 600 python $R/bench.py > $R/gpurun_out/summary/r06_bench_line.json 2>/dev/null
timeout 1500 bash $R/tools/prof_r06.sh "p2 renumbered p2_renumbered" 2>&1 | tail -3
cp $S/r06_p2_pmc_raw.json $R/profiles/r06_p2_pmc_raw.json 2>/dev/null     # (bench.py --workload p2 reads its traffic from it)
timeout 600 bash $R/tools/prof_amg.sh 2 > $R/gpurun_out/prof_amg_r06.log 2>&1
timeout 120 python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_amg $S/r06_amg_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config3_amg_probe.py 2" 2>&1 | tail -1
grep -a "levels" $R/gpurun_out/prof_amg/run.log > $S/r06_amg_setup_solve.txt
timeout 600 bash $R/tools/prof_ns.sh > $R/gpurun_out/prof_ns_r06.log 2>&1
timeout 120 python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_ns $S/r06_ns_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config5_probe.py 43 2" 2>&1 | tail -1
rm -rf $R/gpurun_out/prof_amg $R/gpurun_out/prof_ns $R/gpurun_out/prof
timeout 300 python $R/bench.py --workload p2 --steps 3 --warmup 1 > $S/r06_p2_bench_line.json 2>/dev/null
(cd $R && DBGS="0 1 3 5 7" timeout 900 bash tools/probes/run_latmarch_abl.sh > /dev/null 2>&1; timeout 600 bash tools/probes/run_latmarch_iter.sh > /dev/null 2>&1; bash tools/probes/latmarch_report.sh > $S/r06_p2_latmarch.txt)      # (k_lat_march with parts switched off, the solve with either product)
timeout 300 python $R/bench.py --workload th > $S/r06_th_bench_line.json 2>/dev/null
timeout 300 python $R/bench.py --cells 215 --mesh renumbered --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-case > $S/r06_sell_unstructured_renumbered_bench_line.json 2>/dev/null
timeout 400 bash $R/tools/probes/run_box_probe.sh "216 441" > /dev/null 2>&1; cp $R/gpurun_out/box_probe.txt $S/r06_box_probe.txt
timeout 200 python $R/tools/probes/first_step_probe.py 2>&1 | tail -18 > $S/r06_first_step.txt
(FS_BOX_ASSEMBLY=0 FS_TILE_ROWS=8192 FS_SLICE_ORDER=-2 timeout 100 python $R/tools/assemble_probe.py child 215; FS_TILE_ROWS=8192 FS_SLICE_ORDER=-2 timeout 100 python $R/tools/assemble_probe.py child 215) 2>&1 | grep -a assemble_ms > $S/r06_box_assembly.txt
ls -la $S $R/gpurun_out/summary | tail -40
