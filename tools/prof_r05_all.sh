#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round-5 evidence in one call, every file from the tree as it is.
#   default bench command: kernel trace + FETCH_SIZE / WRITE_SIZE passes, phases split at bench.py's k_profile_marker launches
#   (tools/collect_profiles.sh r05 -> tools/summarize_profiles.py); the unprofiled line afterwards, so that its roofline.traffic
#   names the r05 PMC file just written
#   configs[3] (P2), the 10 M-DOF cube in FILE order with and without the locality renumbering, configs[3] in file order:
#   trace + PMC passes (tools/prof_r05.sh)
#   configs[2] (AMG) and configs[4] (Navier-Stokes): kernel traces;  bench lines of configs[3] / configs[4] / the 86 M-DOF cube
#   the distributed iteration with the rank its own halo neighbour; the shim scale check
# Summaries land in gpurun_out/summary/ and gpurun_out/summary_r05/ (copy both to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$R/gpurun_out/summary_r05
mkdir -p $S
bash $R/tools/collect_profiles.sh r05 2>&1 | tail -4
cp $R/gpurun_out/summary/r05_pmc.json $R/profiles/r05_pmc.json          # (the line below reads the file of THIS collection)
python $R/bench.py > $R/gpurun_out/summary/r05_bench_line.json 2>/dev/null
bash $R/tools/prof_r05.sh "p2 shuffled renumbered p2_renumbered" 2>&1 | tail -3
bash $R/tools/prof_amg.sh 2 > $R/gpurun_out/prof_amg_r05.log 2>&1
python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_amg $S/r05_amg_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config3_amg_probe.py 2" 2>&1 | tail -1
grep -a "levels" $R/gpurun_out/prof_amg/run.log > $S/r05_amg_setup_solve.txt
bash $R/tools/prof_ns.sh > $R/gpurun_out/prof_ns_r05.log 2>&1
python $R/tools/kernel_stats_csv.py $R/gpurun_out/prof_ns $S/r05_ns_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/config5_probe.py 43 2" 2>&1 | tail -1
rm -rf $R/gpurun_out/prof_amg $R/gpurun_out/prof_ns $R/gpurun_out/prof
python $R/bench.py --workload p2 --steps 3 --warmup 1 > $S/r05_p2_bench_line.json 2>/dev/null
python $R/bench.py --workload th > $S/r05_th_bench_line.json 2>/dev/null
python $R/bench.py --n 440 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-case > $S/r05_bench_line_n440_86M_dof.json 2>/dev/null
python $R/tools/probes/rccl_self_halo_probe.py 99 all 2>&1 | grep -a "iteration\|refresh\|rows" > $S/r05_p2p_self_halo_timings.txt
python $R/tools/probes/cg_tail_probe.py 99 0,16,6 1,16,6 2>&1 | tail -2 > $S/r05_cg_tail.txt
FS_LATTICE_TILES=0 python $R/tools/probes/p2_lattice_probe.py 107 2>&1 | tail -5 > $S/r05_p2_lattice_order.txt      # (the work-item product on the lattice order)
bash $R/tools/probes/p2_lattice_tiles.sh > /dev/null 2>&1; cp $R/gpurun_out/r05_p2_lattice_tiles.txt $S/                    # (the tile product)
bash $R/tools/probes/exp_lattice_grid.sh "256 512 768 1024 2048" > /dev/null 2>&1; cp $R/gpurun_out/exp_var.txt $S/r05_p2_lattice_grid.txt      # (tile workgroups behind the column workgroups)
hipcc --offload-arch=gfx950 -O3 $R/tools/probes/mfma_f64_probe.hip -o /tmp/mfma_f64_probe 2>/dev/null && /tmp/mfma_f64_probe > $S/r05_mfma_f64_probe.txt 2>&1
(FS_SPMV4_KSPLIT=0 python $R/tools/probes/spmv4_probe.py; python $R/tools/probes/spmv4_probe.py) 2>&1 | grep -a "KSPLIT\|max" > $S/r05_spmv4_kernels.txt
python $R/tools/probes/first_step_probe.py 2>&1 | tail -18 > $S/r05_first_step.txt
bash $R/tools/shim_scale_check.sh > $S/r05_shim_scale_check.txt 2>&1
ls -la $S $R/gpurun_out/summary | tail -40
