#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel traces of the distributed CG iteration at 1 M rows with the one communicator a
# 1-GPU box allows (the rank is its own neighbour: tools/probes/rccl_self_halo_probe.py) - RCCL send / recv + ncclAllReduce
# against the peer-to-peer exchange - and the plain timings of all variants.  Summaries: gpurun_out/summary_r03/ (copy to profiles/).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$R/gpurun_out/summary_r03
mkdir -p $S
cd /tmp
python $R/tools/probes/rccl_self_halo_probe.py 99 all 2>&1 | grep -a "iteration\|refresh\|rows" > $S/r03_p2p_self_halo_timings.txt
for V in rccl p2p; do
  OUT=$R/gpurun_out/prof_selfhalo_$V
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python $R/tools/probes/rccl_self_halo_probe.py 99 $V > $OUT/trace.log 2>&1
  python $R/tools/kernel_stats_csv.py $OUT/trace $S/r03_selfhalo_${V}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/probes/rccl_self_halo_probe.py 99 $V" || tail -3 $OUT/trace.log
  rm -rf $OUT
done
cat $S/r03_p2p_self_halo_timings.txt
head -12 $S/r03_selfhalo_p2p_kernel_stats.csv | cut -c1-200
head -14 $S/r03_selfhalo_rccl_kernel_stats.csv | cut -c1-200
