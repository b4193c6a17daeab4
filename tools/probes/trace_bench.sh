#!/bin/bash
# ON THE GPU BOX: kernel trace of the default bench command only (no counter passes), live averages of the CG kernels
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
python $R/tools/summarize_profiles.py ${1:-rXX} $R/gpurun_out/summary_tmp > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
grep "k_cg_update_scaled\|k_dict_spmv<3\|k_dict_cg_iter\|k_dia_pair" $R/gpurun_out/summary_tmp/${1:-rXX}_kernel_stats.csv | cut -c1-150
grep -a "^{" $OUT/stats.log | tail -1 | python $R/tools/probes/bench_line_short.py
rm -rf $OUT/stats
