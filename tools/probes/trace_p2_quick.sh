export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
OUT=$R/gpurun_out/prof_p2q; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python $R/bench.py --no-cpu-baseline --no-hbm-case --workload p2 --steps 2 --warmup 1 > $OUT/trace.log 2>&1
python $R/tools/kernel_stats_csv.py $OUT/trace $R/gpurun_out/p2q_kernel_stats.csv "quick" || tail -3 $OUT/trace.log
rm -rf $OUT
head -16 $R/gpurun_out/p2q_kernel_stats.csv | cut -c1-110
