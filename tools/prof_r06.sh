#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round-6 evidence beyond the default bench command (tools/collect_profiles.sh r06):
#   configs[3] (P2, n = 107) and the 10 M-DOF cube uploaded in FILE order (shuffled) with and without the locality renumbering -
#   kernel trace of each command, then FETCH_SIZE / WRITE_SIZE in separate counter-only passes.
# Summaries land in gpurun_out/summary_r06/ (copy to profiles/).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=$R/gpurun_out/summary_r06
mkdir -p $S
cd /tmp
run_case() {   # tag, filter for the PMC average, command...
  local TAG=$1 FLT=$2; shift 2
  local OUT=$R/gpurun_out/prof_$TAG
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- "$@" > $OUT/trace.log 2>&1
  grep -a "^{" $OUT/trace.log | tail -1 > $S/r06_${TAG}_bench_line_under_rocprof.json
  python $R/tools/kernel_stats_csv.py $OUT/trace $S/r06_${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $*" || tail -3 $OUT/trace.log
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- "$@" > $OUT/pmc_$C.log 2>&1 || tail -3 $OUT/pmc_$C.log
  done
  python $R/tools/pmc_average.py $OUT "$FLT" > $S/r06_${TAG}_pmc_raw.json
  rm -rf $OUT
}
P="python $R/bench.py --no-cpu-baseline --no-hbm-case"
CASES=${1:-p2 shuffled renumbered p2_renumbered}
for CASE in $CASES; do
  case $CASE in
    p2) run_case p2 k_ $P --workload p2 --steps 2 --warmup 1 ;;
    p2_split) run_case p2_split_slices k_ $P --workload p2 --steps 2 --warmup 1 ;;
    shuffled) run_case sell_unstructured_shuffled k_ $P --cells 215 --mesh shuffled --steps 1 --warmup 1 ;;
    renumbered) run_case sell_unstructured_renumbered k_ $P --cells 215 --mesh renumbered --steps 1 --warmup 1 ;;
    p2_renumbered) run_case p2_file_order_renumbered k_ $P --workload p2 --mesh renumbered --steps 1 --warmup 1 ;;
  esac
done
ls -la $S
