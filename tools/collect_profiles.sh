#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats of the default bench command and
# separate PMC passes (FETCH_SIZE / WRITE_SIZE never share a pass: TCC has 4 slots, they need 3+2).
# Outputs land in gpurun_out/; tools/summarize_profiles.py condenses them into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cache-free-case"      # (three marker phases: tools/summarize_profiles.py)
echo "== kernel trace + stats"; rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log
mkdir -p $R/gpurun_out/summary
grep -a "^{" $OUT/stats.log | tail -1 > $R/gpurun_out/summary/${1:-r06}_bench_line_under_rocprof.json      # the line printed INSIDE the traced run
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; rocprofv3 --pmc $C -d $OUT/pmc_$C -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cache-free-case > $OUT/pmc_$C.log 2>&1
  tail -1 $OUT/pmc_$C.log
done
# condense on the box: the rocpd databases are tens of MB, gpurun copies back at most 64 MiB
python $R/tools/summarize_profiles.py ${1:-r06} $R/gpurun_out/summary > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $R/gpurun_out/summary
