#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (counters only: no kernel trace in the same run) of one command, averaged per kernel.
# usage: tools/pmc_kernel.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o run -- "$@" > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
  i=$((i+1))
done
python $R/tools/pmc_average.py $OUT ${PMC_FILTER:-k_sell_spmv} > $R/gpurun_out/pmc_$TAG.json
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.json
