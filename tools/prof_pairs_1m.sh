#!/bin/bash
# kernel durations of the CG product at configs[1] with the paired-slice kernel forced on (gpurun)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_pairs
rm -rf $OUT; mkdir -p $OUT
cd /tmp
FS_SPMV_PAIRS=${1:-1} rocprofv3 --kernel-trace --stats -d $OUT -o p -- python $R/tools/solve_overhead_probe.py 99 > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/*.db")[0]); c = db.cursor()
for n, k, t, a, mn in c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by 3 desc limit 8"):
    print("%-90s %7d %9.1f ms avg %7.2f us min %7.2f us" % (n[:90], k, t / 1e6, a / 1e3, mn / 1e3))
PY
