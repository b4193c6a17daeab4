#!/bin/bash
# rocprofv3 kernel stats of configs[4] (2 time steps), run through gpurun
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_ns
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o ns -- python $R/tools/config5_probe.py 43 2 > $OUT/run.log 2>&1
grep "solve \|DOF" $OUT/run.log
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/*.db")[0]); c = db.cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc limit 22"))
for n, k, t, a in rows:
    print("%-72s %7d %10.1f ms %9.1f us" % (n[:72], k, t / 1e6, a / 1e3))
PY
