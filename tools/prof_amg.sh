#!/bin/bash
# rocprofv3 kernel stats of the AMG-preconditioned solve of configs[2] (run through gpurun)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_amg
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o amg -- python $R/tools/config3_amg_probe.py ${1:-2} > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/*.db")[0]); c = db.cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc limit 28"))
tot = sum(r[2] for r in rows)
for n, k, t, a in rows:
    print("%-70s %7d %10.1f ms %9.1f us" % (n[:70], k, t / 1e6, a / 1e3))
PY
