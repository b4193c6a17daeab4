#!/bin/bash
# rocprofv3 kernel stats of configs[4] (2 time steps), run through gpurun
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_ns
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o ns -- env FS_SADDLE_DEBUG=1 python $R/tools/config5_probe.py 43 2 > $OUT/run.log 2>&1
grep "solve\|DOF" $OUT/run.log
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/*.db")[0]); c = db.cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc limit 40"))
for n, k, t, a in rows:
    print("%-72s %7d %10.1f ms %9.1f us" % (n[:72], k, t / 1e6, a / 1e3))
# idle analysis: where does the GPU wait inside the FGMRES iterations?
ks = list(c.execute("select name, start, end from kernels order by start"))
import collections
gap_after = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for (n0, s0, e0), (n1, s1, e1) in zip(ks[:-1], ks[1:]):
    busy += (e0 - s0)
    g = s1 - e0
    if g < 5e6:      # ignore the pauses between python calls
        gap_after[n0[:40] + " -> " + n1[:30]][0] += 1
        gap_after[n0[:40] + " -> " + n1[:30]][1] += g
# one FGMRES solve = from k_sd_diag to the kernel before the next k_assemble_ns / k_sd_diag
cur = None
for n0, s0, e0 in ks:
    if n0.startswith("k_sd_diag"):
        cur = [s0, e0, 0.0, 0]
    if cur is not None:
        if n0.startswith("k_assemble_ns") or n0.startswith("k_axpy"):
            print("solve: span %.1f ms, busy %.1f ms, %d kernels" % ((cur[1] - cur[0]) / 1e6, cur[2] / 1e6, cur[3]))
            cur = None
        else:
            cur[1] = e0; cur[2] += e0 - s0; cur[3] += 1
print("busy %.1f ms, span %.1f ms" % (busy / 1e6, (ks[-1][2] - ks[0][1]) / 1e6))
for k2, (cnt, tot) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:14]:
    print("gap %-74s %6d %9.1f ms %8.1f us" % (k2, cnt, tot / 1e6, tot / cnt / 1e3))
PY

python $R/tools/trace_window.py $OUT
