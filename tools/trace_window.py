"""Print one FGMRES iteration of a rocprofv3 kernel trace (start offset, duration, gap) - used by tools/prof_ns.sh."""
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
c = db.cursor()
ks = list(c.execute("select name, start, end from kernels order by start"))
# the 30th preconditioner application
idx = [i for i, k in enumerate(ks) if k[0].startswith("k_sd_pressure_rows")]
i0 = idx[30] - 2
t0 = ks[i0][1]
prev_end = ks[i0 - 1][2]
for n, s, e in ks[i0:i0 + 64]:
    print("%9.1f us  dur %8.1f  gap %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:60]))
    prev_end = e
try:
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()])
except Exception as ex:
    print(ex)

# gaps > 20 us inside the third FGMRES solve
starts = [i for i, k in enumerate(ks) if k[0].startswith("k_sd_diag")]
if len(starts) > 2:
    i0 = starts[2]
    i1 = next(i for i in range(i0, len(ks)) if ks[i][0].startswith("k_assemble_ns") or ks[i][0].startswith("k_axpy"))
    tot = 0.0
    for i in range(i0 + 1, i1):
        g = (ks[i][1] - ks[i - 1][2]) / 1e3
        if g > 20.0:
            tot += g
            print("gap %8.1f us at +%9.1f us: %s -> %s" % (g, (ks[i][1] - ks[i0][1]) / 1e3, ks[i - 1][0][:40], ks[i][0][:40]))
    print("total of those gaps %.1f ms over %d kernels" % (tot / 1e3, i1 - i0))
