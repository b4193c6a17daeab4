"""rocprofv3 kernel-trace database -> profiles/<tag>_kernel_stats.csv (per-kernel calls / total / avg / min / max).
usage: kernel_stats_csv.py <dir with the rocpd .db> <out.csv> "<header comment>" """
import csv
import glob
import re
import sqlite3
import sys

src, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
db = sqlite3.connect(glob.glob(src + "/*.db")[0])
rows = list(db.cursor().execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


with open(out, "w", newline="") as fh:
    fh.write('"# %s"\n' % note)
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct_of_gpu_time"])
    for n, k, t, a, mn, mx in rows[:40]:
        w.writerow([short(n), k, "%.2f" % (t / 1e6), "%.1f" % (a / 1e3), "%.1f" % (mn / 1e3), "%.1f" % (mx / 1e3), "%.2f" % (100.0 * t / tot)])
print("wrote", out)
