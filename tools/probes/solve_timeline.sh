#!/bin/bash
# ON THE GPU BOX: the kernels of ONE solve in time order (start offset, duration, gap to the previous one) - what the fixed cost of a
# solve consists of.  Traces tools/probes/solve_fixed_cost.py and prints the last solve with max_iter = 1.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/fs_tl; mkdir -p /tmp/fs_tl; cd /tmp
FS_FIXED_ONLY=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fs_tl -o run -- python $R/tools/probes/solve_fixed_cost.py > /tmp/fs_tl/log 2>&1
K=$(find /tmp/fs_tl -name "*kernel_trace.csv" | head -1); M=$(find /tmp/fs_tl -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" <<'P'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "") + " " + r.get("Bytes", "")))
except Exception:
    pass
ev.sort()
# the last solve: from the last k_extract_dinv on
last = max(i for i, e in enumerate(ev) if "k_extract_dinv" in e[2])
t0, prev = ev[last][0], ev[last][0]
for s, e, n in ev[last:]:
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    prev = e
P
