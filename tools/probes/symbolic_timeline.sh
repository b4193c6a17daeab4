#!/bin/bash
# ON THE GPU BOX: the kernels of ONE warm set-up (mesh + sparsity pattern + tables) of the n = 99 box in time order.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/fs_sy; mkdir -p /tmp/fs_sy; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fs_sy -o run -- python $R/tools/probes/symbolic_cold_warm.py > /tmp/fs_sy/log 2>&1
K=$(find /tmp/fs_sy -name "*kernel_trace.csv" | head -1); M=$(find /tmp/fs_sy -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" <<'P'
import csv, sys, re
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"wrapped_(\w+?)_config<[^,]*, ([^>]*?)>", n)
    short = ("rocprim " + m.group(1) + " <" + m.group(2)[:40] + ">") if m else n.split("(")[0][:70]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")[12:] + " " + r.get("Bytes", "")))
except Exception:
    pass
ev.sort()
# the last set-up: from the last k_box_vertices on, up to the assembly kernel after it
last = max(i for i, e in enumerate(ev) if "k_box_vertices" in e[2])
t0 = prev = ev[last][0]
busy = 0.0
for s, e, n in ev[last:]:
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    busy += (e - s) / 1e3
    prev = e
    if "k_assemble" in n: break
print("kernels and copies busy %.1f us of %.1f us" % (busy, (prev - t0) / 1e3))
P
