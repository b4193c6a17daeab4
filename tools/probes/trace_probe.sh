#!/bin/bash
# ON THE GPU BOX: kernel-trace durations of the CG kernels of tools/probes/fused_iter_probe.py (arguments passed on); works for
# timing ablations whose solves break down (the launches before the breakdown are in the trace); PROBE=p2_iter_probe.py: configs[3]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/fs_tr; mkdir -p /tmp/fs_tr; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/fs_tr -o run -- python $R/tools/probes/${PROBE:-fused_iter_probe.py} "$@" > /tmp/fs_tr/log 2>&1
tail -2 /tmp/fs_tr/log
F=$(find /tmp/fs_tr -name "*kernel_trace.csv" | head -1)
python - "$F" <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"]
    if any(k in nm for k in ("k_dict_cg_iter", "k_dict_spmv", "k_cg_update_scaled", "k_dia_pair")):
        d[nm.split("(")[0][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    live = sorted(x for x in v if x > 0.5 * max(v))
    print("  %-48s calls %5d  live %5d  median of live %8.2f us" % (k, len(v), len(live), live[len(live) // 2]))
P
