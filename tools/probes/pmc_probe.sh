#!/bin/bash
# ON THE GPU BOX: hardware counters of the CG kernels of the iteration probe, one rocprofv3 --pmc pass per counter group
# (counter-only passes: no trace options beside them).  bash tools/probes/pmc_probe.sh "<probe args>" GROUP1 GROUP2 ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; shift
cd /tmp
for G in "$@"; do
  rm -rf /tmp/fs_pmc; mkdir -p /tmp/fs_pmc
  FS_CG_FUSED=0 rocprofv3 --pmc $G --output-format csv -d /tmp/fs_pmc -o run -- python $R/tools/probes/fused_iter_probe.py $ARGS > /tmp/fs_pmc/log 2>&1
  F=$(find /tmp/fs_pmc -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "== $G: no counter file"; tail -3 /tmp/fs_pmc/log; continue; fi
  python - "$F" "$G" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].split("(")[0]
    if any(k in nm for k in ("k_dict_spmv<3", "k_cg_update_scaled")):
        acc[nm[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("  %-40s %s" % (k, "  ".join("%s %.4g" % (c, sorted(v)[len(v) * 3 // 4]) for c, v in sorted(d.items()))))
P
done
