#!/bin/bash
# ON THE GPU BOX: hardware counters of kernels matching a pattern in ANY command, one rocprofv3 --pmc pass per counter group
#   bash tools/probes/pmc_any.sh "<kernel name pattern>" "<command>" GROUP1 GROUP2 ...
export TMPDIR=/tmp
PAT=$1; CMD=$2; shift 2
cd /tmp
for G in "$@"; do
  rm -rf /tmp/fs_pmc; mkdir -p /tmp/fs_pmc
  rocprofv3 --pmc $G --output-format csv -d /tmp/fs_pmc -o run -- $CMD > /tmp/fs_pmc/log 2>&1
  F=$(find /tmp/fs_pmc -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "== $G: no counter file"; tail -3 /tmp/fs_pmc/log; continue; fi
  python - "$F" "$PAT" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].split("(")[0]
    if sys.argv[2] in nm:
        acc[nm[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("  %-44s %s" % (k, "  ".join("%s %.4g (n=%d)" % (c, sorted(v)[len(v) // 2], len(v)) for c, v in sorted(d.items()))))
P
done
