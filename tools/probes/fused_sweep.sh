#!/bin/bash
# ON THE GPU BOX: kernel-trace durations of the one-launch iteration at 1 M rows for a few launch geometries / variants
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for CFG in "FS_DICT_BLOCKS=1024 FS_CG_FUSED_VARIANT=0" "FS_DICT_BLOCKS=512 FS_CG_FUSED_VARIANT=0" "FS_DICT_BLOCKS=2048 FS_CG_FUSED_VARIANT=0" "FS_DICT_BLOCKS=768 FS_CG_FUSED_VARIANT=0" "FS_DICT_BLOCKS=1024 FS_CG_FUSED_VARIANT=1" "FS_DICT_BLOCKS=1024 FS_CG_FUSED_VARIANT=2" "FS_DICT_BLOCKS=512 FS_CG_FUSED_VARIANT=2" "FS_DICT_BLOCKS=1024 FS_CG_FUSED_VARIANT=3"; do
  rm -rf /tmp/fs_tr; mkdir -p /tmp/fs_tr
  env $CFG rocprofv3 --kernel-trace --stats -d /tmp/fs_tr -o run -- python $R/tools/probes/fused_iter_probe.py ${1:-99} > /tmp/fs_tr/log 2>&1
  echo "== $CFG"
  grep "fused=1" /tmp/fs_tr/log | tail -1
  F=$(find /tmp/fs_tr -name "*kernel_stats.csv" | head -1)
  python - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    nm = r["Name"]
    if any(k in nm for k in ("k_dict_cg_iter", "k_dict_spmv", "k_cg_update_scaled")):
        print("   %-40s calls %6s avg %8.2f us  min %8.2f  max %8.2f" % (nm[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
done
