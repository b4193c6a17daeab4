#!/bin/bash
# ON THE GPU BOX: the kernels and copies of ONE step of the default bench command in time order, the CG iteration launches folded
# into one line - what a step consists of besides the iterations.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/fs_st; mkdir -p /tmp/fs_st; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fs_st -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/fs_st/log 2>&1
K=$(find /tmp/fs_st -name "*kernel_trace.csv" | head -1); M=$(find /tmp/fs_st -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" <<'P'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "") + " " + r.get("Bytes", "")))
except Exception:
    pass
ev.sort()
# solves = runs of consecutive iteration launches
runs = []
for i, e in enumerate(ev):
    if "k_dict_cg_iter" in e[2]:
        if runs and i - runs[-1][1] <= 4: runs[-1][1] = i        # (a status copy between two batches of launches)
        else: runs.append([i, i])
start = runs[1][1] + 1        # (bench.py goes on to other legs after the timed steps: the window is the tail of solve 2 .. the iterations of solve 3)
end = runs[2][1] + 1
t0 = ev[start][0]; prev = t0
n_it = 0; it_t = 0.0; it_first = None
for s, e, n in ev[start:end]:
    if "k_dict_cg_iter" in n:
        n_it += 1; it_t = (e - it_first) / 1e3 if it_first else 0.0
        if it_first is None: it_first = s
        prev = e
        continue
    if n_it:
        print("          ... %d x k_dict_cg_iter, %.1f us from the first start to the last end" % (n_it, it_t)); n_it = 0; it_first = None
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    prev = e
P
grep -a "^{" /tmp/fs_st/log | tail -1 | cut -c1-200
