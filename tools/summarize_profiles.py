#!/usr/bin/env python3
"""Condense the rocprofv3 rocpd databases written by tools/collect_profiles.sh
(gpurun_out/prof/...) into the small tracked summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   per (phase, kernel): calls, total/avg/min/max us, % of GPU time
  profiles/<tag>_pmc_raw.json       per-launch FETCH_SIZE / WRITE_SIZE of the hot kernels as reported
                                    (the gfx950 correction of the read side is applied in
                                    profiles/<tag>_pmc.json, see profiles/README.md)

The default bench command runs three legs back to back (configs[1] at 1 M DOF; the 10 M-DOF cube with the
row-dictionary product; the same cube with the streaming product).  bench.py enqueues an empty kernel named
k_profile_marker (fs_profile_marker) before each leg; a dispatch belongs to the phase whose marker is the last
one before it.  A trace WITHOUT the markers is refused: splitting on the name of a set-up kernel went wrong once
a round removed that kernel (round 4: every launch of both sizes under one phase, calibration 4.13).
"""
import csv
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
# second argument: output directory (default profiles/).  On the GPU box the summaries are written under gpurun_out/
# and the multi-10-MB databases are deleted before gpurun copies the directory back.
OUT = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
# phase 0 = before the first marker (fs_init, code-object loading); then one phase per marker, in bench.py's order
PHASES = ("init", "n99_1M_dof", "n215_10M_dof", "n215_10M_dof_streaming")
N_DOF = {"n99_1M_dof": 100 ** 3, "n215_10M_dof": 216 ** 3, "n215_10M_dof_streaming": 216 ** 3}
MARKER = "k_profile_marker"
HOT = ("k_sell_spmv", "k_dia_pair_spmv", "k_dict_spmv", "k_box_spmv", "k_lat_march", "k_lattice_spmv", "k_dict_cg_iter", "k_cg_update", "k_assemble", "k_dot", "k_dirichlet", "k_residual", "k_sum_partials")


def short(name):
    name = re.sub(r"^void ", "", name)
    if "rocprim" in name:
        m = re.search(r"detail::(radix_sort_\w+|partition_impl|scan_\w+|lookback_scan\w*|init_\w+|\w+_kernel)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return name.split("(")[0]


def phase_split(cur, table, name_col, start_col):
    """Start times of the marker launches, in order.  Exactly len(PHASES) - 1 of them, or the trace is not one of bench.py's."""
    rows = [r[0] for r in cur.execute("select %s from %s where %s like '%%%s%%' order by %s"
                                      % (start_col, table, name_col, MARKER, start_col)).fetchall()]
    rows = sorted(set(rows))            # (a counter pass lists a dispatch once per counter instance)
    if len(rows) != len(PHASES) - 1:
        sys.exit("summarize_profiles: %d %s launches in %s, expected %d (one per leg of the default bench command) - "
                 "refusing to attribute kernels to phases" % (len(rows), MARKER, table, len(PHASES) - 1))
    return rows


def phase_of(marks, start):
    i = 0
    while i < len(marks) and start >= marks[i]:
        i += 1
    return PHASES[i]


def kernel_stats():
    db = sqlite3.connect(os.path.join(SRC, "stats", "bench_results.db"))
    cur = db.cursor()
    split = phase_split(cur, "kernels", "name", "start")
    agg = {}
    for name, start, dur in cur.execute("select name, start, duration from kernels"):
        if MARKER in name:
            continue
        ph = phase_of(split, start)
        a = agg.setdefault((ph, short(name)), [0, 0.0, 1e30, 0.0, []])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
        a[4].append(dur)
    total = sum(a[1] for a in agg.values())
    out = os.path.join(OUT, TAG + "_kernel_stats.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        # live_* exclude the no-op launches of a CG batch enqueued after convergence (< 35 % of the 90th percentile: a launch of the one-launch
        # iteration that returns on the status word still takes 5 us of a live launch's 20)
        w.writerow(["phase", "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_gpu_time",
                    "live_calls", "live_avg_us"])
        for (ph, k), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            ref = sorted(a[4])[int(0.9 * (len(a[4]) - 1))]          # (90th percentile: one slow first launch must not define `live`)
            live = [d for d in a[4] if d >= 0.35 * ref]
            w.writerow([ph, k, a[0], "%.1f" % (a[1] / 1e3), "%.3f" % (a[1] / a[0] / 1e3), "%.3f" % (a[2] / 1e3),
                        "%.3f" % (a[3] / 1e3), "%.2f" % (100 * a[1] / total), len(live),
                        "%.3f" % (sum(live) / len(live) / 1e3)])
    print("wrote", out)


def pmc(counter):
    """Mean counter value per launch and phase.  Launches enqueued after CG converged are no-ops
    (they return on the status word); they are dropped by discarding values < 1% of the maximum."""
    db = sqlite3.connect(os.path.join(SRC, "pmc_" + counter, "bench_results.db"))
    cur = db.cursor()
    split = phase_split(cur, "counters_collection", "kernel_name", "start")
    vals = {}
    for name, start, val in cur.execute(
            "select kernel_name, start, value from counters_collection where counter_name=?", (counter,)):
        if MARKER in name:
            continue
        vals.setdefault((phase_of(split, start), short(name)), []).append(val)
    mean, cnt = {}, {}
    for k, v in vals.items():
        top = max(v)
        live = [x for x in v if x >= 0.01 * top] if top > 0 else v
        mean[k] = sum(live) / len(live)
        cnt[k] = len(live)
    return mean, cnt


def main():
    os.makedirs(OUT, exist_ok=True)
    kernel_stats()
    fetch, nf = pmc("FETCH_SIZE")
    write, _ = pmc("WRITE_SIZE")
    raw = {"_doc": "per-launch means over live launches; FETCH_SIZE/WRITE_SIZE are KiB as reported by rocprofv3 "
                   "(separate --pmc passes, no other trace domain)", "kernels": {}}
    for key in sorted(fetch):
        ph, k = key
        if not k.startswith(HOT):
            continue
        raw["kernels"]["%s/%s" % (ph, k)] = {"launches": nf[key], "FETCH_SIZE_KiB": round(fetch[key], 1),
                                              "WRITE_SIZE_KiB": round(write.get(key, 0.0), 1)}
    json.dump(raw, open(os.path.join(OUT, TAG + "_pmc_raw.json"), "w"), indent=1)
    # gfx950 correction (guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B.
    # Calibrated in THIS run on kernels of known byte count (see profiles/README.md): reads x2, writes x1.
    cal = {}
    for ph in PHASES[1:]:
        n = N_DOF[ph]
        # streams of known size: dot (1 read), residuals (2 / 3 reads).  The CG update kernel (reads r, w, p, s, x; writes r, p, s, x)
        # calibrates the WRITE side only: the tail of w, written by the product just before it, is still in the L2s (its reads come
        # out at 0.447 instead of 0.500 of the bytes at 10 M rows)
        for k, expect in (("k_dot_partial", 8 * n), ("k_residual", 16 * n), ("k_residual_scaled", 24 * n)):
            if (ph, k) in fetch:
                cal["%s/%s" % (ph, k)] = {"expected_read_bytes": expect,
                                          "FETCH_SIZE_bytes": int(fetch[(ph, k)] * 1024),
                                          "ratio": round(fetch[(ph, k)] * 1024 / expect, 4)}
        for k, expect in (("k_cg_update<true>", 40 * n), ("k_cg_update_scaled<true, true>", 32 * n)):
            if (ph, k) in write:
                cal["%s/%s(write)" % (ph, k)] = {"expected_write_bytes": expect,
                                                 "WRITE_SIZE_bytes": int(write[(ph, k)] * 1024),
                                                 "ratio": round(write[(ph, k)] * 1024 / expect, 4)}
    out = {"_doc": "HBM-side bytes per launch of the dominant kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(read side doubled per the gfx950 rule, confirmed by the calibration block: read ratios 0.50, write ratios 1.0). "
                   "bench.py reports these as roofline.traffic.  Phases = the legs of the default bench command, split at its "
                   "k_profile_marker launches.",
           "phases": sorted({ph for ph, _ in fetch}, key=PHASES.index),
           "calibration": cal}

    def hbm(key):
        return (2 * fetch[key] + write.get(key, 0.0)) * 1024

    def first(ph, *prefixes):
        for pre in prefixes:
            for key in sorted(fetch):
                if key[0] == ph and key[1].startswith(pre):
                    return key
        return None

    for ph, size in (("n99_1M_dof", "n99"), ("n215_10M_dof", "n215"), ("n215_10M_dof_streaming", "n215")):
        # streaming product, DOTS template argument 3 (in-CG kernel of the diagonally scaled solve) / 0 (bare): at HBM-resident
        # sizes one product = k_dia_pair_spmv on the paired DIA slices + k_sell_spmv on the rest, one launch of each
        for dots, tag in (("3", "spmv_fused_"), ("0", "spmv_bare_")):
            keys = [k for k in (first(ph, "k_dia_pair_spmv<%s," % dots), first(ph, "k_sell_spmv<1, %s," % dots)) if k is not None]
            if keys:                                   # (the streaming leg comes last and owns the n215 keys)
                out[tag + size] = int(sum(hbm(k) for k in keys))
                out[tag + size + "_kernels"] = [k[1] for k in keys]
        if ph.endswith("streaming"):
            key = first(ph, "k_cg_update_scaled<true, true>")
            if key is not None:
                out["update_" + size] = int(hbm(key))
            continue
        # row-dictionary form of the product: the marching-window kernel of P1 boxes (round 6), else the work-item kernel
        # (dictionary in LDS / class rows per work item)
        key = first(ph, "k_box_spmv<3,", "k_dict_spmv<3,")
        if key is not None:
            out["spmv_dict_" + size] = int(hbm(key))
            out["spmv_dict_" + size + "_kernel"] = key[1]
        key = first(ph, "k_dict_cg_iter<3")         # the one-launch CG iteration (update k + product k + 1; up to 3 M rows)
        if key is not None:
            out["cg_iter_" + size] = int(hbm(key))
        key = first(ph, "k_assemble_p1_box_gather", "k_assemble_p1_scalar_gather")      # (box meshes: the geometry-free form, round 6)
        if key is not None:
            out["assemble_" + size] = int(hbm(key))
    json.dump(out, open(os.path.join(OUT, TAG + "_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
