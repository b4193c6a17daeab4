#!/usr/bin/env python3
"""Condense the rocprofv3 rocpd databases written by tools/collect_profiles.sh
(gpurun_out/prof/...) into the small tracked summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   per (phase, kernel): calls, total/avg/min/max us, % of GPU time
  profiles/<tag>_pmc_raw.json       per-launch FETCH_SIZE / WRITE_SIZE of the hot kernels as reported
                                    (the gfx950 correction of the read side is applied in
                                    profiles/<tag>_pmc.json, see profiles/README.md)

The default bench command runs two problem sizes back to back; dispatches are attributed to a
phase by time: before / after the last sparsity build (last k_pair_keys dispatch).
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
PHASES = ("n99_1M_dof", "n215_10M_dof")
HOT = ("k_sell_spmv", "k_dia_pair_spmv", "k_dict_spmv", "k_dict_cg_iter", "k_cg_update", "k_assemble", "k_dot", "k_dirichlet", "k_residual", "k_sum_partials")


def short(name):
    name = re.sub(r"^void ", "", name)
    if "rocprim" in name:
        m = re.search(r"detail::(radix_sort_\w+|partition_impl|scan_\w+|lookback_scan\w*|init_\w+|\w+_kernel)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return name.split("(")[0]


def phase_split(cur, table, name_col, start_col):
    rows = cur.execute("select %s from %s where %s like 'k_pair_keys%%' order by %s"
                       % (start_col, table, name_col, start_col)).fetchall()
    # (round 4: bench.py builds the 1 M-DOF pattern twice - cold and warm - so the 10 M-DOF phase starts at the LAST build)
    return rows[-1][0] if len(rows) > 1 else None


def kernel_stats():
    db = sqlite3.connect(os.path.join(SRC, "stats", "bench_results.db"))
    cur = db.cursor()
    split = phase_split(cur, "kernels", "name", "start")
    agg = {}
    for name, start, dur in cur.execute("select name, start, duration from kernels"):
        ph = PHASES[0] if split is None or start < split else PHASES[1]
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
        # live_* exclude the no-op launches of a CG batch enqueued after convergence (< 10 % of the max)
        w.writerow(["phase", "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_gpu_time",
                    "live_calls", "live_avg_us"])
        for (ph, k), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            live = [d for d in a[4] if d >= 0.1 * a[3]]
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
        ph = PHASES[0] if split is None or start < split else PHASES[1]
        vals.setdefault((ph, short(name)), []).append(val)
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
    n_dof = {PHASES[0]: 100 ** 3, PHASES[1]: 216 ** 3}
    cal = {}
    for ph in PHASES:
        n = n_dof[ph]
        # streams of known size: dot (1 read), residuals (2 / 3 reads), CG updates (reads r,w,p,s,x [+z,dinv])
        for k, expect in (("k_dot_partial", 8 * n), ("k_residual", 16 * n), ("k_residual_scaled", 24 * n),
                          ("k_cg_update<true>", 56 * n), ("k_cg_update_scaled<true>", 40 * n)):
            if (ph, k) in fetch:
                cal["%s/%s" % (ph, k)] = {"expected_read_bytes": expect,
                                          "FETCH_SIZE_bytes": int(fetch[(ph, k)] * 1024),
                                          "ratio": round(fetch[(ph, k)] * 1024 / expect, 4)}
        for k, expect in (("k_cg_update<true>", 40 * n), ("k_cg_update_scaled<true>", 32 * n)):
            if (ph, k) in write:
                cal["%s/%s(write)" % (ph, k)] = {"expected_write_bytes": expect,
                                                 "WRITE_SIZE_bytes": int(write[(ph, k)] * 1024),
                                                 "ratio": round(write[(ph, k)] * 1024 / expect, 4)}
    out = {"_doc": "HBM-side bytes per launch of the dominant kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "(read side doubled per the gfx950 rule, confirmed by the calibration block). "
                   "bench.py reports these as roofline.traffic.",
           "calibration": cal}
    for ph, tag in ((PHASES[0], "spmv_fused_n99"), (PHASES[1], "spmv_fused_n215")):
        # DOTS template argument: 3 = in-CG kernel of the diagonally scaled solve, 1 = unscaled CG /
        # fs_spmv_benchmark(fused), 0 = bare SpMV
        # the launch shape (last template argument = entries per round) depends on the problem size
        def find(dots):
            # ... and so does the 4th one (non-temporal matrix loads when the matrix exceeds the caches)
            for un in ("4", "16", "8", "2"):
                for tail in ("", ", false", ", true"):
                    k = (ph, "k_sell_spmv<1, %s, %s%s>" % (dots, un, tail))
                    if k in fetch:
                        return k
            return None
        for dots in ("3", "1"):
            key = find(dots)
            if key is not None:
                out[tag + ("" if dots == "3" else "_dots1")] = int((2 * fetch[key] + write.get(key, 0.0)) * 1024)
            # HBM-resident sizes: the product is k_dia_pair_spmv (paired DIA slices, two rows per lane) + k_sell_spmv on the
            # few unpaired slices - one product = one launch of each
            for nt in ("true", "false"):
                pk = (ph, "k_dia_pair_spmv<%s, %s>" % (dots, nt))
                if pk in fetch:
                    total = (2 * fetch[pk] + write.get(pk, 0.0)) * 1024
                    if key is not None:
                        total += (2 * fetch[key] + write.get(key, 0.0)) * 1024
                    out[tag + ("" if dots == "3" else "_dots1")] = int(total)
                    out[tag + ("" if dots == "3" else "_dots1") + "_kernels"] = [pk[1]] + ([key[1]] if key is not None else [])
        if tag not in out and tag + "_dots1" in out:
            out[tag] = out[tag + "_dots1"]
        key = find("0")
        bare = 0.0 if key is None else (2 * fetch[key] + write.get(key, 0.0)) * 1024
        for nt in ("true", "false"):
            pk = (ph, "k_dia_pair_spmv<0, %s>" % nt)
            if pk in fetch:
                bare += (2 * fetch[pk] + write.get(pk, 0.0)) * 1024
        if bare > 0.0:
            out[tag.replace("fused", "bare")] = int(bare)
        for dk in sorted(fetch):              # row-dictionary form of the same product (dictionary in LDS / class rows per work item;
            if dk[0] == ph and (dk[1].startswith("k_dict_spmv<3, true") or dk[1].startswith("k_dict_spmv<3, false")):    # last argument: run length)
                out[tag.replace("spmv_fused", "spmv_dict")] = int((2 * fetch[dk] + write.get(dk, 0.0)) * 1024)
        for ik in sorted(fetch):                          # the one-launch CG iteration (update k + product k + 1; up to 3 M rows;
            if ik[0] == ph and ik[1].startswith("k_dict_cg_iter<3"):      # second template argument: decomposed space)
                out[tag.replace("spmv_fused", "cg_iter")] = int((2 * fetch[ik] + write.get(ik, 0.0)) * 1024)
        key = (ph, "k_assemble_p1_scalar_gather<false>")
        if key in fetch:
            out[tag.replace("spmv_fused", "assemble")] = int((2 * fetch[key] + write.get(key, 0.0)) * 1024)
    json.dump(out, open(os.path.join(OUT, TAG + "_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
