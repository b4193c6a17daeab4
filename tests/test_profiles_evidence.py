"""The committed measurement files of the CURRENT round must be internally consistent (the round-4 summaries put the
kernels of both problem sizes under one phase and nobody looked).  CPU-only: reads profiles/<round>_*.json / .csv."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def current_round():
    with open(os.path.join(ROOT, "bench.py")) as fh:
        return re.search(r'^PROFILE_ROUND = "(r\d+)"', fh.read(), re.M).group(1)


def test_bench_reads_traffic_from_the_newest_round_only():
    tag = current_round()
    rounds = sorted(re.match(r"(r\d+)_pmc\.json$", f).group(1) for f in os.listdir(PROFILES) if re.match(r"r\d+_pmc\.json$", f))
    assert rounds[-1] == tag, "bench.PROFILE_ROUND = %s but the newest committed PMC summary is %s" % (tag, rounds[-1])
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def committed_traffic"):src.index("def make_roofline")]
    assert "r04_pmc" not in body and "r03_pmc" not in body      # no silent fall-through to an older round's kernel


def test_pmc_summary_of_the_current_round_is_phased_and_calibrated():
    tag = current_round()
    with open(os.path.join(PROFILES, tag + "_pmc.json")) as fh:
        pmc = json.load(fh)
    assert {"n99_1M_dof", "n215_10M_dof", "n215_10M_dof_streaming"} <= set(pmc["phases"])
    cal = pmc["calibration"]
    reads = {k: v["ratio"] for k, v in cal.items() if "expected_read_bytes" in v}
    writes = {k: v["ratio"] for k, v in cal.items() if "expected_write_bytes" in v}
    # FETCH_SIZE tallies 128-byte requests as 64: kernels of known byte count read 0.50 of their bytes, in EVERY phase
    assert any(k.startswith("n99_1M_dof/") for k in reads) and any(k.startswith("n215_10M_dof") for k in reads), reads
    for k, ratio in reads.items():
        assert 0.49 <= ratio <= 0.51, (k, ratio)
    for k, ratio in writes.items():
        assert 0.97 <= ratio <= 1.05, (k, ratio)
    for key in ("spmv_dict_n215", "spmv_fused_n215", "cg_iter_n99"):
        assert pmc.get(key), key
    # the traffic of each kernel against the bytes its storage form must move (bench.py's required-bytes models)
    n99, n215 = 100 ** 3, 216 ** 3
    assert 0.95 <= pmc["cg_iter_n99"] / (90.0 * n99) <= 1.15
    assert 0.95 <= pmc["spmv_dict_n215"] / (26.0 * n215) <= 1.15
    assert pmc["spmv_fused_n215"] > 4 * pmc["spmv_dict_n215"]       # the streaming product moves the matrix


def test_kernel_stats_of_the_current_round_have_both_problem_sizes():
    tag = current_round()
    with open(os.path.join(PROFILES, tag + "_kernel_stats.csv")) as fh:
        rows = [r for r in csv.reader(fh) if r and not r[0].startswith("#")]
    head, rows = rows[0], rows[1:]
    ph, kern, live = head.index("phase"), head.index("kernel"), head.index("live_avg_us")
    phases = {r[ph] for r in rows}
    assert {"n99_1M_dof", "n215_10M_dof", "n215_10M_dof_streaming"} <= phases, phases
    by = {(r[ph], r[kern]): float(r[live]) for r in rows}
    it = [v for (p, k), v in by.items() if p == "n99_1M_dof" and k.startswith("k_dict_cg_iter<3")]
    prod = [v for (p, k), v in by.items() if p == "n215_10M_dof" and k.startswith(("k_box_spmv<3", "k_dict_spmv<3"))]
    assert it and prod
    assert not any(p == "n99_1M_dof" and k.startswith("k_dia_pair_spmv<3") for p, k in by)     # a 10 M-row kernel under the 1 M phase
    assert prod[0] > 2.0 * it[0]


def test_bench_line_of_the_current_round_names_its_own_pmc_file():
    tag = current_round()
    with open(os.path.join(PROFILES, tag + "_bench_line.json")) as fh:
        line = json.loads(fh.read().strip().splitlines()[-1])
    roof = line["roofline"]
    assert tag + "_pmc.json" in roof["traffic_source"]
    with open(os.path.join(PROFILES, tag + "_pmc.json")) as fh:
        assert roof["traffic"] == json.load(fh)["spmv_dict_n215"]
    assert 0.0 < roof["frac"] <= 1.0
