"""bench.py on one GPU: the JSON line the driver parses (contract of the task: metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config + roofline + cpu_baseline), for the default command
and for the side workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_default_command_prints_the_contract_line(gpu):
    d = _bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["unit"] == "DOF/s" and d["dtype"] == "f64" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "configs[1]" in d["config"]["workload"] and d["config"]["n_dof"] == 1000000 and d["config"]["cg_iterations"] == 293
    assert d["config"]["true_rel_residual"] <= 1.01e-8
    assert abs(d["value"] - 1e6 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-3
    # a roofline fraction: the bytes the kernel's storage form has to move over its duration, never above the peak
    assert 0.3 <= r["frac"] <= 1.0 and abs(r["achieved"] - r["required_bytes_per_launch"] / r["avg_launch_ms"] / 1e6) <= 1e-3 * r["achieved"]
    assert r["csr_equivalent_GBps"] > r["achieved"] and 0.3 <= r["iteration"]["frac"] <= 1.0 and 0.5 <= r["update_kernel"]["frac"] <= 1.0
    assert d["one_shot_dof_per_s"] < d["value"] and r["one_shot"]["dof_per_s"] < r["dof_per_s"]
    # the first step of the process is reported by itself, the cold figure includes fs_init
    assert d["first_step_ms"] > d["ms_per_step"] and d["cold_one_shot_dof_per_s"] < d["one_shot_dof_per_s"] and d["init_ms"] > 0
    # the BASELINE operator has repeated rows: the line says so and carries the streaming kernel's roofline beside it
    assert ("k_box_spmv" in r["kernel"] or "k_dict_spmv" in r["kernel"]) and "note_row_dictionary" in r
    # ... and the same kernels where the Infinity Cache cannot help (85.8 M rows), beside the 10 M-row fractions
    cf = r["cache_free_case"]
    assert "error" not in cf, cf
    assert cf["required_bytes_per_launch"] == 26 * 441 ** 3 and 0.3 <= cf["frac"] <= 1.0 and 0.3 <= cf["iteration"]["frac"] <= 1.0
    s = r["streaming_kernel"]
    assert "k_dia_pair_spmv" in s["kernel"] and 0.5 <= s["frac"] <= 1.0 and s["cg_iterations"] == r["cg_iterations"] == 451
    # the step workload itself (1 M rows, cache-resident): ONE launch per CG iteration, priced on the 90 B/row that launch moves
    k = d["dominant_kernel_on_step_workload"]
    assert "k_dict_cg_iter" in k["kernel"] and k["fused_iteration"] == 1 and k["required_bytes_per_launch"] == 90 * 1000000
    assert k["update_kernel"]["frac"] is None and k["iteration"]["required_bytes"] == 90 * 1000000 and 0.0 < k["iteration"]["frac"] <= 1.0
    assert d["update_kernel_ms"] == 0.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "DOF/s" and c["sample"]
    assert d["parity"]["iterations_gpu"] == d["parity"]["iterations_cpu"] and d["parity"]["max_rel_diff_solution"] <= 1e-9


@pytest.mark.parametrize("args,expect", [
    (("--cells", "23", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-case"), "P1 Poisson"),
    (("--workload", "p2", "--cells", "15", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"), "configs[3]"),
    (("--cells", "15", "--mesh", "renumbered", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-case"), "RANDOMLY PERMUTED"),
    (("--workload", "p2", "--cells", "11", "--mesh", "renumbered", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"), "RANDOMLY PERMUTED"),
    (("--workload", "p2", "--cells", "11", "--mesh", "shuffled", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"), "renumbering off"),
])
def test_side_workloads_print_a_line(gpu, args, expect):
    d = _bench(*args)
    assert expect in d["config"]["workload"] and d["value"] > 0 and d["config"]["true_rel_residual"] <= 1.1e-8
    assert "roofline" in d and d["n_gpus"] == 1
    if "--workload" in args:       # the CG2 legs carry their own check: the linear profile is in the space
        assert d["parity"]["max_abs_error_vs_exact_profile"] <= 1e-4
