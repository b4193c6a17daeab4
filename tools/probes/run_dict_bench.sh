python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "dictionary or box_assembly or box_snap" 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>gpurun_out/b1.err > gpurun_out/b1.log
python - <<EOF2
import json
d=json.loads([l for l in open("gpurun_out/b1.log") if l.startswith("{")][0])
print("P1 1M:", d["value"], d["ms_per_step"], "spmv", d["dominant_kernel_on_step_workload"]["avg_launch_ms"], "upd", d["update_kernel_ms"])
r=d["roofline"]; print("P1 10M: spmv", r["avg_launch_ms"], r["frac"], r["dof_per_s"], "iter", r["iteration"]["ms"], r["iteration"]["frac"], "upd", r["update_kernel"]["avg_launch_ms"])
EOF2
python bench.py --workload p2 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/b2.err > gpurun_out/b2.log
python - <<EOF2
import json
d=json.loads([l for l in open("gpurun_out/b2.log") if l.startswith("{")][0])
r=d["roofline"]; print("P2:", d["value"], d["ms_per_step"], "spmv", r["avg_launch_ms"], r["frac"], "iter", r["iteration"]["ms"], r["iteration"]["frac"], "upd", r["update_kernel"]["avg_launch_ms"])
EOF2
