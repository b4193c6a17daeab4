import sys, json
d = json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["dominant_kernel_on_step_workload"]["avg_launch_ms"], d["update_kernel_ms"], d["roofline"]["avg_launch_ms"], d["roofline"].get("dof_per_s"), d["config"]["cg_iterations"], d["config"]["true_rel_residual"])
