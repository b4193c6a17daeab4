for b in 256 512 768 1024 1536; do
FS_DICT_BLOCKS=$b python bench.py --no-cpu-baseline --no-hbm-case --steps 10 --warmup 2 2>/dev/null > gpurun_out/bb.log
python - <<EOF2
import json
d=json.loads([l for l in open("gpurun_out/bb.log") if l.startswith("{")][0])
print("blocks $b P1 1M:", d["value"], d["ms_per_step"], "spmv", d["dominant_kernel_on_step_workload"]["avg_launch_ms"], "upd", d["update_kernel_ms"])
EOF2
done
