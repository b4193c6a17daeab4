# k_cg_update_scaled: which stores of the non-temporal form are ordinary ones (FS_UPDATE_R_PLAIN bits: 1 r, 2 p, 4 s, 8 x; default 1): configs[3] and the default bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in ${1:-0 1 3 5 9 15 1}; do
  echo "== FS_UPDATE_R_PLAIN=$v"
  FS_UPDATE_R_PLAIN=$v python tools/probes/p2_lattice_probe.py 107 2>&1 | grep -E "n=107 lattice 1" | tail -1 | cut -c1-150
  FS_UPDATE_R_PLAIN=$v python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('   1M %.4f ms/step | 10M product %.2f us frac %.3f update %.1f us iteration %.1f us | streaming product %.1f iteration %.1f' % (d['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*r['update_kernel']['avg_launch_ms'], 1e3*r['iteration']['ms'], 1e3*r['streaming_kernel']['avg_launch_ms'], 1e3*r['streaming_kernel']['iteration']['ms']))"
done
