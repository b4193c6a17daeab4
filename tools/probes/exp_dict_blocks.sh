# the row-dictionary product's launch geometry at 10 M rows (FS_DICT_BLOCKS; default 1024): the default bench command's roofline fields
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in ${1:-512 768 1024 1536 2048}; do
  FS_DICT_BLOCKS=$v python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('FS_DICT_BLOCKS=$v: 1M %.4f ms/step | 10M product %.2f us frac %.3f update %.1f us iteration %.1f us' % (d['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*r['update_kernel']['avg_launch_ms'], 1e3*r['iteration']['ms']))"
done
