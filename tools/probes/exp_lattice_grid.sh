# k_lattice_spmv: the number of tile workgroups behind the column / corner workgroups (FS_LATTICE_TILE_WGS), configs[3]:
# the product alone (FS_LATTICE_DEBUG), then the solve without any check
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -f gpurun_out/exp_var.txt
for v in ${1:-1024 1536 2048 3072}; do
  echo "== FS_LATTICE_TILE_WGS=$v" >> gpurun_out/exp_var.txt
  env FS_LATTICE_TILE_WGS=$v FS_LATTICE_DEBUG=2 FS_LATTICE_CHECK=1 python tools/probes/p2_lattice_probe.py 107 2>&1 | grep -E "rows differ|lattice tiles\]   tile" | head -12 >> gpurun_out/exp_var.txt
  env FS_LATTICE_TILE_WGS=$v python tools/probes/p2_lattice_probe.py 107 2>&1 | grep -E "n=107 lattice 1" >> gpurun_out/exp_var.txt
done
cat gpurun_out/exp_var.txt
