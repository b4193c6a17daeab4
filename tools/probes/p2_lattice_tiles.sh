#!/bin/bash
# configs[3] in lattice order: the tile product (k_lattice_spmv) against the work-item product (k_dict_spmv), alone and inside the solve
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
{
echo "# FS_LATTICE_DEBUG=1 FS_LATTICE_CHECK=1 python tools/probes/p2_lattice_probe.py 107  (products alone, no dots, 10 launches each)"
FS_LATTICE_DEBUG=1 FS_LATTICE_CHECK=1 python $R/tools/probes/p2_lattice_probe.py 107 2>&1 | grep -E "lattice tiles" | head -n 6
echo "# python tools/probes/p2_lattice_probe.py 107  (inside the solve: product with the three dots, sampled with events)"
python $R/tools/probes/p2_lattice_probe.py 107 2>&1 | tail -n 5
} > $R/gpurun_out/r05_p2_lattice_tiles.txt 2>&1
cat $R/gpurun_out/r05_p2_lattice_tiles.txt
