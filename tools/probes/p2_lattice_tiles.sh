#!/bin/bash
# configs[3] in lattice order: the tile product (k_lattice_spmv) against the work-item product (k_dict_spmv<.., 12>) on the SAME lattice-ordered
# operator - alone, with the three fused dots, with 1 GB streamed between the launches - and the two NUMBERINGS inside the solve (lattice 0: the
# work-item product in the space's own numbering, lattice 1: the tile product in lattice order)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
{
echo "# FS_LATTICE_DEBUG=2 FS_LATTICE_CHECK=1 python tools/probes/p2_lattice_probe.py 107  (both products on the lattice-ordered operator: 10 launches back to back, then single launches with events)"
FS_LATTICE_DEBUG=2 FS_LATTICE_CHECK=1 python $R/tools/probes/p2_lattice_probe.py 107 2>&1 | grep -E "lattice tiles" | head -n 17
echo "# python tools/probes/p2_lattice_probe.py 107  (inside the solve: product with the three dots, sampled with events)"
python $R/tools/probes/p2_lattice_probe.py 107 2>&1 | tail -n 5
} > $R/gpurun_out/r05_p2_lattice_tiles.txt 2>&1
cat $R/gpurun_out/r05_p2_lattice_tiles.txt
