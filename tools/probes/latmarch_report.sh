#!/bin/bash
# profiles/r06_p2_latmarch.txt from what tools/probes/run_latmarch_abl.sh and run_latmarch_iter.sh left in gpurun_out/latmarch/
#   bash tools/probes/latmarch_report.sh > profiles/r06_p2_latmarch.txt
cd "${GRAFT_REPO_ROOT:-$(dirname $0)/../..}"
echo "# k_lat_march at BASELINE configs[3] (n = 107, 9 984 600 rows in lattice order), one MI355X; tools/prof_r06_p2.sh / prof_r06_all.sh"
echo "# 1. the product alone (FS_LATTICE_DEBUG=2: 10 launches back to back / per-launch events; 'tile product' in the library's"
echo "#    messages = the product of the lattice-ordered operator, here k_lat_march) with parts of the kernel switched off:"
echo "#    FS_LM_DBG bit 1 = no ends of the lines, 2 = no line waves, 4 = no loads (0 = the kernel as shipped; 7 = barriers only)"
grep -a "==\|no dots\|three dots\|differ" gpurun_out/latmarch/abl.txt | awk '!seen[$0]++'
echo "# 2. the solve (466 iterations) with k_lat_march (FS_LATTICE_MARCH=1) and with the tile product k_lattice_spmv (=0)"
cat gpurun_out/latmarch/iter.txt
