#!/bin/bash
# the marching-window product of CG2 box operators (k_lat_march): the lattice tests with every row compared (lattice_check), then
# configs[3] timed in both forms.  Every command under timeout, stdin closed.
exec < /dev/null
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/latmarch
O=gpurun_out/latmarch
export FS_KRYLOV_DEBUG=1
timeout 600 python -m pytest tests/test_gpu_p2.py -k "lattice" -x -q > $O/tests.txt 2>&1
echo "tests exit $?" >> $O/tests.txt
unset FS_KRYLOV_DEBUG
N=${1:-107}
FS_LATTICE_DEBUG=1 FS_LATTICE_CHECK=1 timeout 300 python tools/probes/p2_lattice_probe.py $N > $O/probe_march.txt 2>&1
FS_LATTICE_MARCH=0 FS_LATTICE_DEBUG=1 FS_LATTICE_CHECK=1 timeout 300 python tools/probes/p2_lattice_probe.py $N > $O/probe_tiles.txt 2>&1
tail -5 $O/tests.txt; grep -v "^\[fs_krylov\]" $O/probe_march.txt | tail -30; grep "lattice 1\|tile product" $O/probe_tiles.txt | tail -8
