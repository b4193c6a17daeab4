#!/bin/bash
# ON THE GPU BOX: SQ counters of the tile product k_lattice_spmv and of the work-item product k_dict_spmv in the same run of
# tools/probes/p2_lattice_probe.py 107 (one --pmc pass per group; median per launch; <3> = inside the solve) -> profiles/r05_p2_lattice_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
echo '# bash tools/probes/p2_lattice_pmc.sh = tools/probes/pmc_any.sh "spmv" "python tools/probes/p2_lattice_probe.py 107" <three counter groups>  (median per launch; <3> = inside the solve: lattice 0 runs k_dict_spmv, lattice 1 k_lattice_spmv; FINAL round-5 kernel)'
bash $R/tools/probes/pmc_any.sh "spmv" "python $R/tools/probes/p2_lattice_probe.py 107" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
} > $R/gpurun_out/r05_p2_lattice_pmc.txt 2>&1
cat $R/gpurun_out/r05_p2_lattice_pmc.txt
