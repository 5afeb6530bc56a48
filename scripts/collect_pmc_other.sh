#!/bin/bash
# PMC evidence for the non-KNN legs (VERDICT r02 "missing" 4): separate rocprofv3 passes per counter group around the
# triangulation, BA and SIFT workloads of bench.py.  Usage (via gpurun): bash scripts/collect_pmc_other.sh rNN
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
G1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
G2="FETCH_SIZE"
G3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
G4="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
for wl in tri ba sift; do
  extra=""; [ $wl = sift ] && extra="--pipe-depth 1"
  bash $R/scripts/pmc_cmd.sh pmc_${TAG}_$wl "$G1" "$G2" "$G3" "$G4" -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline $extra > $R/gpurun_out/${TAG}_${wl}_pmc.txt 2>&1
done
ls $R/gpurun_out/${TAG}_*_pmc.txt
