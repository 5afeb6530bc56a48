#!/bin/bash
# PMC passes for the KNN filter kernel (separate passes; counters only with --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r1
mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
P2="FETCH_SIZE"
P3="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o pass$i -- python $R/scripts/run_knn_steps.py 6 > $OUT/pass$i.log 2>&1
  i=$((i+1))
done
ls $OUT
