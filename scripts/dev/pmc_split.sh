cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_split; mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
i=1
for P in "$P1" "$P4"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o pass$i -- python $R/scripts/run_knn_steps.py 6 > $OUT/pass$i.log 2>&1
  i=$((i+1))
done
python $R/scripts/summarize_pmc.py $OUT split | sed -n '/knn_filter_split/,/^$/p' | head -40
python $R/scripts/trace_filter.py
