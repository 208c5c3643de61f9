# rocprofv3 --pmc passes (one group of counters per run) over the split-operand acting kernel at 65 536 rows; summary into gpurun_out/pmc_x3_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_x3; rm -rf $OUT; mkdir -p $OUT
export TVC_ACT_X3=1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/pmc_run.py act 65536 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 tools/pmc_summarize.py $OUT actor_x3 > gpurun_out/pmc_x3_summary.txt 2>&1
cat gpurun_out/pmc_x3_summary.txt
