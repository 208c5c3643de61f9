# rocprofv3 --pmc passes (one group of counters per run) over the one-launch acting kernel at 16 384 rows (one workgroup per CU,
# one wave per SIMD) and 65 536 rows (two per CU); summaries into gpurun_out/pmc_act_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_act; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 tools/pmc_run.py act 16384 65536 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 tools/pmc_summarize.py $OUT actor_rows > gpurun_out/pmc_act_summary.txt 2>&1
cat gpurun_out/pmc_act_summary.txt
