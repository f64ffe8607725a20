cd /tmp && export TMPDIR=/tmp
R=/root/repo
CMD="python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline"
export ESCX_STREAMS=1 ESCX_BENCH_NO_PROFILE=1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -f csv -d $R/gpurun_out/pmc2/p$i -o p -- $CMD > $R/gpurun_out/pmc2_p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/tools/pmc_agg.py $R/gpurun_out/pmc2 --json $R/gpurun_out/pmc2_agg.json --top 12 > $R/gpurun_out/pmc2_agg.txt
find $R/gpurun_out/pmc2 -name "*.csv" ! -name "*counter_collection*" -delete
du -sh $R/gpurun_out/pmc2
