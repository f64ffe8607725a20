#!/bin/bash
# Round-end evidence run (one gpurun call): GPU tests, bench line, rocprofv3 kernel stats of the same command, HBM and SQ counter
# passes, event breakdowns, the other BASELINE configurations, micro-benchmarks.  Everything lands in gpurun_out/round/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -x -q -m gpu --durations=12 -rP > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt
# the reports the parity tests print (per precision mode), and the summary lines
grep -aE "^\[(sweep|fp64|range|unfiltered|clustered|batch36|bench36|voiced|large36|configs)|^[0-9]+ passed|^FAILED|^ERROR|slowest|^[0-9.]+s call" $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>&1 | grep "^#" > $O/event_breakdown_isolated.txt
ESCX_STREAMS=1 ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>&1 | grep "^#" > $O/event_breakdown_1stream.txt
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads"
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o p -- $CMD > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
grep -a "^{" $O/prof.log | tail -1 > $O/prof_bench_line.txt      # the WHOLE JSON line the traced process printed (round 5 kept 300 characters of it)
PC="python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --skip-single-clip --skip-other-workloads"
timeout 900 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/pmc_fetch -o f -- $PC > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/pmc_write -o w -- $PC > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/cal_fetch -o f -- python $R/tools/pmc_calib.py run > $O/cal_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/cal_write -o w -- python $R/tools/pmc_calib.py run > $O/cal_write.log 2>&1
python $R/tools/pmc_calib.py reduce $O/cal_fetch $O/cal_write $O/pmc_calibration.json > /dev/null
python $R/tools/pmc_hbm.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.json $O/pmc_dominant.json $O/pmc_calibration.json > /dev/null
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); ESCX_STREAMS=1 timeout 900 rocprofv3 --pmc $set -f csv -d $O/pmc_sq/p$i -o p -- $PC > $O/pmc_sq_$i.log 2>&1
done
python $R/tools/pmc_agg.py $O/pmc_sq --json $O/sq_counters.json --top 16 > $O/sq_counters.txt
cd $R
ESCX_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rccl_1rank.json
timeout 900 python tools/bench_configs.py > $O/other_configs.json 2>$O/other_configs.err
# strong-scaling mode at N = 1 (BASELINE configs[3] on one GPU) and the training step (BASELINE configs[4], first slice)
timeout 900 python bench.py --global-batch 288 --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>/dev/null | tail -1 > $O/bench_strong_n1.json
ESCX_BENCH_BREAKDOWN=1 timeout 900 python bench.py --mode train --steps 10 --warmup 3 2>$O/train.err | tail -1 > $O/bench_train.json; cut -c1-300 $O/bench_train.json
grep "^#" $O/train.err | awk '!seen[$0]++' > $O/train_breakdown.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_train.log 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null
tail -1 $O/prof_train.log | cut -c1-300 > $O/prof_train_bench_line.txt
rm -rf $O/prof_train
# adversarial step (ESC-Large generator + discriminator): bench line and kernel stats
cd $R
timeout 900 python bench.py --mode train_adv --steps 4 --warmup 2 2>$O/train_adv.err | tail -1 > $O/bench_train_adv.json; cut -c1-300 $O/bench_train_adv.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_adv -o p -- python $R/bench.py --mode train_adv --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_adv.log 2>&1
cp $(find $O/prof_adv -name "*kernel_stats.csv" | head -1) $O/train_adv_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_adv
cd $R
# the all-fp32-MFMA adversarial step (rounds 2-4; the default is the split-operand precision since round 5)
timeout 900 python bench.py --mode train_adv --adv-precision fp32 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_adv_fp32mfma.json; cut -c1-300 $O/bench_train_adv_fp32mfma.json
# opt-in bf16 precision of the discriminator's wide convolutions: bench line + kernel stats of the same step
timeout 900 python bench.py --mode train_adv --adv-precision bf16 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_adv_bf16.json; cut -c1-300 $O/bench_train_adv_bf16.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_adv16 -o p -- python $R/bench.py --mode train_adv --adv-precision bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_adv16.log 2>&1
cp $(find $O/prof_adv16 -name "*kernel_stats.csv" | head -1) $O/train_adv_bf16_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_adv16
cd $R
# round 4: small-batch latencies, datapath utilisation tables of the codec AND of the training step
timeout 600 python tools/small_batch.py > $O/small_batch.txt 2>&1
timeout 1200 bash tools/sq_util.sh > /dev/null 2>&1; cp gpurun_out/sq_util.txt $O/sq_util.txt 2>/dev/null
SQ_BENCH_ARGS="--mode train --steps 2 --warmup 1 --profile-steps 2 --no-cpu-baseline" SQ_OUT=sq_util_train.txt SQ_TOP=45 ESCX_TRAIN_PARTS=1 timeout 1500 bash tools/sq_util.sh > /dev/null 2>&1; cp gpurun_out/sq_util_train.txt $O/sq_util_train.txt 2>/dev/null
# round 5: GEMM-engine micro-benchmark with the 32x32x2 MFMA arm, the fused-PVQ phase trace when the tuning build is present, the wide parity sweeps
# round 6: the wide oracle sweeps run in their own call (tools/r6_sweeps.sh: Base-576 and Large-288 in all three precision modes)
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/cal_fetch $O/cal_write
ls -la $O
