#!/bin/bash
# Codec-only subset of tools/profile_round.sh (after a change that touches only the inference kernels): GPU tests, bench line, rocprofv3 kernel stats of the
# same command, HBM and SQ counter passes, event breakdowns, datapath utilisation table.  Everything lands in gpurun_out/round/ (collect with tools/collect_round.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>&1 | grep "^#" > $O/event_breakdown_isolated.txt
ESCX_STREAMS=1 ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>&1 | grep "^#" > $O/event_breakdown_1stream.txt
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads"
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o p -- $CMD > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
tail -1 $O/prof.log | cut -c1-300 > $O/prof_bench_line.txt
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
timeout 600 python tools/small_batch.py > $O/small_batch.txt 2>&1
timeout 1200 bash tools/sq_util.sh > /dev/null 2>&1; cp gpurun_out/sq_util.txt $O/sq_util.txt 2>/dev/null
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/cal_fetch $O/cal_write
ls -la $O
