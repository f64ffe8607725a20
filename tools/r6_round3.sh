#!/bin/bash
# Round 6, last evidence call (third session): on the final tree, with the PMC traffic of this build's codec kernels in place (profiles/pmc_dominant.json) -
# the GPU suite, the driver's bench command, the training / adversarial evidence after the split-operand MLP kernels of the training step, the wide three-mode oracle sweeps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round3; rm -rf $O; mkdir -p $O
cd $R
s=$SECONDS
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 -rP > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt; echo "pytest wall $((SECONDS - s)) s"
grep -aE "^\[(sweep|fp64|range|unfiltered|clustered|batch36|bench36|voiced|large36|configs|train-x2)|^[0-9]+ passed|^FAILED|^ERROR|slowest|^[0-9.]+s call" $O/pytest_gpu_full.txt > $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; echo
ESCX_BENCH_BREAKDOWN=1 timeout 900 python bench.py --mode train --steps 10 --warmup 3 2>$O/train.err | tail -1 > $O/bench_train.json; cut -c1-200 $O/bench_train.json; echo
grep "^#" $O/train.err | awk '!seen[$0]++' > $O/train_breakdown.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_train -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_train.log 2>&1
cp $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_train
cd $R
timeout 900 python bench.py --mode train_adv --steps 4 --warmup 2 2>$O/train_adv.err | tail -1 > $O/bench_train_adv.json; cut -c1-200 $O/bench_train_adv.json; echo
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_adv -o p -- python $R/bench.py --mode train_adv --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_adv.log 2>&1
cp $(find $O/prof_adv -name "*kernel_stats.csv" | head -1) $O/train_adv_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_adv
cd $R
timeout 900 python bench.py --mode train_adv --adv-precision fp32 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_adv_fp32mfma.json
timeout 900 python bench.py --mode train_adv --adv-precision bf16 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_adv_bf16.json
SQ_BENCH_ARGS="--mode train --steps 2 --warmup 1 --profile-steps 2 --no-cpu-baseline" SQ_OUT=sq_util_train.txt SQ_TOP=45 ESCX_TRAIN_PARTS=1 timeout 1500 bash tools/sq_util.sh > /dev/null 2>&1; cp gpurun_out/sq_util_train.txt $O/sq_util_train.txt 2>/dev/null
s=$SECONDS
timeout 4200 bash tools/r6_sweeps.sh > $O/sweeps_tail.txt 2>&1; cp gpurun_out/r6_sweeps/base576.log $O/parity_sweep_base576.log; cp gpurun_out/r6_sweeps/large288.log $O/parity_sweep_large288.log; cat $O/sweeps_tail.txt; echo "sweeps wall $((SECONDS - s)) s"
ls $O
