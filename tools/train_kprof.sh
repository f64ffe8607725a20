#!/bin/bash
# Kernel statistics of the training step with ONE batch part on ONE stream (kernels alone on the GPU: durations are not stretched by the other part),
# next to the step time of that mode without the profiler: what the launches outside the tagged launch groups cost, and what the gaps between kernels add up to.
#   tools/train_kprof.sh [NAME] [ENV=VAL ...] -> gpurun_out/train_kprof_NAME.csv / .txt
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=${1:-one}; shift
O=$R/gpurun_out/train_kprof_tmp; rm -rf $O; mkdir -p $O
cd $R
env ESCX_TRAIN_PARTS=1 "$@" timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 > $R/gpurun_out/train_kprof_$NAME.txt
cd /tmp; export TMPDIR=/tmp
env ESCX_TRAIN_PARTS=1 "$@" timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O -o p -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/train_kprof_$NAME.csv 2>/dev/null
tail -1 $O/log.txt | cut -c1-330 >> $R/gpurun_out/train_kprof_$NAME.txt
rm -rf $O
cat $R/gpurun_out/train_kprof_$NAME.txt
