#!/bin/bash
# Kernel timeline of a few bench steps (two streams): per-queue gaps between consecutive kernels and per-family durations.
#   tools/ktrace.sh NAME [ENV=VAL ...] -> gpurun_out/ktrace_NAME.csv (raw trace, steps after warm-up only is not separable: keep it short)
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; shift
O=$R/gpurun_out/ktrace_tmp_$NAME; rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace -f csv -d $O -o p -- python $R/bench.py --steps 6 --warmup 2 --profile-steps 1 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads > $O/log.txt 2>&1
cp $(find $O -name "*kernel_trace.csv" | head -1) $R/gpurun_out/ktrace_$NAME.csv 2>/dev/null
tail -1 $O/log.txt | cut -c1-160
head -2 $R/gpurun_out/ktrace_$NAME.csv
rm -rf $O
