#!/bin/bash
# Round 6, first GPU check: the precision contract (three modes), the range-stress checkpoints, bench line, rocprofv3 kernel stats of the same command.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "layer_accuracy or golden or range_stress or fallback_kernel or every_transformer_layer or patch_embed_and_deembed" > $O/pytest_a.txt 2>&1; tail -5 $O/pytest_a.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads"
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o p -- $CMD > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
grep "^{" $O/prof.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/prof
head -12 $O/kernel_stats.csv | cut -c1-200
