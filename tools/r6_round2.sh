#!/bin/bash
# Round 6, second evidence call: the driver's bench command with the PMC traffic of this build's kernels in place, the SQ utilisation table with the matrix-busy column,
# the wide three-mode oracle sweeps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round2; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
timeout 1200 bash tools/sq_util.sh > /dev/null 2>&1; cp gpurun_out/sq_util.txt $O/sq_util.txt 2>/dev/null; head -12 $O/sq_util.txt
timeout 4200 bash tools/r6_sweeps.sh > $O/sweeps_tail.txt 2>&1; cp gpurun_out/r6_sweeps/base576.log $O/parity_sweep_base576.log; cp gpurun_out/r6_sweeps/large288.log $O/parity_sweep_large288.log; cat $O/sweeps_tail.txt
