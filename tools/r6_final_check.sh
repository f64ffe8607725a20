#!/bin/bash
# What the driver runs at round end, on the final tree: the GPU suite, smoke(), the default bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
s=$SECONDS
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt; echo "pytest wall $((SECONDS - s)) s" | tee -a $O/pytest_gpu.txt
s=$SECONDS
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt; echo "smoke wall $((SECONDS - s)) s"
s=$SECONDS
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-240 $O/bench.json; echo; echo "bench wall $((SECONDS - s)) s"
