#!/bin/bash
# Single-clip kernel timeline (profiles/r6_b1_timeline.txt): rocprofv3 kernel trace of tools/b1_trace.py + hipGraph replay against eager
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b1; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/t -o p -- python $R/tools/b1_trace.py > $O/log.txt 2>&1
python $R/tools/b1_trace_stats.py $(find $O/t -name "*kernel_trace.csv" | head -1) | tee $O/b1_trace_stats.txt
rm -rf $O/t
cd $R; timeout 300 python tools/graph_latency.py 2>&1 | tail -3 | tee $O/graph_latency.txt
