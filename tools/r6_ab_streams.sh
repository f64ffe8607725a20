#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
AB_BENCH=1 timeout 1500 python tools/ab.py --rounds 3 --steps 20 "s2:" "s3:ESCX_STREAMS=3" "s4:ESCX_STREAMS=4" "s3_q12:ESCX_STREAMS=3,GPU_MAX_HW_QUEUES=12" > $O/ab_streams.txt 2>&1; cat $O/ab_streams.txt | tail -12
