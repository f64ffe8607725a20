#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
AB_BENCH=1 timeout 900 python tools/ab.py --rounds 3 --steps 20 --groups mlp_x3 "r5loop:ESCX_MLP_X3_PIPE=0" "pipe:" > $O/ab_pipe.txt 2>&1; cat $O/ab_pipe.txt | tail -30
