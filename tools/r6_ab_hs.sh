#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
export ESCX_LIB_TAG=exp
AB_BENCH=1 timeout 1500 python tools/ab.py --rounds 3 --steps 20 "base:" "hs600:ESCX_MLP_HS_TOKENS=600" "hs0:ESCX_MLP_HS_TOKENS=0" "gs0:ESCX_ATTN_GS_TOKENS=0" "hs600_gs0:ESCX_MLP_HS_TOKENS=600,ESCX_ATTN_GS_TOKENS=0" > $O/ab_hs.txt 2>&1; cat $O/ab_hs.txt | tail -8
