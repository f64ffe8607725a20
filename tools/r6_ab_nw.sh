#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
export ESCX_LIB_TAG=exp
AB_BENCH=1 timeout 1500 python tools/ab.py --rounds 2 --steps 20 --groups attn_fused,mlp_x3 "base:" "attn_nw8:ESCX_ATTN_NW=8" "attn_nw4:ESCX_ATTN_NW=4" "mlp_nw8:ESCX_MLP_X3_NW=8" "mlp_nw4:ESCX_MLP_X3_NW=4" > $O/ab_nw.txt 2>&1; cat $O/ab_nw.txt | tail -14
