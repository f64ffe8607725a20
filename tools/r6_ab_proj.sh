#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or every_transformer_layer or layer_accuracy or (range_stress and f16x2) or unfiltered" > $O/pytest_proj.txt 2>&1; tail -4 $O/pytest_proj.txt
AB_BENCH=1 timeout 900 python tools/ab.py --rounds 3 --steps 20 --groups attn_fused "pipe0:ESCX_MLP_X3_PIPE=0" > $O/ab_proj.txt 2>&1; cat $O/ab_proj.txt | tail -30
