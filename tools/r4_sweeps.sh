#!/bin/bash
# Base-576 and Large-288 oracle sweeps for BOTH settings of the C = 384 head-group split (ESCX_ATTN_GS_TOKENS)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_sweeps; mkdir -p $O; cd $R
run() { # name model n gs-tokens (0 = split off, 600 = on: the default since round 4)
  env ESCX_PARITY_SWEEP=$3 ESCX_PARITY_SWEEP_MODEL=$2 ESCX_ATTN_GS_TOKENS=$4 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k parity_sweep 2>&1 | grep -E "^\[sweep|passed|failed|Error|assert" > $O/$1.log
  tail -2 $O/$1.log
}
run base576_gs_off base 288 0
run base576_gs_on base 288 600
run large288_gs_off large 144 0
run large288_gs_on large 144 600
