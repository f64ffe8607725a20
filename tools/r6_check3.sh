#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
# negative control of the range rule: the two-term images without activation scales (tagged build) on the same stress checkpoints
ESCX_LIB_TAG=exp ESCX_X2_NO_ACT_SCALE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "range_stress and f16x2" > $O/range_rule_off.txt 2>&1; grep "^\[range\|passed\|failed\|^FAILED\|non-finite\|AssertionError: clip" $O/range_rule_off.txt | head -20
