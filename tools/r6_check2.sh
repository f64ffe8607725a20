#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "range_stress" > $O/pytest_range.txt 2>&1; grep "^\[range\|passed\|failed\|Error" $O/pytest_range.txt | head -40
