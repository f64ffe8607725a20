#!/bin/bash
# The wide oracle sweeps of the build under test, in ALL THREE precision modes of escx_set_precision (the parametrised always-on tests, widened):
# ESC-Base 288 + 288 clips, ESC-Large 144 + 144 clips; the oracle runs once per clip set (tests/test_gpu_parity.py _SWEEP_ORACLE), the modes share it.
#   tools/r6_sweeps.sh   -> gpurun_out/r6_sweeps/{base576,large288}.log
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_sweeps; mkdir -p $O; cd $R
env ESCX_PARITY_SWEEP=288 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k parity_sweep_base 2>&1 | grep -aE "^\\[sweep|passed|failed|Error|assert" > $O/base576.log; grep -a "^\[sweep\]\|passed\|failed" $O/base576.log
env ESCX_PARITY_SWEEP_LARGE=144 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k parity_sweep_large 2>&1 | grep -aE "^\\[sweep|passed|failed|Error|assert" > $O/large288.log; grep -a "^\[sweep\]\|passed\|failed" $O/large288.log
