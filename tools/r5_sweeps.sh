#!/bin/bash
# The wide oracle sweeps (Base-576, Large-288) of the build under test; the always-on GPU tests run the same code on 144 / 36 clips.
#   tools/r5_sweeps.sh [tag]   -> gpurun_out/r5_sweeps/{base576,large288}[_tag].log   (extra environment is passed through: ESCX_PVQ_FUSED=0 ...)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_sweeps; mkdir -p $O; cd $R
T=${1:+_$1}
env ESCX_PARITY_SWEEP=288 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k parity_sweep_base 2>&1 | grep -E "^\[sweep|passed|failed|Error|assert" > $O/base576$T.log; tail -2 $O/base576$T.log
env ESCX_PARITY_SWEEP_LARGE=144 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k parity_sweep_large 2>&1 | grep -E "^\[sweep|passed|failed|Error|assert" > $O/large288$T.log; tail -2 $O/large288$T.log
