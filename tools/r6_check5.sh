#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6l; rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_train.py tests/test_gpu_parity.py -x -q -m gpu --durations=6 -k "precomputed_spectrum or two_part_training or full_size_invariants or parity_sweep_base or forward_with_precomputed or training_step_is_run" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
