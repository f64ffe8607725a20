#!/bin/bash
# training-step check: the training / adversarial suites, then the training bench with and without the split-operand forward MLP
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6train; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_train.py -x -q -m gpu --durations=5 > $O/pytest_train.txt 2>&1; tail -9 $O/pytest_train.txt
for arm in on off; do
  if [ $arm = off ]; then export ESCX_TRAIN_MLP_X3=0; else unset ESCX_TRAIN_MLP_X3; fi
  for i in 1 2; do timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2>$O/train_$arm.err | tail -1 > $O/train_$arm.json; python -c "
import json,sys; d=json.loads(open('$O/train_$arm.json').read()); print('$arm', d['ms_per_step'], d['value'])"; done
done
