#!/bin/bash
# training-step check (round 6, third session): the training suite, then the training bench with the split-operand MLP kernels of the C = 45 / 72 layers switched
# (default = both on; bwd_off = ESCX_TRAIN_MLPBWD_X2=0; all_off = also ESCX_TRAIN_MLP_X3=0 = the round-5 kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6train; mkdir -p $O
cd $R
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
for thr in default 16 default 16; do
  if [ $thr = default ]; then unset OMP_NUM_THREADS; else export OMP_NUM_THREADS=$thr; fi
  /usr/bin/time -f "threads $thr: %e s wall, %U s user" timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -k "losses_and_every_gradient and tiny" 2>&1 | tail -2
done
unset OMP_NUM_THREADS
timeout 1800 python -m pytest tests/test_train.py -x -q -m gpu --durations=5 > $O/pytest_train.txt 2>&1; tail -9 $O/pytest_train.txt
for i in 1 2 3; do
for arm in default bwd_off all_off; do
  unset ESCX_TRAIN_MLP_X3 ESCX_TRAIN_MLPBWD_X2
  if [ $arm = bwd_off ]; then export ESCX_TRAIN_MLPBWD_X2=0; fi
  if [ $arm = all_off ]; then export ESCX_TRAIN_MLPBWD_X2=0 ESCX_TRAIN_MLP_X3=0; fi
  timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2>$O/train_$arm.err | tail -1 > $O/train_$arm.json
  python -c "
import json; d=json.loads(open('$O/train_$arm.json').read()); print('$arm', d['ms_per_step'], d['value'])" | tee -a $O/train_ab.txt
done
done
unset ESCX_TRAIN_MLP_X3 ESCX_TRAIN_MLPBWD_X2
ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "^# [BT].mlp" | awk '!seen[$0]++' | tee $O/train_mlp_breakdown.txt
