#!/bin/bash
# sample sclk / power while the bench loop runs: tools/clock_probe.sh NAME [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; shift
cd $R
env "$@" python bench.py --steps 2500 --warmup 5 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads > /tmp/cp_$NAME.log 2>&1 &
PID=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics Package Power\|Current Socket" | tr '\n' ' '; echo
  sleep 1
done
wait $PID
tail -1 /tmp/cp_$NAME.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$NAME', 'ms_per_step', d['ms_per_step'])"
