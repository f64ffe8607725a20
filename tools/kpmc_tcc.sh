#!/bin/bash
# L2 hit / miss counters per kernel (single stream, whole batch): tools/kpmc_tcc.sh NAME [ENV=VAL ...]; KPROF_FILTER = kernel-name regex
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; shift
O=$R/gpurun_out/kpmc_tcc_$NAME; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PC="python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads"
env ESCX_STREAMS=1 "$@" timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $O/p1 -o p -- $PC > $O/log1.txt 2>&1
env ESCX_STREAMS=1 "$@" timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/p2 -o p -- $PC > $O/log2.txt 2>&1
python $R/tools/pmc_agg.py $O --json $O/agg.json --top 0 > /dev/null
python - <<PY
import json,re
rows=json.load(open("$O/agg.json"))
flt=re.compile("${KPROF_FILTER:-.}")
for r in rows:
    if not flt.search(r['kernel']): continue
    c=r['counters']; hit=c.get('TCC_HIT_sum',0); miss=c.get('TCC_MISS_sum',0)
    print(f"{r['kernel'][:64]:64s} {r['avg_us']:7.1f} us  L2 hit {hit:12.0f} miss {miss:11.0f} hit-rate {100*hit/max(hit+miss,1):5.1f}%  FETCH_SIZE {c.get('FETCH_SIZE',0):10.0f} KiB (x2 for 16 B/lane reads)")
PY
