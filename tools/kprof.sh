#!/bin/bash
# rocprofv3 kernel-trace stats of the bench step with every kernel alone on the GPU (ESCX_STREAMS=1: whole batch per launch, serial).
#   tools/kprof.sh NAME [ENV=VAL ...]   -> gpurun_out/kprof_NAME.csv (top rows printed, filtered by $KPROF_FILTER)
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; shift
O=$R/gpurun_out/kprof_tmp_$NAME; rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
env ESCX_STREAMS=${KPROF_STREAMS:-1} "$@" timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O -o p -- python $R/bench.py --steps ${KPROF_STEPS:-6} --warmup 2 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads > $O/log.txt 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kprof_$NAME.csv 2>/dev/null
tail -1 $O/log.txt | cut -c1-200
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$R/gpurun_out/kprof_$NAME.csv")))
flt=re.compile("${KPROF_FILTER:-.}")
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows:
    n=re.sub(r"^void ","",r["Name"]).replace("escx::","")
    n=re.sub(r"\(.*$","",n)
    if flt.search(n): print(f"{n[:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf $O
