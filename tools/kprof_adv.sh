#!/bin/bash
# rocprofv3 kernel-trace stats of the adversarial step (bench.py --mode train_adv) in one discriminator precision.
#   tools/kprof_adv.sh NAME PRECISION [ENV=VAL ...]   -> gpurun_out/kprof_adv_NAME.csv (top 25 rows printed)
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; PREC=$2; shift; shift
O=$R/gpurun_out/kprof_tmp_$NAME; rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
env ESCX_BENCH_ADV_PRECISION=$PREC "$@" timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O -o p -- python $R/bench.py --mode train_adv --steps 3 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
cp $(find $O -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kprof_adv_$NAME.csv 2>/dev/null
tail -1 $O/log.txt | cut -c1-260
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$R/gpurun_out/kprof_adv_$NAME.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 4 steps")
for r in rows[:25]:
    n=re.sub(r"^void ","",r["Name"]).replace("escx::","")
    n=re.sub(r"\(.*$","",n)
    print(f"{n[:110]:110s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf $O
