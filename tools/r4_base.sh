R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4base; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json
ESCX_STREAMS=1 ESCX_BENCH_BREAKDOWN=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-other-workloads 2>&1 | grep "^#" > $O/event_breakdown_1stream.txt
cat $O/event_breakdown_1stream.txt
rocminfo | grep -i "compute unit\|max clock" | head -4
