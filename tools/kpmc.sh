#!/bin/bash
# SQ stall counters per kernel (single stream, whole batch): tools/kpmc.sh NAME [ENV=VAL ...]; KPROF_FILTER = kernel-name regex
R=${GRAFT_REPO_ROOT:-/root/repo}; NAME=$1; shift
O=$R/gpurun_out/kpmc_tmp_$NAME; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PC="python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads"
env ESCX_STREAMS=1 "$@" timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -f csv -d $O/p1 -o p -- $PC > $O/log1.txt 2>&1
env ESCX_STREAMS=1 "$@" timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $O/p2 -o p -- $PC > $O/log2.txt 2>&1
python $R/tools/pmc_agg.py $O --json $O/agg.json --top 0 > /dev/null
python - <<PY
import json,re
rows=json.load(open("$O/agg.json"))
flt=re.compile("${KPROF_FILTER:-.}")
for r in rows:
    if not flt.search(r['kernel']): continue
    c=r['counters']
    if not c.get('SQ_WAVES'): continue
    w=c['SQ_WAVES']; wc=c['SQ_WAVE_CYCLES']*4; cyc=c['GRBM_GUI_ACTIVE']/8.0
    mf=c.get('SQ_INSTS_MFMA',0)
    print(f"{r['kernel'][:48]:48s} {r['avg_us']:7.1f} us waves {w:6.0f} life {wc/w/1e3:7.1f}k cyc  resident/SIMD {wc/(1024*cyc):5.2f}  mfma-busy {mf*32/(1024*cyc)*100:5.1f}%  per-wave: mfma {mf/w:6.0f} valu {(c.get('SQ_INSTS_VALU',0)-mf)/w:6.0f} wait_any {c['SQ_WAIT_ANY']*4/w/1e3:6.1f}k wait_inst {c['SQ_WAIT_INST_ANY']*4/w/1e3:6.1f}k active {c['SQ_ACTIVE_INST_ANY']*4/w/1e3:6.1f}k")
PY
rm -rf $O
