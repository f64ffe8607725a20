#!/bin/bash
# datapath utilisation per kernel (SQ_INSTS_MFMA / SQ_INSTS_VALU / GRBM_GUI_ACTIVE), single stream, written to gpurun_out/sq_util.txt
#   SQ_BENCH_ARGS="--mode train ..." SQ_OUT=sq_util_train.txt tools/sq_util.sh     -> the same table for the training step's kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/squ; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
PC="python $R/bench.py ${SQ_BENCH_ARGS:---steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --skip-isolated --skip-single-clip --skip-other-workloads}"
OUT=${SQ_OUT:-sq_util.txt}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1)); ESCX_STREAMS=${SQ_STREAMS:-1} timeout 900 rocprofv3 --pmc $set -f csv -d $O/p$i -o p -- $PC > $O/log$i.txt 2>&1
done
python $R/tools/pmc_agg.py $O --json $O/agg.json --top 0 > /dev/null
python - <<PY
import json
rows=json.load(open("$O/agg.json"))
out=["# per kernel, single stream: share of the SIMD cycles (1024 SIMDs x GRBM_GUI_ACTIVE / 8) spent issuing matrix / other vector instructions.",
     "# mfma%  = SQ_INSTS_MFMA x 32 cycles (the fp32 16x16x4 instruction; kernels on the 16-bit matrix cores issue 16x16x32 at ~17 cycles: see busy%)",
     "# busy%  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles): the matrix pipe's own busy counter, whatever the instruction mix",
     "# valu%  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles;  issue% = busy% + valu% = the vector-issue utilisation of DESIGN.md section 4",
     f"{'kernel':52s} {'us':>7s} {'mfma%':>6s} {'busy%':>6s} {'valu%':>6s} {'issue%':>6s} {'valu/mfma':>9s} {'waves':>7s}"]
for r in rows[:${SQ_TOP:-30}]:
    c=r['counters']
    if not c.get('SQ_INSTS_MFMA'): continue
    cyc=c['GRBM_GUI_ACTIVE']/8.0
    mf=c['SQ_INSTS_MFMA']*32/(1024*cyc); va=(c['SQ_INSTS_VALU']-c['SQ_INSTS_MFMA'])*4/(1024*cyc)
    bz=c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc)
    out.append(f"{r['kernel'][:52]:52s} {r['avg_us']:7.1f} {100*mf:6.1f} {100*bz:6.1f} {100*va:6.1f} {100*(bz+va):6.1f} {(c['SQ_INSTS_VALU']-c['SQ_INSTS_MFMA'])/c['SQ_INSTS_MFMA']:9.2f} {c.get('SQ_WAVES',0):7.0f}")
open("$R/gpurun_out/$OUT","w").write("\n".join(out)+"\n"); print("\n".join(out))
PY
rm -rf $O
