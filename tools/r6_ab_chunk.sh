#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
AB_BENCH=1 timeout 1500 python tools/ab.py --rounds 2 --steps 6 --batch 288 "one_pass:ESCX_CHUNK_FRAMES=0" "c9:ESCX_CHUNK_FRAMES=5409" "c18:ESCX_CHUNK_FRAMES=10818" "c36:ESCX_CHUNK_FRAMES=21636" "c72:ESCX_CHUNK_FRAMES=43272" > $O/ab_chunk288.txt 2>&1; cat $O/ab_chunk288.txt | tail -12
AB_BENCH=1 timeout 900 python tools/ab.py --rounds 2 --steps 10 --batch 72 "one_pass:ESCX_CHUNK_FRAMES=0" "c18:ESCX_CHUNK_FRAMES=10818" > $O/ab_chunk72.txt 2>&1; cat $O/ab_chunk72.txt | tail -6
