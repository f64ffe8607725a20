#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6h; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fallback_kernel or fused_and_unfused or golden or batch_sizes_alternating or node_batch_288 or hip_graph" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-other-workloads 2>/dev/null | tail -1 | cut -c1-260
