#!/usr/bin/env python
"""Phase breakdown of the fused product-VQ kernel (fused_pvq.h) from s_memtime stamps (tuning build):
    ESCX_BUILD_TAG=trace ESCX_EXTRA_CXXFLAGS=-DESCX_PVQ_TRACE python efficient-speech-codec_amd/build.py
    ESCX_LIB_TAG=trace python tools/pvq_trace.py [clips]
One encode of `clips` clips on ONE stream (kernels alone on the GPU); per stream: mean duration of every phase over the workgroups."""
import os, sys, ctypes
os.environ.setdefault("ESCX_LIB_TAG", "trace"); os.environ["ESCX_STREAMS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 18
model, cfg, sd = bench.build_model(dev)
lib, hd = model._handle(dev)
x = bench.synth_batch(B, 0, None).to(dev)
for _ in range(2):
    model.encode(x, 6)
torch.cuda.synchronize()
bufs = [torch.zeros(16 * 4096 * 8, dtype=torch.int64, device=dev) for _ in range(2)]
lib.escx_debug_mlp_trace(ctypes.c_void_p(bufs[0].data_ptr()))
model.encode(x, 6)
torch.cuda.synchronize()
lib.escx_debug_mlp_trace(None)
t = bufs[0].cpu().numpy().reshape(16, 4096, 8)
names = ["P1 down-proj slices", "P2 reduce+normalise", "P3 search", "P3b final argmin", "P4 up-proj+add"]
for s in range(6):
    r = t[s]; r = r[r[:, 0] > 0]
    if not len(r): continue
    has4 = r[:, 5] > 0
    print(f"stream {s}: {len(r)} workgroups; s_memtime ticks (100 MHz = 10 ns)")
    for i, n in enumerate(names):
        if i == 4 and not has4.any(): continue
        d = (r[:, i + 1] - r[:, i])[has4 if i == 4 else slice(None)]
        print(f"   {n:22s} mean {d.mean() * 10 / 1e3:8.2f} us   p10 {np.percentile(d, 10) * 10 / 1e3:7.2f}  p90 {np.percentile(d, 90) * 10 / 1e3:7.2f}")
    end = np.where(has4, r[:, 5], r[:, 4])
    print(f"   workgroup total        mean {(end - r[:, 0]).mean() * 10 / 1e3:8.2f} us;  launch span {(end.max() - r[:, 0].min()) * 10 / 1e3:8.2f} us")
