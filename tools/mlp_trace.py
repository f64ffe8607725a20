#!/usr/bin/env python
"""Cycle breakdown of the fused MLP main loop (C=192 layer) from s_memtime stamps inside the kernel."""
import os, sys, ctypes
os.environ["ESCX_MLP_VARIANT"] = "164"; os.environ["ESCX_STREAMS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from esc import _native
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
lib, hd = model._handle(dev)
B = int(os.environ.get("TRACE_B", "36"))
LAYER = os.environ.get("TRACE_LAYER", "192")
geo = {"192": (5, 4, 192, 48, 12), "384": (6, 2, 384, 96, 24), "45": (0, 64, 45, 12, 3)}[LAYER]     # layer id, H, C, hidden tiles, KK
n_waves = (B * geo[1] * 300 + 15) // 16 + 64
buf = torch.zeros(n_waves * 8, dtype=torch.int64, device=dev)
lib.escx_debug_mlp_trace(ctypes.c_void_p(buf.data_ptr()))
torch.manual_seed(0)
x = (torch.randn(B, geo[1] * 300, geo[2]) * 0.5).to(dev)
y = torch.empty(B, 4 * geo[1] * 300, 384, device=dev); Hn = ctypes.c_int()
for _ in range(3):
    _native.check(lib.escx_transformer_layer(hd, geo[0], ctypes.c_void_p(x.data_ptr()), B, geo[1], 300, ctypes.c_void_p(y.data_ptr()), ctypes.byref(Hn), None))
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 5] > 0]
names = ["barrier+vmcnt", "dma issue", "fc1", "gelu", "fc2", "loop total"]
HTn, KKn = geo[3], geo[4]
print(f"layer C={LAYER} B={B}: {len(t)} waves traced; HT={HTn} hidden tiles; ideal MFMA cycles per tile-phase: fc1 {KKn*4*32}, fc2 {KKn*4*32} (x2 when two waves share a SIMD)")
for i, n in enumerate(names):
    print(f"  {n:14s} mean {t[:, i].mean() / HTn:9.0f} cyc/tile   p10 {np.percentile(t[:, i], 10) / HTn:8.0f}  p90 {np.percentile(t[:, i], 90) / HTn:8.0f}")
span = t[:, 7].max() - t[:, 6].min()
print(f"  kernel span (first begin -> last end): {span} ticks; per-wave loop total mean {t[:,5].mean():.0f}; s_memtime runs at 100 MHz? ratio check: sum phases/total = {t[:, :5].sum(1).mean() / t[:, 5].mean():.3f}")
