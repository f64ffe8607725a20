#!/usr/bin/env python
"""Cycle breakdown of the fused attention head-group loop from s_memtime stamps (needs a -DESCX_ATTN_TRACE build)."""
import os, sys, ctypes
os.environ["ESCX_STREAMS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from esc import _native
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
lib, hd = model._handle(dev)
B = int(os.environ.get("TRACE_B", "18"))
# layer id, H, C, head groups per window, KK, MFMAs per group (TMW windows)
GEO = {"45": (0, 64, 45, 3, 3, 112), "72": (1, 32, 72, 6, 5, 0), "96": (2, 16, 96, 6, 6, 224), "144": (3, 8, 144, 12, 9, 152), "192": (4, 4, 192, 12, 12, 208)}
for LAYER in os.environ.get("TRACE_LAYERS", "45,96,144,192").split(","):
    lid, H, C, groups, KK, mf = GEO[LAYER]
    n_waves = B * ((H + 3) // 4) * 75 + 256
    buf = torch.zeros(n_waves * 8, dtype=torch.int64, device=dev)
    lib.escx_debug_mlp_trace(ctypes.c_void_p(buf.data_ptr()))
    torch.manual_seed(0)
    x = (torch.randn(B, H * 300, C) * 0.5).to(dev)
    y = torch.empty(B, 2 * H * 300, 384, device=dev); Hn = ctypes.c_int()
    for _ in range(2):
        _native.check(lib.escx_transformer_layer(hd, lid, ctypes.c_void_p(x.data_ptr()), B, H, 300, ctypes.c_void_p(y.data_ptr()), ctypes.byref(Hn), None))
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 7] > 0]
    names = ["barrier+consts", "Q,K GEMM", "scores+softmax", "V GEMM", "P.V", "proj"]
    tot = (t[:, 7] - t[:, 6]).mean()
    print(f"layer C={LAYER} B={B}: {len(t)} waves traced, {groups} head groups, loop total mean {tot:.0f} cycles = {tot / groups:.0f} per group; MFMA issue per group {mf * 32}")
    for i, n in enumerate(names):
        print(f"  {n:16s} mean {t[:, i].mean() / groups:8.0f} cyc/group   p10 {np.percentile(t[:, i], 10) / groups:7.0f}  p90 {np.percentile(t[:, i], 90) / groups:7.0f}")
lib.escx_debug_mlp_trace(None)
