"""Kernel timeline of single-clip encode+decode (run under rocprofv3 --kernel-trace): python tools/b1_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import torch
import bench
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
x = bench.synth_batch(1, 0).to(dev)
for _ in range(30):
    c, s = model.encode(x, 6); model.decode(c, s)
torch.cuda.synchronize()
