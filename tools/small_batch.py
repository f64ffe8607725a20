"""Encode+decode latency of small batches (B = 1, 2, 4, 8, 12, 16, 24) of 3 s clips: prints ms and audio-s/s for the current ESCX_STREAMS."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import torch
import bench

dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
for B in (1, 2, 4, 8, 12, 16, 24, 36):
    x = bench.synth_batch(B, 0).to(dev)
    def step():
        c, s = model.encode(x, 6); return model.decode(c, s)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
    print(f"B={B:2d} streams={os.environ.get('ESCX_STREAMS', '2')} {ms:7.3f} ms  {B * 3.0 / ms * 1e3:8.1f} audio-s/s", flush=True)
