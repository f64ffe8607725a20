import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo/efficient-speech-codec_amd"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
for B in (1, 4):
    x = bench.synth_batch(B, 0).to(dev)
    model.reserve(B, 48000, dev)
    codes, shape = model.encode(x, 6); wave = model.decode(codes, shape); torch.cuda.synchronize()
    def eager():
        c, s = model.encode(x, 6); return model.decode(c, s)
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    # graph capture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): eager()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            c2, s2 = model.encode(x, 6)
            w2 = model.decode(c2, s2)
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
        ok = torch.equal(c2, codes) and torch.equal(w2, wave)
        print(f"B={B}: eager {te*1e3:.3f} ms  graph {tg*1e3:.3f} ms  identical={ok}  rtf_graph={B*3/tg:.0f}")
    except Exception as e:
        print(f"B={B}: eager {te*1e3:.3f} ms; graph capture failed: {type(e).__name__}: {str(e)[:300]}")
