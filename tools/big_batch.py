import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficient-speech-codec_amd"))
import torch, bench
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
for B in (48, 64, 96, 144, 288):
    x = bench.synth_batch(B, 0).to(dev)
    def step():
        c, s = model.encode(x, 6); return model.decode(c, s)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:3d} streams={os.environ.get('ESCX_STREAMS', '2')} {ms:8.3f} ms  {B * 3.0 / ms * 1e3:8.1f} audio-s/s", flush=True)
