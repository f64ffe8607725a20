"""Per-kernel event breakdown of encode+decode at a small batch: python tools/small_batch_breakdown.py [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model, cfg, sd = bench.build_model(dev)
x = bench.synth_batch(B, 0).to(dev)
for _ in range(5):
    c, s = model.encode(x, 6); model.decode(c, s)
lib, hd = model._handle(dev)
lib.escx_profile_enable(hd, 1)
N = 20
for _ in range(N):
    c, s = model.encode(x, 6); model.decode(c, s)
torch.cuda.synchronize()
lib.escx_profile_enable(hd, 0)
recs = json.loads(lib.escx_profile_report(hd).decode())
recs.sort(key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in recs)
print(f"B={B}: sum of kernel times {tot / N:.3f} ms per encode+decode, {sum(r['calls'] for r in recs) // N} launches")
for r in recs[:30]:
    print(f"  {r['name']:28s} calls/step {r['calls'] // N:3d}  {r['ms'] / N * 1e3:8.1f} us/step  {r['ms'] / r['calls'] * 1e3:7.1f} us/launch")
