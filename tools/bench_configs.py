#!/usr/bin/env python
"""Throughput of the other BASELINE configurations on one MI355X (informational; bench.py is the contract line).

    python tools/bench_configs.py        # config 3 (B=64, S=1..6), ESC-Large B=36, B=288 on one GPU, B=1 latency
"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from esc import synth
from esc.models import make_model
from esc.models.codecs import state_manifest

dev = torch.device("cuda:0")


def build(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    cfg = json.loads(str(g["config_json"]))
    m = make_model(cfg)
    sd = {k: torch.from_numpy(np.ascontiguousarray(synth.synth_tensor(k, shp) if not k.endswith(".window") else torch.hann_window(shp[0]).numpy()))
          for k, shp in state_manifest(m.cfg).items()}
    m.load_state_dict(sd)
    return m.to(dev).eval()


def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
base = build("base")
x64 = bench.synth_batch(64, 7).to(dev)
base.reserve(64, 48000, dev)
for S in range(1, 7):
    te = timeit(lambda: base.encode(x64, S))
    codes, shape = base.encode(x64, S)
    td = timeit(lambda: base.decode(codes, shape))
    out[f"base_B64_S{S}"] = {"encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2), "audio_s_per_s": round(64 * 3 / (te + td), 1)}
for B in (1, 2, 4, 8, 288):
    x = bench.synth_batch(min(B, 36), 3).to(dev)
    if B > 36: x = x.repeat(B // 36, 1)
    base.reserve(B, 48000, dev)
    def step():
        c, s = base.encode(x, 6); base.decode(c, s)
    t = timeit(step, reps=5 if B > 36 else 20)
    out[f"base_B{B}_S6"] = {"ms": round(t * 1e3, 3), "audio_s_per_s": round(B * 3 / t, 1)}
large = build("large")
x36 = bench.synth_batch(36, 1).to(dev)
large.reserve(36, 48000, dev)
def stepl():
    c, s = large.encode(x36, 6); large.decode(c, s)
t = timeit(stepl)
out["large_B36_S6"] = {"ms": round(t * 1e3, 3), "audio_s_per_s": round(36 * 3 / t, 1), "gflop_per_clip": 99.25,
                       "whole_path_tflops": round(99.25e9 * 36 / t / 1e12, 1)}
print(json.dumps(out, indent=1))
