#!/usr/bin/env python
"""A/B harness for kernel switches (tuning aid): every arm runs in its own process (the library reads its switches once), interleaved
over rounds; per arm: SHA-256 of the codes + decoded audio of the bench batch (are two arms bit-identical?), median / min ms per
encode+decode step, and optionally the per-kernel event breakdown of selected launch groups.

    python tools/ab.py [--rounds 3] [--steps 20] [--batch 36] [--groups merge_fused,split_fused] NAME:ENV=VAL,ENV=VAL  NAME2:...
    python tools/ab.py base: ws4:ESCX_ROWGEMM_WS=4          # "base:" = no extra environment
Child mode (internal): ab.py --child
"""
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    steps, batch = int(os.environ["AB_STEPS"]), int(os.environ["AB_BATCH"])
    groups = [g for g in os.environ.get("AB_GROUPS", "").split(",") if g]
    dev = torch.device("cuda:0")
    model, cfg, sd = bench.build_model(dev)
    x = bench.synth_batch(batch, 0, None).to(dev)
    codes, shape = model.encode(x, bench.NUM_STREAMS)
    wave = model.decode(codes, shape)
    torch.cuda.synchronize()
    h = hashlib.sha256(codes.cpu().numpy().tobytes() + wave.cpu().numpy().tobytes()).hexdigest()[:16]
    if os.environ.get("AB_DUMP"):                       # tests/test_gpu_parity.py: the arrays themselves, for arms that are not meant to be bit-identical
        import numpy as np
        np.savez(os.environ["AB_DUMP"], codes=codes.cpu().numpy(), wave=wave.cpu().numpy())
    for _ in range(3):
        c, s = model.encode(x, bench.NUM_STREAMS); model.decode(c, s)
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, s = model.encode(x, bench.NUM_STREAMS); model.decode(c, s)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    out = {"hash": h, "median_ms": statistics.median(times), "min_ms": min(times)}
    if groups:
        lib, hd = model._handle(dev)
        lib.escx_profile_enable(hd, 2)          # 2: batch parts back to back, kernels alone on the GPU
        n = 5
        for _ in range(n):
            c, s = model.encode(x, bench.NUM_STREAMS); model.decode(c, s)
        recs = json.loads(lib.escx_profile_report(hd).decode())
        lib.escx_profile_enable(hd, 0)
        out["groups"] = {r["name"]: round(r["ms"] / n, 4) for r in recs if any(r["name"].startswith(g) for g in groups)}
        out["isolated_sum_ms"] = round(sum(r["ms"] for r in recs) / n, 3)
    if os.environ.get("AB_BENCH") == "1":               # the driver's own measurement: K back-to-back steps between two synchronisations
        for _ in range(5):
            c, s = model.encode(x, bench.NUM_STREAMS); model.decode(c, s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            c, s = model.encode(x, bench.NUM_STREAMS); model.decode(c, s)
        torch.cuda.synchronize(); out["bench_ms"] = (time.perf_counter() - t0) * 1e3 / steps
    print("AB_RESULT " + json.dumps(out))


def main():
    args = sys.argv[1:]
    rounds, steps, batch, groups = 3, 20, 36, ""
    arms = []
    i = 0
    while i < len(args):
        a = args[i]
        if a == "--rounds": rounds = int(args[i + 1]); i += 2
        elif a == "--steps": steps = int(args[i + 1]); i += 2
        elif a == "--batch": batch = int(args[i + 1]); i += 2
        elif a == "--groups": groups = args[i + 1]; i += 2
        else:
            name, _, envs = a.partition(":")
            arms.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
            i += 1
    res = {name: [] for name, _ in arms}
    for r in range(rounds):
        for name, env in arms:
            e = dict(os.environ, AB_STEPS=str(steps), AB_BATCH=str(batch), AB_GROUPS=groups if r == 0 else "", AB_BENCH="1", **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
            if not line:
                print(f"[{name}] FAILED rc={p.returncode}: {p.stderr[-600:]}")
                continue
            res[name].append(json.loads(line[0][10:]))
    for name, _ in arms:
        rs = res[name]
        if not rs:
            continue
        hashes = sorted({r["hash"] for r in rs})
        print(f"{name:24s} hash {','.join(hashes)}  median ms/step {[round(r['median_ms'], 3) for r in rs]}  min {[round(r['min_ms'], 3) for r in rs]}"
              + (f"  back-to-back {[round(r['bench_ms'], 3) for r in rs]}" if 'bench_ms' in rs[0] else ""))
        if "groups" in rs[0]:
            print(f"{'':24s} isolated sum {rs[0]['isolated_sum_ms']} ms/step; " + "  ".join(f"{k} {v}" for k, v in sorted(rs[0]['groups'].items())))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        main()
