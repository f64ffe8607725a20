"""Is the ESC-Base training step GPU-bound or host-bound?  Runs bench.py's --mode train step with the GPU left asynchronous and prints, per
phase (forward, losses, backward, optimiser), the HOST time to enqueue it next to the wall time of the whole step with a sync at the end; then
the step with a sync after every phase (GPU time per phase), then a cProfile of the enqueue.
Usage (GPU box): python tools/train_host_profile.py [batch]"""
import cProfile, io, os, pstats, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import numpy as np, torch
import bench
from esc import synth
from esc.modules import ComplexSTFTLoss, MelSpectrogramLoss
from esc.optim import FlatAdamW


def main():
    bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    dev = torch.device("cuda:0")
    model, cfg, sd = bench.build_model(dev)
    model.train()
    pcm = np.stack([synth.noise_clip_int16(f"bench-r0-{i}", bench.TRAIN_SAMPLES) for i in range(bsz)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).to(dev)
    mel_fn, stft_fn = MelSpectrogramLoss(), ComplexSTFTLoss()
    opt = FlatAdamW(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, max_grad_norm=0.5, device=dev)
    w = bench.TRAIN_WEIGHTS

    def step(marks=None, sync=False):
        def mark(name):
            if sync:
                torch.cuda.synchronize()
            if marks is not None:
                marks.append((name, time.perf_counter()))
        mark("start")
        out = model(**dict(x=x, x_feat=None, num_streams=bench.NUM_STREAMS, freeze_codebook=False))
        mark("forward")
        loss = (out["cm_loss"] * w["cm_weight"] + out["cb_loss"] * w["cb_weight"] + mel_fn(out["raw_audio"], out["recon_audio"]) * w["mel_weight"]
                + stft_fn(out["raw_feat"], out["recon_feat"]) * w["stft_weight"]).mean()
        mark("losses")
        loss.backward()
        mark("backward")
        opt.step(); opt.zero_grad()
        mark("optimiser")

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for mode in ("async", "sync"):
        acc = {}
        n = 5
        wall = 0.0
        for _ in range(n):
            marks = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            step(marks, sync=(mode == "sync"))
            torch.cuda.synchronize(); wall += time.perf_counter() - t0
            for (a, ta), (b, tb) in zip(marks, marks[1:]):
                acc[b] = acc.get(b, 0.0) + (tb - ta)
        label = "host enqueue per phase (GPU asynchronous)" if mode == "async" else "wall per phase (sync after each)"
        print(f"{label}: " + "  ".join(f"{k} {1e3 * v / n:.2f} ms" for k, v in acc.items()) + f"  | sum {1e3 * sum(acc.values()) / n:.2f} ms, step wall {1e3 * wall / n:.2f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
