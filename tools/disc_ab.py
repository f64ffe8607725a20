#!/usr/bin/env python
"""A/B of discriminator kernel switches (tuning aid): each arm runs in its own process (switches are read once), prints a SHA-256 over every feature map,
every parameter gradient and the waveform gradient of one forward + backward on 8 synthetic 3 s clips - are two arms bit-identical?
    python tools/disc_ab.py NAME:ENV=V,ENV=V NAME2:...      (child mode: --child)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    import numpy as np, torch
    from esc import synth
    from esc.models import Discriminator
    torch.manual_seed(0)
    disc = Discriminator(sample_rate=16000).cuda()
    disc.set_conv_precision(os.environ.get("AB_PRECISION", "fp32"))
    B = int(os.environ.get("AB_BATCH", "8"))
    pcm = np.stack([(synth.voiced_clip_int16 if i % 2 else synth.noise_clip_int16)(f"dab-{i}", 48000) for i in range(B)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda().unsqueeze(1).requires_grad_(True)
    outs = disc(x)
    loss = sum((f[-1] ** 2).mean() for f in outs) + sum(fm.abs().mean() for f in outs for fm in f[:-1])
    loss.backward()
    torch.cuda.synchronize()
    hf, hg = hashlib.sha256(), hashlib.sha256()
    for f in outs:
        for fm in f:
            hf.update(fm.detach().cpu().numpy().tobytes())
    for k, p in disc.named_parameters():
        hg.update(p.grad.cpu().numpy().tobytes())
    hx = hashlib.sha256(x.grad.cpu().numpy().tobytes())
    print(f"AB_RESULT fmaps {hf.hexdigest()[:16]} param-grads {hg.hexdigest()[:16]} d-wave {hx.hexdigest()[:16]} loss {float(loss.detach()):.6f}")


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for a in sys.argv[1:]:
            name, _, envs = a.partition(":")
            e = dict(os.environ, **dict(kv.split("=", 1) for kv in envs.split(",") if kv))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT")]
            print(f"{name:16s} {line[0][10:] if line else 'FAILED: ' + p.stderr[-500:]}")
