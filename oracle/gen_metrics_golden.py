"""TEST INFRASTRUCTURE -- golden values of the reference's evaluation metrics (SISDR, EntropyCounter) computed by the real
reference code (/root/reference/scripts/metrics.py, imported with shims for torchaudio / pesq).  MelSpectrogramDistance
depends on torchaudio.transforms.MelSpectrogram, which is not installed here; the shim in ref_shims.py implements it from the
torchaudio documentation (torch.stft + HTK triangular filterbank), so the mel values are pinned to the REFERENCE's metric code on
top of that shim ("unpinned at the torchaudio boundary" only).  Also runs the reference's own eval_epoch (scripts/test.py:23-55)
with the reference model on fixture clips: per-bitrate SISDR / MelDistance / utilisation for the GPU harness test.

    python oracle/gen_metrics_golden.py     # writes tests/golden/metrics.npz
"""
import os, sys, types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
sys.modules["pesq"] = types.SimpleNamespace(pesq=lambda *a, **k: 0.0)
sys.modules["transformers"] = sys.modules.get("transformers") or types.ModuleType("transformers")
import importlib
metrics = importlib.import_module("scripts.metrics")

g = torch.Generator().manual_seed(7)
x = torch.randn(3, 4000, generator=g) * 0.1
y = x + 0.03 * torch.randn(3, 4000, generator=g)
y[1] = 0.5 * x[1] + 0.01                       # scaled + offset estimate
sisdr = metrics.SISDR()(x, y).numpy()
codes = torch.randint(0, 1024, (4, 2, 3, 50), generator=g)
codes[:, 1, 2] = 5                             # one collapsed codebook
ec = metrics.EntropyCounter(1024, num_streams=2, num_groups=3, device="cpu")
ec.update(codes); ec.update(codes.flip(0))
rate, util = ec.compute_utilization()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics.npz"), x=x.numpy(), y=y.numpy(), sisdr=sisdr,
                    codes=codes.numpy().astype(np.int16), rate=np.float64(rate),
                    util_keys=np.array(list(util.keys())), util_vals=np.array(list(util.values())))
print("sisdr", sisdr, "rate", rate, util)


# ---- MelSpectrogramDistance of the reference (metrics.py:96-121) on the shimmed MelSpectrogram
xm = torch.randn(2, 12000, generator=g) * 0.1
ym = xm + 0.02 * torch.randn(2, 12000, generator=g)
ym[1] = 0.7 * xm[1]
mel = metrics.MelSpectrogramDistance()(xm, ym).numpy()
print("mel distance", mel)

# ---- the reference's eval_epoch on the reference model: Base fixture clips + the first unfiltered clips
import json, yaml
import gen_golden as gg                                     # (puts the product package on sys.path: move the reference back in front)
sys.path.remove(ref_shims.REFERENCE_ROOT); sys.path.insert(0, ref_shims.REFERENCE_ROOT)
for _m in [m for m in sys.modules if m == "esc" or m.startswith("esc.")]:
    del sys.modules[_m]
test_mod = importlib.import_module("scripts.test")
ref_models = importlib.import_module("esc.models")
cfg = yaml.safe_load(open(f"{ref_shims.REFERENCE_ROOT}/configs/9kbps_esc_base.yaml"))["model"]
model, _ = gg.build_reference(ref_models, cfg)
base = np.load(os.path.join(ROOT, "tests", "golden", "base.npz"))
tags = [("noise", "unfiltered-noise-0"), ("voiced", "unfiltered-voiced-0")]
pcm = np.concatenate([base["pcm"], np.stack([(gg.synth.noise_clip_int16 if k == "noise" else gg.synth.voiced_clip_int16)(t, 48000) for k, t in tags])])
xe = torch.from_numpy(gg.synth.pcm_to_float(pcm))[:, :-80]   # EvalSet drops the last 80 samples (scripts/utils.py:40): input and reconstruction have equal length
loader = [xe[:2], xe[2:]]                                   # two batches of two clips (eval_epoch only iterates and calls len())
funcs = {"MelDistance": metrics.MelSpectrogramDistance(), "SISDR": metrics.SISDR()}
ec = metrics.EntropyCounter(cfg["codebook_size"], num_streams=cfg["max_streams"], num_groups=cfg["group_size"], device="cpu")
torch.set_num_threads(8)
perf = test_mod.eval_epoch(model, loader, funcs, ec, "cpu", 1.5, num_streams=None, verbose=False)
print(perf)

old = dict(np.load(os.path.join(ROOT, "tests", "golden", "metrics.npz")))
old.update(mel_x=xm.numpy(), mel_y=ym.numpy(), mel_dist=mel, eval_tags=np.array(json.dumps(tags)),
           eval_json=np.array(json.dumps(perf)))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics.npz"), **old)
