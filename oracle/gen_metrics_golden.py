"""TEST INFRASTRUCTURE -- golden values of the reference's evaluation metrics (SISDR, EntropyCounter) computed by the real
reference code (/root/reference/scripts/metrics.py, imported with shims for torchaudio / pesq).  MelSpectrogramDistance
depends on torchaudio.transforms.MelSpectrogram, which is not installed here: unpinned.

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
