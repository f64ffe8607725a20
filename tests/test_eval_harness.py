"""Evaluation harness (efficient-speech-codec_amd/scripts): metrics pinned to golden values from the reference's own
metrics.py (SISDR, EntropyCounter); mel filterbank / distance properties (torchaudio is unavailable: unpinned)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
from scripts import metrics as M  # noqa: E402


def test_sisdr_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    got = M.SISDR()(torch.from_numpy(g["x"]), torch.from_numpy(g["y"])).numpy()
    np.testing.assert_allclose(got, g["sisdr"], rtol=1e-5, atol=1e-4)
    assert M.SISDR()(torch.from_numpy(g["x"]), torch.from_numpy(g["x"]) * 3.0 + 0.2).min() > 60      # scale / offset invariant


def test_entropy_counter_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "metrics.npz"))
    codes = torch.from_numpy(g["codes"].astype(np.int64))
    ec = M.EntropyCounter(1024, num_streams=2, num_groups=3, device="cpu")
    ec.update(codes); ec.update(codes.flip(0))
    rate, util = ec.compute_utilization()
    assert rate == pytest.approx(float(g["rate"]), abs=1e-4)
    for k, v in zip(g["util_keys"], g["util_vals"]):
        assert util[str(k)] == pytest.approx(float(v), abs=1e-4)
    assert util["stream_1_group_3"] == 0.0                                  # the collapsed codebook
    with pytest.raises(AssertionError):
        ec.update(codes[:, :1])


def test_mel_filterbank_and_distance_properties():
    fb = M.melscale_fbanks(257, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (257, 80) and float(fb.min()) >= 0 and float(fb.max()) <= 1.0
    peaks = fb.argmax(0)
    assert (peaks[1:] >= peaks[:-1]).all()                                 # centre frequencies increase
    # HTK mel spacing: centre of filter m sits at mel_to_hz(linspace)[m+1]
    f_pts = 700.0 * (10.0 ** (np.linspace(0, 2595.0 * np.log10(1 + 8000 / 700.0), 82) / 2595.0) - 1.0)
    assert abs(float(peaks[40]) * 8000 / 256 - f_pts[41]) < 8000 / 256
    d = M.MelSpectrogramDistance()
    x = torch.randn(2, 8000) * 0.1
    assert torch.allclose(d(x, x), torch.zeros(2))
    assert (d(x, x * 0.5) > 0).all() and d(x, x * 0.5).shape == (2,)


@pytest.mark.gpu
def test_eval_cli_on_synthetic_folder(tmp_path):
    """python -m scripts.test equivalent on a folder of synthetic wavs: perf_stats.json layout of scripts/test.py:81."""
    import json
    import subprocess
    from scipy.io import wavfile
    from esc import synth
    d = tmp_path / "wavs"; d.mkdir()
    for i in range(3):
        wavfile.write(d / f"clip{i}.wav", 16000, synth.voiced_clip_int16(f"eval-{i}", 48000))
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "efficient-speech-codec_amd"))
    out = subprocess.run([sys.executable, "-m", "scripts.test", "--eval_folder_path", str(d), "--batch_size", "3", "--synthetic", "base",
                          "--save_path", str(tmp_path / "out"), "--device", "cuda"], capture_output=True, text=True, env=env,
                         cwd=os.path.join(ROOT, "efficient-speech-codec_amd"))
    assert out.returncode == 0, out.stderr[-3000:]
    stats = json.load(open(tmp_path / "out" / "perf_stats.json"))
    assert set(stats) >= {"MelDistance", "SISDR", "utilization"}
    assert all(len(v) == 6 for v in stats.values())
    assert all(0.0 <= u <= 1.0 for u in stats["utilization"])
