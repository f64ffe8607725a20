"""TEST INFRASTRUCTURE ONLY -- import shims that let the *reference* package (/root/reference/esc)
be imported in the build container, where torchaudio / timm / audiotools are not installed.

Used only by oracle/gen_golden.py (golden-vector generation) and oracle/time_reference.py.  Nothing here
travels to the GPU box in a way that matters: /root/reference does not exist there.

Shimmed third-party surface (pins from /root/reference/requirements.txt:1-9):
  * torchaudio.transforms.Spectrogram / InverseSpectrogram (torchaudio 2.0.0): thin wrappers over
    torch.stft / torch.istft with torchaudio's defaults (hann periodic window of win_length, center=True,
    pad_mode="reflect", normalized=False, onesided=True).  torchaudio's Spectrogram(power=None) is a
    direct torch.stft call, so this is the same computation -- but it is NOT the pinned wheel:
    "parity unpinned" at the torchaudio boundary.
  * timm.models.layers.trunc_normal_ / to_2tuple (only used at init).
  * audiotools (only imported by the discriminator, never executed on the encode/decode path).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


class _Spectrogram(nn.Module):
    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window,
                 power=2.0, normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.power = power
        self.register_buffer("window", window_fn(self.win_length), persistent=True)

    def forward(self, x):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        spec = torch.stft(x2, self.n_fft, self.hop_length, self.win_length, self.window, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        spec = spec.reshape(shape[:-1] + spec.shape[-2:])
        if self.power is not None:
            spec = spec.abs().pow(self.power)
        return spec


class _InverseSpectrogram(nn.Module):
    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window,
                 normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.register_buffer("window", window_fn(self.win_length), persistent=True)

    def forward(self, spec, length=None):
        shape = spec.shape
        s2 = spec.reshape(-1, shape[-2], shape[-1])
        wav = torch.istft(s2, self.n_fft, self.hop_length, self.win_length, self.window, center=True,
                          normalized=False, onesided=True, length=length, return_complex=False)
        return wav.reshape(shape[:-2] + wav.shape[-1:])


class _MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram (2.0.0 documentation, defaults): Spectrogram(n_fft, win_length, hop_length, hann
    periodic window, power, center=True, reflect, onesided, un-normalised) followed by MelScale(n_mels, sample_rate, f_min=0,
    f_max=sample_rate//2, n_stft=n_fft//2+1, norm=None, mel_scale="htk"): mel = fb^T . spec with triangular filters whose corner
    frequencies are equally spaced on the HTK mel scale  m = 2595 log10(1 + f/700).  Written filter by filter from that
    description (NOT the pinned wheel: the mel metric / mel loss stay "parity unpinned at the torchaudio boundary")."""

    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None, pad=0, n_mels=128,
                 window_fn=torch.hann_window, power=2.0, normalized=False, wkwargs=None, center=True, pad_mode="reflect",
                 onesided=None, norm=None, mel_scale="htk"):
        super().__init__()
        assert norm is None and mel_scale == "htk" and center and pad_mode == "reflect" and not normalized and pad == 0
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.power = power
        f_max = float(sample_rate // 2) if f_max is None else f_max
        n_freqs = n_fft // 2 + 1
        import math
        mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
        hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
        m_lo, m_hi = mel(f_min), mel(f_max)
        corners = [hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
        fb = torch.zeros(n_freqs, n_mels, dtype=torch.float64)
        for k in range(n_freqs):
            f = (sample_rate // 2) * k / (n_freqs - 1)
            for j in range(n_mels):
                lo, mid, hi = corners[j], corners[j + 1], corners[j + 2]
                rise, fall = (f - lo) / (mid - lo), (hi - f) / (hi - mid)
                fb[k, j] = max(0.0, min(rise, fall))
        self.register_buffer("fb", fb.float(), persistent=False)
        self.register_buffer("window", window_fn(self.win_length), persistent=False)

    def forward(self, x):
        shape = x.shape
        spec = torch.stft(x.reshape(-1, shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
        if self.power != 1:
            spec = spec.pow(self.power)
        mel = torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)
        return mel.reshape(shape[:-1] + mel.shape[-2:])


def install():
    """Install stubs and put the reference on sys.path.  Idempotent."""
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (it never is on the GPU box)")

    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")
    tat.Spectrogram = _Spectrogram
    tat.InverseSpectrogram = _InverseSpectrogram
    tat.MelSpectrogram = _MelSpectrogram
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm_layers.trunc_normal_ = trunc_normal_
    timm_layers.to_2tuple = to_2tuple
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = timm_models
    sys.modules["timm.models.layers"] = timm_layers

    at = types.ModuleType("audiotools")
    at_ml = types.ModuleType("audiotools.ml")
    at_ml.BaseModel = nn.Module
    at.ml = at_ml
    at.AudioSignal = type("AudioSignal", (), {})
    at.STFTParams = type("STFTParams", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["audiotools"] = at
    sys.modules["audiotools.ml"] = at_ml

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # make sure a previously imported product package named `esc` does not shadow the reference
    for name in [m for m in sys.modules if m == "esc" or m.startswith("esc.")]:
        del sys.modules[name]


def load_reference():
    """Returns the reference `esc.models` module."""
    install()
    import importlib
    return importlib.import_module("esc.models")
