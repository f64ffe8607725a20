"""Evaluation metrics of the reference (`/root/reference/scripts/metrics.py`), restated without torchaudio / pesq.

  EntropyCounter          metrics.py:12-77   codebook utilisation = sum of per-codebook entropies / (S * G * log2 K)
  SISDR                   metrics.py:123-171 scale-invariant SDR (zero-mean, scaling)
  MelSpectrogramDistance  metrics.py:96-121  L1 distance of log10(mel^2) over 7 window sizes; torchaudio's MelSpectrogram
                                             (hann, center/reflect, power=1, HTK mel scale, norm=None) is re-implemented here
  PESQ                    metrics.py:79-94   only if the optional `pesq` package is importable

These are evaluation plumbing (plain torch ops on whatever device the tensors live on), not part of the accelerated path.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

MEL_WINDOWS = [32, 64, 128, 256, 512, 1024, 2048]
MEL_BINS = [5, 10, 20, 40, 80, 160, 320]
SR = 16000


class EntropyCounter:
    """Counter maintaining codebook utilisation on a held-out set (metrics.py:12-77)."""

    def __init__(self, codebook_size=1024, num_streams=6, num_groups=3, device="cuda"):
        self.num_groups, self.codebook_size, self.device = num_groups, codebook_size, device
        self.reset_stats(num_streams)

    def reset_stats(self, num_streams):
        self.counts = torch.zeros(num_streams, self.num_groups, self.codebook_size, device=self.device, dtype=torch.float64)
        self.total_counts = 0
        self.num_streams = num_streams
        self.max_entropy_per_book = math.log2(self.codebook_size)
        self.max_total_entropy = num_streams * self.num_groups * self.max_entropy_per_book

    def update(self, codes: torch.Tensor):
        assert codes.size(1) == self.num_streams and codes.size(2) == self.num_groups, "code indices size not match"
        self.total_counts += codes.size(0) * codes.size(-1)
        flat = codes.permute(1, 2, 0, 3).reshape(self.num_streams * self.num_groups, -1).to(self.counts.device)
        offs = torch.arange(flat.size(0), device=flat.device).unsqueeze(1) * self.codebook_size
        self.counts.view(-1).index_add_(0, (flat + offs).reshape(-1), torch.ones(flat.numel(), dtype=torch.float64, device=flat.device))

    def compute_utilization(self):
        assert self.total_counts > 0, "No data collected, please update on a specific dataset"
        dist = (self.counts / self.total_counts).to(torch.float32)
        ent = -(dist * torch.log2(dist + 1e-10)).sum(-1)               # (S, G)
        util = {f"stream_{s}_group_{g + 1}": round(float(ent[s, g]) / self.max_entropy_per_book, 4)
                for s in range(self.num_streams) for g in range(self.num_groups)}
        return round(float(ent.sum()) / self.max_total_entropy, 4), util


class SISDR(nn.Module):
    """Scale-invariant source-to-distortion ratio in dB, (B, L) x (B, L) -> (B,)  (metrics.py:123-171)."""

    def __init__(self, scaling: bool = True, zero_mean: bool = True):
        super().__init__()
        self.scaling, self.zero_mean = scaling, zero_mean

    def forward(self, x, y):
        eps = 1e-8
        ref, est = x.reshape(x.shape[0], -1), y.reshape(y.shape[0], -1)
        if self.zero_mean:
            ref = ref - ref.mean(dim=1, keepdim=True)
            est = est - est.mean(dim=1, keepdim=True)
        proj = (ref ** 2).sum(dim=1) + eps
        cross = (est * ref).sum(dim=1) + eps
        scale = (cross / proj).unsqueeze(1) if self.scaling else 1
        e_true = scale * ref
        e_res = est - e_true
        return 10 * torch.log10((e_true ** 2).sum(dim=1) / (e_res ** 2).sum(dim=1) + eps)


def _hz_to_mel(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def _mel_to_hz(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'): (n_freqs, n_mels) triangular filters."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    f_pts = _mel_to_hz(np.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.from_numpy(np.maximum(0.0, np.minimum(down, up)).astype(np.float32))


class MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, n_mels, power=1) defaults."""

    def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels, power=1.0):
        super().__init__()
        self.n_fft, self.win_length, self.hop_length, self.power = n_fft, win_length, hop_length, power
        self.register_buffer("window", torch.hann_window(win_length), persistent=False)
        self.register_buffer("fb", melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate), persistent=False)

    def forward(self, x):
        spec = torch.stft(x, self.n_fft, self.hop_length, self.win_length, self.window, center=True, pad_mode="reflect",
                          normalized=False, onesided=True, return_complex=True).abs()
        if self.power != 1.0:
            spec = spec.pow(self.power)
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


class MelSpectrogramDistance(nn.Module):
    """L1 log-mel distance summed over 7 resolutions (metrics.py:96-121)."""

    def __init__(self, win_lengths=MEL_WINDOWS, n_mels=MEL_BINS, clamp_eps=1e-5):
        super().__init__()
        self.mel_transf = nn.ModuleList([MelSpectrogram(SR, w, w, w // 4, n_mels[i], power=1.0) for i, w in enumerate(win_lengths)])
        self.clamp_eps = clamp_eps

    def forward(self, raw_audio, recon_audio):
        loss = 0.0
        for mel in self.mel_transf:
            xm, ym = mel(raw_audio), mel(recon_audio)
            loss = loss + F.l1_loss(xm.clamp(self.clamp_eps).pow(2).log10(), ym.clamp(self.clamp_eps).pow(2).log10(),
                                    reduction="none").mean(dim=[1, 2])
        return loss


class PESQ:
    """Wide-band PESQ per clip; needs the optional `pesq` C extension (metrics.py:79-94)."""

    def __init__(self):
        from pesq import pesq            # noqa: F401  (ImportError if unavailable; the harness then skips the metric)
        self._pesq = pesq

    def __call__(self, x, y):
        return torch.tensor([self._pesq(SR, x[b].cpu().numpy(), y[b].cpu().numpy(), "wb") for b in range(x.size(0))])
