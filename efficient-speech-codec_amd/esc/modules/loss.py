"""Generator losses of the reference (`/root/reference/esc/modules/loss/generator_loss.py:12-74`) on the MI355X.

Same classes, constructor arguments and call signatures; the arithmetic (power-law compression, the seven STFTs of the mel loss as
windowed-DFT GEMMs, HTK filterbanks, L1 / log-L1 terms AND their gradients) runs in libescx.so.  Each forward evaluates the per-clip
loss together with d loss_b / d reconstruction; backward only scales that by the incoming per-clip gradient.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from .. import _native

MEL_WINDOWS = [32, 64, 128, 256, 512, 1024, 2048]
MEL_BINS = [5, 10, 20, 40, 80, 160, 320]
SR = 16000
POWER = 0.3


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _scale_rows(unit: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    lib = _native.load()
    g = g.to(torch.float32).contiguous()
    out = torch.empty_like(unit)
    B = unit.shape[0]
    with torch.cuda.device(unit.device):
        _native.check(lib.escx_scale_rows(_ptr(unit), _ptr(g), _ptr(out), B, unit.numel() // max(B, 1), _stream(unit.device)))
    return out


class _StftLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, recon):
        if not recon.is_cuda:
            raise RuntimeError("esc.modules losses run on a HIP device only (no CPU implementation in this package)")
        lib = _native.load()
        raw_c, rec_c = raw.detach().to(torch.float32).contiguous(), recon.detach().to(torch.float32).contiguous()
        if raw_c.shape != rec_c.shape:
            raise RuntimeError(f"The size of raw_feat {tuple(raw_c.shape)} must match recon_feat {tuple(rec_c.shape)}")
        B = rec_c.shape[0]
        loss = torch.empty(B, dtype=torch.float32, device=rec_c.device)
        need = ctx.needs_input_grad[1]
        unit = torch.empty_like(rec_c) if need else None
        with torch.cuda.device(rec_c.device):
            _native.check(lib.escx_stft_loss(_ptr(raw_c), _ptr(rec_c), B, rec_c.numel() // B, _ptr(loss), _ptr(unit), _stream(rec_c.device)))
        ctx.unit = unit
        return loss

    @staticmethod
    def backward(ctx, g):
        return None, (_scale_rows(ctx.unit, g) if ctx.unit is not None else None)


class ComplexSTFTLoss(nn.Module):
    """L2 loss on power-law compressed complex STFTs (generator_loss.py:12-35).  (B,2,F,T) x (B,2,F,T) -> (B,)."""

    def __init__(self, weight=1.0, power_law=True):
        super().__init__()
        if not power_law:
            raise NotImplementedError("only the power-law compressed form the trainers use is implemented")
        self.power_law, self.weight = power_law, weight

    def forward(self, raw_feat, recon_feat):
        return self.weight * _StftLossFn.apply(raw_feat, recon_feat)


class _MelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, recon, sr):
        if not recon.is_cuda:
            raise RuntimeError("esc.modules losses run on a HIP device only (no CPU implementation in this package)")
        lib = _native.load()
        raw_c, rec_c = raw.detach().to(torch.float32).contiguous(), recon.detach().to(torch.float32).contiguous()
        if raw_c.shape != rec_c.shape or rec_c.dim() != 2:
            raise RuntimeError(f"raw_audio {tuple(raw_c.shape)} and recon_audio {tuple(rec_c.shape)} must both be (B, L)")
        B, L = rec_c.shape
        loss = torch.empty(B, dtype=torch.float32, device=rec_c.device)
        need = ctx.needs_input_grad[1]
        unit = torch.empty_like(rec_c) if need else None
        with torch.cuda.device(rec_c.device):
            _native.check(lib.escx_mel_loss(_ptr(raw_c), _ptr(rec_c), B, L, int(sr), _ptr(loss), _ptr(unit), _stream(rec_c.device)))
        ctx.unit = unit
        return loss

    @staticmethod
    def backward(ctx, g):
        return None, (_scale_rows(ctx.unit, g) if ctx.unit is not None else None), None


class MelSpectrogramLoss(nn.Module):
    """Multi-resolution L1 mel + log-mel loss (generator_loss.py:37-74).  (B,L) x (B,L) -> (B,)."""

    def __init__(self, weight=1.0, win_lengths=MEL_WINDOWS, n_mels=MEL_BINS, clamp_eps=1e-5):
        super().__init__()
        if list(win_lengths) != MEL_WINDOWS or list(n_mels) != MEL_BINS or clamp_eps != 1e-5:
            raise NotImplementedError("the HIP mel loss implements the reference's fixed resolutions (generator_loss.py:6-9)")
        self.n_mels, self.clamp_eps, self.weight = n_mels, clamp_eps, weight

    def forward(self, raw_audio, recon_audio):
        return self.weight * _MelLossFn.apply(raw_audio, recon_audio, SR)


class _GanTermFn(torch.autograd.Function):
    """Per-clip loss over the feature maps that live in ONE channels-last buffer (escx_gan_term): mode 0 = mean (target - x)^2, mode 1 = mean |x - ref|,
    summed over the buffer's maps (`metas`: one (C, Cp, D0, D1, P1, off1) per map; a concatenated buffer holds several column slices).
    The forward stores no gradient: the backward re-evaluates each term with the upstream per-clip gradient folded in (escx_gan_term_grad), so a
    buffer's gradient is written once, in one pass, and never zero-filled, scaled or added afterwards (the maps of a buffer tile it completely)."""

    @staticmethod
    def forward(ctx, buf, ref, metas, mode, target, lo=0, n=None):
        """Clips [lo, lo + n) of `buf` (default: all); `ref` already holds just those clips.  Taking the clip range here instead of a sliced tensor keeps
        autograd from materialising a zero-filled full-size gradient per slice (a pass over [reconstruction | real] hands out its first half)."""
        lib = _native.load()
        n = buf.shape[0] - lo if n is None else n
        loss = torch.empty(n, dtype=torch.float32, device=buf.device)
        base = lo * buf.stride(0) * 4
        with torch.cuda.device(buf.device):
            for k, (C, Cp, D0, D1, P1, off1) in enumerate(metas):
                off = 4 * off1 * Cp
                _native.check(lib.escx_gan_term(ctypes.c_void_p(buf.data_ptr() + base + off), None if ref is None else ctypes.c_void_p(ref.data_ptr() + off), None,
                                                n, C, Cp, D0, D1, P1, int(mode), float(target), _ptr(loss), int(k > 0), _stream(buf.device)))
        ctx.save_for_backward(buf, ref)
        ctx.metas, ctx.mode, ctx.target, ctx.lo, ctx.n = metas, int(mode), float(target), lo, n
        ctx.tiles = all(m[4] == metas[0][4] for m in metas) and sum(m[3] for m in metas) == metas[0][4]      # the slices cover every column of the buffer
        return loss

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return (None,) * 7
        buf, ref = ctx.saved_tensors
        lib = _native.load()
        lo, n = ctx.lo, ctx.n
        g = g.to(torch.float32).contiguous()
        if ctx.tiles:
            grad = torch.empty_like(buf, memory_format=torch.contiguous_format)
            grad[:lo].zero_()
            grad[lo + n:].zero_()
        else:
            grad = torch.zeros_like(buf, memory_format=torch.contiguous_format)
        base = lo * grad.stride(0) * 4
        with torch.cuda.device(buf.device):
            for C, Cp, D0, D1, P1, off1 in ctx.metas:
                off = 4 * off1 * Cp
                _native.check(lib.escx_gan_term_grad(ctypes.c_void_p(buf.data_ptr() + lo * buf.stride(0) * 4 + off), None if ref is None else ctypes.c_void_p(ref.data_ptr() + off),
                                                     _ptr(g), ctypes.c_void_p(grad.data_ptr() + base + off), n, C, Cp, D0, D1, P1, ctx.mode, ctx.target, _stream(buf.device)))
        return (grad,) + (None,) * 6


def _by_buffer(entries):
    """[(buffer, [meta, ...])] of feature-map entries (buffer, C, Cp, D0, D1, P1, off1), grouped by buffer in first-seen order."""
    groups = {}
    for buf, *meta in entries:
        groups.setdefault(buf.data_ptr(), (buf, []))[1].append(tuple(meta))
    return list(groups.values())


class GANLoss(nn.Module):
    """Least-squares GAN + feature-matching losses of the adversarial trainer (`/root/reference/esc/modules/loss/gan_loss.py:5-51`)."""

    def __init__(self, discriminator):
        super().__init__()
        self.discriminator = discriminator

    def forward(self, fake, real):
        if fake.dim() == 2:
            fake = fake.unsqueeze(1)
        if real.dim() == 2:
            real = real.unsqueeze(1)
        return self.discriminator(**dict(x=fake)), self.discriminator(**dict(x=real))

    def discriminator_loss(self, fake, real):
        d_fake, d_real = self.forward(fake.clone().detach(), real)
        loss = 0
        for xf, xr in zip(d_fake, d_real):
            bf, *mf = xf.entries[-1]
            br, *mr = xr.entries[-1]
            loss = loss + _GanTermFn.apply(bf, None, (tuple(mf),), 0, 0.0) + _GanTermFn.apply(br, None, (tuple(mr),), 0, 1.0)
        return loss

    def generator_loss(self, fake, real):
        d_fake, d_real = self.forward(fake, real)
        loss_g, loss_f = 0, 0
        for xf, xr in zip(d_fake, d_real):
            bf, *mf = xf.entries[-1]
            loss_g = loss_g + _GanTermFn.apply(bf, None, (tuple(mf),), 0, 1.0)
            for (ba, metas), (bb, _) in zip(_by_buffer(xf.entries[:-1]), _by_buffer(xr.entries[:-1])):
                loss_f = loss_f + _GanTermFn.apply(ba, bb.detach(), tuple(metas), 1, 0.0)
        return loss_g, loss_f

    # ---- one discriminator forward per signal and step (not in the reference, which runs them twice; the values are identical) --------
    def adversarial_forward(self, fake, real):
        """Feature maps of the reconstruction (differentiable w.r.t. the waveform, discriminator parameters held constant) and of the
        real signal (no graph): everything both updates of an adversarial step need."""
        if fake.dim() == 2:
            fake = fake.unsqueeze(1)
        if real.dim() == 2:
            real = real.unsqueeze(1)
        # ONE pass over [reconstructions | real signals] (round 3): the discriminator's weights are the same for both, so the two passes of round 2
        # become launches of twice the rows (half the launches, half the partial-sum reductions of the weight gradients); a clip's feature maps do
        # not depend on the batch it is in (every output element is one k-ordered contraction), so the loss values are bit for bit those of two
        # passes.  Only the leading clips (the reconstructions) can receive gradient.  ESCX_GAN_TWO_PASSES=1: the round-2 form.
        if os.environ.get("ESCX_GAN_TWO_PASSES") == "1" or fake.shape != real.shape:
            d_fake = self.discriminator(fake, detach_params=True)
            with torch.no_grad():
                d_real = self.discriminator(real)
            return d_fake, d_real
        B = fake.shape[0]
        both = self.discriminator(torch.cat([fake, real.detach().to(fake.dtype)], dim=0), detach_params=True, grad_clips=B)
        return both.clips(0, B), both.clips(B, 2 * B)

    def generator_loss_from(self, d_fake, d_real):
        """gan_loss.py:39-51 on feature maps that are already there."""
        loss_g, loss_f = 0, 0
        whole = getattr(d_fake, "whole", None)
        if whole is not None:                   # clips [lo, hi) of a larger pass: hand the terms the pass's own buffers and the clip range
            lo, n = d_fake.lo, d_fake.hi - d_fake.lo
            for xw, xr in zip(whole, d_real):
                bf, *mf = xw.entries[-1]
                loss_g = loss_g + _GanTermFn.apply(bf, None, (tuple(mf),), 0, 1.0, lo, n)
                for (ba, metas), (bb, _) in zip(_by_buffer(xw.entries[:-1]), _by_buffer(xr.entries[:-1])):
                    loss_f = loss_f + _GanTermFn.apply(ba, bb.detach(), tuple(metas), 1, 0.0, lo, n)
            return loss_g, loss_f
        for xf, xr in zip(d_fake, d_real):
            bf, *mf = xf.entries[-1]
            loss_g = loss_g + _GanTermFn.apply(bf, None, (tuple(mf),), 0, 1.0)
            for (ba, metas), (bb, _) in zip(_by_buffer(xf.entries[:-1]), _by_buffer(xr.entries[:-1])):
                loss_f = loss_f + _GanTermFn.apply(ba, bb.detach(), tuple(metas), 1, 0.0)
        return loss_g, loss_f

    @torch.no_grad()
    def discriminator_backward_from(self, d_fake, d_real):
        """gan_loss.py:30-37 + `disc_loss.mean().backward()` (trainer_adv.py:96-105) on the feature maps of adversarial_forward: returns the
        per-clip discriminator loss and ADDS d mean(loss) / d parameter to the discriminator's gradients."""
        lib = _native.load()
        whole = getattr(d_fake, "whole", None)
        if whole is not None and whole is getattr(d_real, "whole", None) and d_fake.lo == 0 and d_real.lo == d_fake.hi:
            # both halves of ONE pass: LS-GAN terms per half (targets 0 / 1), then one backward (dX chain + weight gradients) over all 2B clips
            B = d_fake.hi
            dbufs = [None] * len(whole.bufs)
            scale = torch.full((2 * B,), 1.0 / B, dtype=torch.float32, device=whole.wave.device)      # d mean_b(term_fake_b + term_real_b)
            loss = None
            for sub in whole:
                buf, C, Cp, D0, D1, P1, off1 = sub.entries[-1]
                buf = buf.detach()
                term = torch.empty(2 * B, dtype=torch.float32, device=buf.device)
                grad = torch.empty_like(buf)                     # the last map of a sub-discriminator is a buffer of its own
                with torch.cuda.device(buf.device):
                    for lo, target in ((0, 0.0), (B, 1.0)):
                        _native.check(lib.escx_gan_term(_ptr(buf[lo:lo + B]), None, None, B, C, Cp, D0, D1, P1, 0, float(target), _ptr(term[lo:lo + B]), 0,
                                                        _stream(buf.device)))
                        _native.check(lib.escx_gan_term_grad(_ptr(buf[lo:lo + B]), None, _ptr(scale[lo:lo + B]), _ptr(grad[lo:lo + B]), B, C, Cp, D0, D1, P1, 0, float(target),
                                                             _stream(buf.device)))
                t = term[:B] + term[B:]
                loss = t if loss is None else loss + t
                bi = next(i for i, b in enumerate(whole.bufs) if b.data_ptr() == buf.data_ptr())
                dbufs[bi] = grad
            self.discriminator.accumulate_param_grads(whole, dbufs)
            return loss
        loss = None
        for out, target in ((d_fake, 0.0), (d_real, 1.0)):
            B = out.wave.shape[0]
            dbufs = [None] * len(out.bufs)
            scale = torch.full((B,), 1.0 / B, dtype=torch.float32, device=out.wave.device)
            for sub in out:
                buf, C, Cp, D0, D1, P1, off1 = sub.entries[-1]
                buf = buf.detach()
                term = torch.empty(B, dtype=torch.float32, device=buf.device)
                grad = torch.empty_like(buf)
                with torch.cuda.device(buf.device):
                    _native.check(lib.escx_gan_term(_ptr(buf), None, None, B, C, Cp, D0, D1, P1, 0, float(target), _ptr(term), 0, _stream(buf.device)))
                    _native.check(lib.escx_gan_term_grad(_ptr(buf), None, _ptr(scale), _ptr(grad), B, C, Cp, D0, D1, P1, 0, float(target), _stream(buf.device)))
                loss = term if loss is None else loss + term
                bi = next(i for i, b in enumerate(out.bufs) if b.data_ptr() == buf.data_ptr())
                dbufs[bi] = grad
            self.discriminator.accumulate_param_grads(out, dbufs)
        return loss
