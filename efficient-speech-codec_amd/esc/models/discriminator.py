"""DAC discriminator of the reference's adversarial trainer (`/root/reference/esc/models/discriminator.py:31-221`) on the MI355X.

Same constructor (`rates, periods, fft_sizes, sample_rate, bands`), same parameter names (`discriminators.{i}.convs.{j}.0.{bias,weight_g,
weight_v}`, `...band_convs.{b}.{j}.0...`, `...conv_post...`: reference checkpoints load with `load_state_dict`), same call:
`disc(x)` with x (B, 1, L) returns one list of feature maps (B, C, D0, D1) per sub-discriminator, differentiable w.r.t. the parameters and
the waveform.  The convolutions, weight normalisation, matched-stride STFT and all backward passes run in libescx (csrc/disc.hip); there
is no PyTorch/CPU implementation here.  MSD (`rates`) is not implemented - no ESC configuration uses it.
"""
from __future__ import annotations

import ctypes
import math
from typing import List

import torch
import torch.nn as nn

from .. import _native
from .codecs import _Node, _attach

BANDS = [(0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)]
_PRECISIONS = {"fp32": 0, "bf16": 1, "split": 2}


def _default_conv_precision() -> int:
    """ESCX_DISC_PRECISION (fp32 | bf16 | split) names the default of NEW Discriminator objects; anything else is an error that says so (not a bare
    KeyError at import time).  The Python host defaults to "split" (fp32-grade, three exact bf16 terms); a C host that never calls escx_disc_set_precision gets mode 0
    (fp32 MFMA) - INTEGRATION.md section 2 spells the difference out."""
    v = __import__("os").environ.get("ESCX_DISC_PRECISION", "split")
    if v not in _PRECISIONS:
        raise ValueError(f"ESCX_DISC_PRECISION={v!r}: expected one of {sorted(_PRECISIONS)}")
    return _PRECISIONS[v]


_DEFAULT_CONV_PRECISION = _default_conv_precision()      # see Discriminator.set_conv_precision


def _conv_specs(periods, fft_sizes, n_bands):
    """(prefix, Cout, Cin, T0, T1) of every convolution in the reference's construction order."""
    out, idx = [], 0
    for _ in periods:
        p = f"discriminators.{idx}."
        ch = [1, 32, 128, 512, 1024, 1024]
        out += [(f"{p}convs.{j}.0.", ch[j + 1], ch[j], 5, 1) for j in range(5)] + [(f"{p}conv_post.", 1, 1024, 3, 1)]
        idx += 1
    for _ in fft_sizes:
        p = f"discriminators.{idx}."
        for b in range(n_bands):
            out += [(f"{p}band_convs.{b}.{j}.0.", 32, 2 if j == 0 else 32, 3, 9 if j < 4 else 3) for j in range(5)]
        out.append((f"{p}conv_post.", 1, 32, 3, 3))
        idx += 1
    return out


class FeatureMaps(list):
    """The reference's list of feature maps of one sub-discriminator, plus the channels-last buffers the HIP loss kernels read."""
    entries: list        # per map: (buffer, C, Cp, D0, D1, P1, off1)


class DiscOutput(list):
    """What Discriminator.forward returns: the reference's list (one FeatureMaps per sub-discriminator) plus what a later native backward
    needs (the pre-processed input waveform and the raw buffers), so that ONE forward pass can serve both the generator's and the
    discriminator's update of an adversarial step."""
    wave: torch.Tensor
    bufs: tuple
    layout: list

    def clips(self, lo: int, hi: int) -> "DiscOutput":
        """The same pass restricted to clips [lo, hi): views, no copies (every buffer is batch-major).  `whole` keeps the pass it came from, so that
        one discriminator pass over [reconstruction | real signal] can serve an adversarial step (esc.modules.GANLoss.adversarial_forward)."""
        out = DiscOutput(FeatureMaps(m[lo:hi] for m in fm) for fm in self)
        for fm, src in zip(out, self):
            fm.entries = [(e[0][lo:hi],) + tuple(e[1:]) for e in src.entries]
        out.wave, out.bufs, out.layout = self.wave[lo:hi], tuple(b[lo:hi] for b in self.bufs), self.layout
        out.whole, out.lo, out.hi = self, lo, hi
        return out


class Discriminator(nn.Module):
    def __init__(self, rates: list = [], periods: list = [2, 3, 5, 7, 11], fft_sizes: list = [2048, 1024, 512], sample_rate: int = 44100,
                 bands: list = BANDS):
        super().__init__()
        if rates:
            raise NotImplementedError("MSD (rates) is not implemented in the MI355X build; the ESC configurations use rates=[]")
        self.cfg = dict(rates=list(rates), periods=list(periods), fft_sizes=list(fft_sizes), sample_rate=sample_rate, bands=[tuple(b) for b in bands])
        for pfx, cout, cin, t0, t1 in _conv_specs(periods, fft_sizes, len(bands)):
            fan_in = cin * t0 * t1
            bound = 1.0 / math.sqrt(fan_in)
            v = torch.empty(cout, cin, t0, t1).uniform_(-bound, bound)                   # nn.Conv2d default init, then weight_norm: g = ||v||
            _attach(self, pfx + "bias", torch.empty(cout).uniform_(-bound, bound), False)
            _attach(self, pfx + "weight_g", v.flatten(1).norm(dim=1).view(cout, 1, 1, 1).clone(), False)
            _attach(self, pfx + "weight_v", v, False)
        self._handles, self._flat, self._carry = {}, {}, {}
        self._flat_grad_mode = False
        self._native_updates = 0

    # ---- flat-gradient mode (esc.optim.FlatAdamW), same contract as esc.ESC -------------------------------------------------
    def enable_flat_grads(self, device):
        device = torch.device(device)
        lib, hd = self._handle(device)
        flat = self._ensure_flat(device, lib, hd)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._flat[idx]
        if "gflat" not in st:
            old = self._carry.pop(idx, None)             # a rebuild keeps the gradient buffer and what it holds (same contract as esc.ESC)
            st["gflat"], st["gfresh"] = old if (old is not None and old[0].shape == flat.shape and old[0].device == flat.device) else (torch.zeros_like(flat), True)
        params = dict(self.named_parameters())
        for key, off, n in st["layout"]:
            params[key].grad = st["gflat"][off:off + n].view(params[key].shape)
        self._flat_grad_mode = True

    def flat_buffers(self, device):
        device = torch.device(device)
        lib, hd = self._handle(device)
        flat = self._ensure_flat(device, lib, hd)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if "gflat" not in self._flat[idx]:
            self.enable_flat_grads(device)
        return flat, self._flat[idx]["gflat"]

    def zero_flat_grads(self, device):
        idx = torch.device(device).index
        st = self._flat.get(idx if idx is not None else torch.cuda.current_device())
        if st is not None and "gflat" in st:
            st["gfresh"] = True

    def set_conv_precision(self, precision: str = "split"):
        """Arithmetic of the wide convolutions (include/escx.h escx_disc_set_precision).
        "split" (default since round 5; ESCX_DISC_PRECISION overrides the default): the 128 -> 512 -> 1024 -> 1024 period convolutions (forward, dX, dW) on the
        bf16 MFMA with every fp32 operand split exactly into three bf16 terms and the six leading cross products accumulated in fp32 - fp32-grade results
        (feature maps within 2e-6 of the fp32-MFMA path, every parity test against the reference unchanged) at 1.4-1.7x its rate;
        "fp32": the fp32 MFMA everywhere (the path of rounds 2-4);
        "bf16": operands of those convolutions and of the 32 -> 32 band convolutions rounded to bf16 while staged, fp32 accumulation (BASELINE configs[4]
        names bf16; narrower than the reference's arithmetic).  Feature maps, parameters and gradients stay fp32 tensors in every mode."""
        mode = _PRECISIONS.get(precision)
        if mode is None:
            raise ValueError(f"conv precision {precision!r}: 'fp32', 'bf16' or 'split'")
        self._conv_precision = mode
        if getattr(self, "_handles", None):
            lib = _native.load()
            for hd in self._handles.values():
                _native.check(lib.escx_disc_set_precision(hd, mode))
        return self

    def note_params_updated(self):
        self._native_updates += 1               # a native kernel wrote the flat buffer: autograd's version counters did not move

    def _version(self) -> int:
        """Changes whenever a parameter changes (in-place torch updates bump autograd's counters, native optimiser steps bump ours)."""
        return sum(p._version for p in self.parameters()) + (self._native_updates << 32)

    def _apply(self, fn, *a, **k):
        self._drop()
        return super()._apply(fn, *a, **k)

    def _drop(self):
        if getattr(self, "_handles", None):
            lib = _native.load()
            for hd in self._handles.values():
                lib.escx_disc_destroy(hd)
        self._carry = {i: (st["gflat"], st["gfresh"]) for i, st in getattr(self, "_flat", {}).items() if "gflat" in st}
        self._handles, self._flat = {}, {}

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_handles"], st["_flat"], st["_carry"], st["_flat_grad_mode"] = {}, {}, {}, False
        return st

    def _handle(self, device):
        lib = _native.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in self._handles:
            c = self.cfg
            cc = _native.EscxDiscConfig()
            cc.sample_rate, cc.n_rates, cc.n_periods, cc.n_ffts, cc.n_bands = c["sample_rate"], 0, len(c["periods"]), len(c["fft_sizes"]), len(c["bands"])
            for i, p in enumerate(c["periods"]):
                cc.periods[i] = p
            for i, w in enumerate(c["fft_sizes"]):
                cc.fft_sizes[i] = w
            for i, (lo, hi) in enumerate(c["bands"]):
                cc.bands[i][0], cc.bands[i][1] = lo, hi
            hd = ctypes.c_void_p()
            _native.check(lib.escx_disc_create(ctypes.byref(cc), idx, ctypes.byref(hd)))
            self._handles[idx] = hd
            mode = getattr(self, "_conv_precision", _DEFAULT_CONV_PRECISION)
            if mode:
                _native.check(lib.escx_disc_set_precision(hd, mode))
            if self._flat_grad_mode and idx in self._carry:
                self._flat_grad_mode = False             # (guards the re-entry through enable_flat_grads -> _handle)
                self.enable_flat_grads(device)
        return lib, self._handles[idx]

    def _ensure_flat(self, device, lib, hd):
        """Flat fp32 parameter buffer in the library's order; the nn.Parameters are views of it (same scheme as esc.ESC)."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._flat.get(idx)
        params = dict(self.named_parameters())
        if st is None:
            n = lib.escx_disc_param_count(hd)
            layout = [(lib.escx_disc_param_key(hd, i).decode(), int(lib.escx_disc_param_offset(hd, i)), int(lib.escx_disc_param_numel(hd, i))) for i in range(n)]
            st = {"flat": torch.zeros(int(lib.escx_disc_param_total(hd)), dtype=torch.float32, device=device), "layout": layout}
            self._flat[idx] = st
        flat, base = st["flat"], st["flat"].data_ptr()
        with torch.no_grad():
            for key, off, n in st["layout"]:
                p = params[key]
                if p.data_ptr() != base + 4 * off:
                    flat[off:off + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                    p.data = flat[off:off + n].view(p.shape)
        return flat

    def accumulate_param_grads(self, out: "DiscOutput", dbufs):
        """d loss / d parameters for gradients `dbufs` (one per buffer of `out`, None = zero) of a forward pass already done, added to the
        parameters' .grad (flat-gradient mode: into the flat buffer).  No autograd involved: the discriminator update of an adversarial
        step reuses the feature maps the generator update has just computed."""
        dev = out.wave.device
        lib, hd = self._handle(dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        flat = self._ensure_flat(dev, lib, hd)
        st = self._flat[idx]
        layout = out.layout
        _, where = _buffer_plan(layout)
        n = len(layout)
        B, L = out.wave.shape
        dbufs = [None if g is None else g.to(torch.float32).contiguous() for g in dbufs]
        fm = (ctypes.c_void_p * n)(*[out.bufs[bi].data_ptr() + 4 * off1 * layout[i][2] for i, (bi, off1) in enumerate(where)])
        dfm = (ctypes.c_void_p * n)(*[(None if dbufs[bi] is None else dbufs[bi].data_ptr() + 4 * off1 * layout[i][2]) for i, (bi, off1) in enumerate(where)])
        flat_mode = self._flat_grad_mode and "gflat" in st
        gnew = st["gflat"] if (flat_mode and st["gfresh"]) else torch.empty_like(flat)
        with torch.cuda.device(dev):
            _native.check(lib.escx_disc_backward(hd, ctypes.c_void_p(flat.data_ptr()), self._version(), ctypes.c_void_p(out.wave.data_ptr()), B, L, fm, dfm,
                                                 ctypes.c_void_p(gnew.data_ptr()), None, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if flat_mode:
            if st["gfresh"]:
                st["gfresh"] = False
            else:
                st["gflat"].add_(gnew)
            return
        params = dict(self.named_parameters())
        for k, off, n_ in st["layout"]:
            g = gnew[off:off + n_].view(params[k].shape)
            params[k].grad = g.clone() if params[k].grad is None else params[k].grad + g

    def fmap_layout(self, device, n_samples):
        lib, hd = self._handle(device)
        out = []
        ints = [ctypes.c_int() for _ in range(7)]
        for i in range(lib.escx_disc_num_fmaps(hd, n_samples)):
            _native.check(lib.escx_disc_fmap_shape(hd, n_samples, i, *[ctypes.byref(v) for v in ints]))
            out.append(tuple(v.value for v in ints))          # (sub, C, Cp, D0, D1, P1, off1)
        return out

    def forward(self, x, detach_params: bool = False, grad_clips: int = 0) -> List[FeatureMaps]:
        """detach_params=True treats the parameters as constants (the generator's update needs d loss / d waveform only).
        grad_clips = n > 0: only the first n clips of the batch can receive gradient (the rest is a constant, e.g. the real signals batched behind the
        reconstructions): the backward then runs on those n clips only."""
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError("x must have shape (B, 1, L)")
        if not x.is_cuda:
            raise RuntimeError("esc Discriminator (MI355X build): x must live on a HIP device; this package has no CPU implementation")
        layout = self.fmap_layout(x.device, x.shape[-1])
        wave = x[:, 0].to(torch.float32).contiguous()
        bufs = _DiscFn.apply(self, wave, (layout, bool(detach_params), int(grad_clips)), *self.parameters())
        out, bi = [], 0
        n_sub = len(self.cfg["periods"]) + len(self.cfg["fft_sizes"])
        per_sub = DiscOutput(FeatureMaps() for _ in range(n_sub))
        per_sub.wave, per_sub.bufs, per_sub.layout = wave.detach(), tuple(b.detach() for b in bufs), layout
        for fm in per_sub:
            fm.entries = []
        cat = None
        for sub, C, Cp, D0, D1, P1, off1 in layout:
            if P1 == D1:
                buf = bufs[bi]; bi += 1
            else:
                if off1 == 0:
                    cat = bufs[bi]; bi += 1
                buf = cat
            per_sub[sub].append(buf[:, :, off1:off1 + D1, :C].permute(0, 3, 1, 2))
            per_sub[sub].entries.append((buf, C, Cp, D0, D1, P1, off1))
        return per_sub


def _buffer_plan(layout):
    """Shapes of the distinct output buffers ([D0][P1][Cp], B prepended at allocation) and, per map, (buffer index, column offset)."""
    shapes, where, cat = [], [], -1
    for sub, C, Cp, D0, D1, P1, off1 in layout:
        if P1 == D1:
            shapes.append((D0, P1, Cp)); where.append((len(shapes) - 1, 0))
        else:                                   # a column slice of the concatenated buffer, which is allocated with its first slice
            if off1 == 0:
                shapes.append((D0, P1, Cp)); cat = len(shapes) - 1
            where.append((cat, off1))
    return shapes, where


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disc, wave, layout_flag, *params):
        layout, detach_params, grad_clips = layout_flag
        ctx.grad_clips = grad_clips
        dev = wave.device
        lib, hd = disc._handle(dev)
        flat = disc._ensure_flat(dev, lib, hd)
        B, L = wave.shape
        shapes, where = _buffer_plan(layout)
        bufs = [torch.empty((B,) + s, dtype=torch.float32, device=dev) for s in shapes]
        ptrs = (ctypes.c_void_p * len(layout))(*[bufs[bi].data_ptr() + 4 * off1 * layout[i][2] for i, (bi, off1) in enumerate(where)])
        with torch.cuda.device(dev):
            _native.check(lib.escx_disc_forward(hd, ctypes.c_void_p(flat.data_ptr()), disc._version(), ctypes.c_void_p(wave.data_ptr()), B, L, ptrs,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        # The feature-map buffers are OUTPUTS of this node: kept as plain attributes of ctx they would form a cycle (ctx -> buffer -> grad_fn -> ctx) that runs through
        # autograd's C++ node, invisible to Python's collector - every forward then leaked its 108 maps (13 GB at 72 signals) until the process ended (rounds 2-3).
        # save_for_backward stores outputs without that edge.
        ctx.save_for_backward(wave, *bufs)
        ctx.disc, ctx.layout, ctx.where, ctx.idx = disc, layout, where, (dev.index if dev.index is not None else torch.cuda.current_device())
        ctx.want_wave, ctx.want_params = ctx.needs_input_grad[1], (any(ctx.needs_input_grad[3:]) and not detach_params)
        return tuple(bufs)

    @staticmethod
    def backward(ctx, *dbufs):
        disc, layout, where = ctx.disc, ctx.layout, ctx.where
        wave, bufs = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        dev = wave.device
        lib, hd = disc._handle(dev)
        st = disc._flat[ctx.idx]
        flat = st["flat"]
        B, L = wave.shape
        dbufs = [None if g is None else g.to(torch.float32).contiguous() for g in dbufs]
        n = len(layout)
        fm = (ctypes.c_void_p * n)(*[bufs[bi].data_ptr() + 4 * off1 * layout[i][2] for i, (bi, off1) in enumerate(where)])
        dfm = (ctypes.c_void_p * n)(*[(None if dbufs[bi] is None else dbufs[bi].data_ptr() + 4 * off1 * layout[i][2]) for i, (bi, off1) in enumerate(where)])
        flat_mode = disc._flat_grad_mode and "gflat" in st and ctx.want_params
        gflat = (st["gflat"] if (flat_mode and st["gfresh"]) else torch.empty_like(flat)) if ctx.want_params else None
        if ctx.grad_clips and ctx.grad_clips < B:       # every buffer is batch-major: the first clips of the pass ARE a pass (same base pointers)
            if ctx.want_params:
                raise RuntimeError("grad_clips restricts the backward to the leading clips: parameter gradients of the whole pass are not available")
            dwave = torch.zeros_like(wave) if ctx.want_wave else None
            B = ctx.grad_clips
        else:
            dwave = torch.empty_like(wave) if ctx.want_wave else None
        if gflat is None and dwave is None:
            return (None, None, None) + (None,) * (len(st["layout"]))
        with torch.cuda.device(dev):
            _native.check(lib.escx_disc_backward(hd, ctypes.c_void_p(flat.data_ptr()), disc._version(), ctypes.c_void_p(wave.data_ptr()), B, L, fm, dfm,
                                                 None if gflat is None else ctypes.c_void_p(gflat.data_ptr()),
                                                 None if dwave is None else ctypes.c_void_p(dwave.data_ptr()),
                                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        grads = (None,) * len(st["layout"])
        if flat_mode:                           # p.grad are views of st["gflat"]: written (or accumulated) in place
            if st["gfresh"]:
                st["gfresh"] = False
            else:
                st["gflat"].add_(gflat)
        elif gflat is not None:
            params = dict(disc.named_parameters())
            by_id = {id(params[k]): gflat[off:off + n_].view(params[k].shape) for k, off, n_ in st["layout"]}
            grads = tuple(by_id.get(id(p)) for p in disc.parameters())
        return (None, dwave, None) + grads
