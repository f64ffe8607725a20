"""esc.ESC -- host side of the MI355X-native ESC codec.

Keeps the reference's Python surface (`/root/reference/esc/models/codecs.py:9-94,183-200`): the constructor
kwargs, `.encode(x, num_streams) -> (codes, feat_shape)`, `.decode(codes, feat_shape) -> wave`, the eval-mode
`.forward(x, x_feat, num_streams)` dict, `max_streams`, `max_bps`, `make_model`, and the reference state_dict
key layout (so reference checkpoints load with `load_state_dict`).  The parameters live here as ordinary
`nn.Parameter`s; the arithmetic lives in libescx.so (hand-written HIP for gfx950) reached through ctypes.
There is NO PyTorch/CPU implementation of the path in this package: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Literal, Optional, Tuple

import torch
import torch.nn as nn

from .. import _native

__all__ = ["ESC", "make_model", "model_dict", "state_manifest"]


def _rel_pos_index(ws: int) -> torch.Tensor:
    """(dh+ws-1)*(2ws-1) + (dw+ws-1); reference attention.py:195-205."""
    ih = torch.arange(ws).repeat_interleave(ws)
    iw = torch.arange(ws).repeat(ws)
    return ((ih[:, None] - ih[None, :] + ws - 1) * (2 * ws - 1) + (iw[:, None] - iw[None, :] + ws - 1)).long()


def state_manifest(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """key -> shape of the reference state_dict for this configuration."""
    h, heads, depth, ws = list(cfg["h_dims"]), list(cfg["swin_heads"]), cfg["swin_depth"], cfg["window_size"]
    n, S, G = len(h), cfg["max_streams"], cfg["group_size"]
    pf, pt = cfg["patch_size"]
    win = int(cfg["win_len"] * cfg["sr"] * 1e-3)
    man: Dict[str, Tuple[int, ...]] = {"ft.window": (win,), "ift.window": (win,)}
    dec = h[::-1]
    H0 = cfg["in_freq"] // pf
    for s in range(S):                                                  # base.py:49-69
        C = dec[max(s - 1, 0)]
        Hq = H0 // 2 ** (S - 1) if s == 0 else H0 // 2 ** (S - s)
        D = cfg["overlap"] * Hq * C
        dims = [D // G] * G
        dims[-1] = D - (D // G) * (G - 1)
        d = cfg["codebook_dims"][s]
        for g in range(G):
            man[f"quantizers.{s}.vqs.{g}.embedding.weight"] = (cfg["codebook_size"], d)
        for g in range(G):
            man[f"quantizers.{s}.down_projs.{g}.weight"] = (d, dims[g])
        for g in range(G):
            man[f"quantizers.{s}.up_projs.{g}.weight"] = (dims[g], d)

    def block(p, C, nH):
        hid = int(C * cfg["mlp_ratio"])
        for j in range(depth):
            q = f"{p}swint_blocks.{j}."
            man[q + "norm1.weight"] = (C,); man[q + "norm1.bias"] = (C,)
            man[q + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nH)
            man[q + "attn.relative_position_index"] = (ws * ws, ws * ws)
            man[q + "attn.qkv.weight"] = (3 * C, C); man[q + "attn.qkv.bias"] = (3 * C,)
            man[q + "attn.proj.weight"] = (C, C); man[q + "attn.proj.bias"] = (C,)
            man[q + "norm2.weight"] = (C,); man[q + "norm2.bias"] = (C,)
            man[q + "mlp.linear_1.weight"] = (hid, C); man[q + "mlp.linear_1.bias"] = (hid,)
            man[q + "mlp.linear_2.weight"] = (C, hid); man[q + "mlp.linear_2.bias"] = (C,)

    for i in range(n - 1):
        p = f"encoder.blocks.{i}."
        block(p, h[i], heads[i])
        man[p + "subsample.norm.weight"] = (2 * h[i],); man[p + "subsample.norm.bias"] = (2 * h[i],)
        man[p + "subsample.down.weight"] = (h[i + 1], 2 * h[i])
    man["encoder.patch_embed.proj.weight"] = (h[0], cfg["in_dim"], pf, pt)
    man["encoder.patch_embed.proj.bias"] = (h[0],)
    man["encoder.patch_embed.norm.weight"] = (h[0],); man["encoder.patch_embed.norm.bias"] = (h[0],)
    block("encoder.pre_nn.", h[0], heads[0])
    rh = heads[::-1]
    for j in range(n - 1):
        p = f"decoder.blocks.{j}."
        block(p, dec[j], rh[j])
        man[p + "subsample.norm.weight"] = (dec[j],); man[p + "subsample.norm.bias"] = (dec[j],)
        man[p + "subsample.up.weight"] = (2 * dec[j + 1], dec[j])
    man["decoder.patch_deembed.de_proj1.weight"] = (h[0] * pf * pt, h[0], 5, 5)
    man["decoder.patch_deembed.de_proj1.bias"] = (h[0] * pf * pt,)
    man["decoder.patch_deembed.de_proj2.weight"] = (cfg["in_dim"], h[0], 3, 3)
    man["decoder.patch_deembed.de_proj2.bias"] = (cfg["in_dim"],)
    block("decoder.post_nn.", dec[-1], rh[-1])
    return man


def _init_tensor(key: str, shape) -> torch.Tensor:
    """Default initialisation with the distributions the reference's modules use."""
    if key.endswith("relative_position_index"):
        return _rel_pos_index(int(math.isqrt(shape[0])))
    if key.endswith(".window"):
        return torch.hann_window(shape[0])
    t = torch.empty(shape)
    if ".norm" in key:
        return t.fill_(1.0) if key.endswith("weight") else t.zero_()
    if key.endswith("relative_position_bias_table"):
        return nn.init.trunc_normal_(t, std=0.02)                       # attention.py:212
    if key.endswith("embedding.weight"):
        return nn.init.kaiming_normal_(t)                               # codebook.py:14
    fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1) if len(shape) > 1 else shape[0]
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    return t.uniform_(-bound, bound)


class _Node(nn.Module):
    """Plain container so that parameters register under the reference's dotted key names."""


def _attach(root: nn.Module, key: str, tensor: torch.Tensor, buffer: bool):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor, persistent=True)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


class ESC(nn.Module):
    """Efficient Speech Codec (cross-scale residual VQ + Swin transformers), MI355X-native execution."""

    def __init__(self, in_dim: int = 2, in_freq: int = 192, h_dims: list = [45, 72, 96, 144, 192, 384],
                 max_streams: int = 6, win_len: int = 20, hop_len: int = 5, sr: int = 16000,
                 patch_size: list = [3, 2], swin_heads: list = [3, 6, 12, 24, 24], swin_depth: int = 2,
                 window_size: int = 4, mlp_ratio: float = 4.,
                 overlap: int = 2, group_size: int = 3,
                 codebook_size: int = 1024, codebook_dims: list = [8, 8, 8, 8, 8, 8],
                 l2norm: bool = True, backbone: Literal['transformer', 'convolution'] = 'transformer',
                 kernel_size: list = [5, 2], conv_depth: int = 1) -> None:
        super().__init__()
        if backbone != "transformer":
            raise NotImplementedError("only backbone='transformer' (csvq+swinT) is implemented; the convolution "
                                      "backbone is an ablation outside the accelerated path")
        self.cfg = dict(in_dim=in_dim, in_freq=in_freq, h_dims=list(h_dims), max_streams=max_streams, win_len=win_len,
                        hop_len=hop_len, sr=sr, patch_size=list(patch_size), swin_heads=list(swin_heads),
                        swin_depth=swin_depth, window_size=window_size, mlp_ratio=float(mlp_ratio), overlap=overlap,
                        group_size=group_size, codebook_size=codebook_size, codebook_dims=list(codebook_dims),
                        l2norm=bool(l2norm), backbone=backbone)
        self.in_freq, self.in_dim = in_freq, in_dim                      # base.py:16-20
        self.max_streams = max_streams
        self.enc_h_dims = list(h_dims)
        self.dec_h_dims = list(h_dims)[::-1]
        self.win_length = int(win_len * sr * 1e-3)                       # base.py:23
        self.hop_length = int(hop_len * sr * 1e-3)                       # base.py:24
        self.max_bps = (2 / overlap) * max_streams * math.log2(codebook_size) * group_size // (20 * patch_size[1] // 2)  # base.py:70
        if codebook_size > 32768:
            raise NotImplementedError("codebook_size > 32768: code indices travel as int16 between ranks (esc/distributed.py)")
        if len(codebook_dims) < max_streams or len(swin_heads) < len(h_dims) - 1:
            raise ValueError("codebook_dims / swin_heads shorter than the number of streams / scales")

        for key, shape in state_manifest(self.cfg).items():
            is_buf = key.endswith("relative_position_index") or key.endswith(".window")
            _attach(self, key, _init_tensor(key, shape), is_buf)

        self._handles: Dict[int, ctypes.c_void_p] = {}
        self._flat: Dict[int, dict] = {}                 # per device: flat fp32 parameter buffer the nn.Parameters are views of
        self._packed_version: Dict[int, int] = {}        # parameter fingerprint the packed device layouts were derived from
        self._dirty = True
        self._flat_grad_mode = False
        self._carry: Dict[int, tuple] = {}               # flat gradient buffers that survive a handle rebuild (see _handle)
        self._precision = None                           # None = the library's default (ESCX_PRECISION, else "f16x2"); see set_precision

    # ---- weight management ------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        own = self.state_dict()
        for k in ("ft.window", "ift.window"):            # torchaudio buffers: optional in incoming checkpoints
            if k not in sd:
                sd[k] = own[k]
        for k in own:                                    # the index buffer is regenerated if absent
            if k.endswith("relative_position_index") and k not in sd:
                sd[k] = own[k]
        self._dirty = True
        return super().load_state_dict(sd, strict=strict, **kw)

    # ---- arithmetic of the inference path (include/escx.h escx_set_precision) -----------------------------
    def set_precision(self, mode: str) -> "ESC":
        """Operand form of the K = C contractions (MLPs, Q / K / V, PatchMerge / PatchSplit, PatchDeEmbed) of encode / decode / eval forward:
        "fp32" (fp32 MFMA, the reference's operand precision), "bf16x3" (every fp32 operand split exactly into three bf16 terms: fp32 results up to
        summation order), "f16x2" (two fp16 terms, range-safe, truncation ~1e-7: the default and the fastest).  fp32 accumulation in every mode.
        Not part of the reference's surface - the reference has one arithmetic (ATen fp32)."""
        if mode not in _native.PRECISIONS:
            raise ValueError(f"precision {mode!r}: expected one of {sorted(_native.PRECISIONS)}")
        self._precision = mode
        lib = _native.load()
        for hd in self._handles.values():
            _native.check(lib.escx_set_precision(hd, _native.PRECISIONS[mode]))
        return self

    @property
    def precision(self) -> str:
        """The mode in effect (of the first live handle; before any handle exists: what set_precision chose, or the library default)."""
        if self._handles:
            code = _native.load().escx_get_precision(next(iter(self._handles.values())))
            return {v: k for k, v in _native.PRECISIONS.items()}[code]
        import os
        return self._precision or {"0": "fp32", "3": "bf16x3", "2": "f16x2"}.get(os.environ.get("ESCX_PRECISION", "f16x2"), os.environ.get("ESCX_PRECISION", "f16x2"))

    def refresh_weights(self):
        """Force a re-pack (in-place updates are detected through autograd's version counters; this is for exotic writes, e.g. through
        raw pointers, that bypass them)."""
        self._packed_version = {}

    def _c_config(self) -> _native.EscxConfig:
        c = self.cfg
        cc = _native.EscxConfig()
        cc.in_dim, cc.in_freq, cc.n_scales = c["in_dim"], c["in_freq"], len(c["h_dims"])
        for i, v in enumerate(c["h_dims"]):
            cc.h_dims[i] = v
        cc.max_streams = c["max_streams"]
        cc.win_length, cc.hop_length = self.win_length, self.hop_length
        cc.patch_f, cc.patch_t = c["patch_size"]
        for i, v in enumerate(c["swin_heads"][:_native.MAX_SCALES]):
            cc.swin_heads[i] = v
        cc.swin_depth, cc.window_size, cc.mlp_ratio = c["swin_depth"], c["window_size"], c["mlp_ratio"]
        cc.overlap, cc.group_size, cc.codebook_size = c["overlap"], c["group_size"], c["codebook_size"]
        for i, v in enumerate(c["codebook_dims"][:_native.MAX_SCALES]):
            cc.codebook_dims[i] = v
        cc.l2norm = int(c["l2norm"])
        return cc

    # ---- flat parameter buffer (training step) ------------------------------------------------
    def _param_version(self) -> int:
        """Changes whenever a parameter is modified in place (optimizer.step, param.data.copy_, ...): autograd's version counters."""
        return sum(p._version for p in self.parameters())

    def _named_params(self) -> Dict[str, nn.Parameter]:
        return dict(self.named_parameters())

    def _ensure_flat(self, device: torch.device, lib, hd) -> torch.Tensor:
        """One contiguous fp32 device buffer holding every trainable parameter in the library's canonical order; the nn.Parameters
        become views into it (so optimizer steps update it in place and the library re-derives its packed layouts on the device
        without any host copy).  Re-established lazily after .to() / external re-assignment of a parameter's storage."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._flat.get(idx)
        params = self._named_params()
        if st is None:
            n = lib.escx_flat_param_count(hd)
            layout = [(lib.escx_flat_param_key(hd, i).decode(), int(lib.escx_flat_param_offset(hd, i)), int(lib.escx_flat_param_numel(hd, i)))
                      for i in range(n)]
            flat = torch.zeros(int(lib.escx_flat_param_total(hd)), dtype=torch.float32, device=device)
            st = {"flat": flat, "layout": layout}
            self._flat[idx] = st
        flat = st["flat"]
        base = flat.data_ptr()
        with torch.no_grad():
            for key, off, n in st["layout"]:
                p = params[key]
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    flat[off:off + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                    p.data = flat[off:off + n].view(p.shape)
        return flat

    # ---- flat-gradient mode (esc.optim.FlatAdamW) ----------------------------------------------------
    def enable_flat_grads(self, device):
        """Every p.grad becomes a view of ONE flat gradient buffer that the training backward fills directly (accumulating across
        backward calls until zero_flat_grads), so that the optimiser is two kernels over flat buffers.  autograd's per-parameter
        accumulation (and therefore per-parameter hooks such as DDP's) is bypassed in this mode."""
        device = torch.device(device)
        lib, hd = self._handle(device, for_training=True)
        flat = self._ensure_flat(device, lib, hd)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._flat[idx]
        if "gflat" not in st:
            old = self._carry.pop(idx, None)             # a rebuild (model.to(), load_state_dict) keeps pending gradients: same layout, same buffer
            if old is not None and old[0].shape == flat.shape and old[0].device == flat.device:
                st["gflat"], st["gfresh"] = old
            else:
                st["gflat"], st["gfresh"] = torch.zeros_like(flat), True
        params = self._named_params()
        for key, off, n in st["layout"]:
            params[key].grad = st["gflat"][off:off + n].view(params[key].shape)
        self._flat_grad_mode = True

    def flat_buffers(self, device):
        device = torch.device(device)
        lib, hd = self._handle(device, for_training=True)
        flat = self._ensure_flat(device, lib, hd)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if "gflat" not in self._flat[idx]:
            self.enable_flat_grads(device)
        return flat, self._flat[idx]["gflat"]

    def zero_flat_grads(self, device):
        idx = torch.device(device).index
        idx = idx if idx is not None else torch.cuda.current_device()
        st = self._flat.get(idx)
        if st is not None and "gflat" in st:
            st["gfresh"] = True                      # the next backward overwrites instead of accumulating: no memset needed

    def note_params_updated(self):
        """The flat buffer was written by a native kernel (no autograd version bump): mark every packed layout stale."""
        self._packed_version = {}

    def _handle(self, device: torch.device, for_training: bool = False):
        lib = _native.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._dirty:
            for hd in self._handles.values():
                lib.escx_destroy(hd)
            # flat-gradient mode outlives the rebuild: the gradient buffer (and with it every p.grad view and whatever a backward has already
            # accumulated) is carried over, so that a live FlatAdamW keeps stepping on real gradients
            self._carry = {i: (st["gflat"], st["gfresh"]) for i, st in self._flat.items() if "gflat" in st}
            self._handles, self._flat, self._packed_version = {}, {}, {}
            self._dirty = False
        if idx not in self._handles:
            hd = ctypes.c_void_p()
            cc = self._c_config()
            _native.check(lib.escx_create(ctypes.byref(cc), idx, ctypes.byref(hd)))
            try:
                for key, t in self.state_dict().items():
                    if not t.is_floating_point():
                        continue
                    host = t.detach().to("cpu", torch.float32).contiguous()
                    shape = (ctypes.c_int64 * max(host.dim(), 1))(*host.shape)
                    _native.check(lib.escx_set_param(hd, key.encode(), host.data_ptr(), shape, host.dim()))
                _native.check(lib.escx_finalize_params(hd))
                if self._precision is not None:
                    _native.check(lib.escx_set_precision(hd, _native.PRECISIONS[self._precision]))
            except Exception:
                lib.escx_destroy(hd)
                raise
            self._handles[idx] = hd
            self._packed_version[idx] = self._param_version()
            if self._flat_grad_mode and idx in self._carry:
                self._flat_grad_mode = False             # (guards the re-entry through enable_flat_grads -> _handle)
                self.enable_flat_grads(device)
        hd = self._handles[idx]
        if not for_training and self._packed_version.get(idx) != self._param_version():
            # parameters were updated in place since the last pack (optimizer steps, manual edits): rebuild every derived layout,
            # including the host-folded ones of the inference path, from the current values
            flat = self._ensure_flat(device, lib, hd)
            with torch.cuda.device(device):
                _native.check(lib.escx_load_flat_params(hd, ctypes.c_void_p(flat.data_ptr()), 1, self._stream(device)))
            self._packed_version[idx] = self._param_version()
        return lib, hd

    def __getstate__(self):
        """copy.deepcopy / torch.save of the whole module: native handles are per-process device resources and are rebuilt lazily."""
        state = self.__dict__.copy()
        state["_handles"], state["_flat"], state["_packed_version"], state["_dirty"], state["_carry"] = {}, {}, {}, True, {}
        state["_flat_grad_mode"] = False
        return state

    def __del__(self):
        try:
            lib = _native.load()
            for hd in self._handles.values():
                lib.escx_destroy(hd)
        except Exception:
            pass

    @staticmethod
    def _need_gpu(t: torch.Tensor, what: str):
        if not t.is_cuda:
            raise RuntimeError(f"esc.ESC (MI355X build): {what} must live on a HIP device (got {t.device}); "
                               "this package has no CPU implementation of the encode/decode path")

    @staticmethod
    def _stream(device) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def reserve(self, batch: int, n_samples: int, device=None):
        """Pre-size the workspace (otherwise done on first use; resizing synchronises the device)."""
        device = torch.device(device) if device is not None else next(self.parameters()).device
        lib, hd = self._handle(device)
        _native.check(lib.escx_reserve(hd, batch, n_samples))

    def latent_shape(self, n_samples: int) -> Tuple[int, int]:
        pf, pt = self.cfg["patch_size"]
        T = 1 + n_samples // self.hop_length
        H = self.in_freq // pf
        for _ in range(len(self.enc_h_dims) - 1):
            H = (H + 1) // 2
        return H, T // pt

    # ---- public API (codecs.py:48-94) ---------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, num_streams: int = 6):
        """(Bs, L) waveform -> codes (Bs, num_streams, group_size, W/overlap) int64, feat_shape (H, W)."""
        if x.dim() != 2:
            raise ValueError("x must have shape (Bs, L)")
        self._need_gpu(x, "x")
        x = x.to(torch.float32).contiguous()
        lib, hd = self._handle(x.device)
        B, L = x.shape
        _, W = self.latent_shape(L)
        if W % self.cfg["overlap"] != 0:
            raise AssertionError("Time dimension must be multiple of overlap")       # quantization.py:407
        codes = torch.empty((B, int(num_streams), self.cfg["group_size"], W // self.cfg["overlap"]), dtype=torch.int64,
                            device=x.device)
        if B == 0:                                      # empty batch: the reference returns empty codes and the latent shape
            return codes, self.latent_shape(L)
        fh, fw = ctypes.c_int(), ctypes.c_int()
        with torch.cuda.device(x.device):
            _native.check(lib.escx_encode(hd, x.data_ptr(), B, L, int(num_streams), codes.data_ptr(), ctypes.byref(fh),
                                          ctypes.byref(fw), self._stream(x.device)))
        return codes, (fh.value, fw.value)

    @torch.no_grad()
    def decode(self, codes: torch.Tensor, feat_shape=(2, 1000), return_feat: bool = False):
        """codes (Bs, num_streams, group_size, *) + feat_shape (H, W) -> waveform (Bs, hop*(2W-1))."""
        if codes.dim() != 4:
            raise ValueError("codes must have shape (Bs, num_streams, group_size, T)")
        self._need_gpu(codes, "codes")
        codes = codes.to(torch.int64).contiguous()
        lib, hd = self._handle(codes.device)
        B, S, G, Tq = codes.shape
        H, W = int(feat_shape[0]), int(feat_shape[1])
        if G != self.cfg["group_size"] or Tq * self.cfg["overlap"] != W:
            raise ValueError(f"codes shape {tuple(codes.shape)} does not match feat_shape {(H, W)}")
        pt = self.cfg["patch_size"][1]
        out = torch.empty((B, self.hop_length * (pt * W - 1)), dtype=torch.float32, device=codes.device)
        feat = torch.empty((B, pt * W, self.in_dim, self.in_freq), dtype=torch.float32, device=codes.device) if return_feat else None
        if B == 0:
            return (out, feat.permute(0, 2, 3, 1)) if return_feat else out
        with torch.cuda.device(codes.device):
            _native.check(lib.escx_decode(hd, codes.data_ptr(), B, S, H, W, out.data_ptr(),
                                          feat.data_ptr() if return_feat else None, self._stream(codes.device)))
        if return_feat:
            return out, feat.permute(0, 2, 3, 1)
        return out

    def forward_one_step(self, x, x_feat=None, num_streams=6, freeze_codebook=False):
        if not self.training and freeze_codebook:
            raise ValueError("``freeze_vq`` must be set False during inference")       # quantization.py:43-44
        if self.training:
            return self._forward_train(x, x_feat, int(num_streams), bool(freeze_codebook))
        if x_feat is not None:
            return self._forward_from_feat(x, x_feat, int(num_streams))
        if x.dim() != 2:
            raise ValueError("x must have shape (Bs, L)")
        self._need_gpu(x, "x")
        xin = x.to(torch.float32).contiguous()
        lib, hd = self._handle(xin.device)
        B, L = xin.shape
        S = int(num_streams)
        _, W = self.latent_shape(L)
        if W % self.cfg["overlap"] != 0:
            raise AssertionError("Time dimension must be multiple of overlap")
        pt = self.cfg["patch_size"][1]
        T = 1 + L // self.hop_length
        dev = xin.device
        codes = torch.empty((B, S, self.cfg["group_size"], W // self.cfg["overlap"]), dtype=torch.int64, device=dev)
        recon = torch.empty((B, self.hop_length * (pt * W - 1)), dtype=torch.float32, device=dev)
        raw_feat = torch.empty((B, T, self.in_dim, self.in_freq), dtype=torch.float32, device=dev)
        recon_feat = torch.empty((B, pt * W, self.in_dim, self.in_freq), dtype=torch.float32, device=dev)
        cm = torch.empty((B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if B > 0:                                   # an empty batch returns the (empty) buffers as they are
                _native.check(lib.escx_forward(hd, xin.data_ptr(), B, L, S, codes.data_ptr(), recon.data_ptr(), raw_feat.data_ptr(),
                                               recon_feat.data_ptr(), cm.data_ptr(), self._stream(dev)))
        return {"cm_loss": cm, "cb_loss": cm.clone(), "raw_audio": x, "recon_audio": recon,
                "raw_feat": raw_feat.permute(0, 2, 3, 1), "recon_feat": recon_feat.permute(0, 2, 3, 1), "codes": codes}

    def _forward_from_feat(self, x, x_feat, S):
        """forward(x, x_feat=...) of the reference (codecs.py:33-34): x_feat is the complex STFT as a real tensor laid out
        (Bs, F, T, 2) - what `rearrange(x_feat, "b h w c -> b c h w")` expects, despite the docstring - and replaces
        spec_transform(x).  The permutation to the library's frame-major layout is plain data movement."""
        if x_feat.dim() != 4 or x_feat.shape[1] != self.in_freq or x_feat.shape[3] != self.in_dim:
            raise ValueError(f"x_feat must have shape (Bs, {self.in_freq}, T, {self.in_dim})")
        self._need_gpu(x_feat, "x_feat")
        feat = x_feat.to(torch.float32).permute(0, 2, 3, 1).contiguous()            # (B, T, 2, F)
        B, T = feat.shape[0], feat.shape[1]
        pt = self.cfg["patch_size"][1]
        W = T // pt
        if W < 1 or W % self.cfg["overlap"] != 0:
            raise AssertionError("Time dimension must be multiple of overlap")
        dev = feat.device
        lib, hd = self._handle(dev)
        codes = torch.empty((B, S, self.cfg["group_size"], W // self.cfg["overlap"]), dtype=torch.int64, device=dev)
        recon = torch.empty((B, self.hop_length * (pt * W - 1)), dtype=torch.float32, device=dev)
        recon_feat = torch.empty((B, pt * W, self.in_dim, self.in_freq), dtype=torch.float32, device=dev)
        cm = torch.empty((B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(lib.escx_forward_feat(hd, feat.data_ptr(), B, T, S, codes.data_ptr(), recon.data_ptr(), recon_feat.data_ptr(),
                                                cm.data_ptr(), self._stream(dev)))
        return {"cm_loss": cm, "cb_loss": cm.clone(), "raw_audio": x, "recon_audio": recon,
                "raw_feat": x_feat.permute(0, 3, 1, 2), "recon_feat": recon_feat.permute(0, 2, 3, 1), "codes": codes}

    def _forward_train(self, x, x_feat, S, freeze):
        """Training-mode forward (codecs.py:30-46 with csrvq.py:97-129, codebook.py:57-75, quantization.py:53-64): differentiable w.r.t.
        every parameter through `_TrainStep`; `codes` holds all max_streams streams (every quantiser runs in training mode)."""
        if x.dim() != 2:
            raise ValueError("x must have shape (Bs, L)")
        if x_feat is not None and (x_feat.dim() != 4 or x_feat.shape[1] != self.in_freq or x_feat.shape[3] != self.in_dim or x_feat.shape[0] != x.shape[0]):
            raise ValueError(f"x_feat must have shape (Bs, {self.in_freq}, T, {self.in_dim})")
        self._need_gpu(x, "x")
        feat = None
        if x_feat is not None:          # codecs.py:33-34: the given spectrum replaces the STFT of x; layout (Bs, F, T, 2) as in eval mode (_forward_from_feat)
            self._need_gpu(x_feat, "x_feat")
            feat = x_feat.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()      # frame-major (Bs, T, 2, F): the library's spectrum layout
        params = tuple(self.parameters())
        recon, recon_fm, raw_fm, cm, cb, codes = _TrainStep.apply(self, x.to(torch.float32).contiguous(), feat, S, freeze, *params)
        return {"cm_loss": cm, "cb_loss": cb, "raw_audio": x, "recon_audio": recon,
                "raw_feat": raw_fm.permute(0, 2, 3, 1), "recon_feat": recon_fm.permute(0, 2, 3, 1), "codes": codes}

    def forward(self, x, x_feat, num_streams, freeze_codebook=False):
        """codecs.py:48-66: dict with cm_loss, cb_loss, raw_audio, recon_audio, raw_feat (Bs,2,F,T), recon_feat (Bs,2,F,T'), codes.
        Eval mode: inference path, no autograd graph.  Training mode: differentiable (HIP forward + hand-written HIP backward)."""
        num_streams = self.max_streams if freeze_codebook else num_streams
        if self.training:
            return self.forward_one_step(x, x_feat, num_streams, freeze_codebook)
        with torch.no_grad():
            return self.forward_one_step(x, x_feat, num_streams, freeze_codebook)


class _TrainStep(torch.autograd.Function):
    """One training-mode pass through libescx: forward keeps its activations on the handle's tape, backward consumes them and returns
    d loss / d parameter for every nn.Parameter (views of one flat gradient buffer, the library's canonical order)."""

    @staticmethod
    def forward(ctx, model, x, feat, S, freeze, *params):
        dev = x.device
        lib, hd = model._handle(dev, for_training=True)
        flat = model._ensure_flat(dev, lib, hd)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        model._packed_version[idx] = None               # the training refresh skips the host-folded inference layouts
        c = model.cfg
        B, L = x.shape
        if feat is not None:            # the spectrum decides the geometry (codecs.py:33-34): T frames <-> hop * (T - 1) samples
            L = model.hop_length * (feat.shape[1] - 1)
        _, W = model.latent_shape(L)
        if W % c["overlap"] != 0:
            raise AssertionError("Time dimension must be multiple of overlap")       # quantization.py:407
        pt = c["patch_size"][1]
        T = 1 + L // model.hop_length
        codes = torch.empty((B, c["max_streams"], c["group_size"], W // c["overlap"]), dtype=torch.int64, device=dev)
        recon = torch.empty((B, model.hop_length * (pt * W - 1)), dtype=torch.float32, device=dev)
        raw_fm = feat if feat is not None else torch.empty((B, T, model.in_dim, model.in_freq), dtype=torch.float32, device=dev)
        recon_fm = torch.empty((B, pt * W, model.in_dim, model.in_freq), dtype=torch.float32, device=dev)
        cm = torch.empty((B,), dtype=torch.float32, device=dev)
        cb = torch.empty((B,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if feat is not None:
                _native.check(lib.escx_train_forward_feat(hd, ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(feat.data_ptr()), B, T, int(S), int(bool(freeze)),
                                                          ctypes.c_void_p(codes.data_ptr()), ctypes.c_void_p(recon.data_ptr()), ctypes.c_void_p(recon_fm.data_ptr()),
                                                          ctypes.c_void_p(cm.data_ptr()), ctypes.c_void_p(cb.data_ptr()), model._stream(dev)))
            else:
                _native.check(lib.escx_train_forward(hd, ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(x.data_ptr()), B, L, int(S), int(bool(freeze)),
                                                 ctypes.c_void_p(codes.data_ptr()), ctypes.c_void_p(recon.data_ptr()), ctypes.c_void_p(raw_fm.data_ptr()),
                                                     ctypes.c_void_p(recon_fm.data_ptr()), ctypes.c_void_p(cm.data_ptr()), ctypes.c_void_p(cb.data_ptr()),
                                                     model._stream(dev)))
        ctx.model, ctx.dev, ctx.idx = model, dev, idx
        ctx.tape_generation = int(lib.escx_train_tape_generation(hd))
        ctx.mark_non_differentiable(raw_fm, codes)
        return recon, recon_fm, raw_fm, cm, cb, codes

    @staticmethod
    def backward(ctx, d_recon, d_recon_fm, _d_raw, d_cm, d_cb, _d_codes):
        model, dev = ctx.model, ctx.dev
        lib, hd = model._handle(dev, for_training=True)
        if int(lib.escx_train_tape_generation(hd)) != ctx.tape_generation:
            raise RuntimeError("esc.ESC backward: the model ran another training forward after the one this graph belongs to; the library keeps ONE "
                               "activation tape per model (backward before the next forward, or use a second model instance)")
        st = model._flat[ctx.idx]
        flat_mode = model._flat_grad_mode and "gflat" in st
        gflat = st["gflat"] if (flat_mode and st["gfresh"]) else torch.empty_like(st["flat"])

        def prep(t):
            return None if t is None else t.to(torch.float32).contiguous()
        d_recon, d_recon_fm, d_cm, d_cb = prep(d_recon), prep(d_recon_fm), prep(d_cm), prep(d_cb)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            _native.check(lib.escx_train_backward(hd, p(d_recon), p(d_recon_fm), p(d_cm), p(d_cb), ctypes.c_void_p(gflat.data_ptr()), model._stream(dev)))
        params = model._named_params()
        if flat_mode:                                   # p.grad are views of st["gflat"]: written in place, nothing for autograd to accumulate
            if st["gfresh"]:
                st["gfresh"] = False
            else:
                st["gflat"].add_(gflat)
            return (None, None, None, None, None) + (None,) * len(params)
        by_id = {id(params[k]): gflat[off:off + n].view(params[k].shape) for k, off, n in st["layout"]}
        grads = tuple(by_id.get(id(q)) for q in model.parameters())
        return (None, None, None, None, None) + grads


model_dict = {"csvq+swinT": ESC}


def make_model(model_config, model_name: str = "csvq+swinT"):
    """codecs.py:190-200.  `model_name` defaults to the ESC codec so that scripts/compress.py:22 (which passes only the
    config) works."""
    if model_name not in model_dict:
        raise NotImplementedError(f"{model_name} is not available in the MI355X build (only csvq+swinT); "
                                  "rvq+* and *+conv are ablation models outside the accelerated path")
    m = model_dict[model_name]
    if isinstance(model_config, dict):
        return m(**model_config)
    return m(**vars(model_config))
