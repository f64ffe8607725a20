"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU (PyTorch fp32, eager ATen) restatement of the reference's encode/decode hot path, written as plain
functions over a `state_dict` (reference key names) and the `model:` config block.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product path
(`efficient-speech-codec_amd/esc`) never does and fails loudly if its HIP library is missing.

Pinning: `tests/test_oracle_golden.py` checks this file against golden vectors produced by the real
reference (imported through oracle/ref_shims.py by oracle/gen_golden.py in the build container):
integer codes must match exactly and audio to <=1e-6 RMS, for ESC-Base and ESC-Large, num_streams 1..6,
plus edge shapes and a tiny config with per-layer activations.  The STFT stage of both the shimmed
reference and this oracle is `torch.stft/istft`, not the pinned torchaudio 2.0.0 wheel (not installed,
not vendored): PARITY UNPINNED at the torchaudio boundary only.

Every function cites the reference lines it restates (paths relative to /root/reference/).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

DEFAULT_CFG = dict(  # esc/models/codecs.py:11-18
    in_dim=2, in_freq=192, h_dims=[45, 72, 96, 144, 192, 384], max_streams=6, win_len=20, hop_len=5,
    sr=16000, patch_size=[3, 2], swin_heads=[3, 6, 12, 24, 24], swin_depth=2, window_size=4,
    mlp_ratio=4.0, overlap=2, group_size=3, codebook_size=1024, codebook_dims=[8] * 6, l2norm=True,
    backbone="transformer", kernel_size=[5, 2], conv_depth=1)


def full_config(cfg: dict) -> dict:
    out = dict(DEFAULT_CFG)
    out.update(cfg)
    if out["backbone"] != "transformer":
        raise ValueError("only the swin-transformer backbone is on the hot path")
    return out


# ----------------------------------------------------------------------------------------------
# STFT front / back end -- esc/models/base.py:22-47
# ----------------------------------------------------------------------------------------------
def stft_params(cfg: dict) -> Tuple[int, int, int]:
    n_fft = (cfg["in_freq"] - 1) * 2                       # base.py:22
    win = int(cfg["win_len"] * cfg["sr"] * 1e-3)           # base.py:23
    hop = int(cfg["hop_len"] * cfg["sr"] * 1e-3)           # base.py:24
    return n_fft, win, hop


def spec_transform(x: Tensor, cfg: dict, window: Optional[Tensor] = None) -> Tensor:
    """(B, L) -> (B, 2, F, T).  base.py:29-37; torchaudio Spectrogram(power=None) == torch.stft."""
    n_fft, win, hop = stft_params(cfg)
    if window is None:
        window = torch.hann_window(win, dtype=x.dtype)
    spec = torch.stft(x, n_fft, hop, win, window, center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    return torch.view_as_real(spec).permute(0, 3, 1, 2)


def audio_reconstruct(feat: Tensor, cfg: dict, window: Optional[Tensor] = None) -> Tensor:
    """(B, 2, F, T) -> (B, hop*(T-1)).  base.py:39-47."""
    n_fft, win, hop = stft_params(cfg)
    if window is None:
        window = torch.hann_window(win, dtype=feat.dtype)
    spec = torch.view_as_complex(feat.permute(0, 2, 3, 1).contiguous())
    return torch.istft(spec, n_fft, hop, win, window, center=True, normalized=False, onesided=True,
                       return_complex=False)


# ----------------------------------------------------------------------------------------------
# Patchify / scale changes -- esc/modules/transformer/scale.py
# ----------------------------------------------------------------------------------------------
def patch_embed(x: Tensor, sd: Dict[str, Tensor], pfx: str, patch) -> Tensor:
    """(B,2,F,T) -> (B, H*W, C): strided conv, row-major (h w) tokens, LayerNorm.  scale.py:42-50."""
    y = F.conv2d(x, sd[pfx + "proj.weight"], sd[pfx + "proj.bias"], stride=tuple(patch))
    B, C, H, W = y.shape
    y = y.permute(0, 2, 3, 1).reshape(B, H * W, C)
    return F.layer_norm(y, (C,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5)


def patch_merge(x: Tensor, H: int, sd, pfx: str) -> Tensor:
    """(B,H*W,C) -> (B,ceil(H/2)*W,C'): zero row if H odd, out[h',w,s*C+c]=x[2h'+s,w,c], LN, Linear.
    scale.py:97-115, 7-14."""
    B, L, C = x.shape
    W = L // H
    m = x.view(B, H, W, C)
    if H % 2:
        m = F.pad(m, (0, 0, 0, 0, 0, 1))
    H2 = m.shape[1] // 2
    m = m.view(B, H2, 2, W, C).permute(0, 1, 3, 2, 4).reshape(B, H2 * W, 2 * C)
    m = F.layer_norm(m, (2 * C,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5)
    return F.linear(m, sd[pfx + "down.weight"])


def patch_split(x: Tensor, H: int, sd, pfx: str) -> Tensor:
    """(B,H*W,C) -> (B,2H*W,C'): LN, Linear(C->2C'), out[2h+s,w,c]=y[h,w,s*C'+c].  scale.py:131-145, 16-23."""
    B, L, C = x.shape
    W = L // H
    y = F.layer_norm(x, (C,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5)
    y = F.linear(y, sd[pfx + "up.weight"])
    C2 = y.shape[-1] // 2
    y = y.view(B, H, W, 2, C2).permute(0, 1, 3, 2, 4).reshape(B, 2 * H * W, C2)
    return y


def patch_deembed(x: Tensor, H: int, sd, pfx: str, patch) -> Tensor:
    """(B,H*W,C) -> (B,2,F,T).  conv5x5 -> pixel shuffle (s1,s2) -> conv3x3.  scale.py:73-81."""
    B, L, C = x.shape
    W = L // H
    s1, s2 = patch
    y = x.view(B, H, W, C).permute(0, 3, 1, 2)
    y = F.conv2d(y, sd[pfx + "de_proj1.weight"], sd[pfx + "de_proj1.bias"], padding=2)
    y = y.permute(0, 2, 3, 1)                                   # B H W (s1 s2 C)
    y = y.reshape(B, H, W, s1, s2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s1, W * s2, C)
    y = F.conv2d(y.permute(0, 3, 1, 2), sd[pfx + "de_proj2.weight"], sd[pfx + "de_proj2.bias"], padding=1)
    return y


# ----------------------------------------------------------------------------------------------
# Swin block -- esc/modules/transformer/attention.py
# ----------------------------------------------------------------------------------------------
def window_plan(H: int, W: int, ws: int, shift: int):
    """Index plan for one (H, W, shift): for every slot of every ws x ws window of the padded, rolled map,
    the source token (or -1 for a zero pad token), plus the additive mask of attention.py:56-75.

    Restates pad-after-norm (attention.py:139-143), roll(-s,-s) over the *padded* map (:146-148),
    window order (hWin, wWin) and in-window order (ih, iw) (:246-250).
    """
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    hh = torch.arange(Hp).view(Hp // ws, 1, ws, 1)
    ww = torch.arange(Wp).view(1, Wp // ws, 1, ws)
    src_h = (hh + shift) % Hp                # rolled[i] = padded[(i + shift) mod Hp]
    src_w = (ww + shift) % Wp
    src_h, src_w = torch.broadcast_tensors(src_h, src_w)
    valid = (src_h < H) & (src_w < W)
    src = torch.where(valid, src_h * W + src_w, torch.full_like(src_h, -1)).reshape(-1, ws * ws)
    mask = None
    if shift > 0:
        def region(n, Np):                   # labels of attention.py:61-70 along one axis
            r = torch.zeros(Np, dtype=torch.long)
            r[Np - ws:Np - shift] = 1
            r[Np - shift:] = 2
            return r[n]
        lab = (3 * region(torch.arange(Hp), Hp).view(Hp // ws, 1, ws, 1)
               + region(torch.arange(Wp), Wp).view(1, Wp // ws, 1, ws))
        lab = lab.expand(Hp // ws, Wp // ws, ws, ws).reshape(-1, ws * ws)
        mask = (lab[:, :, None] != lab[:, None, :]).float() * -100.0
    return src, mask


def swin_block(x: Tensor, H: int, W: int, sd, pfx: str, heads: int, ws: int, shift: int) -> Tensor:
    """One W-MSA / SW-MSA block + MLP.  attention.py:129-178 (block), 215-244 (attention), 267-272 (MLP)."""
    B, L, C = x.shape
    assert L == H * W, "input feature has wrong size"          # attention.py:132
    hd = C // heads
    N = ws * ws
    src, mask = window_plan(H, W, ws, shift)
    nW = src.shape[0]

    xn = F.layer_norm(x, (C,), sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"], 1e-5)
    xz = torch.cat([xn, xn.new_zeros(B, 1, C)], dim=1)          # slot L == the zero pad token
    gather = torch.where(src < 0, torch.full_like(src, L), src).reshape(-1)
    win = xz[:, gather].reshape(B * nW, N, C)

    qkv = F.linear(win, sd[pfx + "attn.qkv.weight"], sd[pfx + "attn.qkv.bias"])
    qkv = qkv.view(B * nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4).contiguous()
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]             # attention.py:225
    attn = q @ k.transpose(-2, -1)
    table = sd[pfx + "attn.relative_position_bias_table"]
    index = sd[pfx + "attn.relative_position_index"].reshape(-1)
    bias = table[index].view(N, N, heads).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        attn = (attn.view(B, nW, heads, N, N) + mask.view(1, nW, 1, N, N)).view(B * nW, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B * nW, N, C)
    out = F.linear(out, sd[pfx + "attn.proj.weight"], sd[pfx + "attn.proj.bias"])

    # window reverse + un-roll + crop == scatter every valid slot back to its source token
    out = out.view(B, nW * N, C)
    keep = (src.reshape(-1) >= 0).nonzero().squeeze(1)
    y = torch.empty_like(x)
    y[:, src.reshape(-1)[keep]] = out[:, keep]
    x = x + y                                                   # attention.py:176
    h = F.layer_norm(x, (C,), sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"], 1e-5)
    h = F.linear(h, sd[pfx + "mlp.linear_1.weight"], sd[pfx + "mlp.linear_1.bias"])
    h = F.gelu(h)                                               # exact erf GELU (nn.GELU default)
    h = F.linear(h, sd[pfx + "mlp.linear_2.weight"], sd[pfx + "mlp.linear_2.bias"])
    return x + h                                                # attention.py:177


def transformer_layer(x: Tensor, H: int, W: int, sd, pfx: str, heads: int, depth: int, ws: int,
                      scale: Optional[str]) -> Tuple[Tensor, int, int]:
    """depth x SwinBlock (shift 0, ws//2, 0, ...) then PatchMerge / PatchSplit.  attention.py:48-91."""
    for j in range(depth):
        x = swin_block(x, H, W, sd, f"{pfx}swint_blocks.{j}.", heads, ws, 0 if j % 2 == 0 else ws // 2)
    if scale == "down":
        return patch_merge(x, H, sd, pfx + "subsample."), (H + 1) // 2, W
    if scale == "up":
        return patch_split(x, H, sd, pfx + "subsample."), H * 2, W
    return x, H, W


# ----------------------------------------------------------------------------------------------
# Product VQ -- esc/modules/vq/quantization.py:7-136,380-432 and codebook.py:20-55
# ----------------------------------------------------------------------------------------------
def split_dimension(total: int, num: int) -> List[int]:
    """quantization.py:380-386."""
    base = total // num
    dims = [base] * num
    dims[-1] = total - base * (num - 1)
    return dims


def pvq_frames(z: Tensor, in_freq: int, overlap: int) -> Tensor:
    """(B, H*W, C) -> (B, W/overlap, overlap*C*H), channel-major (c h) flattening.  quantization.py:388-410."""
    B, L, C = z.shape
    W = L // in_freq
    assert W % overlap == 0, "Time dimension must be multiple of overlap"
    v = z.view(B, in_freq, W, C).permute(0, 2, 3, 1).reshape(B, W, C * in_freq)
    return v.reshape(B, W // overlap, overlap * C * in_freq)


def pvq_unframes(v: Tensor, in_freq: int, overlap: int) -> Tensor:
    """inverse of pvq_frames.  quantization.py:412-432."""
    B, Wo, D = v.shape
    fix = D // overlap
    C = fix // in_freq
    z = v.reshape(B, Wo * overlap, C, in_freq).permute(0, 3, 1, 2)
    return z.reshape(B, in_freq * Wo * overlap, C)


def codebook_search(z: Tensor, cb: Tensor, l2norm: bool = True, want_margin: bool = False):
    """(N, d) x (K, d) -> (N,) int64.  codebook.py:20-43 -- same expression, evaluated left to right."""
    if l2norm:
        cb = F.normalize(cb, dim=-1)
        z = F.normalize(z, dim=-1)
    dist = z.pow(2).sum(1, keepdim=True) - (2 * z) @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
    idx = dist.min(1).indices
    if want_margin:
        top2 = dist.topk(2, dim=1, largest=False).values
        return idx, top2[:, 1] - top2[:, 0]
    return idx


def pvq_encode(z: Tensor, sd, pfx: str, in_freq: int, overlap: int, groups: int, l2norm: bool,
               margins: Optional[list] = None, z_e_out: Optional[list] = None) -> Tensor:
    """(B, H*W, C) -> codes (B, groups, W/overlap).  quantization.py:74-91,110-122."""
    v = pvq_frames(z, in_freq, overlap)
    B, T, D = v.shape
    dims = split_dimension(D, groups)
    codes, s = [], 0
    for g in range(groups):
        ze = F.linear(v[..., s:s + dims[g]], sd[f"{pfx}down_projs.{g}.weight"])
        if z_e_out is not None:
            z_e_out.append(ze)
        r = codebook_search(ze.reshape(B * T, -1), sd[f"{pfx}vqs.{g}.embedding.weight"], l2norm,
                            want_margin=margins is not None)
        if margins is not None:
            margins.append(r[1].view(B, T))
            r = r[0]
        codes.append(r.view(B, T))
        s += dims[g]
    return torch.stack(codes, dim=1)


def pvq_decode(codes: Tensor, sd, pfx: str, in_freq: int, overlap: int) -> Tensor:
    """codes (B, groups, T) -> (B, H*W, C): raw (un-normalised) codebook rows, up-projection, un-frame.
    quantization.py:93-108,124-136; codebook.py:45-55."""
    parts = []
    for g in range(codes.shape[1]):
        zq = F.embedding(codes[:, g], sd[f"{pfx}vqs.{g}.embedding.weight"])
        parts.append(F.linear(zq, sd[f"{pfx}up_projs.{g}.weight"]))
    return pvq_unframes(torch.cat(parts, dim=-1), in_freq, overlap)


# ----------------------------------------------------------------------------------------------
# Whole path -- esc/models/base.py:143-158 (Encoder), csrvq.py:97-183, codecs.py:30-94
# ----------------------------------------------------------------------------------------------
@dataclass
class Trace:
    """Optional per-layer record used by the tiny-config fixture and by kernel-level GPU tests."""
    enc_hs: List[Tensor] = field(default_factory=list)
    dec_hs: List[Tensor] = field(default_factory=list)
    margins: List[Tensor] = field(default_factory=list)     # one (B, groups, T) per stream
    feat: Optional[Tensor] = None
    recon_feat: Optional[Tensor] = None


class EscOracle:
    def __init__(self, cfg: dict, state_dict: Dict[str, Tensor], keep_graph: bool = False):
        self.cfg = full_config(cfg)
        # keep_graph: the tensors are used as they are (leaf tensors with requires_grad for the training-step restatement)
        self.sd = dict(state_dict) if keep_graph else \
            {k: (v.detach().to(torch.float32) if v.is_floating_point() else v.detach()) for k, v in state_dict.items()}
        c = self.cfg
        self.S = c["max_streams"]
        self.H0 = c["in_freq"] // c["patch_size"][0]
        self.enc_dims = list(c["h_dims"])
        self.dec_dims = list(c["h_dims"])[::-1]
        self.enc_heads = list(c["swin_heads"])
        self.dec_heads = list(c["swin_heads"])[::-1]          # codecs.py:26
        # base.py:49-69: stream 0 and 1 both sit at the bottom scale
        nb = len(self.dec_dims) - 1
        self.q_freq = [self.H0 // 2 ** (self.S - 1)] + [self.H0 // 2 ** (self.S - i) for i in range(1, self.S)]
        assert nb == self.S - 1 or True
        self.max_bps = (2 / c["overlap"]) * self.S * math.log2(c["codebook_size"]) * c["group_size"] \
            // (20 * c["patch_size"][1] // 2)                  # base.py:70

    # -- encoder: base.py:143-158
    def encoder(self, feat: Tensor):
        c = self.cfg
        H, W = feat.shape[2] // c["patch_size"][0], feat.shape[3] // c["patch_size"][1]
        x = patch_embed(feat, self.sd, "encoder.patch_embed.", c["patch_size"])
        x, H, W = transformer_layer(x, H, W, self.sd, "encoder.pre_nn.", self.enc_heads[0], c["swin_depth"],
                                    c["window_size"], None)
        hs = [x]
        for i in range(len(self.enc_dims) - 1):
            x, H, W = transformer_layer(x, H, W, self.sd, f"encoder.blocks.{i}.", self.enc_heads[i],
                                        c["swin_depth"], c["window_size"], "down")
            hs.append(x)
        return hs, (H, W)

    def _q(self, i):
        c = self.cfg
        return dict(sd=self.sd, pfx=f"quantizers.{i}.", in_freq=self.q_freq[i], overlap=c["overlap"])

    def _dec_block(self, i, x, H, W):
        c = self.cfg
        return transformer_layer(x, H, W, self.sd, f"decoder.blocks.{i}.", self.dec_heads[i], c["swin_depth"],
                                 c["window_size"], "up")

    # -- csrvq.py:131-158
    def csvq_encode(self, enc_hs, num_streams: int, feat_shape, trace: Optional[Trace] = None, force: Optional[Tensor] = None) -> Tensor:
        """`force` (B, S, G, T) int64, -1 = free: codes imposed on the search (test infrastructure: when the device path resolved a reference
        near-tie the other way, the oracle is CONTINUED from the device's choice so that the later streams - which see a different residual -
        can still be compared code for code; tests/gpu_util.attribute_with_continuation)."""
        c = self.cfg
        H, W = feat_shape
        mg = [] if trace is not None else None

        def enc(i, z):
            m = [] if mg is not None else None
            code = pvq_encode(z, groups=c["group_size"], l2norm=c["l2norm"], margins=m, **self._q(i))
            if mg is not None:
                mg.append(torch.stack(m, dim=1))
            if force is not None:
                f = force[:, i]
                code = torch.where(f >= 0, f, code)
            return code

        codes = [enc(0, enc_hs[-1])]
        if num_streams > 1:
            dec = pvq_decode(codes[0], **self._q(0))
            for i in range(num_streams - 1):
                codes.append(enc(i + 1, enc_hs[-1 - i] - dec))              # csrvq.py:15-17,50-54
                if len(codes) == num_streams:
                    break
                dec = pvq_decode(codes[-1], **self._q(i + 1)) + dec          # csrvq.py:19-21,56-60
                dec, H, W = self._dec_block(i, dec, H, W)
        if trace is not None:
            trace.margins = mg
        return torch.stack(codes, dim=1)

    # -- csrvq.py:160-183
    def csvq_decode(self, codes: Tensor, feat_shape, trace: Optional[Trace] = None) -> Tensor:
        c = self.cfg
        H, W = feat_shape
        S = codes.shape[1]
        dec = pvq_decode(codes[:, 0], **self._q(0))
        if trace is not None:
            trace.dec_hs.append(dec)
        for i in range(len(self.dec_dims) - 1):
            if i < S - 1:
                dec = pvq_decode(codes[:, i + 1], **self._q(i + 1)) + dec
            dec, H, W = self._dec_block(i, dec, H, W)
            if trace is not None:
                trace.dec_hs.append(dec)
        dec, H, W = transformer_layer(dec, H, W, self.sd, "decoder.post_nn.", self.dec_heads[-1],
                                      c["swin_depth"], c["window_size"], None)
        if trace is not None:
            trace.dec_hs.append(dec)
        return patch_deembed(dec, H, self.sd, "decoder.patch_deembed.", c["patch_size"])

    # -- codecs.py:68-94
    @torch.no_grad()
    def encode(self, x: Tensor, num_streams: int = 6, trace: Optional[Trace] = None, force: Optional[Tensor] = None):
        feat = spec_transform(x, self.cfg, self.sd.get("ft.window"))
        enc_hs, shape = self.encoder(feat)
        if trace is not None:
            trace.feat, trace.enc_hs = feat, enc_hs
        return self.csvq_encode(enc_hs, num_streams, shape, trace, force), shape

    @torch.no_grad()
    def decode(self, codes: Tensor, feat_shape=(2, 1000), trace: Optional[Trace] = None) -> Tensor:
        feat = self.csvq_decode(codes, feat_shape, trace)
        if trace is not None:
            trace.recon_feat = feat
        return audio_reconstruct(feat, self.cfg, self.sd.get("ift.window"))

    # -- codecs.py:30-66 + csrvq.py:97-129 in eval mode
    @torch.no_grad()
    def forward_eval(self, x: Tensor, x_feat: Optional[Tensor], num_streams: int) -> dict:
        c = self.cfg
        feat = spec_transform(x, c, self.sd.get("ft.window")) if x_feat is None else x_feat.permute(0, 3, 1, 2)
        enc_hs, (H, W) = self.encoder(feat)
        B = feat.shape[0]
        cm = torch.zeros(B)
        codes = []
        dec = None
        for s in range(self.S):
            transmit = s < num_streams                    # stream 0 always; csrvq.py:108-113
            if s >= 1:
                blk = s - 1
            if transmit:
                resid = enc_hs[-1] if s == 0 else enc_hs[-1 - (s - 1)] - dec
                zes: list = []
                code = pvq_encode(resid, groups=c["group_size"], l2norm=c["l2norm"], z_e_out=zes, **self._q(s))
                loss = 0.0
                for g in range(c["group_size"]):          # codebook.py:72-73, quantization.py:71-72
                    zq = F.embedding(code[:, g], self.sd[f"quantizers.{s}.vqs.{g}.embedding.weight"])
                    loss = loss + F.mse_loss(zq, zes[g], reduction="none").mean([1, 2])
                cm = cm + loss / c["group_size"]
                codes.append(code)
                zq_map = pvq_decode(code, **self._q(s))
                dec = zq_map if s == 0 else zq_map + dec
            if s >= 1:
                dec, H, W = self._dec_block(blk, dec, H, W)
        # the loop above runs block (s-1) after stream s; csrvq.py:108-122 -- one block per stream i>=1
        dec, H, W = transformer_layer(dec, H, W, self.sd, "decoder.post_nn.", self.dec_heads[-1],
                                      c["swin_depth"], c["window_size"], None)
        recon_feat = patch_deembed(dec, H, self.sd, "decoder.patch_deembed.", c["patch_size"])
        return {"cm_loss": cm, "cb_loss": cm.clone(), "raw_audio": x,
                "recon_audio": audio_reconstruct(recon_feat, c, self.sd.get("ift.window")),
                "raw_feat": feat, "recon_feat": recon_feat, "codes": torch.stack(codes, dim=1)}


    # -- training mode: codecs.py:30-66, csrvq.py:23-48,97-129, quantization.py:31-72, codebook.py:57-75 (differentiable: plain autograd)
    def forward_train(self, x: Tensor, num_streams: int, freeze_codebook: bool = False) -> dict:
        c = self.cfg
        S = self.S if freeze_codebook else num_streams                      # codecs.py:65
        feat = spec_transform(x, c, self.sd.get("ft.window"))
        enc_hs, (H, W) = self.encoder(feat)

        def csrvq(enc, dec, sid, transmit):                                 # csrvq.py:23-48 with self.training == True
            resid = enc - dec
            zq, code, cb, cm = pvq_forward_train(resid, groups=c["group_size"], l2norm=c["l2norm"], freeze_vq=freeze_codebook, **self._q(sid))
            if not transmit:                                                # masking non-transmitted streams (:42-44)
                cm, cb = cm * 0.0, cb * 0.0
                zq = zq * 0.0
            return zq + dec, cm, cb, code

        dec, cm_loss, cb_loss, code = csrvq(enc_hs[-1], 0.0, 0, True)       # csrvq.py:104-105
        codes = [code]
        for i in range(len(self.dec_dims) - 1):
            dec, cm_i, cb_i, code_i = csrvq(enc_hs[-1 - i], dec, i + 1, i < S - 1)
            cm_loss, cb_loss = cm_loss + cm_i, cb_loss + cb_i
            codes.append(code_i)
            dec, H, W = self._dec_block(i, dec, H, W)
        dec, H, W = transformer_layer(dec, H, W, self.sd, "decoder.post_nn.", self.dec_heads[-1], c["swin_depth"], c["window_size"], None)
        recon_feat = patch_deembed(dec, H, self.sd, "decoder.patch_deembed.", c["patch_size"])
        return {"cm_loss": cm_loss, "cb_loss": cb_loss, "raw_audio": x,
                "recon_audio": audio_reconstruct(recon_feat, c, self.sd.get("ift.window")),
                "raw_feat": feat, "recon_feat": recon_feat, "codes": torch.stack(codes, dim=1)}


def pvq_forward_train(z: Tensor, sd, pfx: str, in_freq: int, overlap: int, groups: int, l2norm: bool, freeze_vq: bool):
    """ProductVectorQuantize.forward in training mode (quantization.py:31-72) with Codebook.forward (codebook.py:57-75):
    straight-through estimator, commitment / codebook losses, frozen-codebook pass-through.  Returns (z_q map, codes, cb_loss, cm_loss)."""
    v = pvq_frames(z, in_freq, overlap)
    B, T, D = v.shape
    dims = split_dimension(D, groups)
    outs, codes, cb_loss, cm_loss, s = [], [], 0.0, 0.0, 0
    for g in range(groups):
        ze = F.linear(v[..., s:s + dims[g]], sd[f"{pfx}down_projs.{g}.weight"])
        cb_w = sd[f"{pfx}vqs.{g}.embedding.weight"]
        code = codebook_search(ze.reshape(B * T, -1), cb_w, l2norm).view(B, T)
        zq = F.embedding(code, cb_w)
        cm = F.mse_loss(zq.detach(), ze, reduction="none").mean([1, 2])            # codebook.py:68
        cb = F.mse_loss(zq, ze.detach(), reduction="none").mean([1, 2])            # codebook.py:69
        zq = ze + (zq - ze).detach()                                               # codebook.py:70
        if freeze_vq:                                                              # quantization.py:56-59
            zq = zq * 0.0 + ze
            cb, cm = cb * 0.0, cm * 0.0
        outs.append(F.linear(zq, sd[f"{pfx}up_projs.{g}.weight"]))
        codes.append(code)
        cm_loss, cb_loss = cm_loss + cm, cb_loss + cb
        s += dims[g]
    return pvq_unframes(torch.cat(outs, dim=-1), in_freq, overlap), torch.stack(codes, dim=1), cb_loss / groups, cm_loss / groups


# ----------------------------------------------------------------------------------------------
# Training losses -- esc/modules/loss/generator_loss.py:12-74
# ----------------------------------------------------------------------------------------------
MEL_WINDOWS = [32, 64, 128, 256, 512, 1024, 2048]
MEL_BINS = [5, 10, 20, 40, 80, 160, 320]


def power_law(stft: Tensor, power: float = 0.3, eps: float = 1e-10) -> Tensor:
    """generator_loss.py:31-35."""
    return (torch.abs(stft) + eps) ** power * torch.sign(stft)


def complex_stft_loss(raw_feat: Tensor, recon_feat: Tensor) -> Tensor:
    """generator_loss.py:19-29 with power_law=True, weight 1: (B,2,F,T) x (B,2,F,T) -> (B,)."""
    return F.mse_loss(power_law(raw_feat), power_law(recon_feat), reduction="none").mean([1, 2, 3])


def htk_filterbank(n_freqs: int, n_mels: int, sample_rate: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk") (documented formula)."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_hi = 2595.0 * math.log10(1.0 + (sample_rate / 2) / 700.0)
    pts = 700.0 * (10.0 ** (torch.linspace(0.0, m_hi, n_mels + 2, dtype=torch.float64) / 2595.0) - 1.0)
    diff = pts[1:] - pts[:-1]
    slopes = pts.unsqueeze(0) - freqs.unsqueeze(1)
    down, up = -slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).float()


def mel_spectrogram(x: Tensor, n_fft: int, n_mels: int, sample_rate: int = 16000) -> Tensor:
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, win_length=n_fft, hop_length=n_fft//4, n_mels, power=1): (B,L)->(B,n_mels,T)."""
    spec = torch.stft(x, n_fft, n_fft // 4, n_fft, torch.hann_window(n_fft, dtype=x.dtype), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs()
    return torch.matmul(spec.transpose(-1, -2), htk_filterbank(n_fft // 2 + 1, n_mels, sample_rate).to(spec.dtype)).transpose(-1, -2)


def mel_spectrogram_loss(raw_audio: Tensor, recon_audio: Tensor, clamp_eps: float = 1e-5) -> Tensor:
    """generator_loss.py:56-74, weight 1: (B,L) x (B,L) -> (B,)."""
    loss = 0.0
    for w, nm in zip(MEL_WINDOWS, MEL_BINS):
        xm, ym = mel_spectrogram(raw_audio, w, nm), mel_spectrogram(recon_audio, w, nm)
        loss = loss + F.l1_loss(xm, ym, reduction="none").mean([1, 2])
        loss = loss + F.l1_loss(xm.clamp(clamp_eps).pow(2).log10(), ym.clamp(clamp_eps).pow(2).log10(), reduction="none").mean([1, 2])
    return loss


LOSS_WEIGHTS = dict(cm_weight=0.25, cb_weight=1.0, mel_weight=0.25, stft_weight=1.0)      # configs/9kbps_esc_base.yaml:29-33


def training_loss(out: dict, w: dict = LOSS_WEIGHTS) -> dict:
    """scripts/trainer_no_adv.py:105-115: the four losses, their weighted sum, and the scalar that is back-propagated."""
    mel = mel_spectrogram_loss(out["raw_audio"], out["recon_audio"])
    stft = complex_stft_loss(out["raw_feat"], out["recon_feat"])
    total = out["cm_loss"] * w["cm_weight"] + out["cb_loss"] * w["cb_weight"] + mel * w["mel_weight"] + stft * w["stft_weight"]
    return {"mel_loss": mel, "stft_loss": stft, "loss": total, "scalar": total.mean()}


# ----------------------------------------------------------------------------------------------
# Adversarial training: DAC discriminator (esc/models/discriminator.py:31-221) and GAN losses (esc/modules/loss/gan_loss.py:5-51)
# ----------------------------------------------------------------------------------------------
DISC_BANDS = [(0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)]
DISC_DEFAULT = dict(sample_rate=16000, rates=[], periods=[2, 3, 5, 7, 11], fft_sizes=[2048, 1024, 512], bands=DISC_BANDS)


def wn_weight(sd, pfx: str) -> Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, the norm over everything but the output channel."""
    v, g = sd[pfx + "weight_v"], sd[pfx + "weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def wn_conv2d(x: Tensor, sd, pfx: str, stride, padding, act: bool = True) -> Tensor:
    """WNConv2d (discriminator.py:23-28): weight-normalised conv, LeakyReLU(0.1) unless act=False."""
    y = F.conv2d(x, wn_weight(sd, pfx), sd[pfx + "bias"], stride=stride, padding=padding)
    return F.leaky_relu(y, 0.1) if act else y


def disc_preprocess(y: Tensor) -> Tensor:
    """discriminator.py:211-216: remove the DC offset, peak-normalise to 0.8."""
    y = y - y.mean(dim=-1, keepdim=True)
    return 0.8 * y / (y.abs().max(dim=-1, keepdim=True)[0] + 1e-9)


def mpd_forward(x: Tensor, sd, pfx: str, period: int) -> List[Tensor]:
    """MPD.forward (discriminator.py:31-66): x (B,1,L) -> list of 6 feature maps."""
    t = x.shape[-1]
    x = F.pad(x, (0, period - t % period), mode="reflect")
    x = x.view(x.shape[0], 1, -1, period)
    fmap = []
    for i, st in enumerate([(3, 1)] * 4 + [(1, 1)]):
        x = wn_conv2d(x, sd, f"{pfx}convs.{i}.0.", st, (2, 0))
        fmap.append(x)
    fmap.append(wn_conv2d(x, sd, f"{pfx}conv_post.", (1, 1), (1, 0), act=False))
    return fmap


def matched_stride_stft(x: Tensor, window_length: int) -> Tensor:
    """audiotools AudioSignal.stft with STFTParams(window_length, hop = window_length // 4, match_stride=True) (discriminator.py:129-133),
    restated from the audiotools source (not installed here: UNPINNED at this boundary): reflect-pad by ((wl - hop) / 2, (wl - hop) / 2 +
    right_pad), right_pad = ceil(L / hop) * hop - L, torch.stft(center=True, hann), drop the first and last two frames.  (B,1,L) -> (B,2,T,F)."""
    hop = window_length // 4
    L = x.shape[-1]
    right = math.ceil(L / hop) * hop - L
    pad = (window_length - hop) // 2
    xp = F.pad(x, (pad, pad + right), "reflect")
    s = torch.stft(xp.reshape(-1, xp.shape[-1]), n_fft=window_length, hop_length=hop, window=torch.hann_window(window_length, dtype=x.dtype),
                   return_complex=True, center=True)[..., 2:-2]
    return torch.view_as_real(s).permute(0, 3, 2, 1)          # (B, F, T, 2) -> (B, 2, T, F)   ["b 1 f t c -> (b 1) c t f"]


def mrd_forward(x: Tensor, sd, pfx: str, window_length: int, bands=DISC_BANDS) -> List[Tensor]:
    """MRD.forward (discriminator.py:105-176): 5 band stacks of 5 convs, concatenated along frequency, conv_post."""
    spec = matched_stride_stft(x, window_length)
    n_fft = window_length // 2 + 1
    fmap, outs = [], []
    for bi, (lo, hi) in enumerate(bands):
        band = spec[..., int(lo * n_fft):int(hi * n_fft)]
        for j, (st, k) in enumerate([((1, 1), 9), ((1, 2), 9), ((1, 2), 9), ((1, 2), 9), ((1, 1), 3)]):
            band = wn_conv2d(band, sd, f"{pfx}band_convs.{bi}.{j}.0.", st, (1, k // 2))
            fmap.append(band)
        outs.append(band)
    fmap.append(wn_conv2d(torch.cat(outs, dim=-1), sd, f"{pfx}conv_post.", (1, 1), (1, 1), act=False))
    return fmap


def discriminator_forward(x: Tensor, sd, cfg: dict = DISC_DEFAULT) -> List[List[Tensor]]:
    """Discriminator.forward (discriminator.py:218-221): x (B,1,L) -> one list of feature maps per sub-discriminator."""
    assert not cfg.get("rates"), "MSD (rates) is not used by the ESC configurations"
    x = disc_preprocess(x)
    out = [mpd_forward(x, sd, f"discriminators.{i}.", p) for i, p in enumerate(cfg["periods"])]
    n = len(cfg["periods"])
    out += [mrd_forward(x, sd, f"discriminators.{n + i}.", w, cfg.get("bands", DISC_BANDS)) for i, w in enumerate(cfg["fft_sizes"])]
    return out


def gan_discriminator_loss(fake: Tensor, real: Tensor, sd, cfg: dict = DISC_DEFAULT) -> Tensor:
    """GANLoss.discriminator_loss (gan_loss.py:30-37): least-squares GAN, per clip."""
    d_fake, d_real = discriminator_forward(fake.detach().unsqueeze(1), sd, cfg), discriminator_forward(real.unsqueeze(1), sd, cfg)
    loss = 0
    for xf, xr in zip(d_fake, d_real):
        loss = loss + torch.mean(xf[-1] ** 2, dim=[1, 2, 3]) + torch.mean((1 - xr[-1]) ** 2, dim=[1, 2, 3])
    return loss


def gan_generator_loss(fake: Tensor, real: Tensor, sd, cfg: dict = DISC_DEFAULT):
    """GANLoss.generator_loss (gan_loss.py:39-51): adversarial term + L1 feature matching, per clip."""
    d_fake, d_real = discriminator_forward(fake.unsqueeze(1), sd, cfg), discriminator_forward(real.unsqueeze(1), sd, cfg)
    loss_g = 0
    for xf in d_fake:
        loss_g = loss_g + torch.mean((1 - xf[-1]) ** 2, dim=[1, 2, 3])
    loss_f = 0
    for df, dr in zip(d_fake, d_real):
        for a, b in zip(df[:-1], dr[:-1]):
            loss_f = loss_f + F.l1_loss(a, b.detach(), reduction="none").mean([1, 2, 3])
    return loss_g, loss_f
