"""Wire / disk format of ESC codes.

The reference just `torch.save`s the int64 tensor (scripts/compress.py:35), i.e. 64 bits per 10-bit code.  Each code
indexes a 1024-entry codebook, so the real payload is 10 bits: 6 streams x 3 groups x 50 frames/s x 10 b = 9000 b/s,
which is where "9 kbps" comes from (esc/models/base.py:70).  `pack_codes` produces exactly that payload (plus a
16-byte header), packed on the GPU.
"""
from __future__ import annotations

import ctypes
import struct
from typing import Tuple

import torch

MAGIC = b"ESC1"


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_codes(codes: torch.Tensor, feat_shape: Tuple[int, int]) -> bytes:
    """(B, S, G, T) int64 device tensor -> bytes: header (magic, B, S, G, T as u16, H, W as u16) + 10-bit payload."""
    from . import _native
    if not codes.is_cuda:
        raise RuntimeError("pack_codes expects the codes on the HIP device they were produced on")
    lib = _native.load()
    c = codes.to(torch.int64).contiguous()
    n = c.numel()
    out = torch.empty(5 * ((n + 3) // 4), dtype=torch.uint8, device=c.device)
    with torch.cuda.device(c.device):
        _native.check(lib.escx_codes_pack10(ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, _stream(c.device)))
    B, S, G, T = c.shape
    return MAGIC + struct.pack("<6H", B, S, G, T, int(feat_shape[0]), int(feat_shape[1])) + out.cpu().numpy().tobytes()


def unpack_codes(blob: bytes, device="cuda"):
    """Inverse of pack_codes: -> (codes int64 (B,S,G,T) on `device`, feat_shape)."""
    from . import _native
    if blob[:4] != MAGIC:
        raise ValueError("not an ESC code stream")
    B, S, G, T, H, W = struct.unpack("<6H", blob[4:16])
    n = B * S * G * T
    payload = torch.frombuffer(bytearray(blob[16:16 + 5 * ((n + 3) // 4)]), dtype=torch.uint8).to(device)
    lib = _native.load()
    codes = torch.empty((B, S, G, T), dtype=torch.int64, device=payload.device)
    with torch.cuda.device(payload.device):
        _native.check(lib.escx_codes_unpack10(ctypes.c_void_p(payload.data_ptr()), ctypes.c_void_p(codes.data_ptr()), n, _stream(payload.device)))
    return codes, (H, W)


def payload_bits_per_second(num_streams: int, group_size: int = 3, frames_per_second: float = 50.0, bits: int = 10) -> float:
    return num_streams * group_size * frames_per_second * bits
