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
HEADER_BYTES = 16
BITS = 10          # bits per code on the wire: codebook_size <= 1024


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_codes(codes: torch.Tensor, feat_shape: Tuple[int, int], codebook_size: int = 1024) -> bytes:
    """(B, S, G, T) int64 device tensor -> bytes: header (magic, B, S, G, T as u16, H, W as u16) + 10-bit payload.
    The format carries 10 bits per code: a model with codebook_size > 1024 (or a code outside [0, 1024)) is refused
    instead of being silently masked."""
    from . import _native
    if not codes.is_cuda:
        raise RuntimeError("pack_codes expects the codes on the HIP device they were produced on")
    if codebook_size > (1 << BITS):
        raise ValueError(f"the ESC1 wire format carries {BITS}-bit codes; codebook_size={codebook_size} does not fit")
    if codes.dim() != 4:
        raise ValueError("codes must have shape (B, S, G, T)")
    if max(codes.shape) > 0xFFFF or max(int(feat_shape[0]), int(feat_shape[1])) > 0xFFFF:
        raise ValueError("a header field exceeds 16 bits")
    lib = _native.load()
    c = codes.to(torch.int64).contiguous()
    n = c.numel()
    if n and (int(c.min()) < 0 or int(c.max()) >= (1 << BITS)):
        raise ValueError(f"code index outside [0, {1 << BITS}): the wire format would corrupt it")
    out = torch.empty(5 * ((n + 3) // 4), dtype=torch.uint8, device=c.device)
    with torch.cuda.device(c.device):
        _native.check(lib.escx_codes_pack10(ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, _stream(c.device)))
    B, S, G, T = c.shape
    return MAGIC + struct.pack("<6H", B, S, G, T, int(feat_shape[0]), int(feat_shape[1])) + out.cpu().numpy().tobytes()


def parse_header(blob: bytes):
    """(B, S, G, T, H, W, payload_bytes) of an ESC1 stream; raises ValueError on a truncated or inconsistent blob."""
    if len(blob) < HEADER_BYTES or blob[:4] != MAGIC:
        raise ValueError("not an ESC code stream")
    B, S, G, T, H, W = struct.unpack("<6H", blob[4:HEADER_BYTES])
    if min(B, S, G, T, H, W) == 0:
        raise ValueError(f"corrupt ESC1 header: zero dimension in {(B, S, G, T, H, W)}")
    need = 5 * ((B * S * G * T + 3) // 4)
    if len(blob) < HEADER_BYTES + need:
        raise ValueError(f"truncated ESC1 stream: header announces {need} payload bytes, {len(blob) - HEADER_BYTES} present")
    return B, S, G, T, H, W, need


def unpack_codes(blob: bytes, device="cuda", model=None):
    """Inverse of pack_codes: -> (codes int64 (B,S,G,T) on `device`, feat_shape).  With `model` given, the header is
    checked against the model (group_size, max_streams, overlap) before anything is decoded."""
    from . import _native
    B, S, G, T, H, W, need = parse_header(blob)
    if model is not None:
        c = model.cfg
        if G != c["group_size"] or S > c["max_streams"] or T * c["overlap"] != W or c["codebook_size"] > (1 << BITS):
            raise ValueError(f"ESC1 header {(B, S, G, T, H, W)} does not match the model (group_size {c['group_size']}, "
                             f"max_streams {c['max_streams']}, overlap {c['overlap']})")
    n = B * S * G * T
    payload = torch.frombuffer(bytearray(blob[HEADER_BYTES:HEADER_BYTES + need]), dtype=torch.uint8).to(device)
    lib = _native.load()
    codes = torch.empty((B, S, G, T), dtype=torch.int64, device=payload.device)
    with torch.cuda.device(payload.device):
        _native.check(lib.escx_codes_unpack10(ctypes.c_void_p(payload.data_ptr()), ctypes.c_void_p(codes.data_ptr()), n, _stream(payload.device)))
    return codes, (H, W)


def payload_bits_per_second(num_streams: int, group_size: int = 3, frames_per_second: float = 50.0, bits: int = 10) -> float:
    return num_streams * group_size * frames_per_second * bits
