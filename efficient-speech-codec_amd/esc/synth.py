"""Deterministic synthetic weights and audio for ESC.

There are no trained checkpoints in this environment (the reference ships them on Google Drive,
/root/reference/README.md:61-68), so parity and throughput are measured on *name-keyed* deterministic
weights: every state_dict tensor is filled from an integer hash of its key and element index.  The
fill uses only integer arithmetic and float scaling (no libm transcendental), so it reproduces
bit-for-bit on any machine and is independent of module construction order.

Both the golden-vector generator (oracle/gen_golden.py, which drives the *reference* model) and the
product path (bench.py, tests) fill their state_dicts from this one function.
"""
import zlib
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hashed_uniform(key: str, n: int, salt: int = 0) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (key, salt, index)."""
    seed = np.uint64(zlib.crc32(key.encode("utf-8")) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
        h = _splitmix64(idx)
    # 53 random bits -> [0,1) -> [-1,1)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return 2.0 * u - 1.0


def _fan_in(shape) -> int:
    if len(shape) == 2:
        return int(shape[1])
    if len(shape) == 4:
        return int(shape[1] * shape[2] * shape[3])
    return int(shape[0])


def relative_position_index(window: int = 4) -> np.ndarray:
    """(Δh+ws-1)*(2ws-1) + (Δw+ws-1) -- /root/reference/esc/modules/transformer/attention.py:195-205."""
    ih, iw = np.meshgrid(np.arange(window), np.arange(window), indexing="ij")
    ch, cw = ih.reshape(-1), iw.reshape(-1)
    dh = ch[:, None] - ch[None, :] + window - 1
    dw = cw[:, None] - cw[None, :] + window - 1
    return (dh * (2 * window - 1) + dw).astype(np.int64)


def hann_periodic(n: int) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    # 0.5 - 0.5*cos(2*pi*k/n), in float64 then rounded once to f32 by the caller
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def synth_tensor(key: str, shape, dtype=np.float32) -> np.ndarray:
    """Deterministic value for one state_dict entry, chosen by the key's suffix."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if key.endswith("relative_position_index"):
        return relative_position_index(int(round(np.sqrt(shape[0]))))
    if key.endswith(".window"):
        return hann_periodic(shape[0]).astype(np.float32)
    u = hashed_uniform(key, n)
    leaf = key.rsplit(".", 2)
    is_norm = ".norm" in key
    if is_norm and key.endswith(".weight"):
        v = 1.0 + 0.1 * u
    elif is_norm and key.endswith(".bias"):
        v = 0.05 * u
    elif key.endswith("relative_position_bias_table"):
        v = 0.2 * u
    elif key.endswith("embedding.weight"):
        # kaiming_normal_ on (K, d): std = sqrt(2/d); uniform with the same variance
        v = np.sqrt(6.0 / shape[1]) * u
    elif key.endswith(".bias"):
        # nn.Linear / nn.Conv2d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in is not
        # recoverable from the bias shape alone, use the output width as a stand-in
        v = u / np.sqrt(max(shape[0], 1))
    else:
        v = u / np.sqrt(max(_fan_in(shape), 1))
    del leaf
    return v.reshape(shape).astype(dtype)


def synth_state_dict(manifest: dict) -> dict:
    """manifest: key -> shape.  Returns key -> numpy array."""
    return {k: synth_tensor(k, shp) for k, shp in manifest.items()}


def cluster_codebook(key: str, emb: np.ndarray) -> np.ndarray:
    """Stress form of one codebook (SURVEY hard part 1: trained codebooks may have tighter margins than random ones):
    the second half of the rows become NEAR-DUPLICATES of the first half - row K/2 + k = row k + rel_k * |row k| * v_k with v_k a hashed unit
    vector and rel_k cycling log-uniformly through 1e-4 ... 1e-6 (eight steps) - so that for most input vectors the best and the second-best
    code sit at relative distance 1e-4 ... 1e-6 of each other, i.e. hundreds of argmin margins per clip fall below 1e-5.  Pure function of the key."""
    K, d = emb.shape
    half = K // 2
    out = emb.astype(np.float64).copy()
    v = hashed_uniform(key + ":near-duplicate", half * d).reshape(half, d)
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
    rel = 10.0 ** (-4.0 - 2.0 * ((np.arange(half) % 8) / 7.0))
    out[half:2 * half] = out[:half] + rel[:, None] * np.linalg.norm(out[:half], axis=1, keepdims=True) * v
    return out.astype(np.float32)


def clustered_state_dict(manifest: dict) -> dict:
    """synth_state_dict with every codebook replaced by its clustered form (tests/golden/clustered.npz, oracle/gen_clustered_golden.py)."""
    sd = synth_state_dict(manifest)
    for k in sd:
        if k.endswith("embedding.weight"):
            sd[k] = cluster_codebook(k, sd[k])
    return sd


def noise_clip_int16(tag: str, n_samples: int, amp: float = 0.1) -> np.ndarray:
    """Approximately Gaussian noise (sum of 4 hashed uniforms), quantised to int16 PCM."""
    s = sum(hashed_uniform(tag, n_samples, salt=i) for i in range(4)) * (np.sqrt(3.0) / 2.0)
    x = np.clip(amp * s, -1.0, 1.0 - 1.0 / 32768)
    return np.round(x * 32768.0).astype(np.int16)


def voiced_clip_int16(tag: str, n_samples: int, sr: int = 16000) -> np.ndarray:
    """Harmonic 'voiced' signal with a gliding pitch and syllabic envelope plus a little noise."""
    t = np.arange(n_samples, dtype=np.float64) / sr
    f0 = 120.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n_samples)
    for h in range(1, 25):
        x += np.sin(h * phase + 0.3 * h * h) / (h ** 1.2) * np.exp(-((h * 130.0 - 900.0) / 1800.0) ** 2)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 3.1 * t + 0.4)
    x = 0.25 * x * env / np.max(np.abs(x)) + 0.004 * hashed_uniform(tag, n_samples, salt=9)
    return np.round(np.clip(x, -1, 1 - 1 / 32768) * 32768.0).astype(np.int16)


def pcm_to_float(x_int16: np.ndarray) -> np.ndarray:
    return (x_int16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
