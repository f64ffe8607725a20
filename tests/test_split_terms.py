"""The arithmetic claims behind csrc/split_terms.h (DESIGN.md sections 10.1 and 10.9), restated in numpy and checked on the CPU: the device kernels run the same
term splits on the 16-bit matrix cores (GPU side: tests/test_gpu_parity.py test_layer_accuracy_against_fp64 and the oracle sweeps)."""
import numpy as np


def split_bf16(a, n=3):
    out, r = [], a.astype(np.float32).copy()
    for _ in range(n):
        u = r.view(np.uint32).astype(np.uint64)
        t = (((u + ((u >> 16) & 1) + 0x7FFF) >> 16) << 16).astype(np.uint32).view(np.float32)      # round to nearest even
        out.append(t)
        r = (r - t).astype(np.float32)
    return out, r


def x2_scale(w):
    """power of two that brings max |w| into [2^13, 2^14) (split_terms.h x2_scale)"""
    m = float(np.abs(w).max())
    return 1.0 if m == 0.0 else 2.0 ** (13 - int(np.floor(np.log2(m))))


def split_f16(a, scale=1.0):
    r = (a.astype(np.float32) * np.float32(scale)).astype(np.float32)
    a1 = r.astype(np.float16).astype(np.float32)
    a2 = (r - a1).astype(np.float32).astype(np.float16).astype(np.float32)
    return [a1, a2], (r - a1 - a2).astype(np.float32)


def test_three_bf16_terms_are_an_exact_split():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(1 << 20) * np.exp(rng.uniform(-20, 20, 1 << 20))).astype(np.float32)
    terms, rest = split_bf16(a)
    assert not rest.any(), "a - a1 - a2 - a3 must be exactly zero for every float32"
    assert np.array_equal(terms[0].astype(np.float64) + terms[1] + terms[2], a.astype(np.float64))


def test_two_fp16_terms_represent_scaled_weights_to_2_pow_minus_22():
    rng = np.random.default_rng(1)
    for std in (1.0, 0.05, 1e-3):
        w = (rng.standard_normal(1 << 18) * std).astype(np.float32)
        s = x2_scale(w)
        assert 2.0 ** 13 <= float(np.abs(w).max()) * s < 2.0 ** 14
        (a1, a2), rest = split_f16(w, s)
        assert np.isfinite(a1).all() and np.isfinite(a2).all()
        big = np.abs(w) * s >= 2.0 ** -2                # both terms are normal fp16 numbers from here up: the relative bound applies
        assert (np.abs(rest[big]) <= np.abs(w[big]) * s * 2.0 ** -22).all()
        assert (np.abs(rest) <= 2.0 ** -25).all() or (np.abs(rest[~big]) <= 2.0 ** -25).all()      # below: absolute bound of the subnormal low term


def test_truncation_of_the_cross_products_against_float64():
    """3 x bf16 / six products: truncation ~1e-8 of the mean result magnitude; 2 x fp16 / three products with scaled weights: ~1e-7 - both below the 1.6e-7 ... 4.5e-7 of an
    fp32 GEMM's accumulation rounding on the same operands."""
    rng = np.random.default_rng(2)
    for K, wstd in ((48, 0.1), (192, 0.05), (1536, 0.02)):
        x = rng.standard_normal((256, K)).astype(np.float32)
        w = (rng.standard_normal((64, K)) * wstd).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64).T
        mag = np.abs(ref).mean()
        err = lambda y: float(np.sqrt(((y - ref) ** 2).mean()) / mag)
        sgemm = err((x @ w.T).astype(np.float64))
        xs, ws = split_bf16(x)[0], split_bf16(w)[0]
        x6 = sum(xs[i].astype(np.float64) @ ws[j].astype(np.float64).T for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))
        s = x2_scale(w)
        xh, wh = split_f16(x)[0], split_f16(w, s)[0]
        x3 = sum(xh[i].astype(np.float64) @ wh[j].astype(np.float64).T for i, j in ((0, 0), (0, 1), (1, 0))) / s
        assert err(x6) < 2e-8 and err(x3) < 1.5e-7 and err(x3) < sgemm, (K, sgemm, err(x6), err(x3))


def test_power_of_two_scaling_commutes_with_fp32_accumulation():
    """The weights' 2^k rides through the fp32 accumulator and is taken out at the end: every partial sum is the unscaled one times 2^k, so no rounding decision changes."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal(4096).astype(np.float32)
    b = (rng.standard_normal(4096) * 0.03).astype(np.float32)
    s = np.float32(x2_scale(b))
    acc0, acc1 = np.float32(0.37), np.float32(0.37) * s           # the bias rides in the accumulator, scaled
    for u, v in zip(a, b):
        acc0 = np.float32(acc0 + np.float32(u * v))
        acc1 = np.float32(acc1 + np.float32(u * np.float32(v * s)))
    assert np.float32(acc1 / s) == acc0
