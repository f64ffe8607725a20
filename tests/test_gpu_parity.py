"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the committed golden vectors.

Tolerances: integer codes bit-exact; audio <= 1e-4 RMS (north_star); intermediate fp32 activations are compared
relative to their own RMS at 2e-5 (pure fp32 re-association noise)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import NEAR_TIE, PRECISIONS, attribute_with_continuation, build_models, code_report, mismatch_summary, rel_rms, rms, unattributable
from conftest import load_golden, synth_state
from esc import synth, _native

pytestmark = pytest.mark.gpu

ACT_TOL = 2e-5
AUDIO_TOL = 1e-4


@pytest.fixture(scope="module")
def tiny():
    return build_models("tiny")


@pytest.fixture(scope="module")
def base():
    return build_models("base")


def _h(model):
    return model._handle(torch.device("cuda:0"))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------- stage level (tiny + base) -------------------
@pytest.mark.parametrize("L", [1280, 1200, 700])
def test_spec_transform_and_inverse(tiny, L):
    from oracle import esc_oracle as O
    model, orc, g, cfg = tiny
    lib, hd = _h(model)
    x = torch.from_numpy(synth.pcm_to_float(synth.noise_clip_int16(f"stft-{L}", 3 * L))).view(3, L)
    ref = O.spec_transform(x, orc.cfg)                          # (B,2,F,T)
    B, _, F, T = ref.shape
    out = torch.empty(B, T, 2, F, device="cuda")
    xg = x.cuda()
    _native.check(lib.escx_spec_transform(hd, _ptr(xg), B, L, _ptr(out), None))
    got = out.permute(0, 2, 3, 1).cpu()
    assert rel_rms(got, ref) < ACT_TOL
    # inverse on the reference spectrum (drop the odd frame like the codec does)
    T2 = (T // 2) * 2
    ref_wave = O.audio_reconstruct(ref[..., :T2].contiguous(), orc.cfg)
    spec_in = ref[..., :T2].permute(0, 3, 1, 2).contiguous().cuda()
    wave = torch.empty(B, ref_wave.shape[1], device="cuda")
    _native.check(lib.escx_audio_reconstruct(hd, _ptr(spec_in), B, T2, _ptr(wave), None))
    assert rel_rms(wave.cpu(), ref_wave) < ACT_TOL


def test_spec_transform_base_size(base):
    from oracle import esc_oracle as O
    model, orc, g, cfg = base
    lib, hd = _h(model)
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"]))
    ref = O.spec_transform(x, orc.cfg)
    B, _, F, T = ref.shape
    out = torch.empty(B, T, 2, F, device="cuda")
    xg = x.cuda()
    _native.check(lib.escx_spec_transform(hd, _ptr(xg), B, x.shape[1], _ptr(out), None))
    assert rel_rms(out.permute(0, 2, 3, 1).cpu(), ref) < ACT_TOL


@pytest.mark.parametrize("which", ["tiny", "base"])
def test_patch_embed_and_deembed(which, request):
    from oracle import esc_oracle as O
    model, orc, g, cfg = request.getfixturevalue(which)
    lib, hd = _h(model)
    torch.manual_seed(3)
    F = cfg["in_freq"]; T = 61 if which == "tiny" else 121
    feat = torch.randn(2, 2, F, T) * 0.5
    ref = O.patch_embed(feat, orc.sd, "encoder.patch_embed.", cfg["patch_size"])
    spec = feat.permute(0, 3, 1, 2).contiguous().cuda()
    tok = torch.empty(ref.shape, device="cuda")
    _native.check(lib.escx_patch_embed(hd, _ptr(spec), 2, T, _ptr(tok), None))
    assert rel_rms(tok.cpu(), ref) < ACT_TOL
    # de-embed: tokens -> spectrum
    H0, W = F // cfg["patch_size"][0], T // 2
    x = torch.randn(2, H0 * W, cfg["h_dims"][0])
    ref2 = O.patch_deembed(x, H0, orc.sd, "decoder.patch_deembed.", cfg["patch_size"])   # (B,2,F,2W)
    out = torch.empty(2, 2 * W, 2, F, device="cuda")
    xg = x.cuda()
    _native.check(lib.escx_patch_deembed(hd, _ptr(xg), 2, W, _ptr(out), None))
    assert rel_rms(out.permute(0, 2, 3, 1).cpu(), ref2) < ACT_TOL


def _layer_cases(cfg):
    """(layer_id, prefix, C, heads, scale, natural H)"""
    h, heads = cfg["h_dims"], cfg["swin_heads"]
    n = len(h)
    H0 = cfg["in_freq"] // cfg["patch_size"][0]
    cases = [(0, "encoder.pre_nn.", h[0], heads[0], None, H0)]
    H = H0
    for i in range(n - 1):
        cases.append((1 + i, f"encoder.blocks.{i}.", h[i], heads[i], "down", H)); H = (H + 1) // 2
    dh, rh = h[::-1], heads[::-1]
    for j in range(n - 1):
        cases.append((n + j, f"decoder.blocks.{j}.", dh[j], rh[j], "up", H)); H *= 2
    cases.append((2 * n - 1, "decoder.post_nn.", dh[-1], rh[-1], None, H))
    return cases


@pytest.mark.parametrize("which,W", [("tiny", 32), ("tiny", 30), ("base", 20), ("base", 22)])
def test_every_transformer_layer(which, W, request):
    """All (C, heads) pairs, shifted + unshifted blocks, W%4 in {0,2} (window padding), H=2 bottom scale, merge/split."""
    from oracle import esc_oracle as O
    model, orc, g, cfg = request.getfixturevalue(which)
    lib, hd = _h(model)
    torch.manual_seed(11)
    for lid, pfx, C, nH, scale, H in _layer_cases(cfg):
        x = torch.randn(2, H * W, C)
        ref, Hr, _ = O.transformer_layer(x, H, W, orc.sd, pfx, nH, cfg["swin_depth"], cfg["window_size"], scale)
        y = torch.empty(ref.shape, device="cuda")
        Hn = ctypes.c_int()
        xg = x.cuda()
        _native.check(lib.escx_transformer_layer(hd, lid, _ptr(xg), 2, H, W, _ptr(y), ctypes.byref(Hn), None))
        assert Hn.value == Hr
        err = rel_rms(y.cpu(), ref)
        assert err < ACT_TOL, f"{which} layer {lid} ({pfx}) H={H} W={W}: rel rms {err:.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_layer_accuracy_against_fp64(precision):
    """Precision of the device arithmetic, measured per mode of escx_set_precision: every TransformerLayer of ESC-Base against the oracle evaluated in FLOAT64,
    next to the error of the oracle's own float32 evaluation (the reference's arithmetic: ATen sgemm) against the same float64 result.  "f16x2" (default): two fp16
    terms per operand, three cross products - NOT an exact split (truncation ~1e-7), measured 0.65-0.87x the reference's own error; "bf16x3": three bf16 terms, exact
    split, six cross products (0.88-1.03x); "fp32": the fp32 MFMA.  Bound (tightened from 2x): device error <= 1.1 x the reference's float32 error + 1e-7."""
    from oracle import esc_oracle as O
    model, orc, g, cfg = build_models("base", precision)
    assert model.precision == precision
    lib, hd = _h(model)
    assert lib.escx_get_precision(hd) == _native.PRECISIONS[precision]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in orc.sd.items()}
    torch.manual_seed(23)
    W = 20
    worst = 0.0
    for lid, pfx, C, nH, scale, H in _layer_cases(cfg):
        x = torch.randn(2, H * W, C)
        ref64, Hr, _ = O.transformer_layer(x.double(), H, W, sd64, pfx, nH, cfg["swin_depth"], cfg["window_size"], scale)
        ref32, _, _ = O.transformer_layer(x, H, W, orc.sd, pfx, nH, cfg["swin_depth"], cfg["window_size"], scale)
        y = torch.empty(ref32.shape, device="cuda"); Hn = ctypes.c_int()
        xg = x.cuda()
        _native.check(lib.escx_transformer_layer(hd, lid, _ptr(xg), 2, H, W, _ptr(y), ctypes.byref(Hn), None))
        e_dev, e_cpu = rel_rms(y.cpu().double(), ref64), rel_rms(ref32.double(), ref64)
        print(f"[fp64 {precision}] layer {lid:2d} {pfx:22s} C={C:3d}: device {e_dev:.2e}   reference float32 {e_cpu:.2e}   ratio {e_dev / max(e_cpu, 1e-30):.2f}")
        worst = max(worst, e_dev / (1.1 * e_cpu + 1e-7))
        assert e_dev <= 1.1 * e_cpu + 1e-7, f"layer {lid} ({pfx}) [{precision}]: device error {e_dev:.3e} vs float32 reference error {e_cpu:.3e}"
    print(f"[fp64 {precision}] worst device error / bound: {worst:.2f}")


def test_merge_with_odd_height(base):
    from oracle import esc_oracle as O
    model, orc, g, cfg = base
    lib, hd = _h(model)
    torch.manual_seed(5)
    H, W, C = 7, 20, cfg["h_dims"][1]                 # encoder.blocks.1 with an odd H -> zero row before the unshuffle
    x = torch.randn(2, H * W, C)
    ref, Hr, _ = O.transformer_layer(x, H, W, orc.sd, "encoder.blocks.1.", cfg["swin_heads"][1], 2, 4, "down")
    y = torch.empty(ref.shape, device="cuda"); Hn = ctypes.c_int()
    xg = x.cuda()
    _native.check(lib.escx_transformer_layer(hd, 2, _ptr(xg), 2, H, W, _ptr(y), ctypes.byref(Hn), None))
    assert Hn.value == Hr == 4 and rel_rms(y.cpu(), ref) < ACT_TOL


@pytest.mark.parametrize("which", ["tiny", "base"])
def test_pvq_encode_decode_every_stream(which, request):
    from oracle import esc_oracle as O
    model, orc, g, cfg = request.getfixturevalue(which)
    lib, hd = _h(model)
    torch.manual_seed(7)
    W, B = 24, 3
    G = cfg["group_size"]
    for s in range(cfg["max_streams"]):
        C = orc.dec_dims[max(s - 1, 0)]; Hq = orc.q_freq[s]
        enc = torch.randn(B, Hq * W, C); dec = torch.randn(B, Hq * W, C) * 0.5
        margins = []
        ref = O.pvq_encode(enc - dec, orc.sd, f"quantizers.{s}.", Hq, cfg["overlap"], G, cfg["l2norm"], margins=margins)
        codes = torch.full((B, G, W // 2), -1, dtype=torch.int64, device="cuda")
        encg, decg, refg = enc.cuda(), dec.cuda(), ref.cuda()
        _native.check(lib.escx_pvq_encode(hd, s, _ptr(encg), _ptr(decg), B, W, _ptr(codes), G * (W // 2), None))
        got = codes.cpu()
        m = torch.stack(margins, 1)
        bad = (got != ref)
        # a mismatch is only acceptable where the reference's own best/second-best gap is at fp32 noise level
        assert not bad.any() or float(m[bad].max()) < NEAR_TIE, f"stream {s}: {int(bad.sum())} mismatches, margins {m[bad][:5]}"
        refq = O.pvq_decode(ref, orc.sd, f"quantizers.{s}.", Hq, cfg["overlap"]) + dec
        out = torch.empty(B, Hq * W, C, device="cuda")
        _native.check(lib.escx_pvq_decode(hd, s, _ptr(refg), G * (W // 2), _ptr(decg), B, W, _ptr(out), None))
        assert rel_rms(out.cpu(), refq) < ACT_TOL
        # residual = enc (dec NULL) path used by stream 0
        margins0 = []
        ref0 = O.pvq_encode(enc, orc.sd, f"quantizers.{s}.", Hq, cfg["overlap"], G, cfg["l2norm"], margins=margins0)
        _native.check(lib.escx_pvq_encode(hd, s, _ptr(encg), None, B, W, _ptr(codes), G * (W // 2), None))
        bad0 = (codes.cpu() != ref0)
        m0 = torch.stack(margins0, 1)
        assert not bad0.any() or float(m0[bad0].max()) < NEAR_TIE, f"stream {s} (dec=NULL): {int(bad0.sum())} mismatches, margins {m0[bad0][:5]}"


def test_search_tie_break_and_degenerate_vectors(tiny):
    """SURVEY appendix B #11 / codebook.py:31-40 against the oracle's `codebook_search` (== torch.min semantics):
      * an EXACT tie (two identical codebook rows, the input equal to them) -> the lower index;
      * an input equal to a code -> that code;
      * a NaN in one vector -> index 0 for that vector (every distance is NaN, the first NaN wins), neighbours untouched;
      * the all-zero residual: every distance is fp32 ||c_hat||^2 ~ 1, the winner is decided by rounding of the normalised
        codebook norms - the emitted code must be a minimiser of the oracle's distances up to that rounding (excess < 2e-6)."""
    from oracle import esc_oracle as O
    from esc.models import make_model
    from conftest import synth_state
    _, orc0, g, cfg = tiny
    G, ov, Ksz = cfg["group_size"], cfg["overlap"], cfg["codebook_size"]
    sd = {k: v.clone() for k, v in synth_state("tiny").items()}
    sid = 1
    C, Hq, d = orc0.dec_dims[max(sid - 1, 0)], orc0.q_freq[sid], cfg["codebook_dims"][sid]
    D = ov * Hq * C
    dims = O.split_dimension(D, G)
    for gi in range(G):                                    # down-projection = "take the first d entries of the group's sub-vector"
        w = torch.zeros(d, dims[gi]); w[torch.arange(d), torch.arange(d)] = 1.0
        sd[f"quantizers.{sid}.down_projs.{gi}.weight"] = w
        cb = sd[f"quantizers.{sid}.vqs.{gi}.embedding.weight"]
        cb[40] = cb[9]                                      # rows 9 and 40 identical -> exact tie
    model = make_model(cfg); model.load_state_dict(sd); model = model.cuda().eval()
    lib, hd = _h(model)
    W, B = 16, 2
    T = W // ov
    v = torch.zeros(B, T, D)
    torch.manual_seed(21)
    v += 0.01 * torch.randn(B, T, D)                        # everything outside the first d entries is ignored by the selector
    starts = np.cumsum([0] + dims[:-1])
    cbs = [sd[f"quantizers.{sid}.vqs.{gi}.embedding.weight"] for gi in range(G)]
    for gi, st in enumerate(starts):
        v[:, :, st:st + d] = torch.randn(B, T, d)
        v[0, 0, st:st + d] = cbs[gi][9]                    # exact tie 9 / 40
        v[0, 1, st:st + d] = cbs[gi][40] * 2.5             # same direction, other norm: still both rows
        v[0, 2, st:st + d] = cbs[gi][17]                   # equal to a code
        v[0, 3, st:st + d] = 0.0                           # degenerate: zero sub-vector
        v[1, 2, st + 1] = float("nan")                     # NaN in one vector
    enc = O.pvq_unframes(v, Hq, ov)
    zes = []
    ref = O.pvq_encode(enc, sd, f"quantizers.{sid}.", Hq, ov, G, cfg["l2norm"], z_e_out=zes)
    codes = torch.full((B, G, T), -1, dtype=torch.int64, device="cuda")
    encg = enc.cuda()
    _native.check(lib.escx_pvq_encode(hd, sid, _ptr(encg), None, B, W, _ptr(codes), G * T, None))
    got = codes.cpu()
    assert (ref[0, :, 0] == 9).all() and (ref[0, :, 1] == 9).all() and (ref[0, :, 2] == 17).all() and (ref[1, :, 2] == 0).all(), ref
    strict = torch.ones(B, T, dtype=torch.bool); strict[0, 3] = False
    for gi in range(G):
        assert torch.equal(got[:, gi][strict], ref[:, gi][strict]), f"group {gi}: {got[:, gi]} vs {ref[:, gi]}"
        # all-zero sub-vector: the chosen code must minimise the oracle's own distance expression up to fp32 rounding
        cbn = torch.nn.functional.normalize(cbs[gi], dim=-1)
        dist0 = (torch.zeros(1, 1) - (2 * torch.zeros(1, d)) @ cbn.t()) + cbn.pow(2).sum(1, keepdim=True).t()
        k = int(got[0, gi, 3])
        assert 0 <= k < Ksz and float(dist0[0, k] - dist0.min()) < NEAR_TIE
    # whole all-zero map through the stream-0 path (dec = NULL): range + every frame of a group sees the same distances
    C0, Hq0 = orc0.dec_dims[0], orc0.q_freq[0]
    z = torch.zeros(1, Hq0 * 8, C0).cuda()
    c0 = torch.full((1, G, 4), -1, dtype=torch.int64, device="cuda")
    _native.check(lib.escx_pvq_encode(hd, 0, _ptr(z), None, 1, 8, _ptr(c0), G * 4, None))
    assert int(c0.min()) >= 0 and int(c0.max()) < Ksz and (c0 == c0[..., :1]).all()


def test_device_math_accuracy():
    """The branch-free erf/GELU and the v_exp_f32 exponential used inside the fused kernels, against fp64."""
    from scipy import special
    lib = _native.load()
    x = torch.cat([torch.linspace(-9, 9, 2_000_001), torch.linspace(-1.5, 1.5, 1_000_001), torch.tensor([0.0, -0.0, 1e-20, -1e-20, 30.0, -30.0])]).float()
    xg = x.cuda(); yg = torch.empty_like(xg)
    x64 = x.double().numpy()
    _native.check(lib.escx_test_math(_ptr(xg), _ptr(yg), xg.numel(), 1, None))
    erf_ref = special.erf(x64)
    ulp = np.spacing(np.abs(erf_ref).astype(np.float32)).astype(np.float64)
    err = np.abs(yg.cpu().double().numpy() - erf_ref) / np.maximum(ulp, 1e-45)
    assert err[np.abs(x64) > 1e-10].max() < 2.0, f"erf max ulp error {err.max()}"
    _native.check(lib.escx_test_math(_ptr(xg), _ptr(yg), xg.numel(), 0, None))
    g_ref = 0.5 * x64 * (1.0 + erf_ref_scaled(x64))
    g_torch = torch.nn.functional.gelu(x).double().numpy()          # what the reference computes on CPU
    got = yg.cpu().double().numpy()
    assert np.abs(got - g_ref).max() <= np.abs(g_torch - g_ref).max() * 1.5 + 1e-9
    rel = np.abs(got - g_torch) / np.maximum(1.0, np.abs(x64))
    k = int(rel.argmax())
    assert rel[k] <= 6e-7, f"gelu differs from torch by {rel[k]:.3e} (scaled) at x={x64[k]!r}: {got[k]!r} vs {g_torch[k]!r} (true {g_ref[k]!r})"
    xe = torch.linspace(-104, 0, 1_000_001).float(); xeg = xe.cuda(); ye = torch.empty_like(xeg)
    _native.check(lib.escx_test_math(_ptr(xeg), _ptr(ye), xeg.numel(), 2, None))
    ref = np.exp(xe.double().numpy())
    assert (np.abs(ye.cpu().double().numpy() - ref) <= 2e-7 * ref * np.maximum(1.0, np.abs(xe.double().numpy())) + 1e-37).all()


def erf_ref_scaled(x64):
    from scipy import special
    return special.erf(x64 / np.sqrt(2.0))


# ---------------------------------------------------------------- whole path vs golden ------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["tiny", "base", "large"])
def test_codes_bit_exact_and_audio_vs_golden(name, precision):
    """The golden vectors of the REAL reference (oracle/gen_golden.py): codes bit-exact for every num_streams, audio within 1e-4 RMS - in every precision mode."""
    model, orc, g, cfg = build_models(name, precision)
    S = cfg["max_streams"]
    if name == "tiny":
        cases = [(torch.from_numpy(synth.pcm_to_float(g[f"L{L}_pcm"])), g[f"L{L}_codes"], g[f"L{L}_margins"],
                  {s: g[f"L{L}_audio_s{s}"] for s in range(1, S + 1)}) for L in (1280, 1200)]
    else:
        cases = [(torch.from_numpy(synth.pcm_to_float(g["pcm"])), g["codes"], g["margins"],
                  {s: g[f"audio_s{s}"] for s in range(1, S + 1)})]
    for x, ref_codes, margins, audio in cases:
        xg = x.cuda()
        for s in range(1, S + 1):
            codes, shape = model.encode(xg, s)
            assert codes.dtype == torch.int64 and codes.shape[1] == s
            ref = ref_codes[:, :s].astype(np.int64)
            assert np.array_equal(codes.cpu().numpy(), ref), f"{name} S={s}: " + code_report(codes.cpu().numpy(), ref, margins[:, :s])
            wave = model.decode(torch.from_numpy(ref).cuda(), shape).cpu().numpy()
            gold = audio[s]
            got = wave if s == S else wave[:, ::8]
            assert got.shape == gold.shape
            assert rms(got, gold) <= AUDIO_TOL, f"{name} S={s}: audio rms {rms(got, gold):.3e}"


@pytest.mark.parametrize("ws", [2, 3, 8])
def test_other_window_sizes_vs_golden(ws):
    """window_size != 4 (attention.py:93-127, 246-256): the unfused launch sequence with window_attention_any_kernel against fixtures of the REAL reference
    (oracle/gen_window_golden.py; tiny configuration, 2 x 2 / 3 x 3 / 8 x 8 windows, W = 32 and 30 frames) and against the oracle's encoder maps; the training step refuses."""
    from esc.models import make_model
    from oracle.esc_oracle import EscOracle, Trace
    g = load_golden("window")
    cfg = json.loads(str(g[f"ws{ws}_config_json"]))
    sd = synth_state(f"window_ws{ws}")
    model = make_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    orc = EscOracle(cfg, sd)
    for precision in PRECISIONS:
        model.set_precision(precision)
        for L in (1280, 1200):
            x = torch.from_numpy(synth.pcm_to_float(g[f"ws{ws}_L{L}_pcm"]))
            ref = g[f"ws{ws}_L{L}_codes"].astype(np.int64)
            for s in (1, 2, 3):
                codes, shape = model.encode(x.cuda(), s)
                assert tuple(shape) == tuple(g[f"ws{ws}_L{L}_feat_shape"])
                assert np.array_equal(codes.cpu().numpy(), ref[:, :s]), f"ws {ws} L {L} S {s} {precision}: " + code_report(codes.cpu().numpy(), ref[:, :s], g[f"ws{ws}_L{L}_margins"][:, :s])
                wave = model.decode(torch.from_numpy(ref[:, :s].copy()).cuda(), shape).cpu().numpy()
                assert rms(wave if s == 3 else wave[:, ::8], g[f"ws{ws}_L{L}_audio_s{s}"]) <= AUDIO_TOL
            out = model(**dict(x=x.cuda(), x_feat=None, num_streams=3))
            assert np.array_equal(out["codes"].cpu().numpy(), ref)
    model.train()
    with pytest.raises(NotImplementedError, match="window_size"):
        model(**dict(x=x.cuda(), x_feat=None, num_streams=3, freeze_codebook=False))


def test_window_size_8_on_the_base_geometry_against_the_oracle():
    """ESC-Base with 8 x 8 windows (the 2- and 4-row maps are padded to 8 rows, 300 frames to 304), two 1 s clips: codes against the oracle - pinned for other window
    sizes by tests/golden/window.npz - identical or attributed near-ties continued through all six streams; audio of the oracle's codes within the bound."""
    from esc.models import make_model
    from esc.models.codecs import state_manifest
    from oracle.esc_oracle import EscOracle, Trace
    cfg = dict(json.loads(str(load_golden("base")["config_json"])), window_size=8)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict({k: list(v) for k, v in state_manifest(cfg).items()}).items()}
    for k in sd:
        if k.endswith(".window"):
            sd[k] = torch.hann_window(sd[k].shape[0])
    model = make_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    orc = EscOracle(cfg, sd)
    pcm = np.stack([synth.noise_clip_int16("ws8-base-0", 16000), synth.voiced_clip_int16("ws8-base-1", 16000)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    tr = Trace()
    ref, shape = orc.encode(x, 6, trace=tr)
    margins = torch.stack(tr.margins, dim=1).numpy()
    codes, gshape = model.encode(x.cuda(), 6)
    assert tuple(gshape) == tuple(shape)
    findings, forced, _ = attribute_with_continuation(orc, x, codes.cpu().numpy(), ref.numpy(), margins, 6)
    assert not findings, findings[:5]
    assert forced <= 2, f"{forced} near-tie codes resolved the other way in two clips"
    wave = model.decode(ref.cuda(), shape).cpu().numpy()
    assert rms(wave, orc.decode(ref, shape).numpy()) <= AUDIO_TOL


@pytest.mark.parametrize("L", [16000, 24000])
def test_edge_lengths_vs_golden(base, L):
    model, orc, g, cfg = base
    e = load_golden("edge")
    x = torch.from_numpy(synth.pcm_to_float(e[f"L{L}_pcm"])).cuda()
    codes, shape = model.encode(x, 6)
    assert tuple(shape) == tuple(e[f"L{L}_feat_shape"])
    ref = e[f"L{L}_codes"].astype(np.int64)
    assert np.array_equal(codes.cpu().numpy(), ref), code_report(codes.cpu().numpy(), ref, e[f"L{L}_margins"])
    for s in (1, 3, 6):
        wave = model.decode(codes[:, :s].contiguous(), shape).cpu().numpy()
        assert rms(wave if s == 6 else wave[:, ::8], e[f"L{L}_audio_s{s}"]) <= AUDIO_TOL


def test_forward_eval_dict(base):
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    for s in (1, 4, 6):
        out = model(**dict(x=x, x_feat=None, num_streams=s))
        assert set(out) == {"cm_loss", "cb_loss", "raw_audio", "recon_audio", "raw_feat", "recon_feat", "codes"}
        codes, shape = model.encode(x, s)
        assert torch.equal(out["codes"], codes)
        wave = model.decode(codes, shape)
        assert torch.equal(out["recon_audio"], wave), "forward(eval) must equal decode(encode()) bit for bit"
        np.testing.assert_allclose(out["cm_loss"].cpu().numpy(), g[f"cm_loss_s{s}"], rtol=1e-4)
        assert out["raw_feat"].shape == (2, 2, 192, 601) and out["recon_feat"].shape == (2, 2, 192, 600)
        ref = orc.forward_eval(x.cpu(), None, s)
        assert rel_rms(out["raw_feat"].cpu(), ref["raw_feat"]) < ACT_TOL
        assert rel_rms(out["recon_feat"].cpu(), ref["recon_feat"]) < 1e-4


def test_runs_on_the_callers_stream_and_is_deterministic(base):
    """Work is enqueued on the caller's current stream (the extra part streams fork from and join it): results produced inside a
    side stream, consumed there without any device-wide sync, equal the default-stream results bit for bit - including the
    eval-mode losses (no atomics anywhere in the path)."""
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    ref = model(**dict(x=x, x_feat=None, num_streams=6))
    again = model(**dict(x=x, x_feat=None, num_streams=6))
    for k in ("codes", "recon_audio", "cm_loss", "cb_loss", "recon_feat", "raw_feat"):
        assert torch.equal(ref[k], again[k]), f"{k} is not run-to-run deterministic"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        xs = x * 1.0                                   # produced on the side stream right before the call
        codes, shape = model.encode(xs, 6)
        wave = model.decode(codes, shape)
        checksum = wave.double().sum()                 # consumed on the side stream right after
    side.synchronize()
    assert torch.equal(codes, ref["codes"]) and torch.equal(wave, ref["recon_audio"])
    assert float(checksum) == float(ref["recon_audio"].double().sum())


def test_encode_decode_capture_into_a_hip_graph(base):
    """After reserve(), a whole encode+decode (two parts on forked streams included) must be capturable into a hipGraph: no
    allocation, no host synchronisation, no host-side reads inside the calls.  Replaying the graph on new input contents gives
    the eager results bit for bit."""
    model, orc, g, cfg = base
    xa = torch.from_numpy(synth.pcm_to_float(np.stack([synth.noise_clip_int16(f"graph-a{i}", 48000) for i in range(4)]))).cuda()
    xb = torch.from_numpy(synth.pcm_to_float(np.stack([synth.voiced_clip_int16(f"graph-b{i}", 48000) for i in range(4)]))).cuda()
    model.reserve(4, 48000, xa.device)
    eager = {}
    for name, x in (("a", xa), ("b", xb)):
        c, shp = model.encode(x, 6)
        eager[name] = (c.clone(), model.decode(c, shp).clone())
    x = xa.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            c, shp = model.encode(x, 6); model.decode(c, shp)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        codes, shp = model.encode(x, 6)
        wave = model.decode(codes, shp)
    for name, src in (("b", xb), ("a", xa), ("b", xb)):
        x.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(codes, eager[name][0]) and torch.equal(wave, eager[name][1]), f"graph replay differs from eager on input {name}"


def test_forward_with_precomputed_spectrum(base):
    """forward(x, x_feat=...) (codecs.py:33-34): the spectrum the library itself returns, fed back as x_feat (Bs,F,T,2), must
    reproduce codes, audio and losses bit for bit (the STFT is the only thing skipped); the oracle agrees on codes and audio."""
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    for s in (6, 2):
        ref = model(**dict(x=x, x_feat=None, num_streams=s))
        x_feat = ref["raw_feat"].permute(0, 2, 3, 1).contiguous()              # (B,2,F,T) -> (B,F,T,2)
        out = model(**dict(x=x, x_feat=x_feat, num_streams=s))
        assert set(out) == set(ref)
        assert torch.equal(out["codes"], ref["codes"]) and torch.equal(out["recon_audio"], ref["recon_audio"])
        assert torch.equal(out["cm_loss"], ref["cm_loss"]) and torch.equal(out["recon_feat"], ref["recon_feat"])
        assert out["raw_feat"].shape == ref["raw_feat"].shape and torch.equal(out["raw_feat"], ref["raw_feat"])
        o = orc.forward_eval(x.cpu(), x_feat.cpu(), s)
        assert torch.equal(o["codes"], out["codes"].cpu()), code_report(out["codes"].cpu().numpy(), o["codes"].numpy())
        assert rms(out["recon_audio"].cpu().numpy(), o["recon_audio"].numpy()) <= AUDIO_TOL
    with pytest.raises(ValueError):
        model(**dict(x=x, x_feat=x_feat[:, :100], num_streams=6))
    with pytest.raises(AssertionError, match="multiple of overlap"):
        model(**dict(x=x, x_feat=x_feat[:, :, :302].contiguous(), num_streams=6))      # W = 151, overlap 2


def test_full_size_invariants_batch36(base):
    """BASELINE config 2 size (B=36, 3 s clips): prefix property, batch invariance, determinism, code range."""
    model, orc, g, cfg = base
    pcm = np.stack([synth.noise_clip_int16(f"bench-{i}", 48000) for i in range(36)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda()
    full, shape = model.encode(x, 6)
    assert full.shape == (36, 6, 3, 150) and tuple(shape) == (2, 300)
    assert int(full.min()) >= 0 and int(full.max()) < 1024
    again, _ = model.encode(x, 6)
    assert torch.equal(full, again)                                        # run-to-run deterministic
    for s in (1, 2, 3, 5):
        part, _ = model.encode(x, s)
        assert torch.equal(part, full[:, :s])                              # prefix property
    sub, _ = model.encode(x[5:9].contiguous(), 6)
    assert torch.equal(sub, full[5:9])                                     # batch invariance
    wave = model.decode(full, shape)
    assert wave.shape == (36, 47920) and torch.isfinite(wave).all()
    w1 = model.decode(full[7:8].contiguous(), shape)
    assert torch.equal(w1, wave[7:8])
    # (the oracle comparison at this size lives in test_bench_inputs_and_voiced_batch36_against_the_oracle - the clips bench.py times - and in the sweeps)


def test_bench_inputs_and_voiced_batch36_against_the_oracle(base):
    """The exact 36 clips bench.py times (`bench-r0-i`) and an 18-clip voiced batch (the first 18 of the 36 earlier rounds checked: GPU-suite time): every clip compared with the oracle; a
    difference must sit on a reference near-tie (margin < 2e-6) in the earliest differing stream of its clip."""
    model, orc, g, cfg = base
    Trace = __import__("oracle.esc_oracle", fromlist=["Trace"]).Trace
    for label, pcm in (("bench", np.stack([synth.noise_clip_int16(f"bench-r0-{i}", 48000) for i in range(36)])),
                       ("voiced", np.stack([synth.voiced_clip_int16(f"voiced36-{i}", 48000) for i in range(18)]))):
        x = torch.from_numpy(synth.pcm_to_float(pcm))
        codes, shape = model.encode(x.cuda(), 6)
        wave = model.decode(codes, shape)
        tr = Trace()
        oc, _ = orc.encode(x, 6, trace=tr)
        m = torch.stack(tr.margins, dim=1).numpy()
        bad, forced, cont = attribute_with_continuation(orc, x, codes.cpu().numpy(), oc.numpy(), m, 6)
        assert not bad, f"{label}: " + "\n".join(bad[:10])
        same = (codes.cpu() == oc).flatten(1).all(1).numpy()
        print(f"[{label}{len(pcm)}] {int(same.sum())}/{len(pcm)} clips bit-exact, min margin {m.min():.2e}; " + mismatch_summary(codes.cpu().numpy(), oc.numpy(), m))
        assert same.sum() == len(pcm), f"{label}: {int(same.sum())}/{len(pcm)} clips bit-exact (every clip is what this build measures on both batches)"
        ow = orc.decode(oc, shape).numpy()
        assert rms(wave.cpu().numpy()[same], ow[same]) <= AUDIO_TOL


def test_large_batch36_against_the_oracle():
    """ESC-Large (depth 4) at the benchmarked batch size: batch invariance inside B=36 and the oracle on 4 of its clips."""
    model, orc, g, cfg = build_models("large")
    pcm = np.stack([synth.noise_clip_int16(f"large36-{i}", 48000) for i in range(36)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    codes, shape = model.encode(x.cuda(), 6)
    wave = model.decode(codes, shape)
    assert codes.shape == (36, 6, 3, 150) and torch.isfinite(wave).all()
    idx = [0, 11, 23, 35]
    sub, _ = model.encode(x[idx].cuda(), 6)
    assert torch.equal(sub, codes[idx])
    tr = __import__("oracle.esc_oracle", fromlist=["Trace"]).Trace()
    oc, _ = orc.encode(x[idx], 6, trace=tr)
    m = torch.stack(tr.margins, dim=1).numpy()
    got = codes[idx].cpu().numpy()
    bad, forced, cont = attribute_with_continuation(orc, x[idx], got, oc.numpy(), m, 6)
    assert not bad, "\n".join(bad[:10])
    same = (codes[idx].cpu() == oc).flatten(1).all(1).numpy()
    print(f"[large36] {int(same.sum())}/4 clips bit-exact, min margin {m.min():.2e}")
    ow = orc.decode(oc, shape).numpy()
    assert same.any() and rms(wave[idx].cpu().numpy()[same], ow[same]) <= AUDIO_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["base", "large"])
def test_unfiltered_reference_clips(name, precision):
    """tests/golden/unfiltered.npz (oracle/gen_unfiltered_golden.py): the first clips by tag, NO margin-based selection, codes and
    margins from the real reference (margins down to 3e-7 for Base).  Requirement: no difference that is not a reference near-tie;
    the report names every difference with its margin."""
    import json
    model, orc, g, cfg = build_models(name, precision)
    u = load_golden("unfiltered")
    tags = json.loads(str(u[f"{name}_tags"]))
    pcm = np.stack([(synth.noise_clip_int16 if k == "noise" else synth.voiced_clip_int16)(t, 48000) for k, t in tags])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda()
    codes, shape = model.encode(x, cfg["max_streams"])
    ref, m = u[f"{name}_codes"].astype(np.int64), u[f"{name}_margins"]
    got = codes.cpu().numpy()
    bad, forced, cont = attribute_with_continuation(orc, x.cpu(), got, ref, m, cfg["max_streams"])
    same = (got == ref).reshape(len(tags), -1).all(1)
    print(f"[unfiltered {name} {precision}] {int(same.sum())}/{len(tags)} clips bit-exact, reference min margin {m.min():.2e}, "
          f"{int((m < 1e-5).sum())} codes under 1e-5; " + mismatch_summary(got, ref, m))
    assert not bad, "\n".join(bad[:10])
    wave = model.decode(torch.from_numpy(ref).cuda(), shape).cpu().numpy()
    assert rms(wave[:, ::16], u[f"{name}_audio_sub"]) <= AUDIO_TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_clustered_codebooks_against_the_reference(precision):
    """VERDICT r4 item 4b / SURVEY hard part 1: parity on trained-like margin statistics.  tests/golden/clustered.npz holds what the REAL reference
    emits for ESC-Base when every codebook row has a near-duplicate at relative distance 1e-4 ... 1e-6 (esc/synth.py cluster_codebook;
    oracle/gen_clustered_golden.py): ~13 000 of its 16 200 argmin margins are below 1e-5, ~5 600 below 1e-6.  Requirement: ZERO unattributable
    differences - a code the HIP path resolves the other way must sit on a reference margin < 2e-6 in the earliest differing stream of its clip, and
    the later streams of that clip must equal the oracle continued from the device's choice; the flip count per margin decade is printed."""
    import json
    from esc.models import make_model
    from oracle.esc_oracle import EscOracle
    from conftest import load_manifest
    g = load_golden("base")
    cfg = json.loads(str(g["config_json"]))
    sd = {}
    for k, v in synth.clustered_state_dict(load_manifest("base")).items():
        sd[k] = torch.hann_window(v.shape[0]) if k.endswith(".window") else torch.from_numpy(np.ascontiguousarray(v))
    model = make_model(cfg); model.load_state_dict(sd, strict=True); model = model.to("cuda:0").eval().set_precision(precision)
    orc = EscOracle(cfg, sd)
    u = load_golden("clustered")
    tags = json.loads(str(u["tags"]))
    pcm = np.stack([(synth.noise_clip_int16 if k == "noise" else synth.voiced_clip_int16)(t, 48000) for k, t in tags])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    codes, shape = model.encode(x.cuda(), cfg["max_streams"])
    got, ref, m = codes.cpu().numpy(), u["codes"].astype(np.int64), u["margins"]
    # flips against the reference in the FIRST stream (every clip sees the reference's own residual there): by decade of the reference margin
    first = got[:, 0] != ref[:, 0]
    by_decade = {f"<1e-{e}": (int((first & (m[:, 0] < 10.0 ** -e) & (m[:, 0] >= 10.0 ** -(e + 1))).sum()), int(((m[:, 0] < 10.0 ** -e) & (m[:, 0] >= 10.0 ** -(e + 1))).sum()))
                 for e in (4, 5, 6, 7, 8)}
    by_decade["exact ties (0)"] = (int((first & (m[:, 0] == 0)).sum()), int((m[:, 0] == 0).sum()))
    bad, forced, cont = attribute_with_continuation(orc, x, got, ref, m, cfg["max_streams"])
    print(f"[clustered {precision}] {int((got != ref).sum())} of {ref.size} codes differ from the reference fixture ({forced} attributed near-ties forced, {len(cont)} clips continued); "
          f"stream-0 flips / codes per reference-margin decade: {by_decade}; reference margins under 1e-5: {int((m < 1e-5).sum())}")
    assert not bad, "\n".join(bad[:10])
    # the near-duplicates decode to (almost) the same vectors: the device's own round trip stays within the audio tolerance of the reference's
    wave = model.decode(codes, shape).cpu().numpy()
    ref_rms = float(np.sqrt(np.mean(u["audio_sub"].astype(np.float64) ** 2)))
    assert rms(wave[:, ::16], u["audio_sub"]) <= 1e-3 * ref_rms + AUDIO_TOL        # flipped near-duplicates (relative distance <= 1e-4) move the audio by ~1e-5 relative
    assert rms(model.decode(torch.from_numpy(ref).cuda(), shape).cpu().numpy()[:, ::16], u["audio_sub"]) <= AUDIO_TOL


def test_decode_partial_streams_and_errors(base):
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    codes, shape = model.encode(x, 6)
    for s in (1, 2, 6):                                                    # csrvq.py:174-177: missing streams pass through
        w = model.decode(codes[:, :s].contiguous(), shape).cpu()
        ref = orc.decode(codes[:, :s].cpu(), shape)
        assert rms(w.numpy(), ref.numpy()) <= AUDIO_TOL
    with pytest.raises(AssertionError, match="multiple of overlap"):        # quantization.py:407 (T=302 -> W=151)
        model.encode(torch.zeros(1, 24080, device="cuda"), 6)


def test_fused_and_unfused_pipelines_agree(base, monkeypatch):
    """The register-resident fused Swin kernels against the plain GEMM pipeline (ESCX_NO_FUSED=1): same codes, audio
    within fp32 re-association noise.  Also checks the int16 transport kernels."""
    import os
    from esc.models import make_model
    from conftest import synth_state
    from esc.distributed import narrow_codes, widen_codes
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    codes, shape = model.encode(x, 6)
    wave = model.decode(codes, shape)
    monkeypatch.setenv("ESCX_NO_FUSED", "1")
    plain = make_model(cfg); plain.load_state_dict(synth_state("base")); plain = plain.cuda().eval()
    c2, s2 = plain.encode(x, 6)
    w2 = plain.decode(c2, s2)
    monkeypatch.delenv("ESCX_NO_FUSED")
    assert torch.equal(codes, c2), code_report(codes.cpu().numpy(), c2.cpu().numpy(), g["margins"])
    assert rms(wave.cpu().numpy(), w2.cpu().numpy()) < 2e-6
    assert narrow_codes(codes).dtype == torch.int16 and torch.equal(widen_codes(narrow_codes(codes)), codes)


def test_reloading_weights_after_use_takes_effect(base):
    """load_state_dict / .to() after the first call must invalidate the packed device copy (compress.py:23-25 loads a checkpoint
    into an already constructed model): a model that has already run with the fixture weights, reloaded with perturbed ones,
    matches a fresh model built from the perturbed weights, and returns to the original results when reloaded back."""
    from esc.models import make_model
    from conftest import synth_state
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    sd0 = synth_state("base")
    used = make_model(cfg); used.load_state_dict(sd0); used = used.cuda().eval()
    c0, shp = used.encode(x, 6); w0 = used.decode(c0, shp)
    sd1 = {k: (v * 1.01 if (v.is_floating_point() and "embedding" not in k and "window" not in k) else v) for k, v in sd0.items()}
    used.load_state_dict(sd1)
    c1, _ = used.encode(x, 6); w1 = used.decode(c1, shp)
    fresh = make_model(cfg); fresh.load_state_dict(sd1); fresh = fresh.cuda().eval()
    cf, _ = fresh.encode(x, 6); wf = fresh.decode(cf, shp)
    assert torch.equal(c1, cf) and torch.equal(w1, wf)
    assert not torch.equal(w1, w0), "the reloaded weights were ignored"
    used.load_state_dict(sd0)
    c2, _ = used.encode(x, 6)
    assert torch.equal(c2, c0) and torch.equal(used.decode(c2, shp), w0)
    moved = used.cpu().cuda()
    c3, _ = moved.encode(x, 6)
    assert torch.equal(c3, c0)


def test_variable_bitrate_sweep_batch64(base):
    """BASELINE config 3: batch 64, num_streams 1..6 (the RVQ early-exit path).  Prefix property against the S=6 run,
    decode of every prefix finite and S-dependent, oracle agreement on a 2-clip sample per S."""
    model, orc, g, cfg = base
    pcm = np.stack([synth.noise_clip_int16(f"vbr-{i}", 48000, amp=0.05 + 0.01 * (i % 7)) for i in range(64)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda()
    full, shape = model.encode(x, 6)
    prev = None
    for s in range(1, 7):
        codes, shp = model.encode(x, s)
        assert tuple(shp) == tuple(shape) and codes.shape == (64, s, 3, 150)
        assert torch.equal(codes, full[:, :s])
        wave = model.decode(codes, shape)
        assert wave.shape == (64, 47920) and torch.isfinite(wave).all()
        if prev is not None:
            assert not torch.equal(wave, prev)
        prev = wave
        oc, _ = orc.encode(x[30:32].cpu(), s)
        assert torch.equal(oc, codes[30:32].cpu()), code_report(codes[30:32].cpu().numpy(), oc.numpy())
        assert rms(wave[30:32].cpu().numpy(), orc.decode(oc, shape).numpy()) <= AUDIO_TOL


def test_node_batch_288_matches_its_36_clip_shards(base):
    """BASELINE config 4 size on one GPU (288 clips = 8 ranks x 36): every 36-clip shard, processed alone, must give the same codes
    and the same audio bit for bit as inside the 288-clip batch (what makes sharding across ranks transparent), and the round trip
    code -> audio -> ... stays finite.  Oracle agreement on one clip of the last shard."""
    model, orc, g, cfg = base
    pcm = np.stack([synth.noise_clip_int16(f"node-r{r}-{i}", 48000, amp=0.04 + 0.02 * (r % 4)) for r in range(8) for i in range(36)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda()
    codes, shape = model.encode(x, 6)
    assert codes.shape == (288, 6, 3, 150) and int(codes.min()) >= 0 and int(codes.max()) < 1024
    wave = model.decode(codes, shape)
    assert wave.shape == (288, 47920) and torch.isfinite(wave).all()
    for r in (0, 3, 7):
        sl = slice(36 * r, 36 * r + 36)
        c_r, _ = model.encode(x[sl].contiguous(), 6)
        assert torch.equal(c_r, codes[sl]), f"shard {r}: codes depend on the batch"
        assert torch.equal(model.decode(c_r, shape), wave[sl]), f"shard {r}: audio depends on the batch"
    oc, _ = orc.encode(x[287:288].cpu(), 6)
    assert torch.equal(oc, codes[287:288].cpu()), code_report(codes[287:288].cpu().numpy(), oc.numpy())


def test_empty_batch(base):
    """Bs = 0 flows through like in the reference (every op there accepts an empty batch): empty codes / audio of the right
    trailing shapes, and the latent shape of the clip length."""
    model, orc, g, cfg = base
    x = torch.zeros(0, 48000, device="cuda")
    codes, shape = model.encode(x, 4)
    assert codes.shape == (0, 4, 3, 150) and codes.dtype == torch.int64 and tuple(shape) == (2, 300)
    assert model.decode(codes, shape).shape == (0, 47920)
    out = model(**dict(x=x, x_feat=None, num_streams=4))
    assert out["codes"].shape == (0, 4, 3, 150) and out["recon_audio"].shape == (0, 47920) and out["cm_loss"].shape == (0,)
    assert out["raw_feat"].shape == (0, 2, 192, 601) and out["recon_feat"].shape == (0, 2, 192, 600)


def test_decode_survives_corrupt_code_indices(base):
    """A transmitted code outside [0, 1024) must not make the de-quantisation gather read outside the codebook: indices are
    clamped (documented in escx.h), the other clips of the batch are untouched."""
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    codes, shape = model.encode(x, 6)
    good = model.decode(codes, shape)
    bad = codes.clone()
    bad[0, 2, 1, 10] = 5000; bad[0, 0, 0, 0] = -7; bad[0, 5, 2, 149] = 2 ** 40
    clamp = codes.clone()
    clamp[0, 2, 1, 10] = 1023; clamp[0, 0, 0, 0] = 0; clamp[0, 5, 2, 149] = 1023
    out = model.decode(bad, shape)
    assert torch.isfinite(out).all()
    assert torch.equal(out, model.decode(clamp, shape))
    assert torch.equal(out[1], good[1])


def test_c_abi_error_paths(base):
    """Status codes and messages of the C ABI (nothing throws across it)."""
    model, orc, g, cfg = base
    lib, hd = _h(model)
    x = torch.zeros(1, 48000, device="cuda"); codes = torch.zeros(1, 6, 3, 150, dtype=torch.int64, device="cuda")
    fh, fw = ctypes.c_int(), ctypes.c_int()
    assert lib.escx_encode(hd, _ptr(x), 1, 48000, 7, _ptr(codes), ctypes.byref(fh), ctypes.byref(fw), None) == _native.ESCX_ERR_INVALID_ARG
    assert b"num_streams" in lib.escx_last_error()
    assert lib.escx_encode(hd, None, 1, 48000, 6, _ptr(codes), None, None, None) == _native.ESCX_ERR_INVALID_ARG
    assert lib.escx_encode(hd, _ptr(x), 1, 100, 6, _ptr(codes), None, None, None) == _native.ESCX_ERR_INVALID_ARG      # shorter than the reflect pad
    assert lib.escx_encode(hd, _ptr(x), 1, 24080, 6, _ptr(codes), None, None, None) == _native.ESCX_ERR_ASSERT          # W = 151 is odd
    assert b"multiple of overlap" in lib.escx_last_error()
    out = torch.zeros(1, 47920, device="cuda")
    assert lib.escx_decode(hd, _ptr(codes), 1, 6, 3, 300, _ptr(out), None, None) == _native.ESCX_ERR_INVALID_ARG         # wrong feat_shape
    assert lib.escx_transformer_layer(hd, 99, _ptr(x), 1, 4, 4, _ptr(out), None, None) == _native.ESCX_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        model.decode(torch.zeros(1, 6, 2, 150, dtype=torch.int64, device="cuda"), (2, 300))
    # the handle still works after errors
    c, s = model.encode(torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda(), 6)
    assert np.array_equal(c.cpu().numpy(), g["codes"].astype(np.int64))


def test_long_clip_10s_and_default_feat_shape(base):
    """A 10 s clip (L=160000 -> W=1000, the decode() default feat_shape) against the oracle; attention is local, so
    cost and correctness are length-independent."""
    model, orc, g, cfg = base
    pcm = synth.voiced_clip_int16("long-0", 160000)
    x = torch.from_numpy(synth.pcm_to_float(pcm))[None]
    codes, shape = model.encode(x.cuda(), 6)
    assert tuple(shape) == (2, 1000) and codes.shape == (1, 6, 3, 500)
    margins = []
    tr = __import__("oracle.esc_oracle", fromlist=["Trace"]).Trace()
    oc, oshape = orc.encode(x, 6, trace=tr)
    m = torch.stack(tr.margins, dim=1).numpy()
    assert torch.equal(oc, codes.cpu()), code_report(codes.cpu().numpy(), oc.numpy(), m)
    wave = model.decode(codes)                     # default feat_shape=(2, 1000)
    assert wave.shape == (1, 80 * 1999)
    assert rms(wave.cpu().numpy(), orc.decode(oc, oshape).numpy()) <= AUDIO_TOL


def test_bitstream_and_compress_harness(base, tmp_path):
    """10-bit wire format round trip (9 kbps payload exactly) and the scipy-based compress harness."""
    import subprocess, sys, os
    from scipy.io import wavfile
    from esc import bitstream
    from conftest import ROOT
    model, orc, g, cfg = base
    x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
    codes, shape = model.encode(x, 6)
    blob = bitstream.pack_codes(codes, shape)
    assert len(blob) == 16 + codes.numel() * 10 // 8
    assert (len(blob) - 16) * 8 / 3.0 / codes.shape[0] == 9000.0 == bitstream.payload_bits_per_second(6)
    back, shp = bitstream.unpack_codes(blob)
    assert torch.equal(back, codes) and tuple(shp) == tuple(shape)
    # the documented layout, checked by an independent host unpacker: little-endian, 4 codes -> 5 bytes (40 bits), code i of a
    # quad occupies bits 10*i .. 10*i+9 of the 40-bit little-endian word
    payload = np.frombuffer(blob[16:], dtype=np.uint8).reshape(-1, 5).astype(np.uint64)
    word = sum(payload[:, k] << np.uint64(8 * k) for k in range(5))
    host = np.stack([(word >> np.uint64(10 * i)) & np.uint64(1023) for i in range(4)], axis=1).reshape(-1)[: codes.numel()]
    assert np.array_equal(host.astype(np.int64), codes.cpu().numpy().reshape(-1))
    assert blob[:4] == b"ESC1" and np.frombuffer(blob[4:16], dtype="<u2").tolist() == [2, 6, 3, 150, 2, 300]
    with pytest.raises(ValueError, match="truncated"):
        bitstream.unpack_codes(blob[:-7])
    with pytest.raises(ValueError):
        bitstream.unpack_codes(b"ESC1" + bytes(12) + blob[16:])                  # zero dimensions in the header
    with pytest.raises(ValueError, match="outside"):
        bitstream.pack_codes(codes + 1024, shape)
    with pytest.raises(ValueError, match="does not match"):
        bitstream.unpack_codes(bitstream.pack_codes(codes[:, :, :2].contiguous(), shape), model=model)
    back_m, _ = bitstream.unpack_codes(blob, model=model)
    assert torch.equal(back_m, codes)
    odd = codes.reshape(-1)[:1001].reshape(1, 1, 1, 1001).contiguous()           # length not a multiple of 4
    b2, _ = bitstream.unpack_codes(bitstream.pack_codes(odd, (2, 2002)))
    assert torch.equal(b2, odd)
    wav = tmp_path / "clip.wav"
    wavfile.write(wav, 16000, g["pcm"][0])
    out = subprocess.run([sys.executable, "-m", "scripts.compress", "--input", str(wav), "--save_path", str(tmp_path / "out"),
                          "--synthetic", "base", "--num_streams", "6"], capture_output=True, text=True, cwd=os.path.join(ROOT, "efficient-speech-codec_amd"))
    assert out.returncode == 0, out.stderr[-2000:]
    saved = torch.load(tmp_path / "out" / "encoded_9.0kbps_clip.pth")
    assert torch.equal(saved, codes[:1].cpu())
    sr, rec = wavfile.read(tmp_path / "out" / "decoded_9.0kbps_clip.wav")
    assert sr == 16000 and rms(rec, g["audio_s6"][0]) <= AUDIO_TOL


def test_allgather_codes_through_the_c_abi_one_rank_rccl(base):
    """escx_allgather_codes with a real RCCL communicator (one rank: this box has one GPU): the int64 -> int16 -> ncclAllGather ->
    int64 path returns the local codes.  The communicator is created through ctypes on the same RCCL instance libescx resolves."""
    import os
    model, orc, g, cfg = base
    lib, hd = _h(model)
    rccl = None
    for name in ("librccl.so.1", "librccl.so", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = ctypes.CDLL(name); break
        except OSError:
            continue
    assert rccl is not None, "no RCCL library found"
    # (libescx's default search resolves the same SONAME, i.e. the same mapped RCCL instance)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        x = torch.from_numpy(synth.pcm_to_float(g["pcm"])).cuda()
        codes, _ = model.encode(x, 6)
        out = torch.full_like(codes, -1)
        st = torch.cuda.current_stream().cuda_stream
        _native.check(lib.escx_allgather_codes(hd, _ptr(codes), codes.numel(), _ptr(out), 1, comm, ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        assert torch.equal(out, codes)
        assert lib.escx_allgather_codes(hd, _ptr(codes), 0, _ptr(out), 1, comm, None) == _native.ESCX_ERR_INVALID_ARG
        assert lib.escx_allgather_codes(hd, _ptr(codes), codes.numel(), _ptr(out), 1, None, None) == _native.ESCX_ERR_INVALID_ARG
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


_SWEEP_ORACLE = {}       # (model, family, first clip, clips) -> (oracle codes, margins): the precision modes sweep the same clips, the oracle runs once


def _parity_sweep(model_name, n, precision=None):
    """n noise + n voiced clips in batches of 36 against the oracle, every code; a difference must sit on a reference near-tie (margin < 2e-6) in the
    earliest differing stream of its clip, and the later streams of such a clip are verified against the oracle CONTINUED from the device's choice."""
    model, orc, g, cfg = build_models(model_name, precision)
    Trace = __import__("oracle.esc_oracle", fromlist=["Trace"]).Trace
    tot = dict(model=model_name, precision=model.precision, clips=0, exact=0, codes=0, diff=0, under_1e5=0, under_2e6=0)
    min_margin = 1.0
    for fam, fn in (("noise", synth.noise_clip_int16), ("voiced", synth.voiced_clip_int16)):
        for lo in range(0, n, 36):
            k = min(36, n - lo)
            pcm = np.stack([fn(f"sweep-{fam}-{lo + i}", 48000) for i in range(k)])
            x = torch.from_numpy(synth.pcm_to_float(pcm))
            codes, shape = model.encode(x.cuda(), 6)
            key = (model_name, fam, lo, k)
            if key not in _SWEEP_ORACLE:
                tr = Trace()
                oc, _ = orc.encode(x, 6, trace=tr)
                _SWEEP_ORACLE[key] = (oc.numpy(), torch.stack(tr.margins, dim=1).numpy())
            ref, m = _SWEEP_ORACLE[key]
            got = codes.cpu().numpy()
            # every differing code must be a reference near-tie - in its own stream, or in the reference CONTINUED from the device's choice at an
            # attributed near-tie of an earlier stream (the later streams of such a clip see a different residual): nothing stays unverified
            bad, forced, cont = attribute_with_continuation(orc, x, got, ref, m, 6)
            assert not bad, f"{fam} clips {lo}..{lo + k}: " + "\n".join(bad[:10])
            tot["forced"] = tot.get("forced", 0) + forced; tot["continued"] = tot.get("continued", 0) + len(cont)
            tot["clips"] += k; tot["exact"] += int((got == ref).reshape(k, -1).all(1).sum()); tot["codes"] += got.size; tot["diff"] += int((got != ref).sum())
            tot["under_1e5"] += int((m < 1e-5).sum()); tot["under_2e6"] += int((m < 2e-6).sum()); min_margin = min(min_margin, float(m.min()))
            print(f"[sweep {model_name} {tot['precision']} {fam} {lo}..{lo + k}] exact {tot['exact']}/{tot['clips']} clips, smallest reference margin so far {min_margin:.2e}", flush=True)
    print(f"[sweep] {tot}; smallest reference margin {min_margin:.2e}")
    return tot


def _near_tie_budget(tot, strict=False):
    """Every differing code has already been attributed (reference margin < 2e-6 in the earliest differing stream, later streams verified against the oracle
    continued from the device's choice) - this guard bounds HOW MANY near-ties resolve the other way.  VERDICT r5 weak #3 / ADVICE r5: tightened to what was
    observed - at most 1 % of the clips AND at most a sixth of the sweep's codes with a reference margin under 2e-6 (never less than one clip: a single flip is
    fp32 summation-order noise under any arithmetic).  Observed (profiles/r5_parity_sweep_*, r6_*): Base-576 4 clips (0.7 %) of 45 such codes and Base-1152 5 clips
    (0.4 %), Large-576 3 clips (0.5 %) with two fp16 terms; 3 / 576 and 1 / 288 with three bf16 terms; 0 / 576 and 2 / 288 (0.7 %, 25 such codes) on the fp32 MFMA.
    strict: the arm that must stay bit-exact on every clip - ESC-Base on the fp32 MFMA over the 72 always-on clips (bit-exact in every run so far).  The WIDE sweep
    (ESCX_PARITY_SWEEP=288) is not strict: with the oracle on 128 host threads that arm was 576 / 576, with the oracle on 16 threads (tests/conftest.py) it is 574 / 576 + 2
    attributed near-ties - the product's codes are the same in both runs, it is the ORACLE's fp32 summation order (torch's CPU matmul blocking) that resolves two reference
    near-ties the other way.  A code on a margin under 2e-6 is not determined by the reference's arithmetic either."""
    flipped = tot["clips"] - tot["exact"]
    if strict:
        assert flipped == 0, tot
    assert flipped <= max(1, min((tot["clips"] + 99) // 100, tot["under_2e6"] // 6)), tot


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
def test_parity_sweep_base_always_on(precision):
    """the sweep is part of every GPU run, in every precision mode - ESC-Base against the oracle, every code: 36 noise + 36 voiced
    3 s clips, the same in every mode (the oracle runs once) (ESCX_PARITY_SWEEP=<clips per family> widens all three: 288 is the sweep
    whose log is committed under profiles/).  The fp32-MFMA arm is the strict one at the always-on width: ESC-Base has been bit-exact on each of these 72 clips in every run."""
    n = int(os.environ.get("ESCX_PARITY_SWEEP", "36"))
    _near_tie_budget(_parity_sweep("base", n, precision), strict=(precision == "fp32" and n <= 36))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
def test_parity_sweep_large_always_on(precision):
    """The same for ESC-Large (depth 4, where the attributed near-tie clips of the fp32-MFMA sweep live): 18 + 18 clips in every mode
    (ESCX_PARITY_SWEEP_LARGE=<clips per family> widens it; 144 = the committed sweep)."""
    _near_tie_budget(_parity_sweep("large", int(os.environ.get("ESCX_PARITY_SWEEP_LARGE", "18")), precision))


def _scale_keys(sd, factor, which):
    """Multiplies the named parameter families of EVERY SwinBlock / scale change in place (range-stress checkpoints)."""
    n = 0
    for k in sd:
        if any(k.endswith(w) for w in which):
            sd[k] = sd[k] * factor
            n += 1
    assert n > 0
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["ln_gain_x1e5", "fc1_x1e6", "all_small_x1e-4", "deembed_in_x1e6"])
def test_range_stress_checkpoints(case, precision):
    """VERDICT r5 weak #2 / ADVICE r5 (fp16 range of the two-term split): checkpoints whose activations leave fp16's comfortable range, in every precision mode,
    against the oracle (which loads the SAME checkpoint and computes in fp32 like the reference):
      ln_gain_x1e5     every LayerNorm gamma / beta x 1e5  -> LayerNorm outputs of 1e5 ... 5e5 (> 65504): an unscaled fp16 split gives inf -> NaN -> code 0
                       (the Q / K / V weights are scaled x 1e-5 with it: the attention logits stay O(1) - 1e10 times larger logits would turn every softmax into an argmax that
                       flips on fp32 rounding noise under ANY arithmetic, the reference's included; the split still sees 1e5-sized activations against 1e-5-sized weights)
      fc1_x1e6         every fc1 weight / bias x 1e6       -> GELU outputs of ~1e6
      all_small_x1e-4  LayerNorm gamma / beta and fc1 x 1e-4 -> every split activation ~1e-4: low terms of an unscaled split would be fp16 subnormals (14-bit operands)
      deembed_in_x1e6  last decoder block's fc2 weight / bias x 1e6 -> the un-normalised tokens entering PatchDeEmbed reach ~1e6
    (tagged builds: ESCX_X2_NO_ACT_SCALE=1 switches the activation scales of the two-term mode off - the first two cases then fail, which is what makes this a test:
    profiles/r6_range_rule_off.txt)
    Requirement (the same rule as everywhere): finite outputs, no code differs from the oracle except on an attributed reference near-tie, audio of the reference's
    codes within 1e-4 RMS relative to the reference audio's own RMS (the stressed networks emit audio of arbitrary scale)."""
    ln = ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "subsample.norm.weight", "subsample.norm.bias")
    fc1 = ("mlp.linear_1.weight", "mlp.linear_1.bias")

    def edit(sd):
        if case == "ln_gain_x1e5":
            _scale_keys(sd, 1e5, ln)
            _scale_keys(sd, 1e-5, ("attn.qkv.weight",))
        elif case == "fc1_x1e6":
            _scale_keys(sd, 1e6, fc1)
        elif case == "all_small_x1e-4":
            _scale_keys(sd, 1e-4, ln + fc1)
        else:
            _scale_keys(sd, 1e6, ("decoder.post_nn.swint_blocks.1.mlp.linear_2.weight", "decoder.post_nn.swint_blocks.1.mlp.linear_2.bias"))

    model, orc, g, cfg = build_models("base", precision, edit)
    Trace = __import__("oracle.esc_oracle", fromlist=["Trace"]).Trace
    pcm = np.stack([synth.noise_clip_int16(f"range-{i}", 48000) for i in range(2)] + [synth.voiced_clip_int16(f"range-v{i}", 48000) for i in range(2)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    codes, shape = model.encode(x.cuda(), 6)
    key = ("range", case)
    if key not in _SWEEP_ORACLE:
        tr = Trace()
        oc, _ = orc.encode(x, 6, trace=tr)
        _SWEEP_ORACLE[key] = (oc.numpy(), torch.stack(tr.margins, dim=1).numpy(), orc.decode(oc, shape).numpy())
    ref, m, ow = _SWEEP_ORACLE[key]
    got = codes.cpu().numpy()
    assert got.min() >= 0 and got.max() < 1024
    bad, forced, cont = attribute_with_continuation(orc, x, got, ref, m, 6)
    same = (got == ref).reshape(len(pcm), -1).all(1)
    print(f"[range {case} {precision}] {int(same.sum())}/{len(pcm)} clips bit-exact, reference min margin {m.min():.2e}; " + mismatch_summary(got, ref, m))
    assert not bad, "\n".join(bad[:10])
    wave = model.decode(torch.from_numpy(ref).cuda(), shape)
    assert torch.isfinite(wave).all(), "non-finite audio"
    ref_rms = float(np.sqrt(np.mean(ow.astype(np.float64) ** 2)))
    assert np.isfinite(ref_rms) and rms(wave.cpu().numpy(), ow) <= AUDIO_TOL * max(ref_rms, 1.0), (rms(wave.cpu().numpy(), ow), ref_rms)


def _ab_arms(arms, tmp_path, extra_env=None):
    import subprocess, sys
    from conftest import ROOT
    ab = os.path.join(ROOT, "tools", "ab.py")
    got = {}
    for name, env in arms.items():
        out = str(tmp_path / f"{name}.npz")
        e = dict(os.environ, AB_STEPS="1", AB_BATCH="8", AB_GROUPS="", AB_DUMP=out, **(extra_env or {}), **env)
        r = subprocess.run([sys.executable, ab, "--child"], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "AB_RESULT" in r.stdout, f"{name}: {r.stderr[-800:]}"
        got[name] = np.load(out)
    return got


@pytest.mark.gpu
def test_fallback_kernel_forms_against_the_default(tmp_path):
    """The PRODUCT library keeps a handful of fallback forms behind switches (each is read once per process, hence one child process per arm,
    tools/ab.py).  Round 5: the fused product-VQ kernel (fused_pvq.h) and the de-quantisation tables must reproduce the three-launch form
    (split-K GEMM -> search -> up-projection on the MFMA) bit for bit, codes AND audio, and so must the GEMM-engine up-projection;
    attn_gs_off (the C = 384 attention without the head-group split: another projection order) keeps the codes, audio within 1e-6 RMS."""
    arms = {"default": {}, "pvq_three_launch": {"ESCX_PVQ_FUSED": "0", "ESCX_PVQ_TABLE": "0"}, "pvq_fused_mfma_up": {"ESCX_PVQ_TABLE": "0"},
            "pvq_tables_only": {"ESCX_PVQ_FUSED": "0"}, "pvq_up_engine": {"ESCX_PVQ_FUSED": "0", "ESCX_PVQ_TABLE": "0", "ESCX_PVQ_UP_KERNEL": "0"},
            "attn_gs_off": {"ESCX_ATTN_GS_TOKENS": "0"}, "all_fp32_mfma": {"ESCX_PRECISION": "fp32"}, "per_family_switches_fp32": {"ESCX_MLP_X3": "0", "ESCX_ATTN_X3": "0", "ESCX_ROWGEMM_X3": "0"},
            "bf16_three_terms": {"ESCX_PRECISION": "bf16x3"}, "bf16_three_terms_r5_spelling": {"ESCX_X3_TERMS": "3"}}
    got = _ab_arms(arms, tmp_path)
    ref = got["default"]
    for name in ("pvq_three_launch", "pvq_fused_mfma_up", "pvq_tables_only", "pvq_up_engine"):
        assert np.array_equal(got[name]["codes"], ref["codes"]) and np.array_equal(got[name]["wave"], ref["wave"]), f"{name} is not bit-identical to the default"
    # all_fp32_mfma: every contraction on the fp32 MFMA; bf16_three_terms: the exact three-term bf16 split instead of the default two fp16 terms: other summation orders
    # the environment variables only set the DEFAULT mode of new handles (escx_set_precision is the contract); both spellings select the same arithmetic
    assert np.array_equal(got["all_fp32_mfma"]["wave"], got["per_family_switches_fp32"]["wave"]) and np.array_equal(got["bf16_three_terms"]["wave"], got["bf16_three_terms_r5_spelling"]["wave"])
    for name in ("attn_gs_off", "all_fp32_mfma", "bf16_three_terms"):
        assert np.array_equal(got[name]["codes"], ref["codes"]), f"{name}: codes differ from the default"
        rms = float(np.sqrt(np.mean((got[name]["wave"].astype(np.float64) - ref["wave"]) ** 2)))
        assert rms <= 1e-6, f"{name}: audio rms {rms}"


@pytest.mark.gpu
def test_experimental_kernel_forms_against_the_default(tmp_path):
    """The measured-and-rejected kernel forms of rounds 3-4 are compiled into TAGGED builds only (tune_env.h: ESCX_BUILD_TAG=exp
    ESCX_EXTRA_CXXFLAGS=-DESCX_EXPERIMENTAL python efficient-speech-codec_amd/build.py -> libescx_exp.so).  When that library is present they are
    checked against the default library: the in-launch combine of the hidden-split MLP, the combine-on-load consumers and the specialised
    down-projection kernel bit for bit; the weight-stationary / shared-rows merge-split kernels pin their LayerNorm contraction explicitly, so
    they may differ in low-order bits: identical codes, audio within 1e-6 RMS."""
    from conftest import ROOT
    if not os.path.exists(os.path.join(ROOT, "efficient-speech-codec_amd", "esc", "lib", "libescx_exp.so")):
        pytest.skip("no tagged experimental build (libescx_exp.so): the product library does not contain the rejected forms")
    # the rejected forms are variants of the fp32-MFMA kernels: both sides run with the split-operand kernels off (ESCX_MLP_X3=0, ESCX_ATTN_X3=0)
    f32 = {"ESCX_MLP_X3": "0", "ESCX_ATTN_X3": "0", "ESCX_ROWGEMM_X3": "0"}
    ref = _ab_arms({"default": {}}, tmp_path, f32)["default"]
    arms = {"mlp_fused_combine": {"ESCX_MLP_FUSED_COMBINE": "1"}, "combine_on_load": {"ESCX_COMBINE_ON_LOAD": "1"},
            "rowgemm_ws": {"ESCX_ROWGEMM_WS": "4"}, "rowgemm_xs": {"ESCX_ROWGEMM_XS": "1"},
            "pvq_down_kernel": {"ESCX_PVQ_FUSED": "0", "ESCX_PVQ_DOWN_KERNEL": "1"}}
    got = _ab_arms(arms, tmp_path, dict(f32, ESCX_LIB_TAG="exp"))
    for name in ("mlp_fused_combine", "combine_on_load", "pvq_down_kernel"):
        assert np.array_equal(got[name]["codes"], ref["codes"]) and np.array_equal(got[name]["wave"], ref["wave"]), f"{name} is not bit-identical to the default"
    for name in ("rowgemm_ws", "rowgemm_xs"):
        assert np.array_equal(got[name]["codes"], ref["codes"]), f"{name}: codes differ from the default"
        rms = float(np.sqrt(np.mean((got[name]["wave"].astype(np.float64) - ref["wave"]) ** 2)))
        assert rms <= 1e-6, f"{name}: audio rms {rms}"


@pytest.mark.gpu
def test_batch_sizes_alternating_between_one_and_two_parts():
    """Batches under 6 clips run as one part, larger ones as two parts on two streams: the clips a part holds are not monotone in the batch
    (8 -> 2 x 4, 5 -> 1 x 5), the workspace has to grow per part; every clip keeps its codes and its audio in every batch it appears in."""
    model, orc, g, cfg = build_models("base")
    pcm = np.stack([synth.noise_clip_int16(f"parts-{i}", 24000) for i in range(12)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).cuda()
    ref_c, shape = model.encode(x, 6)
    ref_w = model.decode(ref_c, shape)
    for B in (8, 5, 4, 7, 1, 6, 3, 12, 2):
        c, shp = model.encode(x[:B].contiguous(), 6)
        w = model.decode(c, shp)
        assert torch.equal(c, ref_c[:B]) and torch.equal(w, ref_w[:B]), B
