"""Training step: training-mode forward, losses and every parameter gradient.

Fixtures: tests/golden/train.npz = the REAL reference's training-mode forward + its own loss classes + loss.mean().backward()
(oracle/gen_train_golden.py): per-clip losses, codes, the gradient norm of every parameter, a few full gradients.
  * CPU (`not gpu`): the oracle's training restatement (plain autograd over oracle/esc_oracle.py) is pinned to those fixtures.
  * GPU: the HIP training step (through esc.ESC in train mode -> libescx escx_train_forward / escx_train_backward, and the HIP loss
    modules) against the fixtures and, parameter by parameter, against the oracle's gradients.
Tolerances: losses 1e-5 relative.  Gradients: 1e-4 relative RMS per parameter wherever the loss is well conditioned (the VQ losses
and any smooth functional of the outputs: measured ~1e-6).  The trainer's spectral losses are NOT well conditioned in fp32: the power-law
|x|^0.3 of ComplexSTFTLoss and the S/|S|, log10(clamp(mel)) terms of the mel loss have unbounded derivatives near zero, so a 2e-7 relative
change of the reconstruction moves d loss / d recon by ~1e-4.  The reference's OWN fp32 gradients differ from its fp64 gradients by a
median 1.5e-4 (max 1.2e-3 per parameter on the tiny config, measured with the oracle; `test_reference_fp32_gradient_noise_floor`
documents it).  For those losses the HIP gradients are therefore compared with the fp64 oracle, per parameter, against a bound tied to
that measured noise floor (and to the fp32 fixtures at the same level), not against an fp32 value taken as exact.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, synth_state
from esc import synth

LOSS_RTOL = 1e-5
GRAD_TOL = 1e-4


def _clips(g, name):
    tags = json.loads(str(g[f"{name}_tags"]))
    n = json.loads(str(g["n_samples_json"]))[name]
    pcm = np.stack([synth.noise_clip_int16(tags[0], n), synth.voiced_clip_int16(tags[1], n)])
    return torch.from_numpy(synth.pcm_to_float(pcm))


def _cfg(name):
    return json.loads(str(load_golden(name)["config_json"]))


def _oracle_step(name, S, freeze, x, dtype=torch.float32, smooth=None):
    """One training step of the oracle under autograd.  smooth = (A, R, wc, wb): replace the trainer's losses by the smooth functional
    mean_b[ wc*cm + wb*cb + <recon_audio, A> + <recon_feat, R> ] (a well-conditioned probe of the whole backward pass)."""
    from oracle import esc_oracle as O
    sd = {k: (v.to(dtype).clone().requires_grad_(True) if (v.is_floating_point() and not k.endswith(".window")) else
              (v.to(dtype) if v.is_floating_point() else v)) for k, v in synth_state(name).items()}
    orc = O.EscOracle(_cfg(name), sd, keep_graph=True)
    out = orc.forward_train(x.to(dtype), S, freeze)
    if smooth is None:
        ls = O.training_loss(out)
    else:
        A, R, wc, wb = smooth
        total = out["cm_loss"] * wc + out["cb_loss"] * wb + (out["recon_audio"] * A.to(dtype)).sum(1) + (out["recon_feat"] * R.to(dtype)).sum((1, 2, 3))
        ls = {"loss": total, "scalar": total.mean()}
    ls["scalar"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.is_floating_point() and v.requires_grad}
    return out, ls, grads


def _rel_rms(a, b, floor):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(float(np.sqrt(np.mean(b ** 2))), floor))


def _check_against_fixture(g, tag, keys, losses, codes, grads, grad_tol):
    for k in ("cm", "cb", "mel", "stft", "loss"):
        np.testing.assert_allclose(losses[k], g[f"{tag}_{k}"], rtol=LOSS_RTOL, atol=1e-7, err_msg=f"{tag} {k}")
    assert np.array_equal(np.asarray(codes), g[f"{tag}_codes"].astype(np.int64)), f"{tag}: codes differ from the reference"
    ref_norm = g[f"{tag}_gnorm"]
    scale = float(np.sqrt((ref_norm ** 2).sum()))
    for k, rn in zip(keys, ref_norm):
        gn = float(np.linalg.norm(np.asarray(grads[k], np.float64)))
        assert abs(gn - rn) <= grad_tol * max(rn, 1e-6 * scale) + 1e-12, f"{tag}: |grad {k}| = {gn} vs reference {rn}"
    for key in [f for f in g.files if f.startswith(f"{tag}_g::")]:
        k = key.split("::", 1)[1]
        ref = g[key]
        floor = 1e-6 * scale / np.sqrt(ref.size)
        err = _rel_rms(grads[k], ref, floor)
        assert err <= grad_tol, f"{tag}: gradient of {k} rel rms {err:.3e}"


# ------------------------------------------------------------------------------------------------ CPU: the oracle is pinned
@pytest.mark.parametrize("name,cases", [("tiny", None), ("base", [(3, False)]), ("large", None)])
def test_oracle_training_step_matches_reference(name, cases):
    g = load_golden("train")
    keys = json.loads(str(g[f"{name}_keys"]))
    x = _clips(g, name)
    for S, freeze in (cases or json.loads(str(g["cases_json"]))[name]):
        out, ls, grads = _oracle_step(name, S, freeze, x)
        losses = {"cm": out["cm_loss"].detach().numpy() if torch.is_tensor(out["cm_loss"]) else np.zeros(2, np.float32),
                  "cb": out["cb_loss"].detach().numpy() if torch.is_tensor(out["cb_loss"]) else np.zeros(2, np.float32),
                  "mel": ls["mel_loss"].detach().numpy(), "stft": ls["stft_loss"].detach().numpy(), "loss": ls["loss"].detach().numpy()}
        _check_against_fixture(g, f"{name}_s{S}_f{int(freeze)}", keys, losses, out["codes"].numpy(), {k: v.numpy() for k, v in grads.items()}, 2e-5)


def test_train_fixture_covers_every_parameter():
    g = load_golden("train")
    from esc.models import make_model
    for name in ("tiny", "base", "large"):
        keys = json.loads(str(g[f"{name}_keys"]))
        model = make_model(_cfg(name))
        assert keys == [k for k, _ in model.named_parameters()] or set(keys) == {k for k, _ in model.named_parameters()}


# ------------------------------------------------------------------------------------------------ GPU: the HIP training step
def _product_step(name, S, freeze, x, w, smooth=None, x_feat=None):
    from esc.models import make_model
    from esc.modules import ComplexSTFTLoss, MelSpectrogramLoss
    model = make_model(_cfg(name))
    model.load_state_dict(synth_state(name))
    model = model.cuda().train()
    out = model(**dict(x=x.cuda(), x_feat=x_feat, num_streams=S, freeze_codebook=freeze))
    losses = {"cm": out["cm_loss"].detach().cpu().numpy(), "cb": out["cb_loss"].detach().cpu().numpy()}
    if smooth is None:
        mel = MelSpectrogramLoss()(out["raw_audio"], out["recon_audio"])
        stft = ComplexSTFTLoss()(out["raw_feat"], out["recon_feat"])
        loss = out["cm_loss"] * w["cm_weight"] + out["cb_loss"] * w["cb_weight"] + mel * w["mel_weight"] + stft * w["stft_weight"]
        losses.update(mel=mel.detach().cpu().numpy(), stft=stft.detach().cpu().numpy())
    else:
        A, R, wc, wb = smooth
        loss = out["cm_loss"] * wc + out["cb_loss"] * wb + (out["recon_audio"] * A.cuda()).sum(1) + (out["recon_feat"] * R.cuda()).sum((1, 2, 3))
    loss.mean().backward()
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in model.named_parameters()}
    losses["loss"] = loss.detach().cpu().numpy()
    return model, out, losses, grads


def _smooth_probe(name, x):
    """Fixed random cotangents for recon_audio / recon_feat (different per clip) and VQ-loss weights."""
    cfg = _cfg(name)
    g = torch.Generator().manual_seed(77)
    hop = int(cfg["hop_len"] * cfg["sr"] * 1e-3)
    T = 1 + x.shape[1] // hop
    A = torch.randn(x.shape, generator=g) / x.shape[1]
    R = torch.randn(x.shape[0], 2, cfg["in_freq"], (T // 2) * 2, generator=g) / (cfg["in_freq"] * T)
    return A, R, 0.7, 1.3


def _noise_floor(name, S, freeze, x, draws=3):
    """Per-parameter relative RMS distance between fp32 evaluations of the reference restatement and its fp64 gradients of the trainer's
    loss = how well ANY fp32 evaluation determines each gradient.  One fp32 run is a single draw of that noise and draws differ by up to 10x
    (measured on base, S=3: medians 1.0e-3 / 3.1e-3 / 5.6e-3 / 1.2e-2 for the default run, two runs with every input sample moved by one
    ulp, and a single-threaded run - tools/grad_noise_diag.py), so the floor is the envelope over `draws` realisations: the default run, the
    inputs moved by one ulp with random signs, and the default inputs summed single-threaded."""
    _, _, g64 = _oracle_step(name, S, freeze, x, dtype=torch.float64)
    scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in g64.values())))
    gen = torch.Generator().manual_seed(5)
    floor, medians = {k: 0.0 for k in g64}, []
    for d in range(draws):
        nt = torch.get_num_threads()
        xd = x if d != 1 else x * (1 + (torch.randint(0, 2, x.shape, generator=gen).float() * 2 - 1) * 2.0 ** -23)
        if d == 2:
            torch.set_num_threads(1)
        try:
            _, _, g32 = _oracle_step(name, S, freeze, xd)
        finally:
            torch.set_num_threads(nt)
        e = {k: _rel_rms(g32[k].numpy(), g64[k].numpy(), 1e-6 * scale / np.sqrt(g64[k].numel())) for k in g64}
        medians.append(float(np.median(list(e.values()))))
        floor = {k: max(floor[k], e[k]) for k in g64}
    return g64, floor, scale, max(medians)


def test_reference_fp32_gradient_noise_floor():
    """Documents why the trainer-loss gradients cannot be held to 1e-4 against an fp32 reference: the reference restatement itself, fp32
    vs fp64, same codes, differs by more than that on most parameters - while the VQ-loss path alone is good to 1e-6."""
    g = load_golden("train")
    x = _clips(g, "tiny")
    _, floor, _, med = _noise_floor("tiny", 3, False, x)
    assert 2e-5 < med < 2e-2, med
    _, _, a = _oracle_step("tiny", 3, False, x, smooth=(torch.zeros_like(x), torch.zeros(2, 2, 48, 64), 0.7, 1.3))
    _, _, b = _oracle_step("tiny", 3, False, x, dtype=torch.float64, smooth=(torch.zeros_like(x), torch.zeros(2, 2, 48, 64), 0.7, 1.3))
    errs = [_rel_rms(a[k].numpy(), b[k].numpy(), 1e-9) for k in b if float(b[k].abs().max()) > 0]
    assert max(errs) < 2e-5, max(errs)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base"])
def test_backward_pass_every_gradient_well_conditioned(name):
    """The whole hand-written backward (de-embedding, inverse STFT, every Swin block, merge/split, patch embedding, every quantiser with
    STE / commitment / codebook terms, frozen and masked streams) probed with a smooth functional of the outputs: every parameter
    gradient within 1e-4 relative RMS of the oracle's autograd (measured ~1e-6)."""
    g = load_golden("train")
    x = _clips(g, name)
    keys = json.loads(str(g[f"{name}_keys"]))
    probe = _smooth_probe(name, x)
    for S, freeze in json.loads(str(g["cases_json"]))[name]:
        model, out, losses, grads = _product_step(name, S, freeze, x, None, smooth=probe)
        oout, ols, ograds = _oracle_step(name, S, freeze, x, smooth=probe)
        assert torch.equal(out["codes"].cpu(), oout["codes"])
        np.testing.assert_allclose(losses["loss"], ols["loss"].detach().numpy(), rtol=LOSS_RTOL)
        scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values())))
        worst = (0.0, "")
        for k in keys:
            ref = ograds[k].numpy()
            err = _rel_rms(grads[k], ref, 1e-6 * scale / np.sqrt(ref.size))
            worst = max(worst, (err, k))
            assert err <= GRAD_TOL, f"{name} S={S} freeze={freeze}: gradient of {k} rel rms {err:.3e} (|ref| {np.linalg.norm(ref):.3e})"
        print(f"[smooth {name} S={S} freeze={int(freeze)}] worst parameter gradient rel rms {worst[0]:.2e} ({worst[1]})")


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,L,S,freeze", [("tiny", 3, 1180, 3, False), ("tiny", 1, 1180, 2, True), ("base", 1, 3440, 6, False), ("base", 3, 5040, 4, False)])
def test_backward_pass_padded_windows_and_other_batches(name, B, L, S, freeze):
    """Clip lengths whose token maps are NOT a multiple of the 4 x 4 window (W % 4 == 2: the Swin blocks pad after the norm, padded
    slots take part as keys and carry no gradient back), one-clip and three-clip batches: same well-conditioned probe, every gradient."""
    pcm = np.stack([(synth.voiced_clip_int16 if i % 2 else synth.noise_clip_int16)(f"train-pad-{name}-{i}", L) for i in range(B)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    probe = _smooth_probe(name, x)
    model, out, losses, grads = _product_step(name, S, freeze, x, None, smooth=probe)
    oout, ols, ograds = _oracle_step(name, S, freeze, x, smooth=probe)
    assert torch.equal(out["codes"].cpu(), oout["codes"])
    np.testing.assert_allclose(losses["loss"], ols["loss"].detach().numpy(), rtol=LOSS_RTOL)
    scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values())))
    worst = (0.0, "")
    for k, ref in ograds.items():
        err = _rel_rms(grads[k], ref.numpy(), 1e-6 * scale / np.sqrt(ref.numel()))
        worst = max(worst, (err, k))
        assert err <= GRAD_TOL, f"{name} B={B} L={L}: gradient of {k} rel rms {err:.3e}"
    print(f"[smooth {name} B={B} L={L} S={S} freeze={int(freeze)}] worst gradient rel rms {worst[0]:.2e} ({worst[1]})")


@pytest.mark.gpu
def test_backward_pass_esc_large():
    """ESC-Large (swin_depth 4, BASELINE configs[4]'s model): the same well-conditioned probe, one case, every parameter gradient."""
    g = load_golden("train")
    n = json.loads(str(g["n_samples_json"]))["base"]
    pcm = np.stack([synth.noise_clip_int16("train-large-0", n), synth.voiced_clip_int16("train-large-1", n)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    probe = _smooth_probe("large", x)
    model, out, losses, grads = _product_step("large", 4, False, x, None, smooth=probe)
    oout, ols, ograds = _oracle_step("large", 4, False, x, smooth=probe)
    assert torch.equal(out["codes"].cpu(), oout["codes"])
    np.testing.assert_allclose(losses["loss"], ols["loss"].detach().numpy(), rtol=LOSS_RTOL)
    scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values())))
    worst = (0.0, "")
    for k, ref in ograds.items():
        err = _rel_rms(grads[k], ref.numpy(), 1e-6 * scale / np.sqrt(ref.numel()))
        worst = max(worst, (err, k))
        assert err <= GRAD_TOL, f"large: gradient of {k} rel rms {err:.3e}"
    print(f"[smooth large S=4] {len(ograds)} parameters, worst gradient rel rms {worst[0]:.2e} ({worst[1]})")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "base", "large"])
def test_training_step_losses_and_every_gradient(name):
    """The trainer's step (trainer_no_adv.py:105-115): losses 1e-5 relative to the reference fixtures, codes identical; every parameter
    gradient against the fp64 oracle within 4x the reference restatement's own fp32 noise floor (envelope over three fp32 realisations, see
    _noise_floor; the larger of that parameter's and the median over parameters, +1e-5), the median HIP error within 2x the median noise
    floor, and the gradient norms of the reference fixtures within the same level."""
    g = load_golden("train")
    w = json.loads(str(g["weights_json"]))
    keys = json.loads(str(g[f"{name}_keys"]))
    x = _clips(g, name)
    cases = json.loads(str(g["cases_json"]))[name]
    for S, freeze in (cases if name == "tiny" else [c for c in cases if tuple(c) in ((3, False), (6, False), (6, True))]):
        tag = f"{name}_s{S}_f{int(freeze)}"
        model, out, losses, grads = _product_step(name, S, freeze, x, w)
        assert out["codes"].shape[1] == model.max_streams and out["recon_audio"].shape == x.shape
        assert out["raw_feat"].shape == out["recon_feat"].shape
        for k in ("cm", "cb", "mel", "stft", "loss"):
            np.testing.assert_allclose(losses[k], g[f"{tag}_{k}"], rtol=LOSS_RTOL, atol=1e-7, err_msg=f"{tag} {k}")
        assert np.array_equal(out["codes"].cpu().numpy(), g[f"{tag}_codes"].astype(np.int64))
        g64, floor, scale, med = _noise_floor(name, S, freeze, x)       # envelope over three fp32 realisations; med = their largest median
        worst, errs = (0.0, "", 0.0), []
        for k, rn in zip(keys, g[f"{tag}_gnorm"]):
            ref = g64[k].numpy()
            err = _rel_rms(grads[k], ref, 1e-6 * scale / np.sqrt(ref.size))
            errs.append(err)
            bound = 4.0 * max(floor[k], med) + 1e-5
            worst = max(worst, (err / bound, k, err))
            assert err <= bound and err <= max(5e-2, 2.0 * floor[k]), f"{tag}: gradient of {k} rel rms {err:.3e} vs fp64, reference fp32 noise floor {floor[k]:.3e}"
            gn = float(np.linalg.norm(np.asarray(grads[k], np.float64)))
            # the fixture is the fp32 reference: both sides carry their own noise (HIP error + the reference's floor)
            assert abs(gn - rn) <= 2.0 * bound * max(rn, 1e-6 * scale) + 1e-12, f"{tag}: |grad {k}| = {gn} vs reference fixture {rn}"
        print(f"[{tag}] vs fp64 oracle: HIP median rel rms {np.median(errs):.2e} max {max(errs):.2e}; reference-fp32 noise floor median {med:.2e} "
              f"max {max(floor.values()):.2e}; worst (error / bound) {worst[0]:.2f} at {worst[1]}")
        assert np.median(errs) <= 2.0 * med + 1e-5          # as a whole, the HIP gradients sit at the fp32 reference's own distance from the truth


@pytest.mark.gpu
def test_loss_modules_against_the_oracle():
    """ComplexSTFTLoss / MelSpectrogramLoss (HIP) vs the oracle's restatement of generator_loss.py: values and d loss / d reconstruction."""
    from oracle import esc_oracle as O
    from esc.modules import ComplexSTFTLoss, MelSpectrogramLoss
    torch.manual_seed(5)
    raw = 0.1 * torch.randn(3, 6000); rec = (raw + 0.03 * torch.randn(3, 6000)).requires_grad_(True)
    rec_g = rec.detach().cuda().requires_grad_(True)
    wts = torch.tensor([0.5, 1.0, 2.0])
    got = MelSpectrogramLoss()(raw.cuda(), rec_g)
    (got * wts.cuda()).sum().backward()
    ref = O.mel_spectrogram_loss(raw, rec)
    (ref * wts).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=LOSS_RTOL)
    assert _rel_rms(rec_g.grad.cpu().numpy(), rec.grad.numpy(), 1e-12) <= GRAD_TOL
    fa = torch.randn(2, 2, 24, 33); fb = (fa + 0.2 * torch.randn(2, 2, 24, 33)).requires_grad_(True)
    fa[0, 0, 0, :4] = 0.0
    fb_g = fb.detach().cuda().requires_grad_(True)
    got = ComplexSTFTLoss()(fa.cuda(), fb_g)
    got.sum().backward()
    ref = O.complex_stft_loss(fa, fb)
    ref.sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=LOSS_RTOL)
    assert _rel_rms(fb_g.grad.cpu().numpy(), fb.grad.numpy(), 1e-12) <= GRAD_TOL


@pytest.mark.gpu
def test_flat_adamw_and_clipping_match_torch():
    """escx_grad_norm_clip + escx_adamw_step on flat buffers vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (trainer_no_adv.py:116-117)."""
    import ctypes
    from esc import _native
    lib = _native.load()
    torch.manual_seed(3)
    n = 100_003
    p0 = torch.randn(n); grads = [torch.randn(n) * s for s in (0.01, 3.0, 0.2)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.8, 0.95), eps=1e-8, weight_decay=0.05)
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    aux = torch.zeros(2 + 1024, device="cuda")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    for step, gcpu in enumerate(grads, 1):
        ref.grad = gcpu.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        gg = gcpu.cuda()
        _native.check(lib.escx_grad_norm_clip(ptr(gg), n, 0.5, ptr(aux), None))
        _native.check(lib.escx_adamw_step(ptr(p), ptr(gg), ptr(m), ptr(v), n, step, 1e-3, 0.8, 0.95, 1e-8, 0.05, ptr(aux), None))
        torch.cuda.synchronize()
        assert abs(float(aux[0]) - float(tn)) <= 1e-5 * float(tn)
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-7)


@pytest.mark.gpu
def test_eval_after_optimizer_steps_uses_the_updated_weights():
    """in-place parameter updates (optimizer.step) must be picked up without refresh_weights(); train -> eval -> train
    round trips keep working and eval after training equals a fresh model loaded with the updated state_dict."""
    from esc.models import make_model
    cfg = _cfg("tiny")
    g = load_golden("train")
    x = _clips(g, "tiny").cuda()
    model = make_model(cfg); model.load_state_dict(synth_state("tiny")); model = model.cuda().eval()
    c0, shp = model.encode(x, 3)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    model.train()
    for _ in range(2):
        out = model(**dict(x=x, x_feat=None, num_streams=3, freeze_codebook=False))
        (out["cm_loss"] + out["cb_loss"] + (out["recon_audio"] - x).pow(2).mean(1)).mean().backward()
        opt.step(); opt.zero_grad()
    model.eval()
    c1, _ = model.encode(x, 3)
    w1 = model.decode(c1, shp)
    fresh = make_model(cfg); fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}); fresh = fresh.cuda().eval()
    cf, _ = fresh.encode(x, 3)
    assert torch.equal(c1, cf) and torch.equal(w1, fresh.decode(cf, shp))
    assert not torch.equal(w1, model.decode(c0, shp)) or not torch.equal(c0, c1)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.0)                                   # in-place write outside any optimizer
    assert torch.equal(model.encode(x, 3)[0], c1)
    import copy
    twin = copy.deepcopy(model)                           # ADVICE: deepcopy / pickling of a used model
    assert torch.equal(twin.encode(x, 3)[0], c1)


@pytest.mark.gpu
def test_flat_gradients_and_adam_moments_survive_a_handle_rebuild():
    """load_state_dict / model.to() between backward and optimiser step rebuilds the native handle and the flat parameter buffer.
    The flat gradient buffer (with what the backward accumulated) and FlatAdamW's moments must carry over: the run with a rebuild in the
    middle of every step equals the undisturbed run bit for bit."""
    from esc.models import make_model
    from esc.modules import ComplexSTFTLoss, MelSpectrogramLoss
    from esc.optim import FlatAdamW
    g = load_golden("train")
    x = _clips(g, "tiny").cuda()
    mel, stft = MelSpectrogramLoss(), ComplexSTFTLoss()
    runs = []
    for disturb in (False, True):
        model = make_model(_cfg("tiny")); model.load_state_dict(synth_state("tiny")); model = model.cuda().train()
        opt = FlatAdamW(model, lr=1e-3, max_grad_norm=0.5)
        for n in range(3):
            out = model(x=x, x_feat=None, num_streams=3, freeze_codebook=False)
            (out["cm_loss"] + out["cb_loss"] + mel(out["raw_audio"], out["recon_audio"]) + stft(out["raw_feat"], out["recon_feat"])).mean().backward()
            if disturb:
                model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})        # dirty: handle + flat buffer are rebuilt lazily
                model.to("cuda")
            opt.step(); opt.zero_grad()
        assert opt.t == 3 and float(opt.last_grad_norm) > 0
        runs.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    assert all(torch.equal(runs[0][k], runs[1][k]) for k in runs[0])
    init = synth_state("tiny")
    assert sum(not torch.equal(v.cpu(), init[k]) for k, v in runs[0].items() if v.is_floating_point() and not k.endswith(".window")) > 50      # it did train


@pytest.mark.gpu
def test_native_state_guards_of_the_training_path():
    """ADVICE r2, C-ABI level: (1) after a device-side weight refresh (a training forward) the inference entry points must not decode with a stale
    folded de-embedding.  Round 4: the refresh re-folds it ON THE DEVICE (fp64, the host fold's summation order), so the decode is valid at once
    and equals - bit for bit - the decode after the host round trip escx_load_flat_params(full=1); (2) the handle keeps ONE tape: a backward that
    belongs to an earlier forward raises instead of consuming the later forward's activations."""
    import ctypes
    from esc import _native
    from esc.models import make_model
    g = load_golden("train")
    x = _clips(g, "tiny").cuda()
    model = make_model(_cfg("tiny")); model.load_state_dict(synth_state("tiny")); model = model.cuda().eval()
    codes, shape = model.encode(x, 3)
    model.train()
    out_a = model(x=x, x_feat=None, num_streams=3, freeze_codebook=False)
    lib, hd = model._handle(x.device, for_training=True)
    wave = torch.empty((x.shape[0], model.hop_length * (2 * shape[1] - 1)), device="cuda")
    _native.check(lib.escx_decode(hd, codes.data_ptr(), codes.shape[0], 3, shape[0], shape[1], wave.data_ptr(), None, None))
    torch.cuda.synchronize()
    wave_device_fold = wave.clone()
    flat, _ = model.flat_buffers(x.device)
    _native.check(lib.escx_load_flat_params(hd, ctypes.c_void_p(flat.data_ptr()), 1, None))          # host fold of the same weights
    assert lib.escx_decode(hd, codes.data_ptr(), codes.shape[0], 3, shape[0], shape[1], wave.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(wave_device_fold, wave), "device-side fold of the de-embedding differs from the host fold"
    assert torch.equal(wave, model.eval().decode(codes, shape))
    # (2) two graphs alive at once
    model.train()
    out_a = model(x=x, x_feat=None, num_streams=3, freeze_codebook=False)
    out_b = model(x=x, x_feat=None, num_streams=2, freeze_codebook=False)
    with pytest.raises(RuntimeError, match="another training forward"):
        out_a["recon_audio"].sum().backward()
    out_b["recon_audio"].sum().backward()                      # the last forward's graph is fine
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_train_harness_schedules_and_stream_sampling():
    """scripts/train.py host logic: quantisation dropout (scripts/utils.py:11-25) and the four learning-rate schedules."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from scripts.train import lr_at, sample_streams
    rng = np.random.default_rng(0)
    assert all(sample_streams(rng, 0.0, 6) == 6 for _ in range(20))
    draws = [sample_streams(rng, 1.0, 6) for _ in range(600)]
    assert set(draws) == {1, 2, 3, 4, 5, 6}
    with pytest.raises(AssertionError):
        sample_streams(rng, 1.5, 6)
    assert lr_at(10, 1e-4, "constant", 100, 0) == 1e-4
    assert lr_at(1, 1e-4, "constant_warmup", 100, 10) == pytest.approx(1e-5) and lr_at(50, 1e-4, "constant_warmup", 100, 10) == 1e-4
    assert lr_at(5, 1e-4, "cosine_warmup", 100, 10) == pytest.approx(5e-5) and lr_at(100, 1e-4, "cosine_warmup", 100, 10) == pytest.approx(0.0, abs=1e-12)
    assert lr_at(55, 1e-4, "cosine_warmup", 100, 10) == pytest.approx(5e-5)
    assert lr_at(1000, 1e-4, "exponential_decay", 0, 0) == pytest.approx(1e-4 * 0.999996 ** 1000)
    with pytest.raises(ValueError):
        lr_at(0, 1e-4, "linear", 1, 0)
    # the schedulers the reference builds (scripts/utils.py:52-65: transformers' schedules and ExponentialLR), stepped once per optimiser step
    transformers = pytest.importorskip("transformers")
    for kind, total, warm in (("constant", 50, 0), ("constant_warmup", 50, 7), ("cosine_warmup", 60, 9), ("exponential_decay", 50, 0)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], 3e-4)
        sch = {"constant": lambda: transformers.get_constant_schedule(opt),
               "constant_warmup": lambda: transformers.get_constant_schedule_with_warmup(opt, num_warmup_steps=warm),
               "cosine_warmup": lambda: transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total),
               "exponential_decay": lambda: torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999996)}[kind]()
        for n in range(total):
            assert opt.param_groups[0]["lr"] == pytest.approx(lr_at(n, 3e-4, kind, total, warm), rel=1e-9, abs=1e-15), (kind, n)
            opt.step(); sch.step()


def test_generator_schedule_follows_the_reference_loop():
    """the generator's learning rate follows the scheduler in BOTH trainers (trainer_no_adv.py:118, trainer_adv.py:92); the optimiser is
    renewed at step pretraining_steps + 1 (trainer_no_adv.py:76-79, trainer_adv.py:140-143) and the renewed one runs at the constant base rate
    (the scheduler stays bound to the discarded optimiser).  Replayed against the reference's own loop structure with torch objects."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from scripts.train import _Schedule, scheduler_state
    transformers = pytest.importorskip("transformers")
    base, total, warm, pre = 2e-4, 30, 6, 4
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], base)
    sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)     # stays bound to the FIRST optimiser
    mine = _Schedule(base, "cosine_warmup", total, warm, pre)
    for n in range(total):                                    # the reference's loop body: renew, step, scheduler.step()
        renewed = pre > 0 and n == pre + 1
        if renewed:
            opt = torch.optim.AdamW([p], base)
        assert mine.renew_at(n) == renewed
        assert mine.lr(n) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-9, abs=1e-15), n
        opt.step(); sch.step()
    assert not any(_Schedule(base, "constant", total, 0, 0).renew_at(n) for n in range(5))
    with pytest.raises(ValueError):
        _Schedule(base, "linear", total, 0, 0)
    st = scheduler_state("cosine_warmup", base, 10, total, warm)
    assert st["last_epoch"] == 10 and st["base_lrs"] == [base] and st["_last_lr"][0] == pytest.approx(sch.state_dict()["base_lrs"][0] * 0.5 * (1 + np.cos(np.pi * 4 / 24)))


@pytest.mark.gpu
def test_checkpoint_layout_is_the_reference_trainers():
    """a checkpoint written by scripts/train.py has the reference's keys (trainer_adv.py:159-168) and its optimiser states are
    torch.optim.AdamW state_dicts: a REAL torch AdamW over model.parameters() loads them (what the reference's --pretrain_ckp path does,
    trainer_adv.py:118) and continues exactly like FlatAdamW; a torch-written state loads back into FlatAdamW."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from esc.models import Discriminator, make_model
    from esc.optim import FlatAdamW
    from scripts.train import AdvStepper, checkpoint
    g = load_golden("train")
    x = _clips(g, "tiny").cuda()
    model = make_model(_cfg("tiny")); model.load_state_dict(synth_state("tiny")); model = model.cuda()
    disc = Discriminator(sample_rate=16000, periods=[2], fft_sizes=[256]).cuda()
    st = AdvStepper(model, disc, lr=1e-3, dropout_rate=0.0, scheduler="constant_warmup", warmup_steps=2, total_steps=10)
    for n in range(3):
        st.step(x, n)
    ck = checkpoint(st, 2, "constant_warmup", 10, 2)
    assert list(ck) == ["step", "model_state_dict", "model_disc_state_dict", "optimizer_state_dict", "optimizer_disc_state_dict", "scheduler_state_dict", "best_perf"]
    assert ck["step"] == 2 and ck["scheduler_state_dict"]["last_epoch"] == 3 and ck["best_perf"] == -1
    for net, key in ((model, "optimizer_state_dict"), (disc, "optimizer_disc_state_dict")):
        osd = ck[key]
        n_params = len(list(net.parameters()))
        assert sorted(osd["state"]) == list(range(n_params)) and osd["param_groups"][0]["params"] == list(range(n_params))
        assert all(v["exp_avg"].shape == p.shape and float(v["step"]) == 3.0 for v, p in zip(osd["state"].values(), net.parameters()))
    # a real torch AdamW takes the state and makes the same next step as FlatAdamW (same gradients, no clipping)
    twin = make_model(_cfg("tiny")); twin.load_state_dict(ck["model_state_dict"]); twin = twin.cuda().train()
    topt = torch.optim.AdamW(twin.parameters(), lr=1e-3)
    topt.load_state_dict(ck["optimizer_state_dict"])
    gen = torch.Generator().manual_seed(5)
    grads = [torch.randn(p.shape, generator=gen).cuda() * 1e-2 for p in model.parameters()]
    st.opt_g.max_grad_norm = None
    st.opt_g.lr = 1e-3
    for p, q, gr in zip(model.parameters(), twin.parameters(), grads):
        p.grad.copy_(gr); q.grad = gr.clone()
    st.opt_g.step(); topt.step()
    worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(model.parameters(), twin.parameters()))
    assert worst < 2e-7, worst
    # and back: torch's state (after its step) into a fresh FlatAdamW
    back = FlatAdamW(twin, lr=1e-3)
    back.load_state_dict(topt.state_dict())
    assert back.t == 4
    for (_, a), (_, b) in zip(sorted(back.state_dict()["state"].items()), sorted(st.opt_g.state_dict()["state"].items())):
        # torch's lerp_ and the flat kernel's beta * m + (1 - beta) * g round differently in the last bit
        assert float((a["exp_avg"] - b["exp_avg"]).abs().max()) <= 1e-6 * float(b["exp_avg"].abs().max()) + 1e-12
        assert float((a["exp_avg_sq"] - b["exp_avg_sq"]).abs().max()) <= 1e-6 * float(b["exp_avg_sq"].abs().max()) + 1e-15
    with pytest.raises(ValueError):
        back.load_state_dict({"state": {}, "param_groups": [dict(osd["param_groups"][0], params=[0, 1])]})


@pytest.mark.gpu
def test_flat_adamw_continues_a_torch_state_with_gradient_less_parameters():
    """torch.optim.AdamW keeps `step` PER PARAMETER and skips parameters whose grad is None (no decay, no state entry before the first
    gradient).  A state written by such a loop - differing step counts, missing entries - loads into FlatAdamW, and the next steps (with the
    same parameters still gradient-less) follow torch's trajectory: per-parameter bias corrections, untouched inactive parameters."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from esc.models import make_model
    from esc.optim import FlatAdamW
    twin = make_model(_cfg("tiny")); twin.load_state_dict(synth_state("tiny")); twin = twin.cuda().train()
    names = [k for k, _ in twin.named_parameters()]
    late = {k for k in names if k.startswith("quantizers.2.")}          # first gradient arrives at the third torch step
    never = {k for k in names if k.startswith("quantizers.1.vqs.")}      # never sees a gradient
    assert late and never
    topt = torch.optim.AdamW(twin.parameters(), lr=1e-3, betas=(0.8, 0.9), weight_decay=0.05)
    gen = torch.Generator().manual_seed(11)
    def grads():
        return {k: torch.randn(p.shape, generator=gen).cuda() * 1e-2 for k, p in twin.named_parameters()}
    for n in range(3):
        g = grads()
        for k, p in twin.named_parameters():
            p.grad = None if (k in never or (k in late and n < 2)) else g[k]
        topt.step()
    tsd = topt.state_dict()
    steps = {float(v["step"]) for v in tsd["state"].values()}
    assert steps == {1.0, 3.0} and len(tsd["state"]) == len(names) - len(never)
    model = make_model(_cfg("tiny")); model.load_state_dict(twin.state_dict()); model = model.cuda().train()
    opt = FlatAdamW(model, lr=5e-4)
    opt.load_state_dict(tsd)
    assert (opt.lr, opt.betas, opt.weight_decay) == (1e-3, (0.8, 0.9), 0.05) and opt.t == 3
    opt.set_inactive(never)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if k in never}
    for n in range(2):
        g = grads()
        for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
            p.grad.copy_(torch.zeros_like(g[k]) if k in never else g[k])
            q.grad = None if k in never else g[k].clone()
        opt.step(); topt.step()
    worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(model.parameters(), twin.parameters()))
    assert worst < 3e-7, worst
    assert all(torch.equal(dict(model.named_parameters())[k].detach(), v) for k, v in before.items())       # no decay on gradient-less parameters
    back, ref = opt.state_dict(), topt.state_dict()
    assert sorted(back["state"]) == sorted(ref["state"])
    assert all(float(back["state"][i]["step"]) == float(ref["state"][i]["step"]) for i in ref["state"])
    # once every parameter has a gradient again the never-seen ones start at step 1 (torch creates their state lazily) - still per parameter
    opt.set_inactive(())
    g = grads()
    for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        p.grad.copy_(g[k]); q.grad = g[k].clone()
    opt.step(); topt.step()
    worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(model.parameters(), twin.parameters()))
    assert worst < 3e-7, worst
    assert {float(v["step"]) for v in opt.state_dict()["state"].values()} == {1.0, 4.0, 6.0}


@pytest.mark.gpu
def test_training_loop_reduces_the_loss():
    """scripts/train.py `Stepper` (quantisation dropout, frozen-codebook pre-training phase, optimiser renewal, HIP losses + FlatAdamW)
    on one fixed batch: finite losses throughout, frozen phase reports zero VQ losses, and the loss goes down."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from esc.models import make_model
    from scripts.train import Stepper
    g = load_golden("train")
    x = _clips(g, "tiny").cuda()
    model = make_model(_cfg("tiny")); model.load_state_dict(synth_state("tiny")); model = model.cuda()
    st = Stepper(model, lr=2e-3, dropout_rate=0.5, pretraining_steps=4, scheduler="constant_warmup", total_steps=40, warmup_steps=3)
    logs = [st.step(x, n) for n in range(40)]
    assert all(np.isfinite(float(l["loss"])) for l in logs)
    assert all(l["frozen"] and float(l["cm_loss"]) == 0.0 for l in logs[:4]) and not logs[4]["frozen"] and float(logs[4]["cm_loss"]) > 0
    full = [float(l["loss"]) for l in logs[4:] if l["streams"] == model.max_streams]
    assert len(full) >= 6 and np.mean(full[-3:]) < np.mean(full[:3]), full


@pytest.mark.gpu
def test_train_cli_checkpoint_and_resume(tmp_path):
    """`python -m scripts.train`: 6 steps in one run == 3 steps + checkpoint + `--resume` for 3 more (model weights bit-identical: the step is
    deterministic, the optimiser moments, the step counter, the stream sampler and the data order all continue)."""
    import os, subprocess, sys
    from conftest import ROOT
    cwd = os.path.join(ROOT, "efficient-speech-codec_amd")
    base = [sys.executable, "-m", "scripts.train", "--synthetic", "tiny", "--batch_size", "2", "--clip_samples", "1260", "--lr", "1e-3", "--log_steps", "1"]

    def run(extra):
        r = subprocess.run(base + extra, capture_output=True, text=True, cwd=cwd)
        assert r.returncode == 0, r.stderr[-3000:]
        return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    full = run(["--steps", "6", "--save_path", str(tmp_path / "a")])
    run(["--steps", "3", "--save_path", str(tmp_path / "b")])
    rest = run(["--steps", "6", "--resume", str(tmp_path / "b" / "checkpoint.pth"), "--save_path", str(tmp_path / "c")])
    assert [l["step"] for l in rest] == [4, 5, 6]
    for a, b in zip(full[3:], rest):
        assert a["streams"] == b["streams"] and a["loss"] == b["loss"], (a, b)
    # periodic evaluation on a validation folder (the reference's evaluate(): eval_epoch at max_streams) leaves training untouched
    from scipy.io import wavfile
    os.makedirs(tmp_path / "val")
    for i in range(3):
        wavfile.write(tmp_path / "val" / f"v{i}.wav", 16000, synth.voiced_clip_int16(f"val-{i}", 15980))          # EvalSet drops 80 samples: 795 hops of the tiny model, an even frame count
    ev = run(["--steps", "6", "--val_data", str(tmp_path / "val"), "--eval_every", "3"])
    evals = [l for l in ev if "eval" in l]
    assert [l["step"] for l in evals] == [3, 6] and all(np.isfinite(l["eval"]["SISDR"]) and 0 < l["eval"]["utilization"] <= 1 for l in evals)
    assert [l["loss"] for l in ev if "loss" in l] == [l["loss"] for l in full]
    wa = torch.load(tmp_path / "a" / "checkpoint.pth", weights_only=False)["model_state_dict"]
    wc = torch.load(tmp_path / "c" / "checkpoint.pth", weights_only=False)["model_state_dict"]
    assert all(torch.equal(wa[k], wc[k]) for k in wa)


@pytest.mark.gpu
def test_training_step_is_run_to_run_deterministic():
    """No atomics anywhere in the backward (fixed-order partial sums): the same step twice gives bit-identical losses and gradients, for the
    generator (base, two clips) and for the discriminator."""
    from esc.models import Discriminator
    from esc.modules import GANLoss
    g = load_golden("train")
    w = json.loads(str(g["weights_json"]))
    x = _clips(g, "base")
    runs = []
    for _ in range(2):
        model, out, losses, grads = _product_step("base", 6, False, x, w)
        runs.append((losses, grads))
    assert all(np.array_equal(runs[0][0][k], runs[1][0][k]) for k in runs[0][0])
    assert all(np.array_equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    torch.manual_seed(0)
    disc = Discriminator(sample_rate=16000).cuda()
    gan = GANLoss(disc)
    real = x[:, :8000].cuda(); fake = (0.9 * x[:, :8000] + 0.01).cuda()
    dg = []
    for _ in range(2):
        disc.zero_grad()
        gan.discriminator_loss(fake, real).mean().backward()
        dg.append({k: p.grad.detach().clone() for k, p in disc.named_parameters()})
    assert all(torch.equal(dg[0][k], dg[1][k]) for k in dg[0])


@pytest.mark.gpu
@pytest.mark.parametrize("parts", ["1", "2"])
def test_training_step_with_a_precomputed_spectrum(monkeypatch, parts):
    """forward(x, x_feat=...) in TRAINING mode (codecs.py:33-34; round 6: escx_train_forward_feat): the spectrum the library's own STFT produces, handed back as x_feat in the
    reference's (Bs, F, T, 2) layout, gives bit for bit the step of the waveform call - losses, codes, every gradient - in the one-part and in the two-part form;
    a spectrum of another clip changes it; raw_feat is the given spectrum."""
    g = load_golden("train")
    w = json.loads(str(g["weights_json"]))
    x = _clips(g, "base")
    x = torch.cat([x, 0.5 * x.flip(0)], dim=0)
    monkeypatch.setenv("ESCX_TRAIN_PARTS", parts)
    monkeypatch.setenv("ESCX_TRAIN_PARTS_MIN_BATCH", "2")
    _, out0, l0, g0 = _product_step("base", 6, False, x, w)
    feat = out0["raw_feat"].detach().permute(0, 2, 3, 1).contiguous()          # (Bs, 2, F, T) -> the reference's x_feat layout (Bs, F, T, 2)
    assert feat.shape[1:] == (192, 1 + x.shape[1] // 80, 2)
    _, out1, l1, g1 = _product_step("base", 6, False, x, w, x_feat=feat)
    assert torch.equal(out1["raw_feat"], out0["raw_feat"]) and torch.equal(out1["codes"], out0["codes"]) and torch.equal(out1["recon_audio"], out0["recon_audio"])
    assert all(np.array_equal(l0[k], l1[k]) for k in l0) and all(np.array_equal(g0[k], g1[k]) for k in g0)
    _, out2, l2, _ = _product_step("base", 6, False, x, w, x_feat=feat.flip(0).contiguous())
    assert not torch.equal(out2["recon_audio"], out0["recon_audio"])
    with pytest.raises(ValueError):
        _product_step("base", 6, False, x, w, x_feat=feat.permute(0, 3, 1, 2).contiguous())


@pytest.mark.gpu
def test_two_part_training_pass_equals_the_one_part_pass(monkeypatch):
    """Large batches run as two parts on two streams (train.hip: TrainRoot): own tapes and gradient arenas, flat gradients added in a fixed order.
    Clips are independent in the training step, so the forward outputs (losses, codes) are bit for bit those of the one-part pass and the gradients
    differ by summation order only; the two-part form is itself run-to-run deterministic."""
    g = load_golden("train")
    w = json.loads(str(g["weights_json"]))
    x = _clips(g, "base")
    x = torch.cat([x, 0.5 * x.flip(0)], dim=0)                    # 4 clips -> parts of 2 + 2
    monkeypatch.setenv("ESCX_TRAIN_PARTS", "1")
    _, _, l1, g1 = _product_step("base", 6, False, x, w)
    monkeypatch.setenv("ESCX_TRAIN_PARTS", "2")
    monkeypatch.setenv("ESCX_TRAIN_PARTS_MIN_BATCH", "2")
    runs = [_product_step("base", 6, False, x, w)[2:] for _ in range(2)]
    l2, g2 = runs[0]
    assert all(np.array_equal(l2[k], runs[1][0][k]) for k in l2) and all(np.array_equal(g2[k], runs[1][1][k]) for k in g2)
    for k in l1:
        np.testing.assert_allclose(l2[k], l1[k], rtol=1e-6, atol=1e-7, err_msg=k)
    num = sum(float(((g2[k].astype(np.float64) - g1[k]) ** 2).sum()) for k in g1)
    den = sum(float((g1[k].astype(np.float64) ** 2).sum()) for k in g1)
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5


@pytest.mark.gpu
def test_split_operand_mlp_kernels_of_the_training_step_against_the_fp32_mfma_forms(monkeypatch):
    """Round 6: the C = 45 / 72 layers run their MLP forward on the exact three-term split-operand kernel (ESCX_TRAIN_MLP_X3) and the two channel contractions of the fused
    MLP backward on two fp16 terms with power-of-two block scaling (ESCX_TRAIN_MLPBWD_X2, train_mlp_fused.h).
    (a) Backward switch alone - same forward, same tape: the whole gradient moves by the arithmetic error of two contractions (measured 1.7e-7 of its norm, worst tensor 1.1e-6 of
        its own; bounds 2e-6 / 2e-5).
    (b) Forward switch: losses to fp32 rounding; the gradient moves like under any fp32 re-association of the forward - the level of the reference's own fp32-vs-fp64 distance
        (median 1e-3 ... 1e-2 per tensor on this model, _noise_floor), so it is held to 1e-2 of its norm, not to 1e-5."""
    g = load_golden("train")
    w = json.loads(str(g["weights_json"]))
    x = _clips(g, "base")

    def rel(ga, gb):
        num = sum(float(((ga[k].astype(np.float64) - gb[k]) ** 2).sum()) for k in gb)
        den = sum(float((gb[k].astype(np.float64) ** 2).sum()) for k in gb)
        worst = max(((float(((ga[k].astype(np.float64) - gb[k]) ** 2).sum()) / max(float((gb[k].astype(np.float64) ** 2).sum()), 1e-30)) ** 0.5, k) for k in gb)
        return (num / den) ** 0.5, worst

    monkeypatch.setenv("ESCX_TRAIN_MLP_X3", "0"); monkeypatch.setenv("ESCX_TRAIN_MLPBWD_X2", "0")
    _, _, l0, g0 = _product_step("base", 6, False, x, w)
    monkeypatch.delenv("ESCX_TRAIN_MLP_X3")
    _, _, lf, gf = _product_step("base", 6, False, x, w)           # split-operand forward, fp32-MFMA backward
    monkeypatch.delenv("ESCX_TRAIN_MLPBWD_X2")
    _, _, l1, g1 = _product_step("base", 6, False, x, w)           # both (the default)
    for k in l0:
        np.testing.assert_allclose(lf[k], l0[k], rtol=2e-6, atol=1e-7, err_msg=k)
        assert np.array_equal(l1[k], lf[k]), k                      # the backward switch does not touch the forward
    rb, wb = rel(g1, gf)
    rf, wf = rel(gf, g0)
    print(f"[train-x2] backward switch: whole gradient {rb:.2e}, worst tensor {wb[0]:.2e} at {wb[1]}; forward switch: whole gradient {rf:.2e}, worst tensor {wf[0]:.2e} at {wf[1]}")
    assert 0 < rb < 2e-6 and wb[0] < 2e-5, (rb, wb)
    assert 0 < rf < 1e-2, (rf, wf)
