"""Adversarial step (trainer_adv.py:60-105): the DAC discriminator and the GAN losses.

Fixtures: tests/golden/disc.npz = the REAL reference Discriminator + GANLoss on name-keyed synthetic weights (oracle/gen_disc_golden.py;
audiotools' matched-stride STFT is shimmed: unpinned at that boundary).  CPU: the oracle restatement is pinned to them.  GPU: the HIP
discriminator (esc.models.Discriminator / esc.modules.GANLoss -> libescx escx_disc_*) against fixtures and oracle.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from esc import synth


def disc_state():
    man = json.load(open(os.path.join(GOLDEN, "disc_manifest.json")))
    out = {}
    for k, shp in man.items():
        u = synth.hashed_uniform(k, int(np.prod(shp))).reshape(shp)
        if k.endswith("weight_v"):
            out[k] = (u / math.sqrt(int(np.prod(shp[1:])))).astype(np.float32)
        elif k.endswith("bias"):
            out[k] = (0.05 * u).astype(np.float32)
    for k, shp in man.items():
        if k.endswith("weight_g"):
            v = out[k[:-1] + "v"]
            nrm = np.sqrt((v.reshape(shp[0], -1).astype(np.float64) ** 2).sum(1)).reshape(shp)
            out[k] = (nrm * (1.0 + 0.5 * synth.hashed_uniform(k, int(np.prod(shp))).reshape(shp))).astype(np.float32)
    return {k: torch.from_numpy(out[k]) for k in man}


def clips(g):
    L = int(g["n_samples"])
    real = torch.from_numpy(synth.pcm_to_float(np.stack([synth.voiced_clip_int16("disc-real-0", L), synth.noise_clip_int16("disc-real-1", L)])))
    noise = torch.from_numpy(synth.pcm_to_float(np.stack([synth.noise_clip_int16("disc-fake-0", L, amp=0.02), synth.noise_clip_int16("disc-fake-1", L, amp=0.02)])))
    return real, 0.8 * real + noise


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30))


def test_oracle_discriminator_matches_reference():
    from oracle import esc_oracle as O
    g = load_golden("disc")
    real, fake = clips(g)
    sd = {k: v.clone().requires_grad_(True) for k, v in disc_state().items()}
    fake = fake.clone().requires_grad_(True)
    ld = O.gan_discriminator_loss(fake, real, sd)
    ld.mean().backward()
    np.testing.assert_allclose(ld.detach().numpy(), g["disc_loss"], rtol=1e-5)
    keys = json.loads(str(g["keys_json"]))
    gn = np.array([float(sd[k].grad.double().norm()) for k in keys])
    np.testing.assert_allclose(gn, g["disc_gnorm"], rtol=2e-4, atol=1e-9)
    for f in [f for f in g.files if f.startswith("dg::")]:
        assert _rel(sd[f[4:]].grad.numpy(), g[f]) < 1e-4, f
    for v in sd.values():
        v.grad = None
    lg, lf = O.gan_generator_loss(fake, real, sd)
    (lg + 2.0 * lf).mean().backward()
    np.testing.assert_allclose(lg.detach().numpy(), g["gen_loss"], rtol=1e-5)
    np.testing.assert_allclose(lf.detach().numpy(), g["feat_loss"], rtol=1e-5)
    assert _rel(fake.grad.numpy(), g["d_fake"]) < 1e-4
    fm = O.discriminator_forward(fake.detach().unsqueeze(1), {k: v.detach() for k, v in sd.items()})
    shapes = json.loads(str(g["fmap_shapes_json"]))
    assert [[list(t.shape) for t in f] for f in fm] == shapes
    for i, f in enumerate(fm):
        np.testing.assert_allclose([float(t.double().pow(2).mean().sqrt()) for t in f], g["fmap_rms"][i][: len(f)], rtol=1e-4)
