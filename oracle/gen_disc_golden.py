"""TEST INFRASTRUCTURE -- golden values of the ADVERSARIAL losses from the REAL reference discriminator (/root/reference/esc/models/
discriminator.py through oracle/ref_shims.py) and its GANLoss (esc/modules/loss/gan_loss.py), on name-keyed synthetic discriminator weights.

    python oracle/gen_disc_golden.py       # writes tests/golden/disc.npz + disc_manifest.json

`audiotools` is not installed: its `AudioSignal.stft(match_stride=True)` is shimmed here from the audiotools source as remembered (reflect pad
by (wl - hop)/2 (+ right pad to a hop multiple), torch.stft(center=True, hann), first/last two frames dropped): the MRD front end is "parity
unpinned at the audiotools boundary".  Stored: per-clip discriminator / generator / feature losses, checksums of every feature map, the gradient
norm of every discriminator parameter (discriminator step), the gradient w.r.t. the fake waveform (generator step), a few full gradients.
"""
import json, math, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def install_audiotools():
    at = sys.modules["audiotools"]

    class STFTParams:
        def __init__(self, window_length=None, hop_length=None, window_type=None, match_stride=None, padding_type=None):
            self.window_length, self.hop_length, self.window_type, self.match_stride, self.padding_type = window_length, hop_length, window_type, match_stride, padding_type

    class AudioSignal:
        def __init__(self, x, sample_rate, stft_params=None):
            self.audio_data, self.sample_rate, self.p = x, sample_rate, stft_params

        def stft(self):
            p = self.p; wl, hop = p.window_length, p.hop_length
            x = self.audio_data; L = x.shape[-1]
            right, pad = 0, 0
            if p.match_stride:
                assert hop == wl // 4
                right = math.ceil(L / hop) * hop - L; pad = (wl - hop) // 2
            xp = torch.nn.functional.pad(x, (pad, pad + right), p.padding_type or "reflect")
            s = torch.stft(xp.reshape(-1, xp.shape[-1]), n_fft=wl, hop_length=hop, window=torch.hann_window(wl), return_complex=True, center=True)
            s = s.reshape(x.shape[0], x.shape[1], s.shape[-2], s.shape[-1])
            return s[..., 2:-2] if p.match_stride else s
    at.STFTParams, at.AudioSignal = STFTParams, AudioSignal


def synth_disc_state(manifest):
    """Deterministic discriminator weights: weight_v uniform with default-init variance, weight_g = 0.5 .. 1.5 x ||v|| (so that the normalised
    weight differs from v), biases small."""
    out = {}
    for k, shp in manifest.items():
        n = int(np.prod(shp))
        u = gg.synth.hashed_uniform(k, n).reshape(shp)
        if k.endswith("weight_v"):
            fan_in = int(np.prod(shp[1:]))
            out[k] = (u / math.sqrt(fan_in)).astype(np.float32)
        elif k.endswith("bias"):
            out[k] = (0.05 * u).astype(np.float32)
    for k, shp in manifest.items():
        if k.endswith("weight_g"):
            v = out[k[:-1] + "v"]
            nrm = np.sqrt((v.reshape(shp[0], -1).astype(np.float64) ** 2).sum(1)).reshape(shp)
            out[k] = (nrm * (1.0 + 0.5 * gg.synth.hashed_uniform(k, int(np.prod(shp))).reshape(shp))).astype(np.float32)
    return out


def main():
    torch.manual_seed(0); torch.set_num_threads(8)
    gg.ref_shims.install()
    install_audiotools()                      # before the reference package is imported: discriminator.py binds AudioSignal at import time
    import importlib
    D = importlib.import_module("esc.models.discriminator")
    G = importlib.import_module("esc.modules.loss.gan_loss")
    disc = D.Discriminator(sample_rate=16000)
    manifest = {k: list(v.shape) for k, v in disc.state_dict().items()}
    sdn = synth_disc_state(manifest)
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()})
    gan = G.GANLoss(disc)
    L = 47920
    real = torch.from_numpy(gg.synth.pcm_to_float(np.stack([gg.synth.voiced_clip_int16("disc-real-0", L), gg.synth.noise_clip_int16("disc-real-1", L)])))
    noise = torch.from_numpy(gg.synth.pcm_to_float(np.stack([gg.synth.noise_clip_int16("disc-fake-0", L, amp=0.02), gg.synth.noise_clip_int16("disc-fake-1", L, amp=0.02)])))
    fake = (0.8 * real + noise).requires_grad_(True)
    out = {"tags_json": np.array(json.dumps(["disc-real-0", "disc-real-1", "disc-fake-0", "disc-fake-1"])), "n_samples": np.int64(L)}
    keys = [k for k, _ in disc.named_parameters()]
    out["keys_json"] = np.array(json.dumps(keys))
    # discriminator step
    disc.zero_grad()
    ld = gan.discriminator_loss(fake, real)
    ld.mean().backward()
    out["disc_loss"] = ld.detach().numpy()
    params = dict(disc.named_parameters())
    out["disc_gnorm"] = np.array([float(params[k].grad.double().norm()) for k in keys])
    for k in ["discriminators.0.convs.0.0.weight_v", "discriminators.0.convs.0.0.weight_g", "discriminators.2.conv_post.weight_v", "discriminators.4.convs.1.0.bias",
              "discriminators.5.band_convs.0.0.0.weight_v", "discriminators.6.band_convs.3.2.0.weight_g", "discriminators.7.conv_post.bias"]:
        out[f"dg::{k}"] = params[k].grad.numpy().astype(np.float32)
    # generator step
    disc.zero_grad(); fake.grad = None
    lg, lf = gan.generator_loss(fake, real)
    (lg * 1.0 + lf * 2.0).mean().backward()
    out["gen_loss"], out["feat_loss"] = lg.detach().numpy(), lf.detach().numpy()
    out["d_fake"] = fake.grad.numpy().astype(np.float32)
    with torch.no_grad():
        fm = disc(fake.detach().unsqueeze(1))
    out["fmap_shapes_json"] = np.array(json.dumps([[list(t.shape) for t in f] for f in fm]))
    out["fmap_rms"] = np.array([[float(t.double().pow(2).mean().sqrt()) for t in f] + [0.0] * (32 - len(f)) for f in fm])
    print("disc_loss", out["disc_loss"], "gen", out["gen_loss"], "feat", out["feat_loss"], "|d_fake|", float(np.linalg.norm(out["d_fake"])))
    np.savez_compressed(os.path.join(gg.GOLD, "disc.npz"), **out)
    json.dump(manifest, open(os.path.join(gg.GOLD, "disc_manifest.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
