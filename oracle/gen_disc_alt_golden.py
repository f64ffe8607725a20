"""TEST INFRASTRUCTURE -- a SECOND discriminator configuration from the REAL reference (other periods, window lengths and band split, a clip
length that is no multiple of a period or a hop, three clips): pins the generic parts of the restatement in oracle/esc_oracle.py that the
default configuration of gen_disc_golden.py does not exercise.

    python oracle/gen_disc_alt_golden.py   # writes tests/golden/disc_alt.npz
"""
import json, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_disc_golden as gd  # noqa: E402

CFG = dict(rates=[], periods=[3, 7], fft_sizes=[512, 256], sample_rate=16000, bands=[[0.0, 0.3], [0.3, 1.0]])
L, B = 6001, 3


def main():
    torch.manual_seed(0); torch.set_num_threads(8)
    gg.ref_shims.install()
    gd.install_audiotools()
    import importlib
    D = importlib.import_module("esc.models.discriminator")
    G = importlib.import_module("esc.modules.loss.gan_loss")
    D.AudioSignal, D.STFTParams = sys.modules["audiotools"].AudioSignal, sys.modules["audiotools"].STFTParams
    disc = D.Discriminator(**{**CFG, "bands": [tuple(b) for b in CFG["bands"]]})
    manifest = {k: list(v.shape) for k, v in disc.state_dict().items()}
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in gd.synth_disc_state(manifest).items()})
    gan = G.GANLoss(disc)
    real = torch.from_numpy(gg.synth.pcm_to_float(np.stack([gg.synth.voiced_clip_int16(f"dalt-real-{i}", L) for i in range(B)])))
    fake = (0.7 * real + torch.from_numpy(gg.synth.pcm_to_float(np.stack([gg.synth.noise_clip_int16(f"dalt-fake-{i}", L, amp=0.03) for i in range(B)])))).requires_grad_(True)
    keys = [k for k, _ in disc.named_parameters()]
    out = {"cfg_json": np.array(json.dumps(CFG)), "manifest_json": np.array(json.dumps(manifest)), "keys_json": np.array(json.dumps(keys)),
           "n_samples": np.int64(L), "batch": np.int64(B)}
    disc.zero_grad()
    ld = gan.discriminator_loss(fake, real)
    ld.mean().backward()
    params = dict(disc.named_parameters())
    out["disc_loss"] = ld.detach().numpy()
    out["disc_gnorm"] = np.array([float(params[k].grad.double().norm()) for k in keys])
    disc.zero_grad(); fake.grad = None
    lg, lf = gan.generator_loss(fake, real)
    (lg + 2.0 * lf).mean().backward()
    out["gen_loss"], out["feat_loss"], out["d_fake"] = lg.detach().numpy(), lf.detach().numpy(), fake.grad.numpy().astype(np.float32)
    with torch.no_grad():
        fm = disc(fake.detach().unsqueeze(1))
    out["fmap_shapes_json"] = np.array(json.dumps([[list(t.shape) for t in f] for f in fm]))
    out["fmap_rms"] = np.array([[float(t.double().pow(2).mean().sqrt()) for t in f] + [0.0] * (32 - len(f)) for f in fm])
    print("disc", out["disc_loss"], "gen", out["gen_loss"], "feat", out["feat_loss"], [len(f) for f in fm])
    np.savez_compressed(os.path.join(gg.GOLD, "disc_alt.npz"), **out)


if __name__ == "__main__":
    main()
