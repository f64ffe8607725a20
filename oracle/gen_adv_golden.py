"""TEST INFRASTRUCTURE -- golden values of ONE ADVERSARIAL TRAINING STEP as the REAL reference runs it (/root/reference through
oracle/ref_shims.py): scripts/trainer_adv.py:61-107 without the optimisers - generator forward (ESC-Base, training mode), mel loss, the
generator's GAN losses through the reference Discriminator (its own two passes), the weighted sum of configs/9kbps_esc_base_adv.yaml,
`loss.mean().backward()`; then `discriminator_loss` (its own two passes on the detached reconstruction) and its backward.

    python oracle/gen_adv_golden.py        # writes tests/golden/adv.npz        (ESC-Base generator: configs/9kbps_esc_base_adv.yaml as it stands)
    python oracle/gen_adv_golden.py large  # writes tests/golden/adv_large.npz  (BASELINE configs[4]: the `model:` block of configs/9kbps_esc_large.yaml
                                           #  under the adversarial yaml's loss / discriminator blocks - the reference ships no large_adv yaml)

Stored: per-clip losses of both updates, the gradient norm of every generator parameter (through the discriminator) and of every
discriminator parameter.  The product shares the discriminator passes between the two updates (two instead of four): this fixture is what
its result has to equal.  audiotools' matched-stride STFT is shimmed as in gen_disc_golden.py (unpinned at that boundary); inputs and weights
are regenerated from tags by esc/synth.py.
"""
import json, os, sys
import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_disc_golden as gd  # noqa: E402

N_SAMPLES = 15920            # 1 s minus one hop: an even frame count
STREAMS = 6


def main():
    large = len(sys.argv) > 1 and sys.argv[1] == "large"
    torch.manual_seed(0); torch.set_num_threads(8)
    ref_models = gg.ref_shims.load_reference()
    gd.install_audiotools()
    import importlib
    D = importlib.import_module("esc.models.discriminator")
    D.AudioSignal, D.STFTParams = sys.modules["audiotools"].AudioSignal, sys.modules["audiotools"].STFTParams      # the module bound the import-time stubs
    G = importlib.import_module("esc.modules.loss.gan_loss")
    losses = importlib.import_module("esc.modules")
    ycfg = yaml.safe_load(open(f"{gg.ref_shims.REFERENCE_ROOT}/configs/9kbps_esc_base_adv.yaml"))
    w = {k: float(v) for k, v in ycfg["loss"].items()}
    if large:
        ycfg["model"] = yaml.safe_load(open(f"{gg.ref_shims.REFERENCE_ROOT}/configs/9kbps_esc_large.yaml"))["model"]
    model, manifest = gg.build_reference(ref_models, ycfg["model"])
    model.train()
    disc = D.Discriminator(**{**ycfg.get("discriminator", {}), "sample_rate": 16000}) if "discriminator" in ycfg else D.Discriminator(sample_rate=16000)
    dman = {k: list(v.shape) for k, v in disc.state_dict().items()}
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in gd.synth_disc_state(dman).items()})
    gan = G.GANLoss(disc)
    mel_fn, stft_fn = losses.MelSpectrogramLoss(), losses.ComplexSTFTLoss()
    tags = ["advL-0", "advL-1"] if large else ["adv-0", "adv-1"]
    pcm = np.stack([gg.synth.noise_clip_int16(tags[0], N_SAMPLES), gg.synth.voiced_clip_int16(tags[1], N_SAMPLES)])
    x = torch.from_numpy(gg.synth.pcm_to_float(pcm))
    # ---- generator update (trainer_adv.py:70-91)
    o = model(**dict(x=x, x_feat=None, num_streams=STREAMS, freeze_codebook=False))
    mel = mel_fn(o["raw_audio"], o["recon_audio"])
    stft = stft_fn(o["raw_feat"], o["recon_feat"])
    gen, feat = gan.generator_loss(fake=o["recon_audio"], real=o["raw_audio"])
    loss = (o["cm_loss"] * w["cm_weight"] + o["cb_loss"] * w["cb_weight"] + mel * w["mel_weight"] + stft * w["stft_weight"]
            + gen * w["gen_weight"] + feat * w["feat_weight"])
    model.zero_grad(); disc.zero_grad()
    loss.mean().backward()
    gkeys = [k for k, _ in model.named_parameters()]
    gp = dict(model.named_parameters())
    out = {"weights_json": np.array(json.dumps(w)), "tags_json": np.array(json.dumps(tags)), "n_samples": np.int64(N_SAMPLES), "streams": np.int64(STREAMS),
           "disc_cfg_json": np.array(json.dumps({k: v for k, v in ycfg.get("discriminator", {}).items()})),
           "model_cfg_json": np.array(json.dumps(ycfg["model"])),
           "gen_keys_json": np.array(json.dumps(gkeys)),
           "cm": o["cm_loss"].detach().numpy(), "cb": o["cb_loss"].detach().numpy(), "mel": mel.detach().numpy(), "stft": stft.detach().numpy(),
           "gen": gen.detach().numpy(), "feat": feat.detach().numpy(), "loss": loss.detach().numpy(), "codes": o["codes"].numpy().astype(np.int16),
           "gen_gnorm": np.array([0.0 if gp[k].grad is None else float(gp[k].grad.double().norm()) for k in gkeys])}
    # ---- discriminator update (trainer_adv.py:96-105)
    dl = gan.discriminator_loss(fake=o["recon_audio"], real=o["raw_audio"])
    disc.zero_grad()
    dl.mean().backward()
    dkeys = [k for k, _ in disc.named_parameters()]
    dp = dict(disc.named_parameters())
    out.update(disc_keys_json=np.array(json.dumps(dkeys)), disc_loss=dl.detach().numpy(), disc_gnorm=np.array([float(dp[k].grad.double().norm()) for k in dkeys]))
    print("loss", out["loss"], "gen", out["gen"], "feat", out["feat"], "disc", out["disc_loss"], "|g gen|", float(np.sqrt((out["gen_gnorm"] ** 2).sum())),
          "|g disc|", float(np.sqrt((out["disc_gnorm"] ** 2).sum())))
    np.savez_compressed(os.path.join(gg.GOLD, "adv_large.npz" if large else "adv.npz"), **out)


if __name__ == "__main__":
    main()
