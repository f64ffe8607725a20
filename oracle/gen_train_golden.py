"""TEST INFRASTRUCTURE -- golden values of ONE TRAINING STEP from the REAL reference (/root/reference through oracle/ref_shims.py):
training-mode `ESC.forward` (esc/models/codecs.py:48-66), the reference's own loss classes (esc/modules/loss/generator_loss.py), the
loss combination of scripts/trainer_no_adv.py:105-115 with the weights of configs/9kbps_esc_base.yaml, then `loss.mean().backward()`.

    python oracle/gen_train_golden.py      # writes tests/golden/train.npz

Per case (config, num_streams, freeze_codebook): per-clip cm / cb / mel / stft losses, the codes, the gradient norm of EVERY parameter
and a handful of full gradients.  The mel loss sits on the MelSpectrogram shim written from the torchaudio documentation
(ref_shims.py): "parity unpinned at the torchaudio boundary", as for the STFT.  Inputs are regenerated from tags by esc/synth.py.
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

WEIGHTS = dict(cm_weight=0.25, cb_weight=1.0, mel_weight=0.25, stft_weight=1.0)        # configs/9kbps_esc_base.yaml:29-33
CASES = {"tiny": [(1, False), (2, False), (3, False), (3, True)], "base": [(1, False), (3, False), (6, False), (6, True)], "large": [(6, False)]}
N_SAMPLES = {"tiny": 1260, "base": 47920, "large": 47920}            # even frame count: raw_feat and recon_feat have the same T (the trainers' clips are cut that way)
FULL_GRADS = {   # parameters whose whole gradient is stored (small ones from every kind of layer)
    "tiny": ["quantizers.1.vqs.0.embedding.weight", "quantizers.0.down_projs.1.weight", "quantizers.2.up_projs.2.weight",
             "encoder.patch_embed.proj.weight", "encoder.pre_nn.swint_blocks.1.attn.relative_position_bias_table",
             "encoder.blocks.0.swint_blocks.0.attn.qkv.weight", "encoder.blocks.1.subsample.down.weight", "decoder.blocks.0.subsample.up.weight",
             "decoder.post_nn.swint_blocks.1.mlp.linear_2.bias", "decoder.patch_deembed.de_proj1.weight", "decoder.patch_deembed.de_proj2.weight",
             "decoder.blocks.1.swint_blocks.1.norm1.weight"],
    "base": ["quantizers.5.vqs.1.embedding.weight", "encoder.patch_embed.proj.weight", "encoder.pre_nn.swint_blocks.1.attn.relative_position_bias_table",
             "encoder.blocks.4.swint_blocks.0.attn.qkv.bias", "decoder.blocks.2.swint_blocks.1.norm2.weight", "decoder.patch_deembed.de_proj2.weight",
             "decoder.post_nn.swint_blocks.0.mlp.linear_1.bias", "quantizers.3.down_projs.0.weight"],
    "large": ["encoder.patch_embed.proj.weight", "decoder.patch_deembed.de_proj2.weight", "quantizers.0.down_projs.0.weight"],       # ESC-Large: the adversarial trainer's generator
}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models = gg.ref_shims.load_reference()
    import importlib
    losses = importlib.import_module("esc.modules")
    mel_fn, stft_fn = losses.MelSpectrogramLoss(), losses.ComplexSTFTLoss()
    out = {"weights_json": np.array(json.dumps(WEIGHTS)), "cases_json": np.array(json.dumps(CASES)), "n_samples_json": np.array(json.dumps(N_SAMPLES))}
    only = sys.argv[1:]                     # e.g. `python oracle/gen_train_golden.py large`: add / refresh these configurations, keep the rest of the file
    old = dict(np.load(os.path.join(gg.GOLD, "train.npz"))) if only and os.path.exists(os.path.join(gg.GOLD, "train.npz")) else {}
    for name in (only or ("tiny", "base", "large")):
        cfg = gg.TINY_CFG if name == "tiny" else yaml.safe_load(open(f"{gg.ref_shims.REFERENCE_ROOT}/configs/9kbps_esc_{name}.yaml"))["model"]
        model, manifest = gg.build_reference(ref_models, cfg)
        model.train()
        tags = [f"train-{name}-0", f"train-{name}-1"]
        pcm = np.stack([gg.synth.noise_clip_int16(tags[0], N_SAMPLES[name]), gg.synth.voiced_clip_int16(tags[1], N_SAMPLES[name])])
        x = torch.from_numpy(gg.synth.pcm_to_float(pcm))
        keys = [k for k, p in model.named_parameters()]
        out[f"{name}_keys"] = np.array(json.dumps(keys))
        out[f"{name}_tags"] = np.array(json.dumps(tags))
        for S, freeze in CASES[name]:
            model.zero_grad()
            o = model(**dict(x=x, x_feat=None, num_streams=S, freeze_codebook=freeze))
            mel = mel_fn(o["raw_audio"], o["recon_audio"])
            stft = stft_fn(o["raw_feat"], o["recon_feat"])
            loss = o["cm_loss"] * WEIGHTS["cm_weight"] + o["cb_loss"] * WEIGHTS["cb_weight"] + mel * WEIGHTS["mel_weight"] + stft * WEIGHTS["stft_weight"]
            loss.mean().backward()
            tag = f"{name}_s{S}_f{int(freeze)}"

            def vec(t):
                t = t if torch.is_tensor(t) else torch.full((x.shape[0],), float(t))
                return t.detach().numpy().astype(np.float32)
            out[f"{tag}_cm"], out[f"{tag}_cb"], out[f"{tag}_mel"], out[f"{tag}_stft"], out[f"{tag}_loss"] = vec(o["cm_loss"]), vec(o["cb_loss"]), vec(mel), vec(stft), vec(loss)
            out[f"{tag}_codes"] = o["codes"].numpy().astype(np.int16)
            out[f"{tag}_recon_rms"] = np.sqrt((o["recon_audio"].detach().numpy().astype(np.float64) ** 2).mean(axis=1))
            params = dict(model.named_parameters())
            out[f"{tag}_gnorm"] = np.array([0.0 if params[k].grad is None else float(params[k].grad.double().norm()) for k in keys], dtype=np.float64)
            for k in FULL_GRADS[name]:
                g = params[k].grad
                out[f"{tag}_g::{k}"] = (torch.zeros_like(params[k]) if g is None else g).numpy().astype(np.float32)
            print(tag, "loss", out[f"{tag}_loss"], "cm", out[f"{tag}_cm"], "mel", out[f"{tag}_mel"], "stft", out[f"{tag}_stft"],
                  "total grad norm", float(np.sqrt((out[f"{tag}_gnorm"] ** 2).sum())))
    np.savez_compressed(os.path.join(gg.GOLD, "train.npz"), **{**old, **out})


if __name__ == "__main__":
    main()
