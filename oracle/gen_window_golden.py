"""TEST INFRASTRUCTURE -- window sizes other than 4: generates tests/golden/window.npz by running the REAL reference
(/root/reference, imported through oracle/ref_shims.py; attention.py:93-127, 246-256 are generic in window_size) on the tiny
configuration of oracle/gen_golden.py with window_size = 2, 3 and 8 (shift 1, 1, 4), name-keyed deterministic weights and
stored int16 PCM inputs.  Run in the build container only:

    python oracle/gen_window_golden.py      # writes tests/golden/window.npz + window_ws{2,3,8}_manifest.json

Lengths: W = 32 and W = 30 frames (30 is a multiple of neither 4 nor 8: window padding in time); H = 16 / 8 / 4 patches, so
window_size 8 pads the 4-row map in frequency and window_size 3 pads every map.  The fixtures are data (inputs, expected codes,
audio, argmin margins, encoder maps); no reference source is stored.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (build_reference, MarginTap, run_all_streams, pick_clips, TINY_CFG)

WINDOW_SIZES = (2, 3, 8)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models = gg.ref_shims.load_reference()
    tap = gg.MarginTap()
    out = {"window_sizes": np.array(WINDOW_SIZES)}
    for ws in WINDOW_SIZES:
        cfg = dict(gg.TINY_CFG, window_size=ws)
        model, manifest = gg.build_reference(ref_models, cfg)
        out[f"ws{ws}_config_json"] = np.array(json.dumps(cfg))
        for L in (1280, 1200):
            (m, tag, p), = gg.pick_clips(model, tap, 3, 3, L, ["noise"], want=1e-4, floor=1e-5)
            p = np.stack([p, gg.synth.voiced_clip_int16(tag + "-v", L)])
            x = torch.from_numpy(gg.synth.pcm_to_float(p))
            o = gg.run_all_streams(model, x, 3, tap, 3)
            with torch.no_grad():
                enc_hs, shape = model.encoder(model.spec_transform(x))
            out[f"ws{ws}_L{L}_pcm"] = p
            for i, hmap in enumerate(enc_hs):
                out[f"ws{ws}_L{L}_enc{i}"] = hmap.detach().numpy()
            for k, v in o.items():
                out[f"ws{ws}_L{L}_{k}"] = v
            print(f"[ws {ws}, L {L}] codes {o['codes'].shape}, min margin {o['margins'].min():.3e}")
        json.dump(manifest, open(os.path.join(gg.GOLD, f"window_ws{ws}_manifest.json"), "w"), indent=0)
    tap.close()
    np.savez_compressed(os.path.join(gg.GOLD, "window.npz"), **out)


if __name__ == "__main__":
    main()
