"""TEST INFRASTRUCTURE -- UNFILTERED golden codes from the REAL reference (/root/reference through oracle/ref_shims.py).

oracle/gen_golden.py keeps, per fixture, the candidate clip with the widest argmin margin (so that a mismatch on those fixtures
is never an fp32 near-tie).  This script does the opposite on purpose: the FIRST n clips by tag, no selection whatsoever, with
the reference's own best/second-best distance margin stored next to every code, so that tests/test_gpu_parity.py can require
zero mismatches and, should one occur, attribute it to a margin at fp32 re-association level instead of staying silent.

    python oracle/gen_unfiltered_golden.py        # writes tests/golden/unfiltered.npz (codes int16 + margins f32; the PCM is
                                                  # regenerated from the tags by esc/synth.py, which is a pure function of the tag)
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the shims' helpers; does not run main)

SETS = {  # fixture name -> (reference yaml, [(kind, first n tags)])
    "base": ("9kbps_esc_base.yaml", [("noise", 8), ("voiced", 4)]),
    "large": ("9kbps_esc_large.yaml", [("noise", 3), ("voiced", 1)]),
}
N_SAMPLES = 48000


def clip(kind, tag):
    return (gg.synth.noise_clip_int16 if kind == "noise" else gg.synth.voiced_clip_int16)(tag, N_SAMPLES)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models = gg.ref_shims.load_reference()
    tap = gg.MarginTap()
    out, summary = {}, {}
    for name, (yml, kinds) in SETS.items():
        cfg = yaml.safe_load(open(f"{gg.ref_shims.REFERENCE_ROOT}/configs/{yml}"))["model"]
        model, _ = gg.build_reference(ref_models, cfg)
        tags = [(kind, f"unfiltered-{kind}-{i}") for kind, n in kinds for i in range(n)]
        pcm = np.stack([clip(k, t) for k, t in tags])
        x = torch.from_numpy(gg.synth.pcm_to_float(pcm))
        codes, shape = model.encode(x, num_streams=cfg["max_streams"])
        margins = tap.pop(cfg["group_size"])
        audio = model.decode(codes, shape)
        out[f"{name}_tags"] = np.array(json.dumps(tags))
        out[f"{name}_codes"] = codes.numpy().astype(np.int16)
        out[f"{name}_margins"] = margins.numpy().astype(np.float32)
        out[f"{name}_audio_rms"] = np.sqrt((audio.numpy().astype(np.float64) ** 2).mean(axis=1))
        out[f"{name}_audio_sub"] = audio.numpy()[:, ::16].astype(np.float32)
        m = margins.numpy()
        summary[name] = dict(clips=len(tags), n_codes=int(codes.numel()), min_margin=float(m.min()),
                             below_1e5=int((m < 1e-5).sum()), below_1e4=int((m < 1e-4).sum()))
        print(name, summary[name])
    tap.close()
    out["summary_json"] = np.array(json.dumps(summary))
    np.savez_compressed(os.path.join(gg.GOLD, "unfiltered.npz"), **out)


if __name__ == "__main__":
    main()
