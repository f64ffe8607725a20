"""TEST INFRASTRUCTURE -- clustered-codebook stress fixture from the REAL reference.

All weights in this repository are synthetic (no checkpoint is reachable), and random codebooks have wide argmin margins; trained codebooks
may not (SURVEY hard part 1).  Here every codebook row gets a near-duplicate at relative distance 1e-4 ... 1e-6 (esc/synth.py
cluster_codebook), the REAL reference (/root/reference through oracle/ref_shims.py) encodes a few clips with those weights, and its codes
and its own best/second-best margins are stored.  tests/test_gpu_parity.py::test_clustered_codebooks_against_the_reference then requires
that every code the HIP path emits is either the reference's or differs on a reference near-tie (margin < 2e-6, the rule of every other
parity test, later streams verified by continuation), and prints the flip count per margin decade.

    python oracle/gen_clustered_golden.py        # writes tests/golden/clustered.npz (ESC-Base; codes int16 + margins f32; PCM from the tags)
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

KINDS = [("noise", 4), ("voiced", 2)]
N_SAMPLES = 48000


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models = gg.ref_shims.load_reference()
    tap = gg.MarginTap()
    cfg = yaml.safe_load(open(f"{gg.ref_shims.REFERENCE_ROOT}/configs/9kbps_esc_base.yaml"))["model"]
    model = ref_models.make_model(dict(cfg), "csvq+swinT").eval()
    sd = model.state_dict()
    new = gg.synth.clustered_state_dict({k: list(v.shape) for k, v in sd.items()})
    for k, v in sd.items():
        if k.endswith(".window"):
            new[k] = v.numpy().copy()       # torch.hann_window's own f32 rounding, as in gen_golden.build_reference
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in new.items()}, strict=True)
    tags = [(kind, f"clustered-{kind}-{i}") for kind, n in KINDS for i in range(n)]
    pcm = np.stack([(gg.synth.noise_clip_int16 if k == "noise" else gg.synth.voiced_clip_int16)(t, N_SAMPLES) for k, t in tags])
    x = torch.from_numpy(gg.synth.pcm_to_float(pcm))
    codes, shape = model.encode(x, num_streams=cfg["max_streams"])
    margins = tap.pop(cfg["group_size"]).numpy()
    audio = model.decode(codes, shape).numpy()
    tap.close()
    decades = {f"<1e-{e}": int((margins < 10.0 ** -e).sum()) for e in (4, 5, 6, 7)}
    summary = dict(clips=len(tags), n_codes=int(codes.numel()), min_margin=float(margins.min()), margins_below=decades,
                   upper_half_codes=int((codes.numpy() >= cfg["codebook_size"] // 2).sum()))
    print(summary)
    np.savez_compressed(os.path.join(gg.GOLD, "clustered.npz"), tags=np.array(json.dumps(tags)), codes=codes.numpy().astype(np.int16),
                        margins=margins.astype(np.float32), audio_sub=audio[:, ::16].astype(np.float32),
                        audio_rms=np.sqrt((audio.astype(np.float64) ** 2).mean(axis=1)), summary_json=np.array(json.dumps(summary)))


if __name__ == "__main__":
    main()
