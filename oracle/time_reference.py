"""TEST INFRASTRUCTURE -- calibration of the oracle as a stand-in for "the reference's CPU path" (BASELINE.md section 4 step 1).

Imports the REAL reference (/root/reference, through oracle/ref_shims.py), loads the same name-keyed weights into it and into
oracle/esc_oracle.py, runs ESC.encode + ESC.decode of both on identical synthetic clips with the same thread count, requires identical
codes (and audio within 1e-6 RMS) and prints the time ratio.  Build container only: /root/reference does not exist on the GPU box, where
bench.py's `cpu_baseline` leg therefore times the oracle ("kind": "port").

    python oracle/time_reference.py [--threads 8] [--batches 1 4] [--reps 5] [--json profiles/r3_oracle_vs_reference_cpu.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))

import ref_shims  # noqa: E402
from gen_golden import build_reference, synth  # noqa: E402  (name-keyed weights into the reference model)


def median_time(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[8, 1])
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--config", default="9kbps_esc_base.yaml")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()

    ref_models = ref_shims.load_reference()
    from oracle.esc_oracle import EscOracle
    cfg = yaml.safe_load(open(f"{ref_shims.REFERENCE_ROOT}/configs/{args.config}"))["model"]
    ref, manifest = build_reference(ref_models, cfg)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    orc = EscOracle(dict(cfg), sd)
    S = cfg["max_streams"]
    rows = []
    for thr in args.threads:
        torch.set_num_threads(thr)
        for B in args.batches:
            pcm = np.stack([synth.noise_clip_int16(f"timeref-{i}", 48000) for i in range(B)])
            x = torch.from_numpy(synth.pcm_to_float(pcm))
            with torch.no_grad():
                rc, rshape = ref.encode(x, S)
                rw = ref.decode(rc, rshape)
                oc, oshape = orc.encode(x, S)
                ow = orc.decode(oc, oshape)
                assert tuple(rshape) == tuple(oshape) and torch.equal(rc, oc), "oracle codes differ from the reference"
                rms = float((rw - ow).pow(2).mean().sqrt())
                assert rms <= 1e-6, rms

                def run_ref():
                    c, s = ref.encode(x, S); ref.decode(c, s)

                def run_orc():
                    c, s = orc.encode(x, S); orc.decode(c, s)
                # interleave the two so that frequency / cache state is shared
                t_ref, t_orc = median_time(run_ref, args.reps), median_time(run_orc, args.reps)
                t_ref2, t_orc2 = median_time(run_ref, args.reps), median_time(run_orc, args.reps)
                t_ref, t_orc = min(t_ref, t_ref2), min(t_orc, t_orc2)
            row = {"config": args.config, "threads": thr, "batch": B, "reference_ms": round(t_ref * 1e3, 1), "oracle_ms": round(t_orc * 1e3, 1),
                   "oracle_over_reference": round(t_orc / t_ref, 3), "reference_audio_s_per_s": round(B * 3.0 / t_ref, 2),
                   "oracle_audio_s_per_s": round(B * 3.0 / t_orc, 2), "codes_identical": True, "audio_rms_diff": rms}
            rows.append(row)
            print(json.dumps(row), flush=True)
    out = {"what": "ESC.encode + ESC.decode (num_streams = max), the real reference vs oracle/esc_oracle.py, same weights / inputs / thread count, "
                   f"median of {args.reps} after a warm-up, best of two rounds", "host": f"{os.cpu_count()} vCPU build container, torch {torch.__version__} CPU",
           "rows": rows}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
