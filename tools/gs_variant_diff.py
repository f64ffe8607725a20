#!/usr/bin/env python
"""Can the head-group split of the C = 384 attention be chosen by batch size?  Direct measurement.

    python tools/gs_variant_diff.py dump out_default.npz                 # codes of 288 noise + 288 voiced sweep clips, batches of 36, default kernels
    ESCX_ATTN_GS_TOKENS=600 python tools/gs_variant_diff.py dump out_gs.npz single    # the same clips ONE AT A TIME with the split on (what a batch-size rule would run for B = 1)
    python tools/gs_variant_diff.py compare out_default.npz out_gs.npz

The switch is read once per process, hence two processes.  `compare` lists every clip whose codes differ: each one is a clip that a batch-size-dependent
rule would encode differently alone than inside a batch - a violation of batch invariance."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT)
import numpy as np


def dump(path, single):
    import torch
    import bench
    from esc import synth
    model, cfg, sd = bench.build_model(torch.device("cuda:0"))
    n = int(os.environ.get("ESCX_GS_DIFF_CLIPS", "288"))
    out, tags = [], []
    for fam, fn in (("noise", synth.noise_clip_int16), ("voiced", synth.voiced_clip_int16)):
        for lo in range(0, n, 36):
            k = min(36, n - lo)
            tg = [f"sweep-{fam}-{lo + i}" for i in range(k)]
            x = torch.from_numpy(synth.pcm_to_float(np.stack([fn(t, 48000) for t in tg]))).cuda()
            if single:
                codes = torch.cat([model.encode(x[i:i + 1].contiguous(), 6)[0] for i in range(k)])
            else:
                codes, _ = model.encode(x, 6)
            out.append(codes.cpu().numpy().astype(np.int16)); tags += tg
    np.savez_compressed(path, codes=np.concatenate(out), tags=np.array(json.dumps(tags)), gs=os.environ.get("ESCX_ATTN_GS_TOKENS", ""), single=int(single))
    print(f"wrote {path}: {len(tags)} clips, ESCX_ATTN_GS_TOKENS={os.environ.get('ESCX_ATTN_GS_TOKENS', '')!r}, {'one clip per call' if single else 'batches of 36'}")


def compare(a, b):
    A, B = np.load(a), np.load(b)
    tags = json.loads(str(A["tags"]))
    assert tags == json.loads(str(B["tags"]))
    ca, cb = A["codes"], B["codes"]
    diff = np.argwhere((ca != cb).reshape(len(tags), -1).any(1)).ravel()
    print(f"{a} (gs={str(A['gs'])!r}, single={int(A['single'])}) vs {b} (gs={str(B['gs'])!r}, single={int(B['single'])}): "
          f"{len(diff)} of {len(tags)} clips differ, {int((ca != cb).sum())} of {ca.size} codes")
    for i in diff:
        w = np.argwhere(ca[i] != cb[i])
        s0 = int(w[:, 0].min())
        first = w[w[:, 0] == s0][0]
        print(f"  clip {tags[i]}: {len(w)} codes differ, earliest at stream {s0} group {first[1]} frame {first[2]}: {int(ca[i][tuple(first)])} vs {int(cb[i][tuple(first)])}")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == "single")
    else:
        compare(sys.argv[2], sys.argv[3])
