import json

import numpy as np
import torch

from conftest import load_golden, synth_state


def have_gpu():
    return torch.cuda.is_available()


def build_models(name):
    """(product model on cuda:0, oracle on CPU, golden npz, cfg) with identical synthetic weights."""
    from esc.models import make_model
    from oracle.esc_oracle import EscOracle
    g = load_golden(name)
    cfg = json.loads(str(g["config_json"]))
    sd = synth_state(name)
    model = make_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    return model, EscOracle(cfg, sd), g, cfg


def rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def rel_rms(a, b):
    b64 = np.asarray(b, np.float64)
    return rms(a, b) / max(float(np.sqrt(np.mean(b64 ** 2))), 1e-30)


def code_report(got, ref, margins=None):
    """Human-readable attribution of code mismatches (count, first stream, reference margin there)."""
    got = np.asarray(got); ref = np.asarray(ref)
    bad = np.argwhere(got != ref)
    if len(bad) == 0:
        return "all codes equal"
    msg = f"{len(bad)}/{got.size} codes differ; first at (b,s,g,t)={tuple(bad[0])}"
    if margins is not None:
        first_stream = bad[:, 1].min()
        sel = bad[bad[:, 1] == first_stream]
        ms = [float(margins[tuple(i)]) for i in sel[:8]]
        msg += f"; earliest stream {first_stream}, reference margins there {ms}"
    return msg


NEAR_TIE = 2e-6          # best/second-best distance gap below which an argmin flip is fp32 re-association noise (SURVEY hard part 1)


def unattributable(got, ref, margins, tol=NEAR_TIE):
    """Mismatches between (B,S,G,T) code tensors that CANNOT be explained by a reference near-tie.

    Within a clip, streams are sequential: a flip in stream s changes the residual every later stream sees, so only the EARLIEST
    stream with a mismatch is judged, and there every mismatching (g, t) must sit on a reference margin < tol.  Returns a list of
    human-readable findings (empty = every difference is attributable; no difference at all = trivially empty)."""
    got = np.asarray(got); ref = np.asarray(ref); margins = np.asarray(margins)
    out = []
    for b in range(got.shape[0]):
        bad = np.argwhere(got[b] != ref[b])
        if len(bad) == 0:
            continue
        s0 = bad[:, 0].min()
        for s, g, t in bad[bad[:, 0] == s0]:
            m = float(margins[b, s, g, t])
            if not m < tol:
                out.append(f"clip {b} stream {s} group {g} frame {t}: got {got[b, s, g, t]} ref {ref[b, s, g, t]} margin {m:.3e}")
    return out


def mismatch_summary(got, ref, margins):
    got = np.asarray(got); ref = np.asarray(ref)
    bad = np.argwhere(got != ref)
    clips = sorted(set(int(i) for i in bad[:, 0]))
    return f"{len(bad)} differing codes in clips {clips}; " + code_report(got, ref, margins)
