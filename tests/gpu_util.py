import json

import numpy as np
import torch

from conftest import load_golden, synth_state


def have_gpu():
    return torch.cuda.is_available()


PRECISIONS = ("f16x2", "bf16x3", "fp32")      # esc.ESC.set_precision / include/escx.h escx_set_precision: default first


def build_models(name, precision=None, edit=None):
    """(product model on cuda:0, oracle on CPU, golden npz, cfg) with identical synthetic weights.  precision: ESC.set_precision mode (None = the
    library default); edit(sd): in-place modification of the state dict BOTH sides load (range-stress checkpoints)."""
    from esc.models import make_model
    from oracle.esc_oracle import EscOracle
    g = load_golden(name)
    cfg = json.loads(str(g["config_json"]))
    sd = synth_state(name)
    if edit is not None:
        edit(sd)
    model = make_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    if precision is not None:
        model.set_precision(precision)
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


def attribute_with_continuation(orc, x, got, ref, margins, num_streams, tol=NEAR_TIE, max_rounds=8):
    """`unattributable` judges only the EARLIEST differing stream of a clip: once a near-tie has been resolved the other way, every later stream sees
    a different residual and its codes legitimately differ.  This closes that gap: for every clip with an attributed flip the
    oracle is re-run with the device's choice FORCED at the flipped positions (EscOracle.encode(force=...)), and the comparison continues on the next
    streams - again code for code, again only near-ties may differ.  Returns (findings, n_forced_codes, clips_continued); findings empty = every code
    of every stream is either identical to the reference or identical to the reference continued from an attributed near-tie."""
    import torch
    from oracle.esc_oracle import Trace
    got = np.asarray(got); ref = np.asarray(ref).copy(); margins = np.asarray(margins).copy()
    B = got.shape[0]
    force = np.full(got.shape, -1, dtype=np.int64)
    findings, n_forced, continued = [], 0, set()
    for _ in range(max_rounds):
        redo = []
        for b in range(B):
            bad = np.argwhere(got[b] != ref[b])
            if len(bad) == 0:
                continue
            s0 = bad[:, 0].min()
            first = bad[bad[:, 0] == s0]
            ok = True
            for s, g, t in first:
                m = float(margins[b, s, g, t])
                if not m < tol:
                    findings.append(f"clip {b} stream {s} group {g} frame {t}: got {got[b, s, g, t]} ref {ref[b, s, g, t]} margin {m:.3e}"
                                    + (" (after continuation)" if b in continued else ""))
                    ok = False
            if ok:
                for s, g, t in first:
                    if force[b, s, g, t] < 0:
                        n_forced += 1
                    force[b, s, g, t] = got[b, s, g, t]
                redo.append(b)
        if findings or not redo:
            break
        continued.update(redo)
        tr = Trace()
        rc, _ = orc.encode(x[redo], num_streams, trace=tr, force=torch.from_numpy(force[redo]))
        ref[redo] = rc.numpy()
        margins[redo] = torch.stack(tr.margins, dim=1).numpy()
    else:
        findings.append("continuation did not converge")
    return findings, n_forced, sorted(continued)
