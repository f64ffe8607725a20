"""Debug aid: one training step of the HIP path vs the oracle's autograd, every loss and every parameter gradient (no early exit)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, synth_state
import test_train as TT

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
freeze = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
g = load_golden("train")
w = json.loads(str(g["weights_json"]))
x = TT._clips(g, name)
probe = TT._smooth_probe(name, x) if os.environ.get("SMOOTH") else None
model, out, losses, grads = TT._product_step(name, S, freeze, x, w, smooth=probe)
oout, ols, ograds = TT._oracle_step(name, S, freeze, x, smooth=probe)
print("codes equal:", bool((out["codes"].cpu() == oout["codes"]).all()), "mismatches", int((out["codes"].cpu() != oout["codes"]).sum()))
for k, ref in (("cm", oout["cm_loss"]), ("cb", oout["cb_loss"]), ("mel", ols.get("mel_loss")), ("stft", ols.get("stft_loss")), ("loss", ols["loss"])):
    if k not in losses:
        continue
    r = ref.detach().numpy() if torch.is_tensor(ref) else np.zeros(2)
    print(f"{k:5s} got {losses[k]} ref {r} rel {np.abs(losses[k] - r).max() / max(np.abs(r).max(), 1e-12):.2e}")
ra = out["recon_audio"].detach().cpu().numpy(); rb = oout["recon_audio"].detach().numpy()
print("recon_audio rel rms", np.sqrt(((ra - rb) ** 2).mean()) / np.sqrt((rb ** 2).mean()))
rf = out["recon_feat"].detach().cpu().numpy(); rg = oout["recon_feat"].detach().numpy()
print("recon_feat rel rms", np.sqrt(((rf - rg) ** 2).mean()) / np.sqrt((rg ** 2).mean()))
scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in ograds.values())))
bad = 0
for k, p in model.named_parameters():
    ref = ograds[k].numpy()
    err = TT._rel_rms(grads[k], ref, 1e-6 * scale / np.sqrt(ref.size))
    flag = "" if err <= 1e-4 else "   <-- BAD"
    bad += err > 1e-4
    if flag or os.environ.get("VERBOSE"):
        print(f"{k:70s} |ref| {np.linalg.norm(ref):.3e} |got| {np.linalg.norm(grads[k]):.3e} rel {err:.2e}{flag}")
print("bad parameters:", bad, "of", len(ograds))
