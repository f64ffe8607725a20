"""How far apart are fp32 evaluations of the trainer's gradient?  For one fixture case: the HIP step and several fp32 realisations of the oracle
(default, inputs moved by one ulp, single-threaded summation) against the fp64 oracle - forward distance of the reconstruction and per-parameter
gradient distance (median / max).  Usage (GPU box): python tools/grad_noise_diag.py base 3 0"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
import test_train as TT

name = sys.argv[1] if len(sys.argv) > 1 else "base"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
freeze = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
g = load_golden("train")
w = json.loads(str(g["weights_json"]))
x = TT._clips(g, name)
o64, l64, g64 = TT._oracle_step(name, S, freeze, x, dtype=torch.float64)
scale = float(np.sqrt(sum(float((v.double() ** 2).sum()) for v in g64.values())))

def report(tag, recon, grads):
    rb = o64["recon_audio"].detach().numpy()
    fr = float(np.sqrt(((np.asarray(recon, np.float64) - rb) ** 2).mean()) / np.sqrt((rb ** 2).mean()))
    errs = {k: TT._rel_rms(np.asarray(grads[k]), g64[k].numpy(), 1e-6 * scale / np.sqrt(g64[k].numel())) for k in g64}
    v = np.array(list(errs.values()))
    worst = max(errs, key=errs.get)
    print(f"{tag:34s} recon rel rms vs fp64 {fr:.2e} | grad rel rms median {np.median(v):.2e} p90 {np.quantile(v, 0.9):.2e} max {v.max():.2e} ({worst})", flush=True)
    return errs

model, out, losses, grads = TT._product_step(name, S, freeze, x, w)
report("HIP", out["recon_audio"].detach().cpu().numpy(), grads)
gen = torch.Generator().manual_seed(5)
for rep in range(2):
    xp = x * (1 + (torch.randint(0, 2, x.shape, generator=gen).float() * 2 - 1) * 2.0 ** -23)
    model, out, losses, grads = TT._product_step(name, S, freeze, xp, w)
    report(f"HIP, input moved 1 ulp (draw {rep})", out["recon_audio"].detach().cpu().numpy(), grads)
o, l, gg = TT._oracle_step(name, S, freeze, x)
report("oracle fp32", o["recon_audio"].detach().numpy(), {k: v.numpy() for k, v in gg.items()})
for rep in range(2):
    xp = x * (1 + (torch.randint(0, 2, x.shape, generator=gen).float() * 2 - 1) * 2.0 ** -23)
    o, l, gg = TT._oracle_step(name, S, freeze, xp)
    report(f"oracle fp32, input moved 1 ulp ({rep})", o["recon_audio"].detach().numpy(), {k: v.numpy() for k, v in gg.items()})
nt = torch.get_num_threads(); torch.set_num_threads(1)
o, l, gg = TT._oracle_step(name, S, freeze, x)
report("oracle fp32, 1 thread", o["recon_audio"].detach().numpy(), {k: v.numpy() for k, v in gg.items()})
torch.set_num_threads(nt)
