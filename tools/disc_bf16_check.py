import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import numpy as np, torch
from esc import synth
from esc.models import Discriminator
dev = torch.device("cuda:0")
torch.manual_seed(0)
disc = Discriminator(sample_rate=16000).to(dev)
B = 8
pcm = np.stack([(synth.voiced_clip_int16 if i % 2 else synth.noise_clip_int16)(f"bf-{i}", 48000) for i in range(B)])
x = torch.from_numpy(synth.pcm_to_float(pcm)).to(dev).unsqueeze(1).requires_grad_(True)
res = {}
for prec in ("fp32", "bf16"):
    disc.set_conv_precision(prec)
    disc.zero_grad(); x.grad = None
    outs = disc(x)
    loss = sum((f[-1] ** 2).mean() for f in outs) + sum(fm.abs().mean() for f in outs for fm in f[:-1])
    loss.backward()
    torch.cuda.synchronize()
    res[prec] = dict(fm=[fm.detach().clone() for f in outs for fm in f], gx=x.grad.detach().clone(), g={k: p.grad.detach().clone() for k, p in disc.named_parameters()}, loss=float(loss))
a, b = res["fp32"], res["bf16"]
print("loss", a["loss"], b["loss"])
worst = 0
for i, (u, v) in enumerate(zip(a["fm"], b["fm"])):
    e = float((u - v).norm() / (u.norm() + 1e-30)); worst = max(worst, e)
print("feature maps: worst rel l2", worst)
print("d wave rel l2", float((a["gx"] - b["gx"]).norm() / a["gx"].norm()))
ws = sorted(((float((a["g"][k] - b["g"][k]).norm() / (a["g"][k].norm() + 1e-30)), k) for k in a["g"]), reverse=True)
print("param grads: worst", ws[:4], "median", ws[len(ws) // 2])
