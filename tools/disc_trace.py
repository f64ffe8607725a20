"""Per-layer timing of the discriminator (ESCX_DISC_TRACE=1: events + a sync around every launch group): one forward and one full backward
at the bench's batch.  Usage (GPU box): ESCX_DISC_TRACE=1 python tools/disc_trace.py [batch] 2> trace.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import numpy as np, torch
import bench
from esc import synth
from esc.models import Discriminator
from esc.modules import GANLoss

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 36
dev = torch.device("cuda:0")
disc = Discriminator(sample_rate=16000).to(dev).train()
gan = GANLoss(disc)
pcm = np.stack([(synth.voiced_clip_int16 if i % 2 else synth.noise_clip_int16)(f"bench-r0-{i}", bench.TRAIN_SAMPLES) for i in range(2 * bsz)])
x = torch.from_numpy(synth.pcm_to_float(pcm)).to(dev)
fake, real = x[:bsz].clone().requires_grad_(True), x[bsz:]
for rep in range(2):
    print(f"[disc] ---- repetition {rep}", file=sys.stderr, flush=True)
    d_fake, d_real = gan.adversarial_forward(fake=fake, real=real)
    lg, lf = gan.generator_loss_from(d_fake, d_real)
    (lg + 2.0 * lf).mean().backward()
    gan.discriminator_backward_from(d_fake, d_real)
    torch.cuda.synchronize()
