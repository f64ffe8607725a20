"""Where the HOST time of an adversarial training step goes: runs a few steps of bench.py's train_adv workload under cProfile with the GPU
left asynchronous, and prints (a) host time to ENQUEUE a step vs wall time per step with a sync, (b) the top cumulative entries.
Usage (GPU box): python tools/adv_host_profile.py [batch]"""
import cProfile, io, os, pstats, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
import numpy as np, torch
import bench
from esc import synth
from esc.models import Discriminator
from scripts.train import AdvStepper

def main():
    bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    dev = torch.device("cuda:0")
    model, cfg, sd = bench.build_model(dev, "large")
    disc = Discriminator(sample_rate=16000).to(dev)
    disc.set_conv_precision(os.environ.get("ADV_PRECISION", "fp32"))
    pcm = np.stack([(synth.voiced_clip_int16 if i % 2 else synth.noise_clip_int16)(f"bench-r0-{i}", bench.TRAIN_SAMPLES) for i in range(bsz)])
    x = torch.from_numpy(synth.pcm_to_float(pcm)).to(dev)
    st = AdvStepper(model, disc, lr=1e-4, dropout_rate=0.0)
    for n in range(2):
        st.step(x, n)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter(); st.step(x, 2 + rep); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"step {rep}: host enqueue {1e3 * (t1 - t0):.1f} ms, until GPU idle {1e3 * (t2 - t0):.1f} ms")
    pr = cProfile.Profile(); pr.enable()
    for n in range(3):
        st.step(x, 4 + n)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:6000])

if __name__ == "__main__":
    main()
