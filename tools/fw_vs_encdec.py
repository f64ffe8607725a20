import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/efficient-speech-codec_amd')
import bench
dev = torch.device('cuda:0')
model, cfg, sd = bench.build_model(dev)
x = bench.synth_batch(36, 0, None).to(dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def encdec():
    c, s = model.encode(x, 6); model.decode(c, s)
def fw():
    with torch.no_grad(): model(x=x, x_feat=None, num_streams=6)
for _ in range(2):
    print('encode+decode %.3f ms   forward %.3f ms' % (t(encdec), t(fw)))
