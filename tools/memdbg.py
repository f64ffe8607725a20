"""Debug aid: device memory that survives bench.run_train_adv (autograd graphs of the discriminator passes must be released with the step)."""
import sys, os, gc, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
def rep(tag):
    torch.cuda.synchronize()
    free, tot = torch.cuda.mem_get_info()
    print(f"[{tag}] torch allocated {torch.cuda.memory_allocated()/2**30:.1f} GiB, max allocated {torch.cuda.max_memory_allocated()/2**30:.1f}, device free {free/2**30:.1f} of {tot/2**30:.1f}", flush=True)
for prec in ("fp32", "bf16"):
    ap = argparse.Namespace(steps=4, warmup=2, no_cpu_baseline=True, profile_steps=2, adv_precision=prec, gpus=1)
    torch.cuda.reset_peak_memory_stats()
    full = bench.run_train_adv(ap, 0, 1, dev, False, emit=False)
    print(prec, full["ms_per_step"], "ms/step")
    del full
    gc.collect(); torch.cuda.empty_cache()
    rep(prec + " after return")
