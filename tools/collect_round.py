"""Copy the evidence of tools/profile_round.sh (gpurun_out/round/) into profiles/ under this round's prefix.
Usage: python tools/collect_round.py r2"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_hbm import csrc_hash

def main():
    pre = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out", "round"), os.path.join(ROOT, "profiles")
    names = {"bench.json": "bench.json", "bench_rccl_1rank.json": "bench_rccl_1rank.json", "bench_strong_n1.json": "bench_strong_n1.json",
             "bench_train.json": "bench_train.json", "bench_train_adv.json": "bench_train_adv.json", "bench_train_adv_bf16.json": "bench_train_adv_bf16.json", "bench_train_adv_fp32mfma.json": "bench_train_adv_fp32mfma.json", "train_adv_bf16_kernel_stats.csv": "train_adv_bf16_kernel_stats.csv", "event_breakdown_isolated.txt": "event_breakdown.txt",
             "event_breakdown_1stream.txt": "event_breakdown_1stream.txt", "kernel_stats.csv": "kernel_stats.csv", "prof_bench_line.txt": "bench_under_rocprof.json",
             "other_configs.json": "other_configs.json", "pmc_hbm.json": "pmc_hbm.json", "pmc_calibration.json": "pmc_calibration.json",
             "sq_counters.txt": "sq_counters.txt", "sq_counters.json": "sq_counters.json", "train_breakdown.txt": "train_breakdown.txt",
             "train_kernel_stats.csv": "train_kernel_stats.csv", "small_batch.txt": "small_batch.txt", "sq_util.txt": "sq_util.txt", "sq_util_train.txt": "sq_util_train.txt", "train_adv_kernel_stats.csv": "train_adv_kernel_stats.csv", "pytest_gpu.txt": "pytest_gpu.txt", "ubench_gemm.txt": "ubench_gemm.txt",
             "parity_sweep_base576.log": "parity_sweep_base576.log", "parity_sweep_large288.log": "parity_sweep_large288.log"}
    for a, b in names.items():
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, f"{pre}_{b}"))
        else:
            print("missing:", a)
    p = os.path.join(src, "pmc_dominant.json")
    if os.path.exists(p):
        d = json.load(open(p))
        if len(sys.argv) > 2 and sys.argv[2] == "--restamp":      # only when the codec kernel sources are unchanged since the run (check git diff)
            d["_csrc_sha256"] = csrc_hash()
        json.dump(d, open(os.path.join(dst, "pmc_dominant.json"), "w"), indent=1)
        print("pmc_dominant stamp", d["_csrc_sha256"][:12], "current", csrc_hash()[:12])

if __name__ == "__main__":
    main()
