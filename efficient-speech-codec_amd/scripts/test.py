"""Evaluation CLI, counterpart of the reference's scripts/test.py:23-82: every bitrate (num_streams 1..max), metrics per
clip, codebook utilisation, `perf_stats.json` with the same layout.

    python -m scripts.test --eval_folder_path ./eval --batch_size 36 --model_path ./esc9kbps --device cuda
    python -m scripts.test --eval_folder_path ./eval --synthetic base --device cuda          # no checkpoint available
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader, default_collate

from esc.models import make_model
from .metrics import EntropyCounter, MelSpectrogramDistance, SISDR
from .utils import EvalSet, read_yaml


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--eval_folder_path", type=str, required=True)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--model_path", type=str, default=None, help="folder with config.yaml and model.pth")
    p.add_argument("--synthetic", type=str, default=None, help="base|large|tiny: config from tests/golden with name-keyed synthetic weights")
    p.add_argument("--save_path", type=str, default=None)
    p.add_argument("--device", type=str, default="cuda")
    return p.parse_args()


@torch.no_grad()
def eval_epoch(model, eval_loader, metric_funcs, e_counter, device, bps_per_stream, num_streams=None, verbose=True):
    model.eval()
    all_perf = {k: [] for k in metric_funcs}
    all_perf["utilization"] = []
    eval_range = range(num_streams, num_streams + 1) if num_streams is not None else range(1, model.max_streams + 1)
    for s in eval_range:
        perf = {k: [] for k in metric_funcs}
        e_counter.reset_stats(num_streams=s)
        for x in eval_loader:
            x = x.to(device)
            outputs = model(**dict(x=x, x_feat=None, num_streams=s))
            recon_x, codes = outputs["recon_audio"], outputs["codes"]
            for k, func in metric_funcs.items():
                perf[k].extend(func(x, recon_x).tolist())
            e_counter.update(codes)
        for k, v in perf.items():
            all_perf[k].append(round(float(np.mean(v)), 4))
        rate, _ = e_counter.compute_utilization()
        all_perf["utilization"].append(rate)
        if verbose:
            print(f"Test Metrics at {s * bps_per_stream:.2f}kbps: " + " | ".join(f"{k}: {np.mean(v):.4f}" for k, v in perf.items())
                  + f" | utilization: {rate:.4f}")
    model.train()
    return all_perf


def load_model(args):
    if args.model_path:
        cfg = read_yaml(f"{args.model_path}/config.yaml")
        model = make_model(cfg["model"], cfg.get("model_name", "csvq+swinT"))
        model.load_state_dict(torch.load(f"{args.model_path}/model.pth", map_location="cpu")["model_state_dict"])
        return model, cfg["model"]
    from esc import synth
    from esc.models.codecs import state_manifest
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cfg = json.loads(str(np.load(os.path.join(root, "tests", "golden", f"{args.synthetic or 'base'}.npz"))["config_json"]))
    model = make_model(cfg)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(synth.synth_tensor(k, s))) for k, s in state_manifest(model.cfg).items()
                           if not k.endswith(".window")})
    return model, cfg


def run(args):
    eval_loader = DataLoader(EvalSet(args.eval_folder_path), batch_size=args.batch_size, shuffle=False, collate_fn=default_collate)
    metric_funcs = {"MelDistance": MelSpectrogramDistance().to(args.device), "SISDR": SISDR().to(args.device)}
    try:
        from .metrics import PESQ
        metric_funcs = {"PESQ": PESQ(), **metric_funcs}
    except ImportError:
        print("pesq is not installed: PESQ is skipped", file=sys.stderr)
    model, mcfg = load_model(args)
    model = model.to(args.device)
    e_counter = EntropyCounter(mcfg["codebook_size"], num_streams=mcfg["max_streams"], num_groups=mcfg["group_size"], device=args.device)
    performances = eval_epoch(model, eval_loader, metric_funcs, e_counter, args.device, num_streams=None, verbose=True, bps_per_stream=1.5)
    save_path = args.save_path or args.model_path or "."
    os.makedirs(save_path, exist_ok=True)
    json.dump(performances, open(f"{save_path}/perf_stats.json", "w"), indent=2)
    print(f"Test statistics saved into {save_path}/perf_stats.json")
    return performances


if __name__ == "__main__":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    run(parse_args())
