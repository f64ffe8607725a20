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


def _bitrate_pass(model, batches, metric_funcs, e_counter, device, s):
    """One sweep of the evaluation set at `s` streams: per-clip metric values and the code utilisation of this bitrate."""
    scores = {name: [] for name in metric_funcs}
    e_counter.reset_stats(num_streams=s)
    for x in batches:
        x = x.to(device)
        out = model(x=x, x_feat=None, num_streams=s)
        for name, fn in metric_funcs.items():
            scores[name] += fn(x, out["recon_audio"]).tolist()
        e_counter.update(out["codes"])
    return scores, e_counter.compute_utilization()[0]


@torch.no_grad()
def eval_epoch(model, eval_loader, metric_funcs, e_counter, device, bps_per_stream, num_streams=None, verbose=True):
    """Counterpart of the reference's eval_epoch (scripts/test.py:23-55): the same arguments and the same result layout
    {metric: [mean at each evaluated bitrate, rounded to 4], "utilization": [...]}, bitrates = one (`num_streams`) or 1..max_streams.
    The model is put in eval mode for the sweep and its previous train/eval flag is restored afterwards (the reference forces
    train(); here a caller that evaluates an inference-only model keeps an inference-only model)."""
    was_training = model.training
    model.eval()
    streams = [num_streams] if num_streams is not None else list(range(1, model.max_streams + 1))
    table = {name: [] for name in metric_funcs}
    table["utilization"] = []
    try:
        for s in streams:
            scores, rate = _bitrate_pass(model, eval_loader, metric_funcs, e_counter, device, s)
            for name, vals in scores.items():
                table[name].append(round(float(np.mean(vals)), 4))
            table["utilization"].append(rate)
            if verbose:
                line = " | ".join(f"{name}: {np.mean(vals):.4f}" for name, vals in scores.items())
                print(f"Test Metrics at {s * bps_per_stream:.2f}kbps: {line} | utilization: {rate:.4f}")
    finally:
        model.train(was_training)
    return table


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
