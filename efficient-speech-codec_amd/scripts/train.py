"""Training loops on the MI355X (non-adversarial by default, `--adv` for the GAN step): the optimisation step of the reference trainer (`/root/reference/scripts/
trainer_no_adv.py:95-124`) without its experiment plumbing (accelerate, wandb, tqdm, checkpoint rotation are not reproduced).

Per step: sample the number of transmitted streams (quantisation dropout, `scripts/utils.py:11-25`), freeze the codebooks during the
pre-training phase (`trainer_no_adv.py:99`), training-mode forward, mel + complex-STFT + VQ losses with the config's weights
(`configs/9kbps_esc_base.yaml:29-33`), backward, clip_grad_norm 0.5, AdamW, learning-rate schedule.  Everything numerical runs in
libescx (HIP): forward / backward (`esc.ESC` in train mode), losses (`esc.modules`), optimiser (`esc.optim.FlatAdamW`).

    python -m scripts.train --synthetic base --steps 20 --batch_size 36                       # synthetic clips, synthetic init
    torchrun --nproc-per-node 8 -m scripts.train --data ./train_wavs --config cfg.yaml        # data parallel: flat-gradient all-reduce
"""
import argparse
import json
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from esc.models import make_model
from esc.modules import ComplexSTFTLoss, MelSpectrogramLoss
from esc.optim import FlatAdamW

DEFAULT_LOSS_WEIGHTS = dict(stft_weight=1.0, cm_weight=0.25, cb_weight=1.0, mel_weight=0.25)
ADV_LOSS_WEIGHTS = dict(stft_weight=0.0, cm_weight=0.25, cb_weight=1.0, mel_weight=15.0, gen_weight=1.0, feat_weight=2.0)     # configs/9kbps_esc_base_adv.yaml:41-47


def sample_streams(rng: np.random.Generator, dropout_rate: float, max_streams: int) -> int:
    """Quantisation dropout: with probability `dropout_rate` a uniform stream count in [1, max_streams], otherwise all of them."""
    if not 0.0 <= dropout_rate <= 1.0:
        raise AssertionError("dropout_rate must be within [0, 1]")
    return int(rng.integers(1, max_streams + 1)) if rng.random() < dropout_rate else max_streams


def lr_at(step: int, base_lr: float, kind: str, total_steps: int, warmup_steps: int) -> float:
    """constant | constant_warmup | cosine_warmup | exponential_decay (the four schedules of scripts/utils.py:52-65)."""
    if kind == "constant":
        return base_lr
    if kind == "constant_warmup":               # transformers.get_constant_schedule_with_warmup: step / warmup while step < warmup (0 at the first step)
        return base_lr * (step / max(1, warmup_steps) if step < warmup_steps else 1.0)
    if kind == "cosine_warmup":
        if step < warmup_steps:
            return base_lr * step / max(1, warmup_steps)
        prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    if kind == "exponential_decay":
        return base_lr * 0.999996 ** step
    raise ValueError(f"{kind} must be in ('constant', 'constant_warmup', 'cosine_warmup', 'exponential_decay')")


def scheduler_state(kind: str, base_lr: float, n_done: int, total_steps: int, warmup_steps: int) -> dict:
    """`scheduler_state_dict` of a checkpoint written after `n_done` scheduler steps: the fields torch's LRScheduler.state_dict() carries that
    determine the schedule's position (the learning rate here is a pure function of the step, `lr_at`)."""
    return {"base_lrs": [base_lr], "last_epoch": n_done, "_step_count": n_done + 1, "_last_lr": [lr_at(n_done, base_lr, kind, total_steps, warmup_steps)],
            "scheduler_type": kind, "num_training_steps": total_steps, "num_warmup_steps": warmup_steps}


class _Schedule:
    """Learning rate of the generator optimiser at step n, as the reference's loop produces it: `scheduler.step()` after every optimiser step
    (trainer_no_adv.py:118, trainer_adv.py:92), and once the optimiser is renewed after the frozen-codebook phase (at step
    pretraining_steps + 1, trainer_no_adv.py:76-79 / trainer_adv.py:140-143) the NEW optimiser runs at the constant base rate - the scheduler
    stays bound to the discarded one."""

    def __init__(self, base_lr, kind, total_steps, warmup_steps, pretraining_steps):
        lr_at(0, base_lr, kind, total_steps, warmup_steps)                              # unknown kinds raise here, not at step n
        self.base_lr, self.kind, self.total, self.warmup, self.pre = base_lr, kind, total_steps, warmup_steps, pretraining_steps

    def renew_at(self, n: int) -> bool:
        return self.pre > 0 and n == self.pre + 1

    def lr(self, n: int) -> float:
        if self.pre > 0 and n > self.pre:
            return self.base_lr
        return lr_at(n, self.base_lr, self.kind, self.total, self.warmup)


class Stepper:
    """Holds model, losses and optimiser; `step(x, n)` is one optimisation step and returns the per-loss batch means."""

    def __init__(self, model, lr, loss_weights=None, dropout_rate=1.0, pretraining_steps=0, scheduler="constant", total_steps=250000,
                 warmup_steps=0, seed=1234, group=None):
        self.model = model.train()
        self.w = dict(DEFAULT_LOSS_WEIGHTS, **(loss_weights or {}))
        self.mel, self.stft = MelSpectrogramLoss(), ComplexSTFTLoss()
        self.opt = FlatAdamW(model, lr=lr, max_grad_norm=0.5, group=group)                 # torch.optim.AdamW defaults otherwise (utils.py:48-49)
        self.base_lr, self.schedule = lr, _Schedule(lr, scheduler, total_steps, warmup_steps, pretraining_steps)
        self.dropout_rate, self.pretraining_steps = dropout_rate, pretraining_steps
        self.rng = np.random.default_rng(seed)

    def step(self, x: torch.Tensor, n: int) -> dict:
        freeze = n < self.pretraining_steps
        s = sample_streams(self.rng, self.dropout_rate, self.model.max_streams)
        if self.schedule.renew_at(n):                                                      # "Optimizer Renewed" (trainer_no_adv.py:76-79)
            self.opt = FlatAdamW(self.model, lr=self.base_lr, max_grad_norm=0.5, group=self.opt.group)
        self.opt.lr = self.schedule.lr(n)
        out = self.model(x=x, x_feat=None, num_streams=s, freeze_codebook=freeze)
        terms = {"cm_loss": out["cm_loss"], "cb_loss": out["cb_loss"], "mel_loss": self.mel(out["raw_audio"], out["recon_audio"]),
                 "stft_loss": self.stft(out["raw_feat"], out["recon_feat"])}
        total = sum(terms[k] * self.w[k.replace("_loss", "_weight")] for k in terms)
        total.mean().backward()
        self.opt.step()
        self.opt.zero_grad()
        terms["loss"] = total
        return {"streams": s, "frozen": freeze, **{k: v.detach().mean() for k, v in terms.items()}}


class AdvStepper:
    """The adversarial step of `/root/reference/scripts/trainer_adv.py:61-107`: generator update (VQ + mel [+ STFT] + LS-GAN + feature matching,
    clip 1e3) followed by the discriminator update on the detached reconstruction (clip 10); during the frozen-codebook pre-training
    phase the discriminator is not involved."""

    def __init__(self, model, disc, lr, loss_weights=None, dropout_rate=1.0, pretraining_steps=0, seed=1234, group=None, scheduler="constant",
                 total_steps=250000, warmup_steps=0, lr_disc=None):
        from esc.modules import GANLoss
        self.model, self.disc = model.train(), disc.train()
        self.w = dict(ADV_LOSS_WEIGHTS, **(loss_weights or {}))
        self.mel, self.stft, self.gan = MelSpectrogramLoss(), ComplexSTFTLoss(), GANLoss(disc)
        self.opt_g = FlatAdamW(model, lr=lr, max_grad_norm=1e3, group=group)
        self.opt_d = FlatAdamW(disc, lr=lr if lr_disc is None else lr_disc, max_grad_norm=10.0, group=group)   # constant: the scheduler drives opt_g only (trainer_adv.py:43-48)
        self.base_lr, self.schedule = lr, _Schedule(lr, scheduler, total_steps, warmup_steps, pretraining_steps)
        self.dropout_rate, self.pretraining_steps = dropout_rate, pretraining_steps
        self.rng = np.random.default_rng(seed)

    def step(self, x: torch.Tensor, n: int) -> dict:
        freeze = n < self.pretraining_steps
        s = sample_streams(self.rng, self.dropout_rate, self.model.max_streams)
        if self.schedule.renew_at(n):                               # "Generator's Optimizer Renewed" (trainer_adv.py:140-143): fresh Adam moments
            self.opt_g = FlatAdamW(self.model, lr=self.base_lr, max_grad_norm=1e3, group=self.opt_g.group)
        self.opt_g.lr = self.schedule.lr(n)                         # scheduler.step() after opt_g.step() (trainer_adv.py:92)
        out = self.model(x=x, x_feat=None, num_streams=s, freeze_codebook=freeze)
        terms = {"cm_loss": out["cm_loss"], "cb_loss": out["cb_loss"], "mel_loss": self.mel(out["raw_audio"], out["recon_audio"])}
        if self.w["stft_weight"] != 0.0:
            terms["stft_loss"] = self.stft(out["raw_feat"], out["recon_feat"])
        if not freeze:                                              # ONE discriminator pass per signal serves both updates of the step
            d_fake, d_real = self.gan.adversarial_forward(fake=out["recon_audio"], real=out["raw_audio"])
            terms["gen_loss"], terms["feat_loss"] = self.gan.generator_loss_from(d_fake, d_real)
        total = sum(terms[k] * self.w[k.replace("_loss", "_weight")] for k in terms)
        total.mean().backward()
        self.opt_g.step(); self.opt_g.zero_grad()
        log = {"streams": s, "frozen": freeze, **{k: v.detach().mean() for k, v in terms.items()}, "loss": total.detach().mean()}
        if not freeze:                                              # discriminator update on the (detached) reconstruction, trainer_adv.py:96-105
            d_loss = self.gan.discriminator_backward_from(d_fake, d_real)
            self.opt_d.step(); self.opt_d.zero_grad()
            log["disc_loss"] = d_loss.mean()
        return log


def _cpu(sd):
    return {k: (_cpu(v) if isinstance(v, dict) else (v.detach().cpu() if torch.is_tensor(v) else v)) for k, v in sd.items()}


def checkpoint(st, step: int, scheduler: str, total_steps: int, warmup_steps: int, best_perf=-1) -> dict:
    """The reference's checkpoint dictionary, key for key (trainer_no_adv.py:155-163; with a discriminator trainer_adv.py:159-168): optimiser
    states in torch.optim.AdamW's own layout, so the reference's `--pretrain_ckp` loader reads a checkpoint written here and vice versa."""
    adv = isinstance(st, AdvStepper)
    ck = {"step": step, "model_state_dict": _cpu(st.model.state_dict())}
    if adv:
        ck["model_disc_state_dict"] = _cpu(st.disc.state_dict())
    ck["optimizer_state_dict"] = _cpu((st.opt_g if adv else st.opt).state_dict())
    if adv:
        ck["optimizer_disc_state_dict"] = _cpu(st.opt_d.state_dict())
    ck["scheduler_state_dict"] = scheduler_state(scheduler, st.base_lr, step + 1, total_steps, warmup_steps)
    ck["best_perf"] = best_perf
    return ck


def _batches(args, device, rank, world):
    if args.data:
        from torch.utils.data import DataLoader, default_collate
        from .utils import EvalSet
        ds = EvalSet(args.data)
        idx = list(range(rank, len(ds), world))
        dl = DataLoader(torch.utils.data.Subset(ds, idx), batch_size=args.batch_size, shuffle=True, collate_fn=default_collate, drop_last=True)
        while True:
            for x in dl:
                yield x.to(device)
    from esc import synth
    k = 0
    while True:
        pcm = np.stack([(synth.voiced_clip_int16 if (k + i) % 2 else synth.noise_clip_int16)(f"train-r{rank}-{k + i}", args.clip_samples) for i in range(args.batch_size)])
        k += args.batch_size
        yield torch.from_numpy(synth.pcm_to_float(pcm)).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="yaml with model: / loss: blocks (reference layout)")
    ap.add_argument("--synthetic", default="base", help="without --config: base|large|tiny model block from tests/golden, synthetic init")
    ap.add_argument("--data", default=None, help="folder of 16 kHz wavs (default: synthetic clips)")
    ap.add_argument("--clip_samples", type=int, default=47920)
    ap.add_argument("--batch_size", type=int, default=9)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--pretraining_steps", type=int, default=0)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--dropout_rate", type=float, default=1.0)
    ap.add_argument("--scheduler_type", default="constant")
    ap.add_argument("--num_warmup_steps", type=int, default=0)
    ap.add_argument("--adv", action="store_true", help="adversarial training (trainer_adv.py): adds the DAC discriminator and its update")
    ap.add_argument("--save_path", default=None)
    ap.add_argument("--val_data", default=None, help="folder of 16 kHz wavs: every --eval_every steps the model is evaluated at max_streams (trainer_no_adv.py:132-145)")
    ap.add_argument("--eval_every", type=int, default=0)
    ap.add_argument("--resume", default=None, help="checkpoint.pth written by --save_path: continues at its step with the optimiser state (accel.load_state in the reference)")
    ap.add_argument("--log_steps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed)
    loss_w = None
    if args.config:
        from .utils import read_yaml
        cfg = read_yaml(args.config)
        model = make_model(cfg["model"], cfg.get("model_name", "csvq+swinT"))
        loss_w = cfg.get("loss")
    else:
        from .test import load_model
        model, _ = load_model(argparse.Namespace(model_path=None, synthetic=args.synthetic))
    model = model.to(device)
    if args.adv:
        from esc.models import Discriminator
        dcfg = (cfg.get("discriminator") if args.config else None) or dict(sample_rate=16000)
        disc = Discriminator(**dcfg).to(device)
        st = AdvStepper(model, disc, args.lr, loss_w, args.dropout_rate, args.pretraining_steps, seed=args.seed, scheduler=args.scheduler_type,
                        total_steps=args.steps, warmup_steps=args.num_warmup_steps)
    else:
        st = Stepper(model, args.lr, loss_w, args.dropout_rate, args.pretraining_steps, args.scheduler_type, args.steps, args.num_warmup_steps,
                     seed=args.seed)                                    # same seed on every rank: the ranks agree on the stream count
    data = _batches(args, device, rank, world)
    first = 0
    if args.resume:
        # the reference's checkpoint layout (trainer_no_adv.py:155-163, trainer_adv.py:159-168): a checkpoint written by either loads here
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["model_state_dict"])
        first = int(ck["step"]) + 1
        if args.adv:
            if "model_disc_state_dict" in ck:
                disc.load_state_dict(ck["model_disc_state_dict"])
                st.opt_d.load_state_dict(ck["optimizer_disc_state_dict"])
            st.opt_g.load_state_dict(ck["optimizer_state_dict"])      # a non-adversarial checkpoint = the reference's --pretrain_ckp start
        else:
            st.opt.load_state_dict(ck["optimizer_state_dict"])
        if "scheduler_state_dict" in ck and int(ck["scheduler_state_dict"].get("last_epoch", first)) != first:
            # legitimate in reference checkpoints: Accelerate steps a prepared scheduler num_processes times per optimiser step (an N-GPU
            # checkpoint has last_epoch = N * (step + 1)), and trainer_adv.py restarts `step` at 0 on a pretrain checkpoint.  This loop keys
            # its schedule on `step` (one advance per optimiser step on any world size), so it continues from `step` and says so.
            import warnings
            warnings.warn(f"checkpoint step {first - 1} and scheduler position {ck['scheduler_state_dict'].get('last_epoch')} differ; "
                          f"continuing the schedule from step {first}")
        for _ in range(first):                                          # the stream sampler and the data order continue where they stopped
            sample_streams(st.rng, st.dropout_rate, model.max_streams); next(data)
    evaluate = None
    if args.val_data and args.eval_every > 0 and rank == 0:              # the reference's evaluate(): eval_epoch at the full bitrate, no PESQ here
        from torch.utils.data import DataLoader, default_collate
        from .metrics import EntropyCounter, MelSpectrogramDistance, SISDR
        from .test import eval_epoch
        from .utils import EvalSet
        val_dl = DataLoader(EvalSet(args.val_data), batch_size=args.batch_size, shuffle=False, collate_fn=default_collate)
        mfs = {"MelDistance": MelSpectrogramDistance().to(device), "SISDR": SISDR().to(device)}
        mc = model.cfg
        ec = EntropyCounter(mc["codebook_size"], num_streams=mc["max_streams"], num_groups=mc["group_size"], device=device)

        def evaluate(step):
            perf = eval_epoch(model, val_dl, mfs, ec, device, bps_per_stream=1.5, num_streams=model.max_streams, verbose=False)
            print(json.dumps({"step": step, "eval": {k: v[0] for k, v in perf.items()}}))
    t0 = time.perf_counter()
    for n in range(first, args.steps):
        log = st.step(next(data), n)
        if evaluate and (n + 1) % args.eval_every == 0:
            evaluate(n + 1)
        if rank == 0 and ((n + 1) % args.log_steps == 0 or n == 0):
            torch.cuda.synchronize()
            vals = {k: (round(float(v), 5) if torch.is_tensor(v) else v) for k, v in log.items()}
            print(json.dumps({"step": n + 1, "s_per_step": round((time.perf_counter() - t0) / (n + 1 - first), 4), **vals}))
    if rank == 0 and args.save_path:
        os.makedirs(args.save_path, exist_ok=True)
        torch.save(checkpoint(st, args.steps - 1, args.scheduler_type, args.steps, args.num_warmup_steps), os.path.join(args.save_path, "checkpoint.pth"))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    main()
