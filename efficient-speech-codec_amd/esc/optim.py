"""Optimiser step of the reference trainers on the MI355X: `clip_grad_norm_(params, max_norm)` + `AdamW.step()`
(`/root/reference/scripts/trainer_no_adv.py:116-117`, `scripts/utils.py:48-49`: torch.optim.AdamW(lr, betas=(0.8, 0.9)... whatever the
caller passes) as TWO kernels over the model's flat parameter / gradient buffers instead of ~430 per-tensor launches.

    opt = FlatAdamW(model, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, max_grad_norm=0.5)
    loss.backward(); opt.step(); opt.zero_grad()

`FlatAdamW` switches the model to flat-gradient mode: the training backward writes d loss / d parameter straight into one flat buffer
that every `p.grad` is a view of, so nothing is copied between backward and step.  The update rule is torch.optim.AdamW's
(decoupled weight decay, bias-corrected moments); tests/test_train.py checks it against torch step by step.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native


class FlatAdamW:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None, device=None, group=None):
        self.model, self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = model, lr, betas, eps, weight_decay, max_grad_norm
        self.group = group                   # data-parallel process group: gradients are averaged over it before clipping (DDP semantics)
        self.force_allreduce = False         # run the gradient all-reduce even on a one-rank group (exercises RCCL on a 1-GPU box)
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW works on the HIP device the model lives on")
        self.device = dev
        self.t = 0                           # number of step() calls (= every parameter's step unless a checkpoint or set_inactive() says otherwise)
        self._steps = None                   # per-parameter step counts, model.parameters() order (torch.optim.AdamW keeps `step` per parameter)
        self._inactive = set()               # parameter positions that currently have no gradient (torch: p.grad is None -> the parameter is skipped)
        self._state = None
        self.last_grad_norm = None
        model.enable_flat_grads(dev)

    def _buffers(self):
        flat, gflat = self.model.flat_buffers(self.device)
        if self._state is None:
            self._state = (flat, torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(2 + 1024, dtype=torch.float32, device=self.device))
        elif self._state[0].data_ptr() != flat.data_ptr():
            # the model rebuilt its flat buffer (model.to(), load_state_dict): the moments are indexed by the library's layout, which depends on
            # the configuration only, so they stay valid; a different size means a different model
            if self._state[1].shape != flat.shape or self._state[1].device != flat.device:
                raise RuntimeError("the model's flat parameter buffer changed size or device under a live FlatAdamW; build a new optimiser")
            self._state = (flat,) + self._state[1:]
        return flat, gflat, self._state[1], self._state[2], self._state[3]

    @torch.no_grad()
    def step(self):
        lib = _native.load()
        flat, gflat, m, v, aux = self._buffers()
        from .distributed import all_reduce_gradients
        all_reduce_gradients(gflat, self.group, force=self.force_allreduce)     # no-op unless torch.distributed is initialised with more than one rank
        self.t += 1
        p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            clip = None
            if self._inactive:        # torch does not count a None gradient in clip_grad_norm_: whatever the backward wrote for an inactive parameter is dropped first
                slices = self._param_slices()
                for i in self._inactive:
                    if slices[i] is not None:
                        gflat[slices[i][0]:slices[i][0] + slices[i][1]].zero_()
            if self.max_grad_norm is not None:
                # inactive parameters hold zero gradients in the flat buffer (zeroed above): they add nothing to the norm, as a None gradient adds nothing in torch
                _native.check(lib.escx_grad_norm_clip(p(gflat), gflat.numel(), float(self.max_grad_norm), p(aux), st))
                clip = p(aux)
                self.last_grad_norm = aux[0]
            for off, n, t in self._runs(flat.numel()):
                _native.check(lib.escx_adamw_step(p(flat, off), p(gflat, off), p(m, off), p(v, off), n, t, float(self.lr), float(self.betas[0]),
                                                  float(self.betas[1]), float(self.eps), float(self.weight_decay), clip, st))
        self.model.note_params_updated()

    # ---- per-parameter step counts / gradient-less parameters (torch.optim.AdamW semantics) -----------------------------------------------
    def set_inactive(self, names=()):
        """Parameters that have NO gradient in the coming steps (torch: `p.grad is None`, e.g. a module left out of the graph): torch's AdamW
        skips them entirely - no weight decay, no moment decay, `step` not advanced - and so does step() here.  The training backward of this
        package writes a (possibly zero) gradient for every parameter, which is what the reference's graph produces too (masked streams are
        multiplied by 0., csrvq.py:42-44), so nothing is inactive unless the caller says so; reference checkpoints written by other loops can
        carry such parameters: load_state_dict() keeps their step counts and loads entry-less parameters ACTIVE (step 0, zero moments), as torch does."""
        pos = {k: i for i, (k, _) in enumerate(self.model.named_parameters())}
        self._inactive = {pos[k] for k in names}

    def _runs(self, total):
        """[(flat offset, numel, step to apply)] covering the active parameters: ONE launch when every parameter shares a step count."""
        if self._steps is None and not self._inactive:
            return [(0, total, self.t)]
        slices = self._param_slices()
        if self._steps is None:
            self._steps = [self.t - 1] * len(slices)
        items = []
        for i, sl in enumerate(slices):
            if sl is None or i in self._inactive:
                continue
            self._steps[i] += 1
            items.append((sl[0], sl[1], self._steps[i]))
        items.sort()
        runs = []
        for off, n, t in items:
            if runs and runs[-1][2] == t and runs[-1][0] + runs[-1][1] == off:
                runs[-1] = (runs[-1][0], runs[-1][1] + n, t)
            else:
                runs.append((off, n, t))
        if not self._inactive and len({t for _, _, t in runs}) == 1 and sum(n for _, n, _ in runs) == total:
            self._steps = None; self.t = runs[0][2]           # every parameter is level again: back to the one-launch form
            return [(0, total, self.t)]
        return runs

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_flat_grads(self.device)

    # ---- checkpoint interchange: torch.optim.AdamW's state_dict layout (what the reference's trainers save and load) --------------------
    def _param_slices(self):
        """[(offset, numel, shape)] in `model.parameters()` order = the parameter indices of a torch optimiser built from them."""
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._buffers()
        where = {k: (off, n) for k, off, n in self.model._flat[idx]["layout"]}
        return [(where[k] + (tuple(p.shape),)) if k in where else None for k, p in self.model.named_parameters()]

    def state_dict(self):
        """`torch.optim.AdamW(model.parameters(), ...).state_dict()` layout: per-parameter `step` / `exp_avg` / `exp_avg_sq` cut out of the flat
        moment buffers, one param group.  A reference trainer's `optimizer.load_state_dict` reads it (trainer_no_adv.py:62, trainer_adv.py:118)."""
        flat, gflat, m, v, _ = self._buffers()
        slices = self._param_slices()
        state = {}
        for i, sl in enumerate(slices):
            t = self.t if self._steps is None else self._steps[i]
            if sl is not None and t > 0:                  # torch creates a parameter's state at its first step WITH a gradient
                off, n, shp = sl
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": m[off:off + n].view(shp).clone(), "exp_avg_sq": v[off:off + n].view(shp).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(slices))),
                 "max_grad_norm": self.max_grad_norm}            # extra key: torch ignores what it does not know
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the torch AdamW layout (a reference checkpoint's `optimizer_state_dict`) and the flat layout earlier builds of this package wrote."""
        flat, gflat, m, v, _ = self._buffers()
        if "param_groups" not in sd:                        # flat layout {step, exp_avg, exp_avg_sq, lr, ...}
            self.t = int(sd["step"]); self._steps = None; m.copy_(sd["exp_avg"]); v.copy_(sd["exp_avg_sq"])
            self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
            self.max_grad_norm = sd.get("max_grad_norm")
            return
        groups = sd["param_groups"]
        slices = self._param_slices()
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(slices):
            raise ValueError(f"optimizer state has {len(ids)} parameters, the model has {len(slices)}")
        if len({(g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]) for g in groups}) != 1 or any(g.get("amsgrad") for g in groups):
            raise NotImplementedError("FlatAdamW holds ONE hyper-parameter set (the reference builds AdamW(params, lr) with a single group, no amsgrad)")
        g0 = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
        self.max_grad_norm = g0.get("max_grad_norm", self.max_grad_norm)
        m.zero_(); v.zero_()
        steps = [0] * len(slices)                          # a parameter without a state entry never saw a gradient: step 0, zero moments (torch
        for pos, pid in enumerate(ids):                    # creates the entry lazily); torch maps saved ids to parameters by position
            st = sd["state"].get(pid)
            if st is None or slices[pos] is None:
                continue
            off, n, shp = slices[pos]
            m[off:off + n].copy_(st["exp_avg"].reshape(-1)); v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps[pos] = int(float(st["step"]))
        # torch semantics: a parameter without a state entry is ACTIVE - torch creates its state lazily and trains it as soon as it receives a gradient
        # (the reference's own `pretrained.pth`, written while the quantisers were out of the graph, is resumed exactly so: scripts/train.py --pretrain_ckp).  It starts
        # at step 0 with zero moments.  Which parameters are gradient-less in the COMING steps is the loop's knowledge, not the checkpoint's: set_inactive() says so;
        # loading a checkpoint clears whatever an earlier call had set.
        self._inactive = set()
        live = {t for t, sl in zip(steps, slices) if sl is not None}
        self.t = max(live) if live else 0
        self._steps = None if len(live) <= 1 else steps   # torch keeps `step` per parameter: differing counts get their own bias corrections
