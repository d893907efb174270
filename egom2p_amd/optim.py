"""Optimiser side of the train loop over the engine's flat buffers.

Mirrors, for the hot path, `create_optimizer` / `get_parameter_groups` (egom2p/utils/optim_factory.py:97-230:
AdamW betas (0.9, 0.95), eps 1e-8, two groups - `no_decay` iff the name contains "norm." / ".norm" or
ends with ".bias") and `NativeScalerWithGradNormCount` (egom2p/utils/native_scaler.py:21-52: backward,
clip_grad_norm_, step).  The clip coefficient, the data-parallel 1/world factor and zero_grad are folded
into the AdamW kernel: gradients are read once and written once per step.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops


def is_no_decay(name: str, skip_list=()) -> bool:
    """optim_factory.py:113"""
    return "norm." in name or ".norm" in name or name.endswith(".bias") or name in skip_list


class FusedAdamW:
    """Duck-types the torch optimiser surface the reference loop touches: `param_groups` (with "lr",
    "weight_decay", "lr_scale"), `step`, `zero_grad`, `state_dict`, `load_state_dict`."""

    def __init__(self, engine, lr: float = 1e-4, weight_decay: float = 0.05, betas=(0.9, 0.95), eps: float = 1e-8,
                 named_parameters=None, world_size: int = 1):
        self.engine = engine
        self.betas, self.eps = betas, eps
        self.world_size = world_size
        self.m = torch.zeros_like(engine.P)
        self.v = torch.zeros_like(engine.P)
        self.sqnorm = torch.zeros(1, device=engine.dev, dtype=torch.float64)
        self.t = 0
        named = list(named_parameters) if named_parameters is not None else []
        self._named = named
        decay = [p for n, p in named if not is_no_decay(n)]
        nodecay = [p for n, p in named if is_no_decay(n)]
        self.param_groups: List[Dict] = [
            {"params": decay, "weight_decay": weight_decay, "lr": lr, "lr_scale": 1.0, "name": "decay"},
            {"params": nodecay, "weight_decay": 0.0, "lr": lr, "lr_scale": 1.0, "name": "no_decay"},
        ]
        self._frozen_sig = None
        self._runs = [(lo, hi, nd, 0) for lo, hi, nd in engine.opt_runs]
        self._frozen_ranges: List = []          # [lo, hi) ranges of the flat buffers that belong to frozen tensors
        self._frozen_names: List[str] = []
        # torch.optim.AdamW keeps state["step"] PER PARAMETER and skips parameters without a gradient (requires_grad=False), so a
        # tensor that is unfrozen after k steps (run_training_egom2p.py --frozen_model_epochs) starts its bias correction at 1,
        # not at k + 1: `skipped[name]` = optimiser steps the tensor sat out; its effective step is t - skipped[name]
        self.skipped: Dict[str, int] = {}
        self.last_grad_norm: Optional[torch.Tensor] = None

    def _active_runs(self):
        """Honour requires_grad=False (freeze_* methods of the model): frozen tensors are left untouched."""
        if not self._named:
            return self._runs
        sig = tuple(p.requires_grad for _, p in self._named)
        if sig != self._frozen_sig:
            self._frozen_sig = sig
            if all(sig) and not any(self.skipped.values()):
                self._runs = [(lo, hi, nd, 0) for lo, hi, nd in self.engine.opt_runs]
                self._frozen_ranges, self._frozen_names = [], []
            else:
                eng, runs, fro, fnames = self.engine, [], [], []
                frozen = {eng._canon_key(n) for (n, p) in self._named if not p.requires_grad} | set(eng.never_grad)
                for name, (o, n, _) in eng.offsets.items():
                    n4 = (n + 3) // 4 * 4
                    if name in frozen:
                        fnames.append(name)
                        if fro and fro[-1][1] == o:
                            fro[-1][1] = o + n4
                        else:
                            fro.append([o, o + n4])
                        continue
                    nd, sk = is_no_decay(name), self.skipped.get(name, 0)
                    if runs and runs[-1][2] == nd and runs[-1][3] == sk and runs[-1][1] == o:
                        runs[-1][1] = o + n4
                    else:
                        runs.append([o, o + n4, nd, sk])
                self._runs = [tuple(r) for r in runs]
                self._frozen_ranges = [tuple(r) for r in fro]
                self._frozen_names = [n for n in fnames if n not in eng.never_grad]
        return self._runs

    @torch.no_grad()
    def step(self, clip_grad: Optional[float] = None, zero_grad: bool = True):
        """One AdamW step on every (unfrozen) parameter.  Returns the global gradient norm (device tensor,
        of the world-averaged gradients, as clip_grad_norm_ would report it) if clipping was requested."""
        eng = self.engine
        self.t += 1
        gscale = 1.0 / self.world_size
        norm = None
        runs = self._active_runs()
        if clip_grad is not None:
            # clip_grad_norm_ of the reference (native_scaler.py:33) sees only tensors that have a gradient: the engine's
            # backward writes every tensor's gradient, so frozen ranges are left out of the norm here
            self.sqnorm.zero_()
            if not self._frozen_ranges:
                ops.grad_sqnorm(eng.G, self.sqnorm)
            else:
                for lo, hi, _, _ in runs:
                    ops.grad_sqnorm(eng.G[lo:hi], self.sqnorm)
            norm = self.sqnorm.sqrt().to(torch.float32) * gscale
        if zero_grad:
            for lo, hi in self._frozen_ranges:      # never consumed: must not accumulate from step to step
                eng.G[lo:hi].zero_()
        decay, nodecay = self.param_groups
        for name in self._frozen_names:             # this step passes them by, as torch's per-parameter step counter would
            self.skipped[name] = self.skipped.get(name, 0) + 1
        for lo, hi, nd, sk in runs:
            grp = nodecay if nd else decay
            ops.adamw_step(eng.P[lo:hi], eng.G[lo:hi], self.m[lo:hi], self.v[lo:hi], float(grp["lr"]), float(grp["weight_decay"]),
                           self.t - sk, self.betas[0], self.betas[1], self.eps, gscale=gscale,
                           max_norm=float(clip_grad) if clip_grad else 0.0, sqnorm=self.sqnorm if clip_grad else None,
                           zero_grad=zero_grad)
        eng.weights_dirty = True
        self.last_grad_norm = norm
        return norm

    def zero_grad(self, set_to_none: bool = False):
        self.engine.G.zero_()

    def state_dict(self):
        # m / v are the engine's PHYSICAL flat layout (padded heads, padded F): `layout` names it so that a resume under another
        # layout (EGOM2P_HEAD_PAD, a round-3 ego-L checkpoint) fails with a message instead of a size mismatch or, worse, a fit
        return {"m": self.m, "v": self.v, "t": self.t, "skipped": dict(self.skipped), "layout": self.engine.layout_tag(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        have, want = sd.get("layout"), self.engine.layout_tag()
        if have is None:
            # (before round 5 the state carried no tag, and round 5 moved the decoder's context_norm weights inside the flat buffers:
            #  the element counts agree, the element ORDER does not - copying would misalign every moment behind `bridge` silently)
            raise RuntimeError("optimizer state without a storage-layout tag (saved before round 5): the flat parameter order changed since; "
                               "restart the optimizer state (the model state dict, keyed by name, still loads)")
        if tuple(have) != tuple(want):
            raise RuntimeError(f"optimizer state was saved for storage layout {tuple(have)}, this engine uses {tuple(want)} "
                               "(layout version, n_flat, D, heads stored, head pitch, padded F): resume with the same EGOM2P_HEAD_PAD / model, "
                               "or restart the optimizer state")
        if sd["m"].numel() != self.m.numel():
            raise RuntimeError(f"optimizer state has {sd['m'].numel()} moments, this engine {self.m.numel()} parameters")
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.t = int(sd["t"])
        self.skipped = {str(k): int(v) for k, v in sd.get("skipped", {}).items()}
        self._frozen_sig = None                     # runs are rebuilt with the restored per-tensor step offsets
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


def create_optimizer(args, model, **_):
    """`create_optimizer(args, model_without_ddp)` of the reference (optim_factory.py:157-230), AdamW only."""
    if getattr(args, "opt", "adamw").lower().split("_")[-1] != "adamw":
        raise NotImplementedError("only AdamW is on the accelerated path")
    betas = tuple(args.opt_betas) if getattr(args, "opt_betas", None) else (0.9, 0.999)
    eps = args.opt_eps if getattr(args, "opt_eps", None) else 1e-8
    world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
    return FusedAdamW(model.engine, lr=args.lr, weight_decay=args.weight_decay, betas=betas, eps=eps,
                      named_parameters=model.named_parameters(), world_size=world)


class NativeScalerWithGradNormCount:
    """bf16 needs no loss scaling (GradScaler disabled at run_training_egom2p.py:518); same call surface."""
    state_dict_key = "amp_scaler"

    def __init__(self, enabled: bool = False):
        self.enabled = enabled

    def __call__(self, loss, optimizer, clip_grad=None, skip_grad=None, parameters=None, create_graph=False,
                 update_grad=True, compute_grad_norm=True):
        loss.backward()
        norm = None
        if update_grad:
            if skip_grad is not None:
                raise NotImplementedError("skip_grad needs a host sync per step; not on the accelerated path")
            norm = optimizer.step(clip_grad=clip_grad if clip_grad is not None else (1e30 if compute_grad_norm else None))
        return norm

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass
