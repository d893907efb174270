"""One optimiser step of the hot path: micro-batched forward/backward with gradient accumulation,
bucketed RCCL all-reduce overlapped with the last micro-batch's backward, then fused clip + AdamW.

Mirrors the body of `train_one_epoch` (run_training_egom2p.py:701-746) without its per-step host syncs
(`loss.item()` x5 and `torch.cuda.synchronize()`): losses stay on the device until the caller reads them.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence

import torch

from .dp import GradBucketReducer
from .engine import Engine
from .optim import FusedAdamW


class TrainStep:
    def __init__(self, engine: Engine, lr: float = 1e-4, weight_decay: float = 0.05, clip_grad: Optional[float] = 1.0,
                 process_group=None, world_size: int = 1, seed: int = 0, force_reducer: bool = False):
        self.engine = engine
        self.opt = FusedAdamW(engine, lr=lr, weight_decay=weight_decay, world_size=world_size)
        self.clip = clip_grad
        self.reducer = GradBucketReducer(engine.G, process_group, force=force_reducer) if (world_size > 1 or force_reducer) else None
        self.rng = random.Random(seed)           # decoder modality shuffle (egom2p_model.py:312), per forward
        self.loss_sum = torch.zeros(1 + engine.n_mods, device=engine.dev)

    def __call__(self, micro_batches: Sequence[Dict[str, Dict[str, torch.Tensor]]], lr: Optional[float] = None,
                 weight_decay: Optional[float] = None):
        eng = self.engine
        names = [m.name for m in eng.mods]
        k = len(micro_batches)
        self.loss_sum.zero_()
        for i, mb in enumerate(micro_batches):
            order = self.rng.sample(names, len(names))
            eng.forward(mb, dec_order=order)
            self.loss_sum += eng.loss_out
            last = i == k - 1
            eng.backward(1.0 / k, bucket_done=self.reducer.on_bucket if (last and self.reducer is not None) else None)
        if self.reducer is not None:
            self.reducer.finish()
        if lr is not None:
            for g in self.opt.param_groups:
                g["lr"] = lr * g["lr_scale"]
        if weight_decay is not None:
            self.opt.param_groups[0]["weight_decay"] = weight_decay
        norm = self.opt.step(clip_grad=self.clip, zero_grad=True)
        return self.loss_sum / k, norm
