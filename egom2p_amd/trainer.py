"""One optimiser step of the hot path: micro-batched forward/backward with gradient accumulation,
bucketed RCCL all-reduce overlapped with the last micro-batch's backward, then fused clip + AdamW.

Mirrors the body of `train_one_epoch` (run_training_egom2p.py:701-746) without its per-step host syncs
(`loss.item()` x5 and `torch.cuda.synchronize()`): losses stay on the device until the caller reads them.
"""
from __future__ import annotations

import random
from typing import Dict, Optional, Sequence

import torch

from .dp import GradBucketReducer, SparseTableExchange
from .engine import Engine
from .optim import FusedAdamW


class TrainStep:
    def __init__(self, engine: Engine, lr: float = 1e-4, weight_decay: float = 0.05, clip_grad: Optional[float] = 1.0,
                 process_group=None, world_size: int = 1, seed: int = 0, force_reducer: bool = False,
                 clips_per_step: Optional[int] = None, sparse_tables: str = "auto", dp_algo: Optional[str] = None,
                 dp_backend: str = "torch"):
        """clips_per_step (per rank) bounds the table rows one step can touch (clips x kept encoder tokens); with
        sparse_tables = "auto" the encoder tables' gradients go through the row-list exchange when that moves fewer bytes
        than the dense all-reduce ("on" / "off" force it).  dp_algo: "allreduce" | "rs_ag" (GradBucketReducer); dp_backend:
        "torch" (torch.distributed's RCCL) or "cabi" (the library's own communicator, ego_dp_*; GPU groups only)."""
        self.engine = engine
        self.opt = FusedAdamW(engine, lr=lr, weight_decay=weight_decay, world_size=world_size)
        self.clip = clip_grad
        self.sparse = None
        self.clips_per_step = clips_per_step
        skip = ()
        if (world_size > 1 or force_reducer) and sparse_tables != "off" and clips_per_step is not None:
            cap = clips_per_step * engine.N
            V = max(m.vocab_size for m in engine.mods)
            if sparse_tables == "on" or SparseTableExchange.worth_it(V, engine.D, cap, world_size):
                # small tables (cam / gaze: 256 rows) stay on the dense path: only tables the rule favours are exchanged
                picked = [(g, fl, m) for (g, fl), m in zip(engine.track_touched_table_rows(True), engine.mods)
                          if sparse_tables == "on" or SparseTableExchange.worth_it(m.vocab_size, engine.D, cap, world_size)]
                for i, m in enumerate(engine.mods):
                    if all(m is not pm for _, _, pm in picked):
                        engine.touched[i] = None
                if picked:
                    self.sparse = SparseTableExchange([(g, fl) for g, fl, _ in picked], cap, process_group)
                    skip = tuple(f"enc_table.{m.name}" for _, _, m in picked)
        cabi = None
        if dp_backend == "cabi" and (world_size > 1 or force_reducer):
            from .dp import CabiComm
            cabi = CabiComm(engine.dev, process_group)
        elif dp_backend not in ("torch", "cabi"):
            raise ValueError(f"TrainStep: unknown dp_backend {dp_backend!r}")
        self.reducer = (GradBucketReducer(engine.G, process_group, force=force_reducer, skip=skip, algo=dp_algo, cabi_comm=cabi)
                        if (world_size > 1 or force_reducer) else None)
        self.rng = random.Random(seed)           # decoder modality shuffle (egom2p_model.py:312), per forward
        self.loss_sum = torch.zeros(1 + engine.n_mods, device=engine.dev)

    def __call__(self, micro_batches: Sequence[Dict[str, Dict[str, torch.Tensor]]], lr: Optional[float] = None,
                 weight_decay: Optional[float] = None):
        eng = self.engine
        names = [m.name for m in eng.mods]
        k = len(micro_batches)
        if self.sparse is not None:
            # the row lists are sized for clips_per_step clips: more would drop touched rows from the exchange (replicas diverge)
            total = sum(next(iter(mb.values()))["input_mask"].shape[0] for mb in micro_batches)
            if total > self.clips_per_step:
                raise ValueError(f"TrainStep: {total} clips in this step, sparse table exchange sized for clips_per_step={self.clips_per_step}")
        self.loss_sum.zero_()
        for i, mb in enumerate(micro_batches):
            order = self.rng.sample(names, len(names))
            eng.forward(mb, dec_order=order, loss_grad=1.0 / k)      # the CE forms d logits in the same pass
            self.loss_sum += eng.loss_out
            last = i == k - 1
            eng.backward(1.0 / k, bucket_done=self.reducer.on_bucket if (last and self.reducer is not None) else None)
        if self.reducer is not None:
            self.reducer.finish()
        if self.sparse is not None:
            self.sparse.exchange()
        if lr is not None:
            for g in self.opt.param_groups:
                g["lr"] = lr * g["lr_scale"]
        if weight_decay is not None:
            self.opt.param_groups[0]["weight_decay"] = weight_decay
        norm = self.opt.step(clip_grad=self.clip, zero_grad=True)
        return self.loss_sum / k, norm
