"""Data-parallel gradient exchange for the flat gradient buffer: one process per GPU, RCCL
(`torch.distributed` backend "nccl") over xGMI, no data-path collective other than the gradient
all-reduce.

Replaces `torch.nn.parallel.DistributedDataParallel` as used by the reference
(run_training_egom2p.py:514-515 wrap, :723 `no_sync()` on non-update micro-steps, `.module`): the
hand-ordered backward reports each parameter bucket (an encoder/decoder layer, an embedding table) as
soon as its last gradient kernel is enqueued; the bucket's all-reduce is issued on a side stream behind
an event, so it overlaps with the rest of the backward.  Gradients are SUMMED here; the 1/world factor
is folded into the AdamW kernel (each rank normalises its loss by its own token counts and gradients
are then averaged - the reference's DDP semantics).

`GradBucketReducer` only needs a flat tensor and a process group, so the N>1 logic is covered on CPU
with the gloo backend (tests/test_dp_gloo.py).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, flat_grad: torch.Tensor, process_group=None, bucket_cap_mb: float = 32.0, force: bool = False):
        self.G = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: rehearse the path with one rank
        self.cap = int(bucket_cap_mb * 1024 * 1024 / flat_grad.element_size())
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self._pending: Optional[Tuple[int, int]] = None
        self._works: List = []
        self.launched: List[Tuple[int, int]] = []     # for tests / introspection
        self.last_launched: List[Tuple[int, int]] = []

    # buckets arrive tail-first and contiguous: [lo, hi) then [lo', lo) ...; merge until >= cap
    def on_bucket(self, name: str, lo: int, hi: int):
        if not self.active:
            return
        if self._pending is None:
            self._pending = (lo, hi)
        elif hi == self._pending[0]:
            self._pending = (lo, self._pending[1])
        else:                                   # not adjacent: flush what we have
            self._launch(*self._pending)
            self._pending = (lo, hi)
        if self._pending[1] - self._pending[0] >= self.cap:
            self._launch(*self._pending)
            self._pending = None

    def _launch(self, lo: int, hi: int):
        view = self.G[lo:hi]
        self.launched.append((lo, hi))
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Flush the tail bucket and make the compute stream wait for every outstanding all-reduce
        (host does not block with the NCCL/RCCL backend)."""
        if not self.active:
            return
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for w in self._works:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._works.clear()
        self.last_launched, self.launched = self.launched, []


class DataParallel(torch.nn.Module):
    """DDP-shaped wrapper for `egom2p_amd.model.EgoM2P` (`.module`, `no_sync()`, forward passthrough)."""

    def __init__(self, module, device_ids=None, process_group=None, bucket_cap_mb: float = 32.0, **_):
        super().__init__()
        self.module = module
        eng = module.engine
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.broadcast(eng.P, src=0, group=process_group)      # DDP ctor broadcast (run_training_egom2p.py:514)
            eng.weights_dirty = True
        self.reducer = GradBucketReducer(eng.G, process_group, bucket_cap_mb)
        module._bucket_done = self.reducer.on_bucket
        module._after_backward = self.reducer.finish

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        old = self.module._sync_grads
        self.module._sync_grads = False
        try:
            yield
        finally:
            self.module._sync_grads = old


def init_distributed_mode(args=None, backend: Optional[str] = None):
    """Env-var rank discovery + init_process_group (egom2p/utils/dist.py:78-100); "nccl" is RCCL on ROCm."""
    import datetime
    import os
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        if args is not None:
            args.distributed = False
        return False
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", 0))
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=4800))
    dist.barrier()
    if args is not None:
        args.rank, args.world_size, args.gpu, args.distributed, args.dist_backend = rank, world, local, True, backend
    return True
