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
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class CabiComm:
    """The C-ABI's own RCCL communicator (`ego_dp_*`, include/egom2p_hip.h): what a host without torch.distributed would use
    for the gradient exchange.  torch.distributed is only the bootstrap channel here - rank 0's 128-byte unique id travels to
    the other ranks through the existing process group's store (any backend); the exchange itself is RCCL called from the
    library on the streams given."""

    def __init__(self, device, process_group=None):
        import ctypes as C
        from . import _lib as L
        self.lib = L.load()
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        torch.cuda.set_device(device)
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            L.check(self.lib.ego_dp_unique_id(buf), "ego_dp_unique_id")
        box = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=process_group)
        self.handle = C.c_void_p()
        L.check(self.lib.ego_dp_comm_create(C.c_char_p(box[0]), self.rank, self.world, C.byref(self.handle)), "ego_dp_comm_create")

    def allreduce_begin(self, view: torch.Tensor, algo: str, compute_stream, comm_stream):
        from . import _lib as L
        assert view.is_cuda and view.dtype == torch.float32 and view.is_contiguous()
        L.check(self.lib.ego_dp_allreduce_begin(self.handle, view.data_ptr(), view.numel(), 1 if algo == "rs_ag" else 0,
                                                compute_stream.cuda_stream, comm_stream.cuda_stream), "ego_dp_allreduce_begin")

    def wait(self, compute_stream, comm_stream):
        from . import _lib as L
        L.check(self.lib.ego_dp_wait(self.handle, compute_stream.cuda_stream, comm_stream.cuda_stream), "ego_dp_wait")

    def close(self):
        if self.handle:
            self.lib.ego_dp_comm_destroy(self.handle)
            self.handle = None


class GradBucketReducer:
    def __init__(self, flat_grad: torch.Tensor, process_group=None, bucket_cap_mb: float = 32.0, force: bool = False,
                 skip: Sequence[str] = (), algo: Optional[str] = None, cabi_comm: Optional[CabiComm] = None):
        """algo: "allreduce" (RCCL's all-reduce: a ring, bound by one xGMI link per hop - fine while a step has ~1 s of
        backward to hide it) or "rs_ag": reduce-scatter + all-gather of the same bucket, in place - every rank sums one
        1/world shard and the shards travel over all 7 links at once (~2.6 ms against ~18 ms ring for
        the 1.585 GB of gradients; what a step at the released yaml's 4 clips per GPU needs).  Default: $EGOM2P_DP_ALGO or
        "allreduce".  Same sums either way up to the order of the floating-point additions across ranks."""
        import os
        self.algo = algo or os.environ.get("EGOM2P_DP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"GradBucketReducer: unknown algo {self.algo!r}")
        self.G = flat_grad
        self.skip = set(skip)                   # bucket names exchanged by other means (SparseTableExchange)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: rehearse the path with one rank
        self.cap = int(bucket_cap_mb * 1024 * 1024 / flat_grad.element_size())
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self.cabi = cabi_comm                    # exchange through the library's own RCCL communicator instead of torch.distributed
        self._pending: Optional[Tuple[int, int]] = None
        self._works: List = []
        self.launched: List[Tuple[int, int]] = []     # for tests / introspection
        self.last_launched: List[Tuple[int, int]] = []
        # optional timing of the exchange (bench.py: `exchange_ms` / `overlap_frac`): event pairs around every bucket's
        # collectives on the comm stream + one pair around the compute stream's final wait; off by default (no events)
        self.timing = False
        self._t_buckets: List = []
        self._t_wait = None
        self._t_cpu = 0.0
        self._last_timing = None

    # buckets arrive tail-first and contiguous: [lo, hi) then [lo', lo) ...; merge until >= cap
    def on_bucket(self, name: str, lo: int, hi: int):
        if not self.active or name in self.skip:
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

    def _exchange(self, lo: int, hi: int):
        """the collective(s) of one bucket, issued on the current stream (the comm stream on the GPU)"""
        view = self.G[lo:hi]
        cnt = (hi - lo) // self.world
        if self.algo == "rs_ag" and cnt > 0:
            main = view[:cnt * self.world]
            shard = main[self.rank * cnt:(self.rank + 1) * cnt]            # in place: this rank's shard of the bucket
            w = dist.reduce_scatter_tensor(shard, main, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if not self.cuda:
                w.wait()                                                   # gloo: no stream order between two collectives
            else:
                self._works.append(w)
            self._works.append(dist.all_gather_into_tensor(main, shard, group=self.pg, async_op=True))
            if cnt * self.world < hi - lo:                                 # fewer than `world` elements left over
                self._works.append(dist.all_reduce(view[cnt * self.world:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _launch(self, lo: int, hi: int):
        self.launched.append((lo, hi))
        t0 = t1 = None
        if self.cuda and self.timing:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._t_buckets.append((t0, t1, hi - lo))
        if self.cuda and self.cabi is not None:
            if t0 is not None:
                # the library makes the comm stream wait for the compute stream itself: mirror that wait here so that the
                # bucket's clock starts when its gradients are final, not when the previous bucket ended
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm_stream.wait_event(ev)
                t0.record(self.comm_stream)
            self.cabi.allreduce_begin(self.G[lo:hi], self.algo, torch.cuda.current_stream(), self.comm_stream)
            if t1 is not None:
                t1.record(self.comm_stream)
        elif self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if t0 is not None:
                    t0.record(self.comm_stream)
                self._exchange(lo, hi)
                if t1 is not None:
                    # torch's NCCL / RCCL group runs a collective on ITS OWN stream (ordered after the current one when it
                    # is issued); Work.wait() orders the current stream - here the comm stream - after it again, without
                    # blocking the host: only then does an event on the comm stream close the bucket's interval
                    for w in self._works:
                        w.wait()
                    self._works.clear()
                    t1.record(self.comm_stream)
        else:
            import time
            c0 = time.perf_counter()
            self._exchange(lo, hi)
            if self.timing:
                for w in self._works:                # gloo: the collectives of this bucket, synchronously
                    w.wait()
                self._works.clear()
                self._t_cpu += time.perf_counter() - c0
                self._t_buckets.append((None, None, hi - lo))

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
        w0 = w1 = None
        if self.cuda and self.timing:
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record(torch.cuda.current_stream())
        if self.cuda and self.cabi is not None:
            self.cabi.wait(torch.cuda.current_stream(), self.comm_stream)
        elif self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if w0 is not None:
            w1.record(torch.cuda.current_stream())
            self._t_wait = (w0, w1)
        self._works.clear()
        self.last_launched, self.launched = self.launched, []
        if self.timing:
            self._last_timing = (self._t_buckets, self._t_wait, self._t_cpu)
            self._t_buckets, self._t_wait, self._t_cpu = [], None, 0.0

    def timing_stats(self):
        """Timing of the LAST finished exchange (needs `timing = True` before the step; synchronises the device):
        exchange_ms = time the buckets' collectives took on the comm stream (summed over buckets), exposed_ms = how long the
        compute stream then still had to wait for them at the end of the backward, overlap_frac = the share of the exchange
        that ran under the backward (1 - exposed / exchange).  On a CPU group (gloo rehearsal) the collectives are
        synchronous: exposed = exchange, overlap 0."""
        if self._last_timing is None:
            return None
        buckets, wait, cpu = self._last_timing
        nbytes = sum(n for _, _, n in buckets) * self.G.element_size()
        if self.cuda:
            torch.cuda.synchronize(self.G.device)
            ex = sum(a.elapsed_time(b) for a, b, _ in buckets)
            exposed = wait[0].elapsed_time(wait[1]) if wait is not None else 0.0
            if self.cabi is None and dist.get_backend(self.pg) == "gloo":
                exposed = ex          # gloo on device tensors (one-GPU rehearsal): Work.wait() blocks the HOST - nothing overlaps
        else:
            ex = exposed = cpu * 1e3
        frac = 0.0 if ex <= 0 else max(0.0, min(1.0, 1.0 - exposed / ex))
        return {"exchange_ms": ex, "exposed_ms": exposed, "overlap_frac": frac, "buckets": len(buckets), "bytes": nbytes,
                "algo": self.algo, "backend": "cabi" if self.cabi is not None else "torch"}


class SparseTableExchange:
    """Row-list exchange of embedding-table gradients.

    DDP all-reduces the dense 64000 x 768 encoder tables (run_training_egom2p.py:514: 2 x 197 MB of the 1.585 GB), although
    a step with few clips per GPU touches only (clips x kept tokens) rows.  Here every rank compacts the rows it touched
    (flags set by the embedding backward), all-gathers (row ids, rows) padded to `cap_rows`, and sums all ranks' rows in
    rank order into the table - the same result as the dense sum (identical on every rank: one fixed order), moving
    (world - 1) * cap_rows rows instead of 2 (world - 1) / world * V.  `worth_it` is that byte comparison.

    The compaction / gather / scatter are the ego_rows_* HIP kernels; `prims` replaces them only in the CPU rehearsal of
    the collective pattern (tests/test_dp_gloo.py passes its own torch-indexing object) - the product has no CPU path."""

    def __init__(self, tables, cap_rows: int, process_group=None, prims=None):
        """tables: list of (grad view [V, D] fp32, touched flags uint8 [V])."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if prims is None:
            from . import ops                                  # raises if the HIP library is missing
            if not all(g.is_cuda for g, _ in tables):
                raise RuntimeError("SparseTableExchange: table gradients must live on the GPU (HIP row kernels)")
            prims = ops
        self.prims = prims
        self.tables = []
        for g, touched in tables:
            V, D = g.shape
            cap = min(int(cap_rows), V)
            dev = g.device
            self.tables.append(dict(g=g, touched=touched, cap=cap,
                                    rows=torch.full((cap,), -1, dtype=torch.int32, device=dev), count=torch.zeros(1, dtype=torch.int32, device=dev),
                                    send=torch.zeros(cap, D, dtype=torch.float32, device=dev),
                                    all_rows=torch.empty(self.world, cap, dtype=torch.int32, device=dev),
                                    all_count=torch.empty(self.world, 1, dtype=torch.int32, device=dev),
                                    all_send=torch.empty(self.world, cap, D, dtype=torch.float32, device=dev)))

    @staticmethod
    def worth_it(V: int, D: int, cap_rows: int, world: int) -> bool:
        gather = (world - 1) * min(cap_rows, V) * (D * 4 + 4)
        dense = 2.0 * (world - 1) / world * V * D * 4
        return world > 1 and gather < dense

    def _compact(self, t):
        self.prims.rows_compact(t["touched"], t["cap"], t["rows"], t["count"])

    def _gather(self, t):
        self.prims.rows_gather(t["g"], t["rows"], t["count"], t["cap"], t["send"])

    def _scatter(self, t, rows, count, src, add):
        self.prims.rows_scatter(t["g"], rows, count, t["cap"], src, add)

    def exchange(self):
        """Call once per optimiser step, after the last backward (the touched flags accumulate over micro-batches)."""
        for t in self.tables:
            self._compact(t)
            self._gather(t)
            if self.world > 1:
                dist.all_gather_into_tensor(t["all_rows"].view(-1), t["rows"], group=self.pg)
                dist.all_gather_into_tensor(t["all_count"].view(-1), t["count"], group=self.pg)
                dist.all_gather_into_tensor(t["all_send"].view(-1), t["send"].view(-1), group=self.pg)
            else:
                t["all_rows"][0].copy_(t["rows"]); t["all_count"][0].copy_(t["count"]); t["all_send"][0].copy_(t["send"])
            self._scatter(t, t["rows"], t["count"], None, False)            # own rows: cleared, then every rank's rows in rank order
            for r in range(self.world):
                self._scatter(t, t["all_rows"][r], t["all_count"][r], t["all_send"][r], True)

    def overflowed(self) -> bool:
        """Host-side check (one sync): did any rank touch more rows than `cap_rows` in the last exchange?"""
        return any(bool((t["all_count"] & 0x40000000).any().item()) for t in self.tables)


class DataParallel(torch.nn.Module):
    """DDP-shaped wrapper for `egom2p_amd.model.EgoM2P` (`.module`, `no_sync()`, forward passthrough)."""

    def __init__(self, module, device_ids=None, process_group=None, bucket_cap_mb: float = 32.0, algo: Optional[str] = None, **_):
        super().__init__()
        self.module = module
        eng = module.engine
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            dist.broadcast(eng.P, src=0, group=process_group)      # DDP ctor broadcast (run_training_egom2p.py:514)
            eng.weights_dirty = True
        self.reducer = GradBucketReducer(eng.G, process_group, bucket_cap_mb, algo=algo)
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this pool (no effect once HIP is up)
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
