"""N>1 path on CPU: the bucketed gradient reducer (egom2p_amd/dp.py) with the gloo backend, world_size 2.

Rank r computes the oracle's gradients for its half of a batch (each rank normalises its loss by its own
token counts, like the reference under DDP); the reducer sums buckets tail-first as a backward pass would
report them; sum / world must equal the full-batch gradient when per-rank token counts are equal."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egom2p_amd import synth
from egom2p_amd.config import MODEL_CFGS
from egom2p_amd.dp import GradBucketReducer
from oracle import egom2p_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(cfg, sd, md, order):
    leaf = O.make_leaf_state(sd)
    loss, _ = O.forward(leaf, cfg, md, 30, 30, dec_order=order, mode="fp32")
    loss.backward()
    seen, parts, names = set(), [], []
    for k, v in leaf.items():
        if v.requires_grad and id(v) not in seen:
            seen.add(id(v))
            parts.append(v.grad.reshape(-1))
            names.append((k, v.numel()))
    return torch.cat(parts), names


def _worker(rank, world, port, ret, algo="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = MODEL_CFGS["ego_tiny_2e_2d"]
    sd = synth.build_state_dict(cfg, seed=1)
    md = synth.make_clip_batch(cfg, 4, {"tok_cam": (15, 15), "tok_gaze": (15, 15)}, seed=1)
    order = ["tok_gaze", "tok_cam"]
    mine = {k: {kk: vv[2 * rank:2 * rank + 2] for kk, vv in v.items()} for k, v in md.items()}
    G, names = _flat_grads(cfg, sd, mine, order)
    red = GradBucketReducer(G, bucket_cap_mb=0.25, algo=algo)
    # report buckets tail-first, tensor by tensor, as the hand-ordered backward does
    hi = G.numel()
    for name, n in reversed(names):
        red.on_bucket(name, hi - n, hi)
        hi -= n
    red.finish()
    assert hi == 0
    covered = sorted(red.last_launched)
    assert covered[0][0] == 0 and covered[-1][1] == G.numel()
    assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))       # every element reduced exactly once
    assert len(covered) > 1                                                   # really bucketed
    G /= world
    if rank == 0:
        full, _ = _flat_grads(cfg, sd, md, order)
        ret["err"] = float((G - full).norm() / full.norm())
        ret["buckets"] = len(covered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_bucketed_allreduce_matches_full_batch(algo):
    """both exchange algorithms of GradBucketReducer: RCCL-style all-reduce per bucket, and reduce-scatter + all-gather in
    place (bucket sizes are not multiples of the world size here: the leftover elements take the all-reduce path)"""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret, algo), nprocs=2, join=True)
    assert ret["err"] < 1e-5, dict(ret)


def test_reducer_is_noop_single_process():
    g = torch.arange(10.0)
    red = GradBucketReducer(g.clone())
    red.on_bucket("x", 5, 10)
    red.on_bucket("y", 0, 5)
    red.finish()
    assert red.launched == [] and red.last_launched == []


class TorchRowPrims:
    """CPU stand-ins for the ego_rows_* kernels (same contracts: egom2p_hip.h), used ONLY to rehearse the collective
    pattern of dp.SparseTableExchange over gloo; the kernels themselves are tested on the GPU (test_frontend_gpu.py)."""

    @staticmethod
    def rows_compact(touched, cap, rows, count):
        idx = torch.nonzero(touched).flatten().to(torch.int32)
        n = min(idx.numel(), cap)
        rows.fill_(-1)
        rows[:n] = idx[:n]
        count[0] = n | (0x40000000 if idx.numel() > cap else 0)
        touched.zero_()

    @staticmethod
    def rows_gather(table, rows, count, cap, out):
        n = int(count[0]) & 0x3fffffff
        out.zero_()
        out[:n] = table[rows[:n].long()]

    @staticmethod
    def rows_scatter(table, rows, count, cap, src, add):
        n = int(count[0]) & 0x3fffffff
        idx = rows[:n].long()
        if src is None:
            table[idx] = 0
        elif add:
            table[idx] += src[:n]
        else:
            table[idx] = src[:n]


def _sparse_worker(rank, world, port, ret):
    from egom2p_amd.dp import SparseTableExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    V, D, cap = 3000, 48, 256
    gen = torch.Generator().manual_seed(100 + rank)
    tables = []
    for t in range(2):                                       # two tables, overlapping and rank-private rows
        g = torch.zeros(V, D)
        touched = torch.zeros(V, dtype=torch.uint8)
        rows = torch.randperm(V, generator=gen)[:150 + 30 * t]
        rows = torch.cat([rows, torch.tensor([7, 11, 13 + t])])          # rows every rank touches
        for _ in range(2):                                   # two micro-batches accumulate into the same rows
            g[rows] += torch.randn(rows.numel(), D, generator=gen)
            touched[rows] = 1
        tables.append((g, touched))
    dense = [g.clone() for g, _ in tables]
    for d in dense:
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
    ex = SparseTableExchange(tables, cap_rows=cap, prims=TorchRowPrims)
    assert ex.world == 2
    ex.exchange()
    assert not ex.overflowed()
    worst = 0.0
    for (g, touched), d in zip(tables, dense):
        assert int(touched.sum()) == 0                       # flags consumed
        worst = max(worst, float((g - d).abs().max()))
    # the result is bitwise the same on both ranks (one fixed summation order)
    chk = torch.cat([g.reshape(-1) for g, _ in tables]).double().sum().reshape(1)
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, chk)
    # a cap that is too small is reported, not silently truncated
    tiny = SparseTableExchange([(torch.zeros(V, D), torch.ones(V, dtype=torch.uint8))], cap_rows=16, prims=TorchRowPrims)
    tiny.exchange()
    if rank == 0:
        ret["worst"] = worst
        ret["same"] = float(both[0]) == float(both[1])
        ret["overflow_seen"] = tiny.overflowed()
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_table_exchange_equals_dense_allreduce():
    """(row id, row) lists instead of the dense all-reduce of an embedding table."""
    from egom2p_amd.dp import SparseTableExchange
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sparse_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["worst"] < 1e-6 and ret["same"] and ret["overflow_seen"], dict(ret)
    # the byte rule: 4 clips x 2048 kept rows per GPU at 8 GPUs favours the lists, 256 clips per GPU the dense all-reduce
    assert SparseTableExchange.worth_it(64000, 768, 4 * 2048, 8)
    assert not SparseTableExchange.worth_it(64000, 768, 256 * 2048, 8)
    assert not SparseTableExchange.worth_it(256, 768, 4 * 2048, 8)          # cam / gaze tables stay dense


def _timing_worker(rank, world, port, ret):
    """the bench's multi-rank branch without the GPU: a timed bucketed exchange + bench.py's dp_report"""
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_dp", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {}
    for algo in ("allreduce", "rs_ag"):
        G = torch.full((300_003,), float(rank + 1))
        red = GradBucketReducer(G, bucket_cap_mb=0.25, algo=algo)
        red.timing = True
        hi = G.numel()
        for i, n in enumerate([100_001, 50_000, 150_002]):          # three buckets, tail first
            red.on_bucket(f"b{i}", hi - n, hi)
            hi -= n
        red.finish()
        assert torch.equal(G, torch.full_like(G, 3.0))               # 1 + 2 on every element, every bucket exchanged exactly once
        par, extra = bench.dp_report(red, world)
        out[algo] = (par, extra)
    assert bench.dp_report(None, 1) == ({"parallelism": "dp1"}, {})
    if rank == 0:
        ret.put(out)
    dist.destroy_process_group()


def test_bench_dp_report_over_gloo_world2():
    """`bench.py --gpus N` records WHICH exchange ran and how it overlapped (`config.parallelism` = dp<N>:<algo>:<backend>,
    `exchange_ms`, `exposed_ms`, `overlap_frac`, bytes and bucket count from event / wall-clock pairs around every bucket's
    collectives): rehearsed with two gloo ranks on CPU tensors - the same reducer code path the GPU ranks take."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timing_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    out = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for algo in ("allreduce", "rs_ag"):
        par, extra = out[algo]
        assert par == {"parallelism": f"dp2:{algo}:torch"}
        assert extra["exchange_bytes"] == 300_003 * 4 and extra["exchange_buckets"] >= 2
        assert extra["exchange_ms"] > 0 and extra["exposed_ms"] == extra["exchange_ms"] and extra["overlap_frac"] == 0.0   # gloo: synchronous
        assert extra["exchange_gbs"] > 0
