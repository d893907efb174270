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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = MODEL_CFGS["ego_tiny_2e_2d"]
    sd = synth.build_state_dict(cfg, seed=1)
    md = synth.make_clip_batch(cfg, 4, {"tok_cam": (15, 15), "tok_gaze": (15, 15)}, seed=1)
    order = ["tok_gaze", "tok_cam"]
    mine = {k: {kk: vv[2 * rank:2 * rank + 2] for kk, vv in v.items()} for k, v in md.items()}
    G, names = _flat_grads(cfg, sd, mine, order)
    red = GradBucketReducer(G, bucket_cap_mb=0.25)
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


def test_bucketed_allreduce_matches_full_batch():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["err"] < 1e-5, dict(ret)


def test_reducer_is_noop_single_process():
    g = torch.arange(10.0)
    red = GradBucketReducer(g.clone())
    red.on_bucket("x", 5, 10)
    red.on_bucket("y", 0, 5)
    red.finish()
    assert red.launched == [] and red.last_launched == []
