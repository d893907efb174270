"""Two ranks on the GPU: two processes share the box's one MI355X and exchange gradients over gloo
(RCCL refuses two ranks on one device; the RCCL launch path itself is covered with a one-rank group in
test_robustness_gpu.py).  What this adds over the CPU rehearsal in test_dp_gloo.py: the REAL engine on each rank - the
hand-ordered backward reporting buckets while kernels are still in flight, the reducer's comm stream and events against
device gradients, 1/world folded into the fused AdamW - and the sparse table exchange through the ego_rows_* kernels.

Property: two ranks with 2 clips each take the same optimiser steps as one process that accumulates the same two
micro-batches (per-rank loss normalisation + gradient averaging = accumulation with 1/k, section 8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = "ego_b_2e_2d"
B, STEPS = 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clips(cfg, step, rank):
    from egom2p_amd import synth
    return synth.make_clip_batch_device(cfg, B, None, seed=40 + step, sample_offset=rank * B, device="cuda:0")


def _worker(rank, world, port, sparse, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egom2p_amd.config import MODEL_CFGS
    from egom2p_amd.engine import Engine
    from egom2p_amd.trainer import TrainStep
    cfg = MODEL_CFGS[CFG]
    eng = Engine(cfg, "cuda:0", max_batch=B, n_enc=2048, n_dec=2048)
    eng.init_random(9)
    step = TrainStep(eng, lr=1e-3, weight_decay=0.05, clip_grad=1.0, world_size=world, seed=rank,
                     clips_per_step=B, sparse_tables="on" if sparse else "off")
    assert step.reducer is not None and step.reducer.active and (step.sparse is not None) == sparse
    losses, norms = [], []
    for s in range(STEPS):
        l, norm = step([_clips(cfg, s, rank)])
        losses.append(float(l[0]))
        norms.append(float(norm))
    torch.cuda.synchronize()
    if sparse:
        assert not step.sparse.overflowed()
    assert len(step.reducer.last_launched) >= 1
    if rank == 0:
        ret["P"] = eng.P.cpu()
        ret["loss"] = losses
        ret["norm"] = norms
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse", [False, True])
def test_two_ranks_equal_one_process_with_accumulation(sparse):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, sparse, ret), nprocs=2, join=True)
    # the same steps in one process: micro-batches = the two ranks' clips, accumulated
    from egom2p_amd.config import MODEL_CFGS
    from egom2p_amd.engine import Engine
    from egom2p_amd.trainer import TrainStep
    cfg = MODEL_CFGS[CFG]
    eng = Engine(cfg, "cuda:0", max_batch=B, n_enc=2048, n_dec=2048)
    eng.init_random(9)
    step = TrainStep(eng, lr=1e-3, weight_decay=0.05, clip_grad=1.0, world_size=1, seed=0)
    norms = []
    for s in range(STEPS):
        _, norm = step([_clips(cfg, s, 0), _clips(cfg, s, 1)])
        norms.append(float(norm))
    torch.cuda.synchronize()
    # the global gradient norm (after the exchange, with the 1/world of the fused AdamW) is the scale-sensitive witness: a
    # bucket reduced twice, not at all, or a missing 1/world would move it by tens of percent.  AdamW itself is blind to a
    # uniform gradient scale and turns every sign flip of a tiny gradient (bf16 kernels, different decoder orders on the two
    # sides) into a 2 x lr move, so the parameters are compared loosely.
    for a, b in zip(ret["norm"], norms):
        assert abs(a - b) <= 2e-3 * b, (ret["norm"], norms)
    p_ref, p_dp = eng.P.cpu().double(), ret["P"].double()
    upd = float((p_dp - p_ref).norm() / (p_ref - _init_params(cfg)).norm())
    assert upd < 3e-2, upd


def _init_params(cfg):
    from egom2p_amd.engine import Engine
    e = Engine(cfg, "cuda:0", max_batch=1, n_enc=64, n_dec=64)
    e.init_random(9)
    return e.P.cpu().double()
