#!/usr/bin/env python
"""EgoM2P pre-training entry point on MI355X - same launch line, YAML/CLI surface and loop structure as
the reference's `run_training_egom2p.py` (`python -m torch.distributed.run ... run_training_egom2p.py
--config cfgs/default/egom2p/models/main/<x>.yaml`, README_TRAINING.md:40-43), for the hot path:

  get_args (yaml -> set_defaults -> CLI, reference :224-239) -> init_distributed_mode -> get_model via
  create_model(args.model, encoder_embeddings, decoder_embeddings, modality_info) (:354-389) ->
  DataParallel wrap (:514) -> create_optimizer (:517) -> cosine schedule by tokens (:533-561) ->
  train_one_epoch (:678-797).

The data pipeline (webdataset tars, tokenizers) is outside the hot-path scope: `--data synthetic` feeds
clips in the reference's `mod_dict` contract (masking.py:236-266) from the counter-based generator.
Unlike the reference, the loop reads the loss from the device only every `--print_freq` steps.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from egom2p_amd import synth  # noqa: E402
from egom2p_amd.config import MODALITIES, MODEL_CFGS, ModelCfg  # noqa: E402
from egom2p_amd.dp import DataParallel, init_distributed_mode  # noqa: E402
from egom2p_amd.model import MODALITY_INFO, create_model  # noqa: E402
from egom2p_amd.optim import NativeScalerWithGradNormCount, create_optimizer  # noqa: E402


def get_args(argv=None):
    cfg_parser = argparse.ArgumentParser(add_help=False)
    cfg_parser.add_argument("-c", "--config", default="", type=str)
    p = argparse.ArgumentParser("EgoM2P pre-training (MI355X hot path)")
    p.add_argument("--run_name", default="auto")
    p.add_argument("--batch_size", default=4, type=int, help="per GPU")
    p.add_argument("--epochs", default=-1, type=int)
    p.add_argument("--total_tokens", default=-1, type=float, help="billions of (input+target) tokens")
    p.add_argument("--accum_iter", default=1, type=int)
    p.add_argument("--model", default="egom2p_base_12e_12d_swiglu_nobias")
    p.add_argument("--in_domains", default="tok_rgb-tok_depth-tok_cam-tok_gaze")
    p.add_argument("--out_domains", default="tok_rgb-tok_depth-tok_cam-tok_gaze")
    p.add_argument("--num_input_tokens", default=2048, type=int)
    p.add_argument("--num_target_tokens", default=2048, type=int)
    p.add_argument("--loss_type", default="mod")
    p.add_argument("--num_register_tokens", default=0, type=int)
    p.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "bf16"])
    p.add_argument("--opt", default="adamw")
    p.add_argument("--opt_eps", default=1e-8, type=float)
    p.add_argument("--opt_betas", default=[0.9, 0.95], type=float, nargs="+")
    p.add_argument("--clip_grad", default=None, type=float)
    p.add_argument("--weight_decay", default=0.05, type=float)
    p.add_argument("--blr", default=1e-4, type=float, help="base lr: lr = blr * global_batch / 256")
    p.add_argument("--min_blr", default=0.0, type=float)
    p.add_argument("--warmup_epochs", default=-1, type=int)
    p.add_argument("--warmup_tokens", default=-1, type=float)
    p.add_argument("--epoch_size", default=1000, type=int, help="samples per 'epoch'")
    p.add_argument("--data", default="synthetic")
    p.add_argument("--data_path", default="", help="token shards in the reference's layout, e.g. "
                   "'root/[rgb,depth,cam,gaze]/holoassist/token/shard-{000000..000195}.tar' (README_DATA.md); masked on the device")
    p.add_argument("--mask", default="device", choices=["device", "host"],
                   help="where the synthetic clips' token budgets / masks are drawn (device: ego_budget_dirichlet + ego_clip_synth)")
    p.add_argument("--data_config", default="")
    p.add_argument("--output_dir", default="")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--print_freq", default=10, type=int)
    p.add_argument("--max_steps", default=-1, type=int)
    p.add_argument("--resume", default="", help="checkpoint written by this script (model + optimiser state + epoch)")
    p.add_argument("--auto_resume", action="store_true", help="resume from the newest checkpoint-N.pth in --output_dir, if any")
    known, rest = cfg_parser.parse_known_args(argv)
    if known.config:
        with open(known.config) as f:
            p.set_defaults(**{k: v for k, v in yaml.safe_load(f).items() if k in {a.dest for a in p._actions}})
    args = p.parse_args(rest)
    args.in_domains = args.in_domains.split("-")
    args.out_domains = args.out_domains.split("-")
    return args


def get_model(args):
    """reference :354-389"""
    enc = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in args.in_domains}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in args.out_domains}
    return create_model(args.model, encoder_embeddings=enc, decoder_embeddings=dec,
                        modality_info={m: MODALITY_INFO[m] for m in set(args.in_domains) | set(args.out_domains)},
                        num_register_tokens=args.num_register_tokens)


def cosine_scheduler(base, final, total_steps, warmup_steps):
    """per-step values: linear warm-up then cosine (egom2p/utils/scheduler.py semantics)"""
    it = np.arange(total_steps)
    warm = np.linspace(0.0, base, max(warmup_steps, 1))[:warmup_steps] if warmup_steps > 0 else np.array([])
    rest = np.arange(total_steps - len(warm))
    cos = final + 0.5 * (base - final) * (1 + np.cos(math.pi * rest / max(len(rest), 1)))
    return np.concatenate([warm, cos])[:total_steps] if total_steps > 0 else it


class SyntheticClips:
    """Iterable of batched `mod_dict`s: ragged budgets from the reference's Dirichlet mixture (UnifiedMasking,
    egom2p/data/masking.py:181-266) or the canonical split.  mask="device" (default): budgets, permutations, masks and ids are
    made by the HIP kernels (ego_budget_dirichlet + ego_clip_synth) - no host tensors, no H2D copy; mask="host": the host
    generator the parity fixtures were made with."""

    def __init__(self, model_cfg: ModelCfg, batch, n_in, n_tgt, steps, seed, ragged=True, mask="device", device="cuda"):
        self.cfg, self.batch, self.n_in, self.n_tgt, self.steps, self.seed, self.ragged = model_cfg, batch, n_in, n_tgt, steps, seed, ragged
        self.mask, self.device = mask, device

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            if self.mask == "device":
                if self.ragged:
                    yield synth.make_clip_batch_device_masked(self.cfg, self.batch, self.n_in, self.n_tgt, seed=self.seed,
                                                              sample_offset=i * self.batch, device=self.device)
                else:
                    yield synth.make_clip_batch_device(self.cfg, self.batch, None, seed=self.seed, sample_offset=i * self.batch, device=self.device)
            else:
                b = synth.dirichlet_budgets(self.cfg, self.batch, self.n_in, self.n_tgt, seed=self.seed * 7919 + i) if self.ragged else None
                yield synth.make_clip_batch(self.cfg, self.batch, b, seed=self.seed, sample_offset=i * self.batch)


class ShardClips:
    """Real token shards (egom2p_amd/data.py: the reference's modified-WebDataset layout) -> GPU -> `UnifiedMasking` on the
    device: what the reference does with webdataset + CPU workers (`build_wds_fm_pretraining_dataloader`)."""

    def __init__(self, args, model, epoch, rank, world, steps, seed, device):
        from egom2p_amd.data import TokenShards
        from egom2p_amd.masking import UnifiedMasking
        info = {m: MODALITY_INFO[m] for m in args.in_domains}
        # the shard shuffle must be the same on every rank (it is dealt rank::world afterwards): args.seed, not seed + rank
        self.ds = TokenShards(args.data_path, args.batch_size, rank=rank, world=world, shuffle_seed=args.seed,
                              vocab={m: int(i["vocab_size"]) for m, i in info.items()})
        self.ds.set_epoch(epoch)
        missing = set(info) - set(self.ds.names.values())
        if missing:
            raise ValueError(f"--data_path has no folder for {sorted(missing)} (folders: {self.ds.folders})")
        self.mask = UnifiedMasking(info, None, args.num_input_tokens, args.num_target_tokens, seed=seed * 1000 + epoch, device=device)
        self.steps, self.device = steps, device

    def __len__(self):
        return self.steps

    def __iter__(self):
        for batch in self.ds.batches(self.steps):     # the same step count on every rank (cycles over the rank's shards)
            yield self.mask({k: v.to(self.device, non_blocking=True) for k, v in batch.items() if k in self.mask.names})


def train_one_epoch(model, loader, optimizer, scaler, args, epoch, start_steps, lr_values, device):
    """reference :678-797, minus the per-step host syncs"""
    model.train()
    t0, seen = time.time(), 0
    for step, x in enumerate(loader):
        it = start_steps + step // args.accum_iter
        update = (step + 1) % args.accum_iter == 0
        if step % args.accum_iter == 0 and it < len(lr_values):
            for grp in optimizer.param_groups:
                grp["lr"] = float(lr_values[it]) * grp["lr_scale"]
        mod_dict = {m: {k: v.to(device, non_blocking=True) for k, v in d.items()} for m, d in x.items()}
        ctx = torch.autocast("cuda", dtype=torch.bfloat16)      # harmless: the engine is always the bf16 recipe
        with ctx:
            if update:
                loss, mod_loss = model(mod_dict, num_encoder_tokens=args.num_input_tokens,
                                       num_decoder_tokens=args.num_target_tokens, loss_type=args.loss_type)
                grad_norm = scaler(loss / args.accum_iter, optimizer, clip_grad=args.clip_grad, parameters=model.parameters(), update_grad=True)
            else:
                with model.no_sync():
                    loss, mod_loss = model(mod_dict, num_encoder_tokens=args.num_input_tokens,
                                           num_decoder_tokens=args.num_target_tokens, loss_type=args.loss_type)
                    grad_norm = scaler(loss / args.accum_iter, optimizer, clip_grad=args.clip_grad, parameters=model.parameters(), update_grad=False)
        seen += args.batch_size
        if step % args.print_freq == 0:
            lv = loss.item()                                     # the only host sync, every print_freq steps
            if not math.isfinite(lv):
                print(f"Loss is {lv}, stopping training", file=sys.stderr)
                sys.exit(1)
            dt = time.time() - t0
            gn_txt = f"{grad_norm.item():.3f}" if grad_norm is not None else "-"        # accumulation micro-steps have no norm yet
            print(f"Epoch: [{epoch}] step {step} loss {lv:.4f} " + " ".join(f"{m}_loss {v.item():.3f}" for m, v in mod_loss.items()) +
                  f" grad_norm {gn_txt} lr {optimizer.param_groups[0]['lr']:.3e} clips/s/gpu {seen / max(dt, 1e-9):.1f}", flush=True)
        if args.max_steps > 0 and step + 1 >= args.max_steps:
            break
    torch.cuda.synchronize()
    if seen == 0:
        raise RuntimeError("train_one_epoch: the loader produced no batch")
    return {"loss": loss.item(), "clips_per_s": seen / (time.time() - t0)}


def main(args):
    distributed = init_distributed_mode(args)
    rank = args.rank if distributed else 0
    world = args.world_size if distributed else 1
    device = torch.device("cuda", torch.cuda.current_device())
    seed = args.seed + rank                                      # reference :397-399
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    model = get_model(args)
    global_batch = args.batch_size * args.accum_iter * world
    args.lr = args.blr * global_batch / 256                      # reference :498-505
    model = DataParallel(model)
    optimizer = create_optimizer(args, model.module)
    scaler = NativeScalerWithGradNormCount(enabled=False)        # bf16: GradScaler disabled (:518)
    tokens_per_step = global_batch * (args.num_input_tokens + args.num_target_tokens)
    if args.total_tokens > 0:
        total_steps = int(args.total_tokens * 1e9 / tokens_per_step)
    else:
        total_steps = max(1, args.epochs) * args.epoch_size // global_batch
    warmup_steps = int(args.warmup_tokens * 1e9 / tokens_per_step) if args.warmup_tokens > 0 else 0
    lr_values = cosine_scheduler(args.lr, args.min_blr * global_batch / 256, total_steps, min(warmup_steps, total_steps))
    steps_per_epoch = max(1, args.epoch_size // (args.batch_size * world)) if args.max_steps < 0 else args.max_steps
    epochs = max(1, math.ceil(total_steps * args.accum_iter / steps_per_epoch))
    mcfg = model.module.cfg
    if rank == 0:
        print(f"model {args.model}: {model.module.engine.num_params() / 1e6:.1f} M params; world {world}; global batch {global_batch}; "
              f"lr {args.lr:.3e}; {total_steps} optimiser steps", flush=True)
    start, first_epoch = 0, 0
    # resume (the reference's auto_load_model, egom2p/utils/checkpoint.py:123-157: explicit --resume, else the newest
    # checkpoint-N.pth of the output directory); only this script's own files, read with weights_only=True
    if not args.resume and args.auto_resume and args.output_dir and os.path.isdir(args.output_dir):
        have = [int(f[len("checkpoint-"):-4]) for f in os.listdir(args.output_dir)
                if f.startswith("checkpoint-") and f.endswith(".pth") and f[len("checkpoint-"):-4].isdigit()]
        if have:
            args.resume = os.path.join(args.output_dir, f"checkpoint-{max(have)}.pth")
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=True)
        model.module.load_state_dict(ck["model"])
        if "optimizer" in ck:
            optimizer.load_state_dict(ck["optimizer"])
        first_epoch = int(ck.get("epoch", -1)) + 1
        start = first_epoch * (steps_per_epoch // args.accum_iter)
        if rank == 0:
            print(f"resumed {args.resume}: continuing at epoch {first_epoch}, optimiser step {start}", flush=True)
    for epoch in range(first_epoch, epochs):
        # the decoder-modality shuffle draws from python's global `random` (egom2p_model.py:312): one stream per
        # (seed, epoch), so a resumed run continues with the orders an uninterrupted run would have drawn
        random.seed(seed * 1000 + epoch)
        if args.data_path:
            loader = ShardClips(args, model.module, epoch, rank, world, steps_per_epoch, seed, device)
        else:
            loader = SyntheticClips(mcfg, args.batch_size, args.num_input_tokens, args.num_target_tokens, steps_per_epoch,
                                    seed=seed * 1000 + epoch, mask=args.mask, device=device)
        stats = train_one_epoch(model, loader, optimizer, scaler, args, epoch, start, lr_values, device)
        start += steps_per_epoch // args.accum_iter
        if rank == 0:
            print(json.dumps({"epoch": epoch, **stats}), flush=True)
            if args.output_dir:
                os.makedirs(args.output_dir, exist_ok=True)
                torch.save({"model": model.module.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
                            "args": {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, list, type(None)))}},
                           os.path.join(args.output_dir, f"checkpoint-{epoch}.pth"))
        if args.max_steps > 0:
            break


if __name__ == "__main__":
    main(get_args())
