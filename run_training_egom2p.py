#!/usr/bin/env python
"""EgoM2P pre-training entry point on MI355X - same launch line, YAML/CLI surface and loop structure as
the reference's `run_training_egom2p.py` (`python -m torch.distributed.run ... run_training_egom2p.py
--config cfgs/default/egom2p/models/main/<x>.yaml`, README_TRAINING.md:40-43), for the hot path:

  get_args (yaml -> set_defaults -> CLI, reference :224-239) -> init_distributed_mode -> get_model via
  create_model(args.model, encoder_embeddings, decoder_embeddings, modality_info) (:354-389) ->
  DataParallel wrap (:514) -> create_optimizer (:517) -> lr / weight-decay schedules per loader step (cosine or
  inverse_sqrt, a constant frozen-model phase in front: :524-561, egom2p_amd/scheduler.py) -> per epoch train_one_epoch
  (:678-797: freezes the shared parameters during the frozen-model epochs, :686-693) and, every --eval_freq epochs,
  evaluate (:800-835) on held-out clips.

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
from egom2p_amd.scheduler import build_schedules  # noqa: E402


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
    p.add_argument("--weight_decay_end", default=None, type=float, help="final weight decay (default: constant)")
    p.add_argument("--blr", default=1e-4, type=float, help="base lr: lr = blr * global_batch / 256")
    p.add_argument("--min_blr", default=0.0, type=float)
    p.add_argument("--frozen_model_blr", default=-1, type=float, help="base lr of the frozen-model phase (default: blr)")
    p.add_argument("--scheduler", default="cosine", help="cosine | inverse_sqrt-<timescale>")
    p.add_argument("--warmup_epochs", default=-1, type=int)
    p.add_argument("--warmup_steps", default=-1, type=int)
    p.add_argument("--warmup_tokens", default=-1, type=float)
    p.add_argument("--cooldown_epochs", default=0, type=int)
    p.add_argument("--cooldown_steps", default=-1, type=int)
    p.add_argument("--cooldown_tokens", default=-1, type=float)
    p.add_argument("--frozen_model_epochs", default=0, type=int, help="epochs in which only the input / output embeddings train")
    p.add_argument("--frozen_model_tokens", default=0, type=float, help="the same in billions of tokens")
    p.add_argument("--frozen_embedding_domain", default=None, type=str, help="'-'-joined modalities whose embeddings stay frozen too")
    p.add_argument("--eval_freq", default=1, type=int, help="evaluate every this many epochs (and after the last one)")
    p.add_argument("--eval_steps", default=0, type=int, help="held-out batches per evaluation (0 = no evaluation)")
    p.add_argument("--eval_data_path", default="", help="held-out token shards (layout of --data_path); synthetic held-out clips if empty")
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


def train_one_epoch(model, loader, optimizer, scaler, args, epoch, start_steps, lr_values, device, wd_values=None):
    """reference :678-797, minus the per-step host syncs.  `start_steps` = epoch * steps per epoch: the schedules are indexed
    by the loader step like the reference's (:702); during the frozen-model epochs only the embeddings train (:686-693)."""
    model.train()
    if args.frozen_model_epochs > 0 and epoch < args.frozen_model_epochs:
        if args.frozen_embedding_domain is None:
            model.module.freeze_shared_params()
        else:
            model.module.freeze_params_except_specific_embeddings(args.frozen_embedding_domain)
    else:
        model.module.unfreeze_all()
    t0, seen = time.time(), 0
    for step, x in enumerate(loader):
        it = start_steps + step
        update = (step + 1) % args.accum_iter == 0
        if step % args.accum_iter == 0 and it < len(lr_values):
            for grp in optimizer.param_groups:                   # :707-713
                grp["lr"] = float(lr_values[it]) * grp["lr_scale"]
                if wd_values is not None and grp["weight_decay"] > 0:
                    grp["weight_decay"] = float(wd_values[it])
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
                  f" grad_norm {gn_txt} lr {optimizer.param_groups[0]['lr']:.3e} wd {optimizer.param_groups[0]['weight_decay']:.4f} "
                  f"clips/s/gpu {seen / max(dt, 1e-9):.1f}", flush=True)
        if args.max_steps > 0 and step + 1 >= args.max_steps:
            break
    torch.cuda.synchronize()
    if seen == 0:
        raise RuntimeError("train_one_epoch: the loader produced no batch")
    return {"loss": loss.item(), "clips_per_s": seen / (time.time() - t0)}


@torch.no_grad()
def evaluate(model, loader, device, args, prefix="[Eval] "):
    """reference :800-835: a no-grad pass over held-out batches in eval mode, the loss and the per-modality losses averaged over
    the batches and over the ranks (MetricLogger.global_avg + synchronize_between_processes, logger.py:52-63).  The sums stay
    on the device; one host read at the end."""
    model.eval()
    tot, names, n = None, None, 0
    for x in loader:
        mod_dict = {m: {k: v.to(device, non_blocking=True) for k, v in d.items()} for m, d in x.items()}
        loss, mod_loss = model(mod_dict, num_encoder_tokens=args.num_input_tokens, num_decoder_tokens=args.num_target_tokens,
                               loss_type=args.loss_type)
        vec = torch.stack([loss.detach().float()] + [v.detach().float() for v in mod_loss.values()])
        tot = vec.double() if tot is None else tot + vec.double()
        names = list(mod_loss.keys())
        n += 1
    if n == 0:
        return {}
    cnt = torch.tensor([float(n)], device=tot.device, dtype=torch.float64)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        both = torch.cat([tot, cnt])
        torch.distributed.all_reduce(both)
        tot, cnt = both[:-1], both[-1:]
    avg = (tot / cnt).cpu().tolist()
    out = {prefix + "loss": avg[0]}
    out.update({prefix + f"{m}_loss": v for m, v in zip(names, avg[1:])})
    print("Eval averaged stats: " + "  ".join(f"{k}: {v:.4f}" for k, v in out.items()), flush=True)
    model.train()
    return out


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
    # ---- epochs / warm-up / cool-down / frozen phase from token budgets (reference :433-470); an "epoch" is epoch_size clips
    tok_per_clip = args.num_input_tokens + args.num_target_tokens
    per_loader_step = tok_per_clip * args.batch_size * world
    if args.epochs < 0:
        # (the reference stops when neither --epochs nor --total_tokens is given; here that means one epoch)
        args.epochs = 1 if args.total_tokens < 0 else math.ceil(args.total_tokens * 1e9 / (tok_per_clip * args.epoch_size))
    elif args.total_tokens > 0:
        raise SystemExit("Epochs and total tokens are both non-negative, stopping training.")
    if args.warmup_epochs < 0 and args.warmup_steps < 0:
        args.warmup_steps = math.ceil(args.warmup_tokens * 1e9 / per_loader_step) if args.warmup_tokens > 0 else 0
    if args.cooldown_epochs < 0 and args.cooldown_steps < 0:
        if args.cooldown_tokens < 0 and "inverse_sqrt" in args.scheduler:
            raise SystemExit("Cooldown epochs, steps and total tokens all set to negative values, stopping training.")
        args.cooldown_steps = math.ceil(args.cooldown_tokens * 1e9 / per_loader_step)
    if args.frozen_model_epochs <= 0:
        if args.frozen_model_tokens > 0:
            args.frozen_model_epochs = math.ceil(args.frozen_model_tokens * 1e9 / (tok_per_clip * args.epoch_size))
    elif args.frozen_model_tokens > 0:
        raise SystemExit("Frozen_model_epochs and frozen_model_tokens are both non-negative, stopping training.")
    args.min_lr = args.min_blr * global_batch / 256
    args.frozen_model_lr = (args.frozen_model_blr if args.frozen_model_blr > 0 else args.blr) * global_batch / 256
    steps_per_epoch = max(1, args.epoch_size // (args.batch_size * world)) if args.max_steps < 0 else args.max_steps
    epochs = max(1, args.epochs)
    args.epochs = epochs
    lr_values, wd_values = build_schedules(args, steps_per_epoch)                 # indexed by the loader step (:702)
    total_steps = epochs * steps_per_epoch // args.accum_iter
    mcfg = model.module.cfg
    if rank == 0:
        print(f"model {args.model}: {model.module.engine.num_params() / 1e6:.1f} M params; world {world}; global batch {global_batch}; "
              f"lr {args.lr:.3e}; {total_steps} optimiser steps", flush=True)
    first_epoch = 0
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
        if rank == 0:
            print(f"resumed {args.resume}: continuing at epoch {first_epoch}, loader step {first_epoch * steps_per_epoch}", flush=True)
    for epoch in range(first_epoch, epochs):
        # the decoder-modality shuffle draws from python's global `random` (egom2p_model.py:312): one stream per
        # (seed, epoch), so a resumed run continues with the orders an uninterrupted run would have drawn
        random.seed(seed * 1000 + epoch)
        if args.data_path:
            loader = ShardClips(args, model.module, epoch, rank, world, steps_per_epoch, seed, device)
        else:
            loader = SyntheticClips(mcfg, args.batch_size, args.num_input_tokens, args.num_target_tokens, steps_per_epoch,
                                    seed=seed * 1000 + epoch, mask=args.mask, device=device)
        stats = train_one_epoch(model, loader, optimizer, scaler, args, epoch, epoch * steps_per_epoch, lr_values, device, wd_values)
        if args.eval_steps > 0 and ((epoch + 1) % max(1, args.eval_freq) == 0 or epoch + 1 == epochs):
            # held-out clips (reference :641-649): shards of --eval_data_path, or synthetic clips from a seed stream no training
            # epoch uses; the decoder-order stream is re-seeded so that every evaluation sees the same batches and orders
            random.seed(seed * 1000 + 999_983)
            if args.eval_data_path:
                ev_args = argparse.Namespace(**{**vars(args), "data_path": args.eval_data_path})
                ev_loader = ShardClips(ev_args, model.module, 0, rank, world, args.eval_steps, seed + 7919, device)
            else:
                ev_loader = SyntheticClips(mcfg, args.batch_size, args.num_input_tokens, args.num_target_tokens, args.eval_steps,
                                           seed=(seed + 7919) * 1000 + 999, mask=args.mask, device=device)
            stats.update(evaluate(model, ev_loader, device, args))
        if rank == 0:
            print(json.dumps({"epoch": epoch, **stats}), flush=True)
            if args.output_dir:
                os.makedirs(args.output_dir, exist_ok=True)
                with open(os.path.join(args.output_dir, "log.txt"), "a", encoding="utf-8") as f:       # reference :669-671
                    f.write(json.dumps({"epoch": epoch, **stats}) + "\n")
            if args.output_dir:
                os.makedirs(args.output_dir, exist_ok=True)
                torch.save({"model": model.module.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
                            "args": {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, list, type(None)))}},
                           os.path.join(args.output_dir, f"checkpoint-{epoch}.pth"))
        if args.max_steps > 0:
            break


if __name__ == "__main__":
    main(get_args())
