"""Per-step learning-rate / weight-decay value arrays of the train loop, indexed by the loader step `it = epoch *
steps_per_epoch + step` exactly like the reference's (egom2p/utils/scheduler.py:21-82; consumed at
run_training_egom2p.py:700-713): linear warm-up + cosine, constant (the frozen-model phase), linear warm-up + inverse
square root + linear cool-down.  Host-side numpy, no kernel.  tests/golden/schedules.npz holds the reference functions' own
outputs (oracle/make_goldens_schedules.py)."""
from __future__ import annotations

import math

import numpy as np


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if (warmup_epochs > 0 or warmup_steps > 0) else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    if n < 0:
        raise ValueError(f"warm-up ({warmup_iters} steps) is longer than the schedule ({epochs * niter_per_ep} steps)")
    i = np.arange(n)
    cos = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * k / n)) for k in i]) if n else np.array([])
    out = np.concatenate((warm, cos))
    assert len(out) == epochs * niter_per_ep
    return out


def constant_scheduler(base_value, epochs, niter_per_ep):
    return base_value * np.ones(epochs * niter_per_ep)


def inverse_sqrt_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1,
                           cooldown_epochs=0, cooldown_steps=-1, timescale=10_000):
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    cooldown_iters = cooldown_steps if cooldown_steps > 0 else cooldown_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if (warmup_epochs > 0 or warmup_steps > 0) else np.array([])
    n = epochs * niter_per_ep - warmup_iters - cooldown_iters
    if n <= 0:
        raise ValueError("warm-up + cool-down leave no step for the inverse-sqrt phase")
    i = np.arange(n)
    body = base_value * np.ones(n) if base_value == final_value else base_value / np.sqrt((i + timescale) / timescale)
    cool = np.linspace(body[-1], final_value, cooldown_iters) if (cooldown_epochs > 0 or cooldown_steps > 0) else np.array([])
    out = np.concatenate((warm, body, cool))
    assert len(out) == epochs * niter_per_ep
    return out


def build_schedules(args, steps_per_epoch: int):
    """(lr values, wd values) over args.epochs * steps_per_epoch loader steps: the frozen-model phase's constant values first
    (run_training_egom2p.py:524-531), then the main schedule (:533-561).  Needs args.lr / min_lr / frozen_model_lr /
    weight_decay / weight_decay_end / scheduler / warmup_* / cooldown_* / frozen_model_epochs / epochs."""
    wd_end = args.weight_decay if args.weight_decay_end is None else args.weight_decay_end
    fe = max(0, int(args.frozen_model_epochs))
    main_epochs = args.epochs - fe
    if main_epochs < 0:
        raise ValueError("frozen_model_epochs exceeds epochs")
    f_lr = constant_scheduler(args.frozen_model_lr, fe, steps_per_epoch) if fe > 0 else np.array([])
    f_wd = constant_scheduler(args.weight_decay, fe, steps_per_epoch) if fe > 0 else np.array([])
    we, ws = max(0, args.warmup_epochs), args.warmup_steps
    if args.scheduler == "cosine":
        lr = cosine_scheduler(args.lr, args.min_lr, main_epochs, steps_per_epoch, warmup_epochs=we, warmup_steps=ws)
        wd = cosine_scheduler(args.weight_decay, wd_end, main_epochs, steps_per_epoch)
    elif "inverse_sqrt" in args.scheduler:
        try:
            timescale = int(args.scheduler.split("-")[-1])
        except ValueError:
            timescale = 10_000
        ce, cs = max(0, args.cooldown_epochs), args.cooldown_steps
        lr = inverse_sqrt_scheduler(args.lr, args.min_lr, main_epochs, steps_per_epoch, warmup_epochs=we, warmup_steps=ws,
                                    cooldown_epochs=ce, cooldown_steps=cs, timescale=timescale)
        wd = inverse_sqrt_scheduler(args.weight_decay, wd_end, main_epochs, steps_per_epoch, cooldown_epochs=ce, cooldown_steps=cs,
                                    timescale=timescale)
    else:
        raise NotImplementedError(f"Scheduler {args.scheduler} not implemented.")
    return np.concatenate((f_lr, lr)), np.concatenate((f_wd, wd))
