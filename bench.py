#!/usr/bin/env python
"""Throughput of the EgoM2P hot path on MI355X: training steps (forward + backward + gradient all-reduce
+ clip + AdamW) of ego-b (400M) on synthetic 10,300-position clips, one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 3 --warmup 1

Prints ONE JSON line on rank 0.  `value` = clip positions (10,300 per clip) per second over all GPUs,
inputs resident in HBM.  A "step" = one optimiser step over `--clips-per-gpu` clips (default 256: the
global batch of 2048 of BASELINE.json at 8 GPUs), micro-batched with gradient accumulation.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); normally already exported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def flops_per_clip(cfg, budgets, n_enc, n_dec):
    """Algorithmic forward FLOPs per clip, block-sparse decoder self-attention."""
    D, F, Le, Ld = cfg.dim, cfg.mlp_hidden, cfg.encoder_depth, cfg.decoder_depth
    N = sum(budgets[m.name][0] for m in cfg.mods)
    M = sum(budgets[m.name][1] for m in cfg.mods)
    s_dec = sum(budgets[m.name][1] ** 2 for m in cfg.mods)
    enc = Le * (8 * N * D * D + 4 * N * N * D + 6 * N * D * F)
    dec = Ld * (8 * M * D * D + 4 * s_dec * D + (4 * M * D * D + 4 * N * D * D + 4 * M * N * D) + 6 * M * D * F)
    ctx = 2 * N * D * D
    logits = sum(2 * budgets[m.name][1] * D * m.vocab_size for m in cfg.mods)
    return float(enc + dec + ctx + logits)


def cpu_baseline(cfg, sd, synth, budgets, rank, order, n_enc, n_dec, protocol):
    """The oracle (CPU restatement of the reference's forward, validated against the reference's own outputs) timed on
    the host cores: forward + backward of whole clips of the same workload.  SURVEY.md section 8(d) protocol = one
    warm-up, then the median of 5 runs, for B in {1, 4}, in fp32 and in the autocast-emulating bf16 mode ("full",
    ~20 min); the default ("survey-fp32") runs the fp32 legs of that protocol under a budget of CPU seconds (~5 min); "quick":
    one warm-up + median of 3 at B = 1 fp32, one run of bf16 mode.  `value` is always the B = 1 fp32 median."""
    import statistics
    from oracle import egom2p_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("EGOM2P_CPU_THREADS", "16"))))   # the GPU box's CPU share per GPU is 16
    torch.set_num_threads(cores)

    def once(md, mode):
        leaf = O.make_leaf_state(sd)
        t0 = time.time()
        loss, _ = O.forward(leaf, cfg, md, n_enc, n_dec, dec_order=order, mode=mode)
        loss.backward()
        return time.time() - t0, float(loss.item())

    full = [(1, "fp32", 1, 5), (4, "fp32", 1, 5), (1, "bf16", 1, 5), (4, "bf16", 1, 5)]
    plan = {"quick": [(1, "fp32", 1, 3), (1, "bf16", 0, 1)], "full": full, "survey-fp32": full[:2],
            "full-fp32": full[:2], "full-bf16": [full[0][:3] + (1,)] + full[2:]}[protocol]   # halves of "full" for a 20-minute box limit
    # "survey-fp32" (the default): SURVEY 8(d)'s fp32 legs - one warm-up + median of 5 at B = 1 and at B = 4 - under a budget of CPU
    # seconds so that the default run stays within minutes on a slow host: a leg stops early (never below 3 timed runs) when
    # its next run would not fit, and the B = 4 leg is left out (and said so) when not even warm-up + 3 runs of it fit
    budget = float(os.environ.get("EGOM2P_CPU_BASELINE_BUDGET_S", "340")) if protocol == "survey-fp32" else float("inf")
    runs, total, notes = [], 0.0, []
    t_clip = None                                            # seconds per clip seen so far (B = 1 fp32)
    for B, mode, n_warm, n_timed in plan:
        if t_clip is not None and total + (n_warm + 3) * B * t_clip > budget:
            notes.append(f"B={B} {mode} leg left out: warm-up + 3 runs (~{(n_warm + 3) * B * t_clip:.0f} s) do not fit the {budget:.0f} s budget")
            continue
        md = synth.make_clip_batch(cfg, B, budgets, seed=100 + rank, sample_offset=0)      # the GPU run's first clips, host copy
        for _ in range(n_warm):
            total += once(md, mode)[0]
        ts, loss = [], None
        for _ in range(n_timed):
            if len(ts) >= 3 and total + sum(ts) + ts[-1] > budget:
                notes.append(f"B={B} {mode}: {len(ts)} of {n_timed} timed runs (budget {budget:.0f} s)")
                break
            t, loss = once(md, mode)
            ts.append(t)
            print(f"[cpu_baseline] B={B} {mode}: {t:.1f} s", file=sys.stderr, flush=True)
        total += sum(ts)
        med = statistics.median(ts)
        if B == 1 and mode == "fp32":
            t_clip = med
        runs.append({"batch": B, "mode": mode, "warmup": n_warm, "timed": len(ts), "median_s": round(med, 3),
                     "clip_positions_per_s": round(B * 10300.0 / med, 1), "loss": loss})
    head = runs[0]
    what = {"quick": " (a bounded sample: 1 warm-up + median of 3 at B = 1 fp32 and one bf16-mode run; SURVEY 8(d)'s protocol is 'survey-fp32' / 'full')",
            "survey-fp32": " (SURVEY 8(d): 1 warm-up + median of 5, B in {1, 4}, fp32, under a CPU-seconds budget; the bf16-mode legs are '--cpu-baseline full')"}
    return {"value": head["clip_positions_per_s"], "unit": "clip-positions/s", "cores": torch.get_num_threads(), "kind": "port",
            "protocol": protocol + what.get(protocol, ""),
            "sample": f"protocol '{protocol}': B=1 fp32 fwd+bwd of one 10300-position clip (N=M={n_enc}), {head['warmup']} warm-up + "
                      f"median of {head['timed']} ({head['median_s']} s each); all legs {total:.0f} s of CPU work" + ("; " + "; ".join(notes) if notes else ""),
            "loss": head["loss"], "runs": runs}


def dp_report(reducer, world, sparse=None):
    """Data-parallel fields of the JSON line (world > 1, or the one-rank rehearsal): which exchange ran and how it overlapped.
    `exchange_ms` = the buckets' collectives on the comm stream, `exposed_ms` = what the compute stream still waited for at
    the end of the backward, `overlap_frac` = 1 - exposed / exchange (GradBucketReducer.timing_stats)."""
    st = reducer.timing_stats() if reducer is not None else None
    if st is None:
        return {"parallelism": f"dp{world}"}, {}
    tag = f"dp{world}:{st['algo']}:{st['backend']}" + (":sparse-tables" if sparse is not None else "")
    extra = {"exchange_ms": round(st["exchange_ms"], 3), "exposed_ms": round(st["exposed_ms"], 3), "overlap_frac": round(st["overlap_frac"], 4),
             "exchange_buckets": st["buckets"], "exchange_bytes": st["bytes"],
             "exchange_gbs": round(st["bytes"] / max(st["exchange_ms"], 1e-9) / 1e6, 3)}
    return {"parallelism": tag}, extra


def generation_flops(cfg, n_cond, n_target, steps, target_vocab):
    """Algorithmic forward FLOPs of one ROAR + CFG generation of `n_target` tokens conditioned on `n_cond` tokens
    (generate.py:747-817): per step a conditional pass (encoder over the conditioning + everything
    decoded so far) and an unconditional one (encoder over the decoded tokens only; step 0: empty context, the
    cross-attention is skipped), decoder over the step's tokens with unmasked self-attention, logits for them."""
    D, F, Le, Ld = cfg.dim, cfg.mlp_hidden, cfg.encoder_depth, cfg.decoder_depth
    per = [n_target // steps + (1 if i < n_target % steps else 0) for i in range(steps)]
    tot, done = 0.0, 0
    for M in per:
        for N in (n_cond + done, done):
            enc = Le * (8 * N * D * D + 4 * N * N * D + 6 * N * D * F) + 2 * N * D * D
            cross = (4 * M * D * D + 4 * N * D * D + 4 * M * N * D) if N > 0 else 0
            dec = Ld * (8 * M * D * D + 4 * M * M * D + cross + 6 * M * D * F)
            tot += enc + dec + 2 * M * D * target_vocab
        done += M
    return float(tot)


def extra_config4(device, repeats_b1=5, repeats_b8=3):
    """BASELINE config 4 (rgb -> depth, `eval_model_rgb2depth.py`: ROAR 3 steps, CFG 2.0, top-p 0.8, the whole schedule one
    hipGraph) on a fresh full-depth ego-b with random weights: batch-1 latency and batch-8 throughput, in this process."""
    import time as _t
    from egom2p_amd.config import MODEL_CFGS
    from egom2p_amd.eval_generation import TASKS
    from egom2p_amd.generate import GenerationSampler, build_chained_generation_schedules, init_empty_target_modality, init_full_input_modality
    from egom2p_amd.model import MODALITY_INFO, create_model
    from egom2p_amd import synth
    from egom2p_amd.profiler import PEAK_BF16_TFLOPS
    t = TASKS["rgb2depth"]
    mods = ["tok_rgb", "tok_depth", "tok_cam", "tok_gaze"]
    name = "egom2p_base_12e_12d_swiglu_nobias"
    torch.set_grad_enabled(False)
    try:
        model = create_model(name, encoder_embeddings={m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods},
                             decoder_embeddings={m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}, modality_info=MODALITY_INFO)
        model.eval()
        sampler = GenerationSampler(model, use_graphs=False)
        schedule = build_chained_generation_schedules(
            cond_domains=[t["cond"]], target_domains=[t["target"]], tokens_per_target=[t["tokens"]], autoregression_schemes=["roar"],
            decoding_steps=[t["steps"]], token_decoding_schedules=["linear"], temps=[0.01], temp_schedules=["constant"],
            cfg_scales=[2.0], cfg_schedules=["constant"], cfg_grow_conditioning=True)
        f_clip = generation_flops(MODEL_CFGS[name], 5120, t["tokens"], t["steps"], 64000)
        out = {"workload": "rgb2depth generation: ROAR 3 steps, CFG 2.0, top-p 0.8, 5120 rgb -> 5120 depth tokens, ego-b 12e/12d, one hipGraph per schedule",
               "algorithmic_tflop_per_clip": round(f_clip / 1e12, 3)}
        for B, reps in ((1, repeats_b1), (8, repeats_b8)):
            ids = synth.randint("bench.cfg4", (B, 5, 32, 32), 64000, seed=0)
            sample = {t["cond"]: {"tensor": ids.to(device)}}
            sample = init_empty_target_modality(sample, MODALITY_INFO, t["target"], B, t["tokens"], device)
            sample = init_full_input_modality(sample, MODALITY_INFO, t["cond"], device)
            sampler.generate_graphed(sample, schedule, seed=0, top_p=0.8, top_k=0.0)          # capture + warm-up
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(reps):
                sampler.generate_graphed(sample, schedule, seed=0, top_p=0.8, top_k=0.0)
            torch.cuda.synchronize()
            dt = (_t.perf_counter() - t0) / reps
            if B == 1:
                out["ms_per_clip_b1"] = round(dt * 1e3, 3)
                out["frac_b1"] = round(f_clip / dt / 1e12 / PEAK_BF16_TFLOPS, 4)
            else:
                out["clips_per_s_b8"] = round(B / dt, 3)
                out["frac_b8"] = round(f_clip * B / dt / 1e12 / PEAK_BF16_TFLOPS, 4)
        del sampler, model
        return out
    finally:
        torch.set_grad_enabled(True)
        torch.cuda.empty_cache()


def extra_config5(device, clips=64, mb=32, steps=2):
    """BASELINE config 5 (ego-L ~1.2 B: D = 1152, 18 heads of 64, F = 3072, 24e / 24d; bf16 and bf16 + fp8 forward linears) as
    training steps of `clips` clips at micro-batch `mb` on fresh engines, in this process."""
    import time as _t
    from egom2p_amd import synth
    from egom2p_amd.config import MODEL_CFGS
    from egom2p_amd.engine import Engine
    from egom2p_amd.profiler import PEAK_BF16_TFLOPS
    from egom2p_amd.trainer import TrainStep
    cfg = MODEL_CFGS["ego_L_1152"]
    f_fwd = flops_per_clip(cfg, synth.CANONICAL_BUDGETS, 2048, 2048)
    out = {"workload": f"ego_L_1152 (1.19 B) training step, {clips} clips, micro-batch {mb}, canonical 10300-position clips",
           "algorithmic_tflop_per_clip": round(3 * f_fwd / 1e12, 3)}
    # "fp8": forward linears on e4m3; "fp8_bwd": forward AND dgrad GEMMs on e4m3 (weight gradients bf16) - DESIGN section 4f
    for tag, fp8, fp8b in (("bf16", False, False), ("fp8", True, False), ("fp8_bwd", True, True)):
        eng = Engine(cfg, device, max_batch=mb, n_enc=2048, n_dec=2048, fp8_forward=fp8, fp8_backward=fp8b)
        eng.init_random(seed=0)
        mbs = [synth.make_clip_batch_device(cfg, mb, synth.CANONICAL_BUDGETS, seed=100, sample_offset=i * mb, device=device) for i in range(2)]
        step = TrainStep(eng, lr=1e-4, weight_decay=0.05, clip_grad=1.0)
        n_mb = clips // mb
        run = lambda i: step([mbs[(i * n_mb + j) % 2] for j in range(n_mb)])
        run(0)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for i in range(steps):
            run(1 + i)
        torch.cuda.synchronize()
        dt = (_t.perf_counter() - t0) / steps
        out[f"clips_per_s_{tag}"] = round(n_mb * mb / dt, 3)
        out[f"frac_{tag}"] = round(3 * f_fwd * n_mb * mb / dt / 1e12 / PEAK_BF16_TFLOPS, 4)
        del step, eng, mbs
        torch.cuda.empty_cache()
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process (which has not touched the
    GPU and never will) starts the N ranks as ONE child `python -m torch.distributed.run` (one rank per GPU, RCCL), lets
    the child's stderr through, relays rank 0's single JSON line and exits with the child's code.  No exec: a process
    that has initialised the GPU must never be replaced, and the parent stays GPU-free so that it could be.
    Reference launch: torchrun (README_TRAINING.md:40-43), egom2p/utils/dist.py:78-100, DDP at run_training_egom2p.py:514."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()            # counting devices does not initialise HIP on this image
    one_dev = os.environ.get("EGOM2P_ONE_DEVICE") == "1"
    if n_dev < args.gpus and not one_dev:
        print(f"[bench] --gpus {args.gpus} but only {n_dev} GPU(s) visible; refusing to time fewer ranks than asked "
              f"(rehearsal on one device: EGOM2P_DIST_BACKEND=gloo EGOM2P_ONE_DEVICE=1)", file=sys.stderr)
        sys.exit(2)
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = p.communicate()
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.lstrip().startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(out.decode(errors="replace"))
        print(f"[bench] self-launch failed: rc {p.returncode}, {len(lines)} JSON line(s)", file=sys.stderr)
        sys.exit(p.returncode or 1)
    rec = json.loads(lines[0])
    if rec.get("ranks_seen") != args.gpus:
        print(f"[bench] ranks_seen {rec.get('ranks_seen')} != --gpus {args.gpus}", file=sys.stderr)
        sys.exit(1)
    rec["launcher"] = "self (bench.py started torch.distributed.run)"
    sys.stdout.write(json.dumps(rec) + "\n")
    sys.stdout.flush()
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="egom2p_base_12e_12d_swiglu_nobias")
    ap.add_argument("--clips-per-gpu", type=int, default=256)
    ap.add_argument("--micro-batch", type=int, default=64,
                    help="clips per forward/backward (accumulated to --clips-per-gpu); 64 measured best on MI355X: 32 -1.5 %, 128 -0.6 %")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["survey-fp32", "quick", "full", "full-fp32", "full-bf16"], default="survey-fp32",
                    help="survey-fp32 (default): SURVEY 8(d)'s fp32 legs - warm-up + median of 5 at B = 1 and B = 4 - within a budget of "
                         "$EGOM2P_CPU_BASELINE_BUDGET_S (340) CPU seconds; quick: warm-up + median of 3 (B=1 fp32) + one bf16-mode run; "
                         "full: fp32 and bf16 mode (~20 min)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--fp8", action="store_true", help="forward linears on e4m3 operands (BASELINE config 5: bf16 + fp8 MFMA GEMMs)")
    ap.add_argument("--fp8-bwd", action="store_true", help="with --fp8: the dgrad GEMMs on e4m3 operands too (weight gradients stay bf16)")
    ap.add_argument("--dp-algo", choices=["allreduce", "rs_ag"], default=None,
                    help="gradient exchange per bucket: RCCL all-reduce (ring) or in-place reduce-scatter + all-gather (default: $EGOM2P_DP_ALGO or allreduce)")
    ap.add_argument("--dp-backend", choices=["torch", "cabi"], default="torch",
                    help="torch.distributed's RCCL process group, or the C-ABI's own RCCL communicator (ego_dp_*)")
    ap.add_argument("--sparse-tables", choices=["auto", "on", "off"], default="auto",
                    help="row-list exchange of the encoder tables' gradients (auto: when it moves fewer bytes than the dense exchange)")
    ap.add_argument("--preset", choices=["yaml4"], default=None,
                    help="yaml4: the released yaml's regime - 4 clips per GPU and step, micro-batch 4 (batch_size: 4 of "
                         "ego-b_mod4_500b_clariden_2048_camcv_depthdenoise.yaml), where the exchange is NOT hidden by the backward")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-4 / config-5 measurements appended after the timed region (N = 1)")
    args = ap.parse_args()
    if args.preset == "yaml4":
        args.clips_per_gpu, args.micro_batch = 4, 4
    under_launcher = "WORLD_SIZE" in os.environ and ("TORCHELASTIC_RUN_ID" in os.environ or int(os.environ["WORLD_SIZE"]) > 1)
    if args.gpus > 1 and not under_launcher:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):     # a stray WORLD_SIZE=1 from a wrapper shell is not a launcher
            os.environ.pop(k, None)
        self_launch(args, sys.argv[1:])

    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL prints its version block at
    # communicator creation) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from egom2p_amd import synth
    from egom2p_amd.config import MODEL_CFGS
    from egom2p_amd.engine import Engine
    from egom2p_amd.profiler import PEAK_BF16_TFLOPS, PEAK_HBM_GBS, KernelTimer, kernel_source_sha
    from egom2p_amd.trainer import TrainStep

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a one-GPU box: EGOM2P_DIST_BACKEND=gloo EGOM2P_ONE_DEVICE=1 puts every rank on
    # device 0 and exchanges over gloo (RCCL refuses two ranks on one device); never set by the driver
    if os.environ.get("EGOM2P_ONE_DEVICE") == "1":
        local = 0
    if world > 1 or os.environ.get("EGOM2P_FORCE_REDUCER") == "1":
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("EGOM2P_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    if world != args.gpus:
        # a line labelled --gpus N must come from N ranks: never time fewer (or more) ranks than asked
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run", file=sys.stderr)
        sys.exit(2)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    ranks_seen, rank_devices = 1, [f"{dev} ({torch.cuda.get_device_name(local)})"]
    if dist.is_initialized():
        ranks_seen = dist.get_world_size()
        gathered = [None] * ranks_seen
        prop = torch.cuda.get_device_properties(local)
        dist.all_gather_object(gathered, f"rank{rank}:cuda:{local}:{prop.name}:{getattr(prop, 'uuid', '')}")
        rank_devices = gathered

    cfg = MODEL_CFGS[args.model]
    n_enc = n_dec = 2048
    mb = min(args.micro_batch, args.clips_per_gpu)
    n_mb = args.clips_per_gpu // mb
    clips = n_mb * mb
    eng = Engine(cfg, dev, max_batch=mb, n_enc=n_enc, n_dec=n_dec, fp8_forward=args.fp8, fp8_backward=args.fp8 and args.fp8_bwd)
    eng.init_random(seed=0)                      # same weights on every rank (DDP broadcast semantics)
    budgets = synth.CANONICAL_BUDGETS
    pool = 2                                     # distinct micro-batches per rank, cycled (inputs stay in HBM)
    # clips are synthesised on the device (ego_clip_synth: bit-identical to the host generator, no H2D copy)
    mbs = [synth.make_clip_batch_device(cfg, mb, budgets, seed=100 + rank, sample_offset=i * mb, device=dev) for i in range(pool)]
    step = TrainStep(eng, lr=args.lr, weight_decay=0.05, clip_grad=1.0, world_size=world, seed=rank,
                     force_reducer=os.environ.get("EGOM2P_FORCE_REDUCER") == "1", clips_per_step=clips,
                     sparse_tables=args.sparse_tables, dp_algo=args.dp_algo, dp_backend=args.dp_backend)

    def run_step(i):
        return step([mbs[(i * n_mb + j) % pool] for j in range(n_mb)])

    for i in range(args.warmup):
        run_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses, gnorm = run_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(losses[0].item())

    # the timed steps above ran the exchange exactly as TrainStep runs it (no per-bucket event pairs, no extra
    # waits); exchange_ms / exposed_ms / overlap_frac come from ONE more, instrumented step outside the timed region
    if step.reducer is not None:
        step.reducer.timing = True
        run_step(args.warmup + args.steps)
        torch.cuda.synchronize()
        step.reducer.timing = False
    par, dp_extra = dp_report(step.reducer, world, step.sparse)       # that instrumented step's exchange (event-timed on the comm stream)
    if dp_extra:
        dp_extra["exchange_timing"] = "one instrumented step after the timed region (the timed steps run uninstrumented)"
    ms_per_step = dt / args.steps * 1e3
    tokens = world * clips * 10300 * args.steps
    value = tokens / dt
    f_fwd = flops_per_clip(cfg, budgets, n_enc, n_dec)
    f_step_per_gpu = 3.0 * f_fwd * clips                     # fwd + bwd (2x) algorithmic
    e2e_tflops = f_step_per_gpu / (dt / args.steps) / 1e12

    out = {
        "metric": ("multimodal tokens/sec (ego-b 400M, 10300-tok clips), training fwd+bwd+allreduce+AdamW"
                   if args.model == "egom2p_base_12e_12d_swiglu_nobias" else
                   f"multimodal tokens/sec ({args.model}, 10300-tok clips), training fwd+bwd+allreduce+AdamW"),
        "value": value, "unit": "clip-positions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16+fp8fwd+fp8dgrad" if args.fp8_bwd else "bf16+fp8fwd") if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} mod4, synthetic 10300-position clips (1009+1009 rgb, 1009+1009 depth, 15+15 cam, "
                               f"15+15 gaze kept -> N=M=2048), bf16 MFMA GEMM/attention, fp32 residual/LN/CE/AdamW",
                   "clips_per_gpu_per_step": clips, "micro_batch": mb, "global_batch": clips * world, **par},
        "ranks_seen": ranks_seen, "rank_devices": rank_devices,
        "clips_per_s": value / 10300.0, "final_loss": final_loss,
        "algorithmic_tflops_per_gpu": e2e_tflops, "mfma_frac_end_to_end": e2e_tflops / PEAK_BF16_TFLOPS,
    }
    out.update(dp_extra)

    if rank == 0 and not args.no_kernel_profile:
        # live HIP-event timing of one micro-batch forward+backward, per kernel class
        kt = KernelTimer()
        pair, eng.pair_stream = eng.pair_stream, None      # per-kernel times: every launch alone (the timed steps above pair wgrad GEMMs with LayerNorm backward launches)
        with kt.capture(cfg.num_heads):
            eng.forward(mbs[0], dec_order=[m.name for m in cfg.mods], loss_grad=1.0)
            eng.backward(1.0)
            if world == 1:
                # the optimiser side of a step (clip norm, AdamW with zero_grad folded in, the bf16 weight copies): HBM-bound too
                step.opt.step(clip_grad=1.0, zero_grad=True)
                eng.refresh_weights()
        eng.pair_stream = pair
        eng.zero_grad()
        summ = kt.summary()
        mfma = {k: v for k, v in summ.items() if v["flops"] > 0}
        dom = max(mfma.items(), key=lambda kv: kv[1]["ms"])
        out["roofline"] = {"bound": "mfma", "kernel": dom[0], "achieved": dom[1]["tflops"], "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": dom[1]["tflops"] / PEAK_BF16_TFLOPS, "traffic": None,
                           "launches": dom[1]["calls"], "avg_launch_ms": dom[1]["ms"] / dom[1]["calls"]}
        # HBM-side bytes per launch of that kernel: PMC counters need a rocprofv3 wrapper around the process, so they are
        # collected by the command recorded in profiles/pmc_latest.json (same workload and micro-batch) and read here -
        # only while that file was measured on THESE kernels (sha of the kernel sources), else traffic stays null
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            ent = pmc.get("abi", {}).get(dom[0])
            same = pmc.get("kernel_src_sha") == kernel_source_sha()
            if ent and same and pmc.get("micro_batch") == mb and args.model == "egom2p_base_12e_12d_swiglu_nobias":
                out["roofline"]["traffic"] = ent["traffic_bytes_per_launch"]
                out["roofline"]["traffic_unit"] = "bytes/launch (rocprofv3 PMC, profiles/pmc_latest.json)"
            elif ent and same:
                out["roofline"]["traffic_note"] = f"profiles/pmc_latest.json holds micro-batch {pmc.get('micro_batch')}, this run {mb}"
            elif ent and not same:
                out["roofline"]["traffic_note"] = ("profiles/pmc_latest.json was measured on other kernel sources "
                                                   f"({pmc.get('kernel_src_sha')} vs {kernel_source_sha()}): re-run tools/pmc_run.sh")
        out["roofline"]["algorithmic_bytes_per_launch"] = dom[1]["bytes"] / dom[1]["calls"]
        # north_star's HBM-bound paths (embedding / masking / scatter / normalisation / loss): algorithmic GB/s vs the HBM roof
        # `gbs` = ALGORITHMIC bytes / HIP-event time (what the path has to move); `traffic` = bytes per call that left the L2s by the
        # PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, profiles/pmc_latest.json, same kernel sources and micro-batch only) and
        # `frac` = traffic / time / 8 TB/s.  Requests served by the 256 MB Infinity Cache are counted by FETCH_SIZE
        # (MI355X_MICROARCH.md, HBM): a frac above ~0.79 (6.3 TB/s is what HBM delivers) says part of the bytes never reached HBM.
        hbm = {k: v for k, v in summ.items() if v["flops"] == 0 and v["bytes"] > 0}
        pmc_abi = {}
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if (pmc.get("kernel_src_sha") == kernel_source_sha() and pmc.get("micro_batch") == mb
                    and args.model == "egom2p_base_12e_12d_swiglu_nobias"):
                pmc_abi = pmc.get("abi", {})
        out["hbm_paths"] = {}
        for k, v in sorted(hbm.items(), key=lambda kv: -kv[1]["ms"]):
            us = 1e3 * v["ms"] / v["calls"]
            ent = {"gbs": round(v["gbs"], 1), "algorithmic_bytes_per_call": round(v["bytes"] / v["calls"]), "us_per_call": round(us, 1),
                   "frac_algorithmic_of_8TBs": round(v["gbs"] / PEAK_HBM_GBS, 3), "traffic": None, "frac": None}
            pe = pmc_abi.get(k)
            if pe is not None:
                tb = pe["traffic_bytes_per_launch"]
                ent["traffic"] = round(tb)
                ent["frac"] = round(tb / (us * 1e-6) / (PEAK_HBM_GBS * 1e9), 3)
                if ent["frac"] > 0.79:
                    ent["note"] = "above the 6.3 TB/s HBM delivers: part of these L2-side bytes were served by the Infinity Cache"
            out["hbm_paths"][k] = ent
        out["hbm_paths_note"] = ("gbs / frac_algorithmic_of_8TBs: algorithmic bytes over HIP-event time; traffic: PMC bytes per call "
                                 "(rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes, profiles/pmc_latest.json; null when that file "
                                 "is of other kernel sources / micro-batch); frac = traffic / time / 8 TB/s")
        out["kernel_breakdown"] = {k: {"ms": round(v["ms"], 3), "calls": v["calls"], "tflops": round(v["tflops"], 1),
                                        "gbs": round(v["gbs"], 1)} for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.detach().float().cpu().clone() for k, v in eng.state_dict().items()}
    if rank == 0 and world == 1 and not args.no_extras and args.model == "egom2p_base_12e_12d_swiglu_nobias" and not args.fp8:
        # Driver-witnessed numbers for BASELINE configs 4 and 5 (the headline fields above stay config 2): measured AFTER the
        # timed region, in this process, on fresh models - the ego-b training engine's 110 GB are released first
        del step, mbs
        eng_keep_cfg = eng.cfg
        del eng
        torch.cuda.empty_cache()
        out["extra"] = {}
        for key, fn in (("config4", extra_config4), ("config5", extra_config5)):
            try:
                out["extra"][key] = fn(dev)
            except Exception as e:                                   # an extra must never cost the headline line
                out["extra"][key] = {"error": f"{type(e).__name__}: {e}"}
        eng = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, synth, budgets, rank, [m.name for m in cfg.mods], n_enc, n_dec, args.cpu_baseline)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
