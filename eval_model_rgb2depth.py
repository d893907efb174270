#!/usr/bin/env python
"""rgb -> depth generation on MI355X: same flow as the reference's `eval_model_rgb2depth.py`
(:46-96: ROAR, 3 steps, linear token schedule, temperature 0.01, CFG 2.0, top-p 0.8, one clip per call),
on the HIP engine.  The Cosmos video tokenizer (external TorchScript blobs) and the depth decoding / plotting
are outside the hot-path scope: the conditioning clip is given as Cosmos token ids (an .npz with a (5,32,32)
int array, e.g. the reference's example_data/rgb2cam_egoexo.npz), or synthetic ids, and the predicted depth
token ids are written to an .npz.

    python eval_model_rgb2depth.py [--ckpt checkpoint-main.pth] [--tokens clip.npz] [--out depth_tokens.npz] [--bench 5]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from egom2p_amd import synth  # noqa: E402
from egom2p_amd.generate import (GenerationSampler, build_chained_generation_schedules, init_empty_target_modality,  # noqa: E402
                                 init_full_input_modality)
from egom2p_amd.model import MODALITY_INFO, create_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="egom2p_base_12e_12d_swiglu_nobias")
    ap.add_argument("--ckpt", default="", help="reference-format checkpoint ({'model': state_dict}); random init if empty")
    ap.add_argument("--tokens", default="", help=".npz with (5,32,32) Cosmos ids of the rgb clip; synthetic if empty")
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--bench", type=int, default=0, help="time this many generate() calls after one warm-up")
    ap.add_argument("--no-graphs", action="store_true", help="launch kernels one by one instead of replaying hipGraphs")
    ap.add_argument("--graph", choices=["schedule", "pass"], default="schedule",
                    help="schedule: all 6 passes + samplers + scatters of a clip batch are ONE captured graph; pass: one graph per pass")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    device = "cuda"
    mods = ["tok_rgb", "tok_depth", "tok_cam", "tok_gaze"]
    model = create_model(args.model, encoder_embeddings={m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods},
                         decoder_embeddings={m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}, modality_info=MODALITY_INFO)
    if args.ckpt:
        # weights_only=True: nothing from the file is executed
        model.load_state_dict(torch.load(args.ckpt, map_location="cpu", weights_only=True)["model"])
    model.eval()
    sampler = GenerationSampler(model, use_graphs=(not args.no_graphs) and args.graph == "pass")
    whole = (not args.no_graphs) and args.graph == "schedule"

    def generate(sample):
        if whole:
            return sampler.generate_graphed(sample, schedule, seed=0, top_p=top_p, top_k=top_k)
        return sampler.generate(sample, schedule, verbose=False, seed=0, top_p=top_p, top_k=top_k)

    cond_domains, target_domains, tokens_per_target = ["tok_rgb"], ["tok_depth"], [5120]
    schedule = build_chained_generation_schedules(
        cond_domains=cond_domains, target_domains=target_domains, tokens_per_target=tokens_per_target,
        autoregression_schemes=["roar"], decoding_steps=[3], token_decoding_schedules=["linear"], temps=[0.01],
        temp_schedules=["constant"], cfg_scales=[2.0], cfg_schedules=["constant"], cfg_grow_conditioning=True)
    top_p, top_k = 0.8, 0.0

    if args.tokens:
        z = np.load(args.tokens, allow_pickle=False)
        ids = torch.from_numpy(np.asarray(z[z.files[0]]).astype(np.int64)).reshape(1, 5, 32, 32).repeat(args.batch, 1, 1, 1)
    else:
        ids = synth.randint("eval.rgb", (args.batch, 5, 32, 32), 64000, seed=0)
    sample = {"tok_rgb": {"tensor": ids.to(device)}}
    for t, n in zip(target_domains, tokens_per_target):
        sample = init_empty_target_modality(sample, MODALITY_INFO, t, args.batch, n, device)
    for c in cond_domains:
        sample = init_full_input_modality(sample, MODALITY_INFO, c, device)

    out = generate(sample)
    torch.cuda.synchronize()
    if args.out:
        np.savez_compressed(args.out, tok_depth=out["tok_depth"]["tensor"].cpu().numpy().astype(np.int32))
    if args.bench > 0:
        t0 = time.perf_counter()
        for _ in range(args.bench):
            generate(sample)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.bench
        # forward FLOPs of the 6 passes: dense attention, N = 5120/6827/8534 (cond), 0/1707/3414 (uncond)
        print(json.dumps({"metric": "rgb2depth generation (ROAR 3 steps, CFG 2.0, top-p 0.8)", "model": args.model,
                          "graph": "none" if args.no_graphs else args.graph, "batch": args.batch, "s_per_clip": dt / args.batch, "clips_per_s": args.batch / dt,
                          "passes_per_clip": 6, "ms_per_pass": dt / 6 * 1e3}))
    print("done: depth tokens", tuple(out["tok_depth"]["tensor"].shape))


if __name__ == "__main__":
    main()
