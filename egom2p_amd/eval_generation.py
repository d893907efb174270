"""The reference's four generation scripts on MI355X - `eval_model_rgb2depth.py` (:46-96), `eval_model_rgb2cam.py` (:40-95),
`eval_model_rgb2gaze.py` (:41-96), `eval_model_depth2rgb.py` (:34-91): one conditioning modality, one target modality, ROAR
with a linear token schedule, temperature 0.01, CFG 2.0 over the growing conditioning, top-p 0.8, one clip batch per call -
on the HIP engine.  The Cosmos video tokenizer (external TorchScript blobs) and the decoding / plotting of the predicted
tokens are outside the hot-path scope: the conditioning clip is given as Cosmos token ids (an .npz with a (5,32,32) int
array, e.g. the reference's example_data/rgb2cam_egoexo.npz), or synthetic ids, and the predicted token ids are written to
an .npz.  The root-level `eval_model_<task>.py` scripts call `main(<task>)`.
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from . import synth
from .generate import (GenerationSampler, build_chained_generation_schedules, init_empty_target_modality,
                       init_full_input_modality)
from .model import MODALITY_INFO, create_model

# per script: conditioning / target modality, tokens to generate, decoding steps (the reference's constants)
TASKS = {
    "rgb2depth": dict(cond="tok_rgb", target="tok_depth", tokens=5120, steps=3),      # eval_model_rgb2depth.py:46-60
    "rgb2cam": dict(cond="tok_rgb", target="tok_cam", tokens=30, steps=3),            # eval_model_rgb2cam.py:40-55
    "rgb2gaze": dict(cond="tok_rgb", target="tok_gaze", tokens=30, steps=5),          # eval_model_rgb2gaze.py:41-56
    "depth2rgb": dict(cond="tok_depth", target="tok_rgb", tokens=5120, steps=6),      # eval_model_depth2rgb.py:34-49
}


def main(task: str = "rgb2depth"):
    t = TASKS[task]
    ap = argparse.ArgumentParser(description=f"{task} generation (ROAR {t['steps']} steps, CFG 2.0, top-p 0.8)")
    ap.add_argument("--model", default="egom2p_base_12e_12d_swiglu_nobias")
    ap.add_argument("--ckpt", default="", help="reference-format checkpoint ({'model': state_dict}); random init if empty")
    ap.add_argument("--tokens", default="", help=".npz with (5,32,32) Cosmos ids of the conditioning clip; synthetic if empty")
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--bench", type=int, default=0, help="time this many generate() calls after one warm-up")
    ap.add_argument("--no-graphs", action="store_true", help="launch kernels one by one instead of replaying hipGraphs")
    ap.add_argument("--graph", choices=["schedule", "pass"], default="schedule",
                    help="schedule: all passes + samplers + scatters of a clip batch are ONE captured graph; pass: one graph per pass")
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

    cond_domains, target_domains, tokens_per_target = [t["cond"]], [t["target"]], [t["tokens"]]
    schedule = build_chained_generation_schedules(
        cond_domains=cond_domains, target_domains=target_domains, tokens_per_target=tokens_per_target,
        autoregression_schemes=["roar"], decoding_steps=[t["steps"]], token_decoding_schedules=["linear"], temps=[0.01],
        temp_schedules=["constant"], cfg_scales=[2.0], cfg_schedules=["constant"], cfg_grow_conditioning=True)
    top_p, top_k = 0.8, 0.0

    def generate(sample):
        if whole:
            return sampler.generate_graphed(sample, schedule, seed=0, top_p=top_p, top_k=top_k)
        return sampler.generate(sample, schedule, verbose=False, seed=0, top_p=top_p, top_k=top_k)

    if args.tokens:
        z = np.load(args.tokens, allow_pickle=False)
        ids = torch.from_numpy(np.asarray(z[z.files[0]]).astype(np.int64)).reshape(1, 5, 32, 32).repeat(args.batch, 1, 1, 1)
    else:
        ids = synth.randint(f"eval.{t['cond']}", (args.batch, 5, 32, 32), 64000, seed=0)
    sample = {t["cond"]: {"tensor": ids.to(device)}}
    for tg, n in zip(target_domains, tokens_per_target):
        sample = init_empty_target_modality(sample, MODALITY_INFO, tg, args.batch, n, device)
    for c in cond_domains:
        sample = init_full_input_modality(sample, MODALITY_INFO, c, device)

    out = generate(sample)
    torch.cuda.synchronize()
    if args.out:
        np.savez_compressed(args.out, **{t["target"]: out[t["target"]]["tensor"].cpu().numpy().astype(np.int32)})
    if args.bench > 0:
        t0 = time.perf_counter()
        for _ in range(args.bench):
            generate(sample)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.bench
        passes = 2 * t["steps"]                                 # conditional + unconditional encoder/decoder pass per step
        print(json.dumps({"metric": f"{task} generation (ROAR {t['steps']} steps, CFG 2.0, top-p 0.8)", "model": args.model,
                          "graph": "none" if args.no_graphs else args.graph, "batch": args.batch, "s_per_clip": dt / args.batch,
                          "clips_per_s": args.batch / dt, "passes_per_clip": passes, "ms_per_pass": dt / passes * 1e3}))
    print(f"done: {t['target']} tokens", tuple(out[t["target"]]["tensor"].shape))
