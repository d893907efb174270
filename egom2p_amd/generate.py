"""ROAR + classifier-free-guidance generation over the HIP engine (BASELINE config 4).

Mirrors the part of `egom2p/models/generate.py` the reference's eval scripts use
(`eval_model_rgb2depth.py:43-96`): `build_chained_generation_schedules` (:197-322),
`init_empty_target_modality` (:83-115), `init_full_input_modality` (:117-151), `empty_img_modality`
(:30-37) and `GenerationSampler.generate` (:1031-1099) for token modalities with scheme 'roar', with or
without guidance (`roar_step_batched` :768-783, `guided_roar_step_batched` :785-817).  MaskGIT,
autoregressive text and multi-guided generation are outside the hot-path scope and raise.

Per schedule step: two encoder-decoder passes on the engine (conditional / unconditional; the
unconditional one sees an empty context on the first step), then ONE HIP kernel per step does the CFG
mix, nucleus filtering, temperature softmax and sampling for all rows (`ego_sample_cfg_topp`).  The
random ROAR order and the sampling uniforms come from torch's generator seeded with `seed + step`
like the reference (:1050, :484-485); they are explicit inputs of the kernels, so a test can pin them.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .engine import warmup_stream


# ---------------------------------------------------------------------------------------------- schedules
def linear_schedule(num_steps: int, total_tokens: int) -> np.ndarray:
    """egom2p/utils/generation.py: tokens per step, descending, zeros trimmed."""
    edges = np.linspace(0, total_tokens, num_steps + 1, dtype=int)
    steps = np.sort(np.diff(edges))[::-1]
    return np.trim_zeros(steps, "b")


def cosine_schedule(num_steps: int, total_tokens: int) -> np.ndarray:
    s = np.array([0.5 * (1 + math.cos(math.pi * i / num_steps)) for i in range(num_steps)])
    toks = [round(total_tokens * d) for d in (s[:-1] - s[1:])]
    toks.append(total_tokens - sum(toks))
    return np.array(toks)


def build_chained_generation_schedules(cond_domains: List[str], target_domains: List[str], tokens_per_target: List[int],
                                       autoregression_schemes: List[str], decoding_steps: List[int],
                                       token_decoding_schedules: List[str], temps: List[float], temp_schedules: List[str],
                                       cfg_scales: List[float], cfg_schedules: List[str], cfg_grow_conditioning: bool = False,
                                       modality_info: Optional[dict] = None) -> List[dict]:
    out, cond = [], list(cond_domains)
    for i, target in enumerate(target_domains):
        scheme, ntoks = autoregression_schemes[i], tokens_per_target[i]
        if scheme == "roar":
            sched = linear_schedule(decoding_steps[i], ntoks)
        elif scheme == "maskgit":
            sched = cosine_schedule(decoding_steps[i], ntoks) if token_decoding_schedules[i] == "cosine" else linear_schedule(decoding_steps[i], ntoks)
        else:
            raise NotImplementedError(f"scheme {scheme} is outside the hot-path scope")
        n = len(sched)
        if temp_schedules[i] == "constant":
            t_s = temps[i] * np.ones(decoding_steps[i])
        elif temp_schedules[i] == "linear":
            t_s = np.concatenate([[temps[i]], (temps[i] * (sched.sum() - sched.cumsum()) / sched.sum())[:-1]]).clip(min=1e-9)
        else:
            raise NotImplementedError(f"temperature schedule {temp_schedules[i]}")
        if cfg_schedules[i] != "constant":
            raise NotImplementedError(f"guidance schedule {cfg_schedules[i]}")
        c_s = cfg_scales[i] * np.ones(decoding_steps[i])
        for tok, t, c in zip(sched, t_s, c_s):
            out.append({"target_domain": target, "scheme": scheme, "num_tokens": int(tok), "temperature": float(t),
                        "cfg_scale": float(c), "cfg_cond_domains": list(cond)})
        if cfg_grow_conditioning:
            cond.append(target)
    return out


# ---------------------------------------------------------------------------------------------- mod_dict helpers
def empty_img_modality(mod_dict, key):
    mod_dict[key]["input_mask"][:] = True
    mod_dict[key]["target_mask"][:] = False
    return mod_dict


def init_empty_target_modality(mod_dict, modality_info, domain, batch_size, num_tokens, device):
    if modality_info[domain]["type"] not in ("img", "gaze", "cam", "keypoints"):
        raise NotImplementedError("sequence modalities are outside the hot-path scope")
    mod_dict[domain] = {"tensor": torch.zeros((batch_size, num_tokens), dtype=torch.int64, device=device),
                        "input_mask": torch.ones((batch_size, num_tokens), dtype=torch.bool, device=device),
                        "target_mask": torch.zeros((batch_size, num_tokens), dtype=torch.bool, device=device)}
    return empty_img_modality(mod_dict, domain)


def init_full_input_modality(mod_dict, modality_info, domain, device, eos_id=3):
    if modality_info[domain]["type"] not in ("img", "gaze", "cam", "keypoints"):
        raise NotImplementedError("sequence modalities are outside the hot-path scope")
    d = mod_dict[domain]
    shape = (d["tensor"].shape[0], int(np.prod(d["tensor"].shape[1:])))
    d.setdefault("input_mask", torch.zeros(shape, dtype=torch.bool, device=device))
    d.setdefault("target_mask", torch.ones(shape, dtype=torch.bool, device=device))
    d.setdefault("decoder_attention_mask", torch.zeros(shape, dtype=torch.bool, device=device))
    d["input_mask"][:] = False
    d["target_mask"][:] = True
    return mod_dict


# ---------------------------------------------------------------------------------------------- sampler
class GenerationSampler:
    def __init__(self, model, use_graphs: bool = False):
        self.model = model
        self.engine = model.engine if hasattr(model, "engine") else model
        self.use_graphs = use_graphs           # replay each encoder-decoder pass from a captured hipGraph

    # one encoder-decoder pass (forward_enc_dec_roar_batched, generate.py:747-766) ---------------------
    def _enc_inputs(self, mod_dict, n_enc: Optional[int] = None):
        eng = self.engine
        enc = {}
        for m in eng.mods:
            if m.name not in mod_dict:
                continue
            d = mod_dict[m.name]
            B = d["tensor"].shape[0]
            ids = d["tensor"].reshape(B, -1).to(eng.dev, torch.int64)
            mask = d["input_mask"].reshape(B, -1).to(eng.dev, torch.bool)
            enc[m.name] = (ids, mask)
        # rows kept by forward_mask_encoder_generation = max unmasked count over the batch (:413-415)
        if n_enc is None:
            n_enc = int(torch.stack([(~v[1]).sum(1) for v in enc.values()]).sum(0).max().item()) if enc else 0
        return enc, n_enc

    def _logits(self, mod_dict, target_mod, mod_pos, n_enc: Optional[int] = None, ws: Optional[dict] = None):
        """n_enc given: no host sync (the caller knows the kept-row count, e.g. from the schedule); ws: workspace to use."""
        eng = self.engine
        enc, n_enc = self._enc_inputs(mod_dict, n_enc)
        if ws is not None:
            return eng.infer_logits(enc, n_enc, target_mod, mod_pos, ws=ws)
        if self.use_graphs:
            return eng.infer_logits_graphed(enc, n_enc, target_mod, mod_pos).clone()
        return eng.infer_logits(enc, n_enc, target_mod, mod_pos)

    def roar_order(self, target_mask: torch.Tensor, num_select: int, seed: Optional[int], noise: Optional[torch.Tensor] = None,
                   n_dec: Optional[int] = None) -> torch.Tensor:
        """Positions decoded in this step (forward_mask_decoder_roar, :481-516): unmasked targets in a random
        order shared by the batch, the first `num_select` of them.  `noise` ([T] uniforms) / `n_dec` given: nothing
        is drawn or read back here (graph capture)."""
        dev = target_mask.device
        if noise is None:
            if seed is not None:
                torch.manual_seed(seed)
            noise = torch.rand(target_mask.shape[1], device=dev)
        if n_dec is None:
            n_dec = min(int(num_select), int((~target_mask[0]).sum().item()))
        ids_shuffle = torch.argsort(target_mask.float() + noise.unsqueeze(0) * 1e-6, dim=1)
        return ids_shuffle[:, :n_dec]

    def roar_step(self, mod_dict, target_mod, num_select, temperature, top_k, top_p, conditioning=(), guidance_scale=1.0,
                  seed=None, mod_pos: Optional[torch.Tensor] = None, uniforms: Optional[torch.Tensor] = None,
                  return_logits: bool = False, forced_samples: Optional[torch.Tensor] = None,
                  static: Optional[dict] = None):
        """static (graph capture): {"noise", "n_dec", "n_enc_cond", "n_enc_uncond", "ws"} - every count comes from the
        schedule, every random number from a buffer filled before the replay: no host sync, no generator call."""
        eng = self.engine
        st = static or {}
        d = mod_dict[target_mod]
        if mod_pos is None:
            mod_pos = self.roar_order(d["target_mask"].to(eng.dev), num_select, seed, noise=st.get("noise"), n_dec=st.get("n_dec"))
        mod_pos = mod_pos.to(eng.dev)
        guided = guidance_scale != 1.0 and len(conditioning) > 0
        uncond = None
        if guided:
            uncond = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in mod_dict.items()}
            for mod in conditioning:
                uncond = empty_img_modality(uncond, mod)
        ws = st.get("ws")
        pair = guided and getattr(eng, "cfg_pair", False) and (ws is None or ws.get("groups", 1) >= 2)
        logits_uncond = None
        if pair:
            # the two passes of a guided step decode the same rows: one decoder pass over both contexts (engine.infer_logits_cfg)
            enc_c, n_c = self._enc_inputs(mod_dict, st.get("n_enc_cond"))
            enc_u, n_u = self._enc_inputs(uncond, st.get("n_enc_uncond"))
            pair = n_c > 0
        if pair and ws is None and self.use_graphs:
            logits_cond, logits_uncond = (t.clone() for t in eng.infer_logits_cfg_graphed(enc_c, n_c, enc_u, n_u, target_mod, mod_pos))
        elif pair:
            logits_cond, logits_uncond = eng.infer_logits_cfg(enc_c, n_c, enc_u, n_u, target_mod, mod_pos, ws=ws)
        else:
            logits_cond = self._logits(mod_dict, target_mod, mod_pos, n_enc=st.get("n_enc_cond"), ws=ws)
            if static is not None:
                logits_cond = logits_cond.clone()
            if guided:
                logits_uncond = self._logits(uncond, target_mod, mod_pos, n_enc=st.get("n_enc_uncond"), ws=ws)
        B, M, V = logits_cond.shape
        if uniforms is None:
            uniforms = torch.rand(B * M, device=eng.dev)
        samples = torch.empty(B * M, device=eng.dev, dtype=torch.int32)
        ops.sample_cfg_topp(logits_cond.view(B * M, V), None if logits_uncond is None else logits_uncond.view(B * M, V), V,
                            float(guidance_scale), float(top_p), float(temperature), uniforms, samples, ld=V,
                            top_k=ops.top_k_count(top_k, V))          # int: a count, float: a share of V (generate.py:335-342)
        samples = samples.view(B, M).to(torch.int64)
        drawn = samples
        if forced_samples is not None:             # teacher forcing for parity tests: scatter the given tokens instead
            samples = forced_samples.to(eng.dev, torch.int64).view(B, M)
        d["tensor"] = torch.scatter(d["tensor"].reshape(B, -1).to(eng.dev), -1, mod_pos, samples)
        d["input_mask"] = torch.scatter(d["input_mask"].to(eng.dev), -1, mod_pos, torch.zeros_like(samples, dtype=torch.bool))
        d["target_mask"] = torch.scatter(d["target_mask"].to(eng.dev), -1, mod_pos, torch.ones_like(samples, dtype=torch.bool))
        if return_logits:
            return mod_dict, dict(logits_cond=logits_cond, logits_uncond=logits_uncond, mod_pos=mod_pos, samples=drawn)
        return mod_dict

    @torch.no_grad()
    def generate(self, mod_dict, schedule, top_k=0.0, top_p=0.0, text_tokenizer=None, verbose=False, seed=None):
        mod_dict = copy.deepcopy(mod_dict)
        for step, info in enumerate(schedule):
            target = info["target_domain"]
            if info["scheme"].lower() != "roar":
                raise NotImplementedError(f"scheme {info['scheme']} is outside the hot-path scope")
            mod_dict = self.roar_step(mod_dict, target, info["num_tokens"], info["temperature"], top_k, top_p,
                                      conditioning=info.get("cfg_cond_domains", []), guidance_scale=info.get("cfg_scale", 1.0),
                                      seed=None if seed is None else seed + step)
        return mod_dict

    # ---- the whole schedule as ONE hipGraph (BASELINE config 4: "hipGraph-captured decode") ---------------------------
    @torch.no_grad()
    def generate_graphed(self, mod_dict, schedule, top_k=0.0, top_p=0.0, seed: int = 0):
        """`generate` with every encoder-decoder pass, the device sampler and the token scatters of ALL schedule steps
        replayed from one captured graph: one host call per clip batch, one host read (the initial mask counts) per call.

        What makes that possible: with ROAR the kept-row counts of every pass follow from the schedule and the initial
        number of unmasked inputs (conditional pass of step s: inputs + tokens decoded so far; unconditional: tokens
        decoded so far, 0 on the first step), and the random ROAR order / sampling uniforms are drawn BEFORE the replay
        with torch's generator seeded `seed + step` exactly as the eager path does - same tokens, bit for bit.
        One graph per (batch, schedule, initial counts, top_p), kept in the engine's bounded graph cache."""
        eng = self.engine
        names = [m.name for m in eng.mods if m.name in mod_dict]
        B = mod_dict[names[0]]["tensor"].shape[0]
        flat = {n: {"tensor": mod_dict[n]["tensor"].reshape(B, -1).to(eng.dev, torch.int64),
                    "input_mask": mod_dict[n]["input_mask"].reshape(B, -1).to(eng.dev, torch.bool),
                    "target_mask": mod_dict[n]["target_mask"].reshape(B, -1).to(eng.dev, torch.bool)} for n in names}
        # the one host read: unmasked inputs per sample and modality, open targets per modality (of sample 0, like :498-500)
        cnt = torch.stack([torch.cat([(~flat[n]["input_mask"]).sum(1), (~flat[n]["target_mask"][0]).sum().reshape(1)]) for n in names]).cpu().numpy()
        n_in = {n: cnt[i, :B].astype(np.int64) for i, n in enumerate(names)}            # [B] each
        n_open = {n: int(cnt[i, B]) for i, n in enumerate(names)}
        plan, dec_so_far = [], {n: 0 for n in names}
        for info in schedule:
            if info["scheme"].lower() != "roar":
                raise NotImplementedError(f"scheme {info['scheme']} is outside the hot-path scope")
            t = info["target_domain"]
            n_dec = min(int(info["num_tokens"]), n_open[t] - dec_so_far[t])
            cond = list(info.get("cfg_cond_domains", []))
            guided = info.get("cfg_scale", 1.0) != 1.0 and len(cond) > 0
            # rows kept by a pass = max over the batch of the unmasked inputs of all its modalities (:413-415)
            total = int(sum(n_in[n] + dec_so_far[n] for n in names).max())
            unc = [n for n in names if n not in cond]
            total_u = int(sum(n_in[n] + dec_so_far[n] for n in unc).max()) if unc else 0
            plan.append(dict(target=t, n_dec=n_dec, n_enc_cond=total, guided=guided, n_enc_uncond=total_u, info=info))
            dec_so_far[t] += n_dec
        key = ("generate", bool(eng.cfg_pair), B, tuple(names), tuple((n, tuple(int(x) for x in n_in[n])) for n in names), tuple(sorted(n_open.items())), float(top_p), (type(top_k).__name__, float(top_k or 0)),
               tuple((p["target"], p["n_dec"], p["info"]["temperature"], p["info"].get("cfg_scale", 1.0),
                      tuple(p["info"].get("cfg_cond_domains", []))) for p in plan))
        graphs = eng.__dict__.setdefault("_graphs", {})
        g = graphs.pop(key, None)
        if g is None:
            while len(graphs) >= eng.max_graphs:
                graphs.pop(next(iter(graphs)))
            if eng.weights_dirty:
                eng.refresh_weights()
            n_max = max(max(p["n_enc_cond"] for p in plan), 1) + getattr(eng, "R", 0)
            m_max = max(p["n_dec"] for p in plan)
            st = {"in": {n: {k: v.clone() for k, v in flat[n].items()} for n in names},
                  "noise": [torch.zeros(flat[p["target"]]["target_mask"].shape[1], device=eng.dev) for p in plan],
                  "uni": [torch.zeros(B * p["n_dec"], device=eng.dev) for p in plan],
                  "ws": eng._alloc_infer(B, n_max, m_max, fresh=True,
                                         groups=2 if (eng.cfg_pair and any(p["guided"] for p in plan)) else 1), "out": None}

            def run():
                md = {n: {k: v.clone() for k, v in st["in"][n].items()} for n in names}
                for i, p in enumerate(plan):
                    info = p["info"]
                    md = self.roar_step(md, p["target"], p["n_dec"], info["temperature"], top_k, top_p,
                                        conditioning=info.get("cfg_cond_domains", []), guidance_scale=info.get("cfg_scale", 1.0),
                                        uniforms=st["uni"][i],
                                        static=dict(noise=st["noise"][i], n_dec=p["n_dec"], n_enc_cond=p["n_enc_cond"],
                                                    n_enc_uncond=p["n_enc_uncond"], ws=st["ws"]))
                return md

            side = warmup_stream(eng.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up outside the capture (lazy initialisations)
                run()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["out"] = run()
            g = (graph, st)
        graphs[key] = g
        graph, st = g
        for n in names:
            for k in ("tensor", "input_mask", "target_mask"):
                st["in"][n][k].copy_(flat[n][k])
        for i, p in enumerate(plan):                            # the eager path's draws, in its order (roar_order, then roar_step)
            torch.manual_seed(seed + i)
            st["noise"][i].copy_(torch.rand(st["noise"][i].shape[0], device=eng.dev))
            st["uni"][i].copy_(torch.rand(st["uni"][i].shape[0], device=eng.dev))
        graph.replay()
        out = copy.deepcopy({n: dict(mod_dict[n]) for n in mod_dict if n not in names})
        for n in names:
            out[n] = {k: v.clone() for k, v in st["out"][n].items()}
            for k, v in mod_dict[n].items():
                out[n].setdefault(k, v)
        return out
