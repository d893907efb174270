"""ORACLE tooling - golden vectors for the top-k (+ top-p) token filter of the generation path from the REAL reference
`GenerationSampler.top_k_top_p_filtering` (egom2p/models/generate.py:332-359), run in the build container only.

    python oracle/make_goldens_topk.py        ->  tests/golden/topk_filter.npz

The reference's files are loaded by path (see make_goldens.py).  Logits come from the counter-based generator (bf16-representable
conditional / unconditional rows, mixed as guided_roar_step_batched does: uncond + (cond - uncond) * 2.0, :805), so the GPU box
regenerates them bit-identically; the fixture holds, per case and row, the ids of the tokens the reference's filter keeps.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from egom2p_amd import synth                      # noqa: E402
import make_goldens as MG                         # noqa: E402

# name: (V, top_k (int: a count, float: a share of V), top_p, logit scale)
CASES = {
    "k50": (64000, 50, 0.0, 1.0),
    "kshare": (64000, 0.001, 0.0, 1.0),          # float: int(0.001 * 64000) = 64 tokens
    "k7_v256": (256, 7, 0.0, 1.0),
    "k200_p06": (64000, 200, 0.6, 3.0),          # top-k, then the nucleus of the renormalised survivors
    "kall_v256": (256, 300, 0.0, 1.0),           # k >= V: nothing removed
    "k1": (64000, 1, 0.0, 1.0),
}
ROWS, CFG = 4, 2.0


def case_logits(name, V, scale):
    """(cond, uncond) bf16 [ROWS, V] and the fp32 mixed logits torch computes from them"""
    c = (synth.normal(f"topk.{name}.cond", (ROWS, V), scale, 0)).bfloat16()
    u = (synth.normal(f"topk.{name}.uncond", (ROWS, V), scale, 0)).bfloat16()
    mixed = u.float() + (c.float() - u.float()) * CFG
    return c, u, mixed


def main():
    MG.load_reference()
    REF = MG.REF

    def _load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    _load("egom2p.utils.generation", "egom2p/utils/generation.py")
    sys.modules.setdefault("egom2p.utils.tokenizer", type(sys)("egom2p.utils.tokenizer"))
    tt = _load("egom2p.utils.tokenizer.text_tokenizer", "egom2p/utils/tokenizer/text_tokenizer.py")
    sys.modules["egom2p.utils"].get_sentinel_to_id_mapping = tt.get_sentinel_to_id_mapping
    sys.modules["egom2p.utils"].merge_span_masking = tt.merge_span_masking
    G = _load("egom2p.models.generate", "egom2p/models/generate.py")
    sampler = G.GenerationSampler(torch.nn.Identity())           # the filter touches no model state

    gold = {"meta": np.array(repr(dict(rows=ROWS, cfg_scale=CFG, cases={k: list(v) for k, v in CASES.items()})))}
    for name, (V, top_k, top_p, scale) in CASES.items():
        _, _, mixed = case_logits(name, V, scale)
        out = sampler.top_k_top_p_filtering(mixed.clone(), top_k=top_k, top_p=top_p)
        kept = torch.isfinite(out)
        n = int(kept.sum(1).max())
        ids = np.full((ROWS, n), -1, np.int32)
        for r in range(ROWS):
            k = kept[r].nonzero()[:, 0].numpy()
            ids[r, :len(k)] = k
        gold[f"kept.{name}"] = ids
        # where the reference's choice is arbitrary: tokens that tie (in the fp32 mixed logit) with the smallest kept one
        lo = torch.where(kept, mixed, torch.full_like(mixed, float("inf"))).min(1).values
        gold[f"ties.{name}"] = (mixed == lo[:, None]).sum(1).numpy().astype(np.int32)
        print(f"[goldens] topk {name}: kept per row {kept.sum(1).tolist()}, ties at the boundary {gold[f'ties.{name}'].tolist()}")
    path = os.path.join(ROOT, "tests", "golden", "topk_filter.npz")
    np.savez_compressed(path, **gold)
    print(f"[goldens] -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


if __name__ == "__main__":
    main()
