"""`UnifiedMasking` of the reference (egom2p/data/masking.py:131-266, 519-560) on the device, for whole batches.

The reference masks one sample at a time in CPU dataloader workers: mixture component and token counts (:530-536), Dirichlet
token budgets (`input_token_budget` / `target_token_budget`, :181-234), then per modality a random permutation whose first
`input_budget` positions become encoder inputs and the next `target_budget` decoder targets, with `decoder_attention_mask`
carrying the target count at the first target position (`image_mask`, :236-266).  Here the same contract is produced for a
batch `{modality: tokens [B, ...]}` by two HIP kernels (ego_budget_dirichlet, ego_clip_synth with ids = NULL): the token
tensors are passed through untouched, nothing but one 64-bit stream key per clip crosses the host-device boundary.

Scope: the modality types the EgoM2P hot path uses - 'img', 'cam', 'gaze', 'keypoints' (the reference's `image_mask`
branch, :546-547).  Sequence types ('seq', 'seq_token', 'seq_emb': span masking with a text tokenizer) raise.
Randomness is a counter-based stream (seed, call index, sample index), not torch's global generator: the budgets follow the
reference's distribution (tests/test_frontend_gpu.py checks their moments), individual draws differ.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import ops
from .synth import MOD4_MIXTURE_ALPHAS, _key

_WHOLE = ("img", "cam", "gaze", "keypoints")


def _pair(x) -> Tuple[int, int]:
    return (int(x), int(x)) if isinstance(x, (int, float)) else (int(x[0]), int(x[1]))


class UnifiedMasking:
    def __init__(self, modality_info: Dict[str, Dict], text_tokenizer=None,
                 input_tokens_range: Union[int, Tuple[int, int]] = 2048,
                 target_tokens_range: Optional[Union[int, Tuple[int, int]]] = 2048, max_tries: int = 100,
                 sampling_weights: Optional[Sequence[float]] = None, seed: int = 0, device: str = "cuda"):
        if target_tokens_range is None:
            raise NotImplementedError("target_tokens_range=None (all non-input tokens are targets) is outside the hot-path scope")
        self.names = list(modality_info)
        for n in self.names:
            if modality_info[n].get("type", "img") not in _WHOLE:
                raise NotImplementedError(f"modality {n}: type {modality_info[n].get('type')!r} needs the reference's sequence masking")
        self.max_tokens = [int(modality_info[n]["max_tokens"]) for n in self.names]
        self.min_tokens = [int(modality_info[n].get("min_tokens", 0)) for n in self.names]
        # per modality a list of alphas, one per mixture component (the reference sets them from the alphas yaml:
        # cfgs/default/egom2p/alphas_mixture/main/mix_mod4_all2all_uni.yaml); default: that file's symmetric mixture
        def alphas(key):
            per_mod = [list(modality_info[n].get(key, MOD4_MIXTURE_ALPHAS)) for n in self.names]
            n_mix = len(per_mod[0])
            assert all(len(a) == n_mix for a in per_mod), f"{key}: every modality needs one alpha per mixture component"
            return [[per_mod[i][j] for i in range(len(self.names))] for j in range(n_mix)]          # [n_mix][n_mods]
        self.in_alphas, self.tgt_alphas = alphas("input_alphas"), alphas("target_alphas")
        assert len(self.in_alphas) == len(self.tgt_alphas)
        self.weights = [1.0] * len(self.in_alphas) if sampling_weights is None else [float(w) for w in sampling_weights]
        assert len(self.weights) == len(self.in_alphas)
        self.in_range, self.tgt_range = _pair(input_tokens_range), _pair(target_tokens_range)
        self.max_tries, self.seed, self.device, self.calls = int(max_tries), int(seed), device, 0

    def __call__(self, mod_dict: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
        """mod_dict: {modality: token tensor [B, ...] with prod(...) == max_tokens}, on the device.  Returns the batched
        `{modality: {tensor, input_mask, target_mask, decoder_attention_mask}}` the model's forward consumes."""
        B = next(iter(mod_dict.values())).shape[0]
        base = self.calls * B
        self.calls += 1
        dev = self.device
        keys = np.array([_key(f"mask{base + b}.budget", self.seed) for b in range(B)], dtype=np.uint64)
        kd = torch.from_numpy(keys.view(np.int64)).to(dev)
        k_in = torch.empty(len(self.names), B, dtype=torch.int32, device=dev)
        k_tg = torch.empty(len(self.names), B, dtype=torch.int32, device=dev)
        ops.budget_dirichlet(kd, self.in_alphas, self.tgt_alphas, self.weights, self.max_tokens, self.min_tokens,
                             [True] * len(self.names), self.in_range, self.tgt_range, k_in, k_tg, max_tries=self.max_tries)
        out: Dict[str, Dict[str, torch.Tensor]] = {}
        for j, name in enumerate(self.names):
            t = mod_dict[name]
            n = self.max_tokens[j]
            assert t.is_cuda and t.shape[0] == B and t[0].numel() == n, (name, tuple(t.shape), n)
            pk = np.array([_key(f"mask{base + b}.{name}.perm", self.seed) for b in range(B)], dtype=np.uint64)
            pkd = torch.from_numpy(pk.view(np.int64)).to(dev)
            in_mask = torch.empty((B, n), dtype=torch.bool, device=dev)
            tg_mask = torch.empty((B, n), dtype=torch.bool, device=dev)
            dam = torch.empty((B, n), dtype=torch.int32, device=dev)
            ops.clip_synth(pkd, pkd, k_in[j], k_tg[j], n, 1, None, in_mask, tg_mask, dam)
            out[name] = {"tensor": t, "input_mask": in_mask, "target_mask": tg_mask, "decoder_attention_mask": dam}
        return out
