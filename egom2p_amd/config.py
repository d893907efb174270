"""Model / modality configuration for the EgoM2P hot path.

Mirrors the information the reference keeps in `egom2p/data/modality_info.py:59-141`
(vocab size, max tokens, type, sha256-derived id) and the registry entries of
`egom2p/models/egom2p_model.py:882-1196` (depth / width / heads per name).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Dict, List, Tuple


def uint15_hash(name: str) -> int:
    """Modality id = sha256(name) mod 2^15 (reference: `egom2p/utils/misc.py:39-41`)."""
    return int(hashlib.sha256(name.encode("utf-8")).hexdigest(), 16) % (2 ** 15)


@dataclass(frozen=True)
class Modality:
    name: str
    vocab_size: int
    max_tokens: int          # number of clip positions of this modality
    kind: str                # 'video' (3-D sin-cos, 5x32x32) or 'seq1d' (1-D sin-cos, 30)
    type: str                # reference 'type' field: img / cam / gaze
    grid: Tuple[int, int, int] = (0, 0, 0)   # (t, h, w) for video

    @property
    def id(self) -> int:
        return uint15_hash(self.name)


# The four modalities of the mod4 configs (reference `modality_info.py:59-69,75-85,116-141`).
TOK_RGB = Modality("tok_rgb", 64000, 5120, "video", "img", (5, 32, 32))
TOK_DEPTH = Modality("tok_depth", 64000, 5120, "video", "img", (5, 32, 32))
TOK_CAM = Modality("tok_cam", 256, 30, "seq1d", "cam")
TOK_GAZE = Modality("tok_gaze", 256, 30, "seq1d", "gaze")

MODALITIES: Dict[str, Modality] = {m.name: m for m in (TOK_RGB, TOK_DEPTH, TOK_CAM, TOK_GAZE)}
# Build-owned parity-test modalities: six more 30-token sequence modalities of the cam / gaze embedding class
# (`GazeCamToken{Encoder,Decoder}Embedding`), so that a model with EGO_MAX_MODS = 8 modalities can be pinned against
# the reference (tests/golden/tiny8.npz).
for _i in range(6):
    MODALITIES[f"tok_aux{_i}"] = Modality(f"tok_aux{_i}", 256, 30, "seq1d", "cam" if _i % 2 == 0 else "gaze")


@dataclass(frozen=True)
class ModelCfg:
    """Shape of one EgoM2P variant (SwiGLU, bias-free, LayerNorm(no bias, eps 1e-6))."""
    name: str
    dim: int
    encoder_depth: int
    decoder_depth: int
    num_heads: int
    mlp_ratio: float = 4.0
    modalities: Tuple[str, ...] = ("tok_rgb", "tok_depth", "tok_cam", "tok_gaze")
    share_embedding: bool = True       # decoder to_logits tied to decoder token_emb
    eps: float = 1e-6
    num_register_tokens: int = 0       # learned [1, R, dim] rows in front of every sample's encoder tokens (egom2p_model.py:170-171, 381-387)

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads

    @property
    def mlp_hidden(self) -> int:
        # GatedMlp: int(2 * int(dim * mlp_ratio) / 3)  (reference `egom2p_utils.py:161,349`)
        return int(2 * int(self.dim * self.mlp_ratio) / 3)

    @property
    def mods(self) -> List[Modality]:
        return [MODALITIES[m] for m in self.modalities]

    @property
    def total_positions(self) -> int:
        return sum(m.max_tokens for m in self.mods)


# Registered SwiGLU/no-bias variants (reference `egom2p_model.py:982-1120`) plus the two
# build-owned configs named by BASELINE.json (ego-tiny plumbing config, ego-L MFMA-aligned).
MODEL_CFGS: Dict[str, ModelCfg] = {
    "egom2p_tiny_6e_6d_swiglu_nobias": ModelCfg("egom2p_tiny_6e_6d_swiglu_nobias", 384, 6, 6, 6),
    "egom2p_small_8e_8d_swiglu_nobias": ModelCfg("egom2p_small_8e_8d_swiglu_nobias", 512, 8, 8, 8),
    "egom2p_base_12e_12d_swiglu_nobias": ModelCfg("egom2p_base_12e_12d_swiglu_nobias", 768, 12, 12, 12),
    "egom2p_large_24e_24d_swiglu_nobias": ModelCfg("egom2p_large_24e_24d_swiglu_nobias", 1020, 24, 24, 15),
    # build-owned
    "ego_tiny_2e_2d": ModelCfg("ego_tiny_2e_2d", 128, 2, 2, 2, modalities=("tok_cam", "tok_gaze")),
    # eight modalities (the C-ABI's EGO_MAX_MODS) at plumbing size
    "ego_tiny8_2e_2d": ModelCfg("ego_tiny8_2e_2d", 128, 2, 2, 2,
                                modalities=("tok_cam", "tok_gaze") + tuple(f"tok_aux{i}" for i in range(6))),
    "ego_b_2e_2d": ModelCfg("ego_b_2e_2d", 768, 2, 2, 12),
    # the same with four register tokens (`--num_register_tokens 4`, run_training_egom2p.py:70, 386): tests/golden/b2_reg4.npz
    "ego_b_2e_2d_reg4": ModelCfg("ego_b_2e_2d_reg4", 768, 2, 2, 12, num_register_tokens=4),
    # untied decoder head (`share_embedding=False`, the FM wrapper's setting: egom2p_model.py:856-858) at parity-test depth
    "ego_b_2e_2d_untied": ModelCfg("ego_b_2e_2d_untied", 768, 2, 2, 12, share_embedding=False),
    "ego_L_1152": ModelCfg("ego_L_1152", 1152, 24, 24, 18),
    "ego_L_1152_2e_2d": ModelCfg("ego_L_1152_2e_2d", 1152, 2, 2, 18),       # ego-L width (BASELINE config 5) at parity-test depth
    # the REGISTERED ego-L geometry (egom2p_model.py:1080-1092: dim 1020, 15 heads of 68, F = 2720) at parity-test depth;
    # stored in rows of 1024 with heads padded to 128 (engine.py) and run by the ego_attn_*_hd kernels
    "ego_L_1020_2e_2d": ModelCfg("ego_L_1020_2e_2d", 1020, 2, 2, 15),
    # the registered ego-XL geometry (egom2p_model.py:1100-1118: dim 2046, 31 heads of 66, F = 5456) at parity-test depth:
    # rows of 2048, heads of 128, F padded to 5504 (the fused SwiGLU backward needs F % 256: this shape takes the two-call form)
    "ego_XL_2046_1e_1d": ModelCfg("ego_XL_2046_1e_1d", 2046, 1, 1, 31),
    "ego_gen_384_2e_2d": ModelCfg("ego_gen_384_2e_2d", 384, 2, 2, 6, modalities=("tok_rgb", "tok_depth")),
    "ego_gen_384_2e_2d_reg4": ModelCfg("ego_gen_384_2e_2d_reg4", 384, 2, 2, 6, modalities=("tok_rgb", "tok_depth"), num_register_tokens=4),
    # config 4 (rgb -> depth generation) at ego-b width: D = 768, 12 heads of 64, F = 2048, at parity-test depth
    "ego_b_gen_2e_2d": ModelCfg("ego_b_gen_2e_2d", 768, 2, 2, 12, modalities=("tok_rgb", "tok_depth")),
}
