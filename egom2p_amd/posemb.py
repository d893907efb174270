"""Fixed sin-cos positional tables (buffers, never trained).

Same functions as the reference's builders, written independently:
  * 1-D, cam / gaze (30 positions): `[sin(n w_d) || cos(n w_d)]`, w_d = T^-(d/(D/2))
    (reference `egom2p/models/egom2p_utils.py:32-44`).
  * 3-D, video (t,h,w = 5,32,32): per axis D/3 channels with sin/cos *interleaved*,
    axes concatenated t | h | w, positions flattened `(t h w)`
    (reference `egom2p/models/egom2p_utils.py:63-86`).
Checked bit-for-bit against the reference in `tests/test_oracle_vs_goldens.py`.
"""
from __future__ import annotations

import torch

from .config import Modality


def posemb_1d(n: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    if dim % 2:
        raise ValueError("1-D sin-cos needs dim % 2 == 0")
    half = dim // 2
    w = 1.0 / (temperature ** (torch.arange(half, dtype=torch.float32) / half))
    ang = torch.arange(n, dtype=torch.float32)[:, None] * w[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=1)[None]                      # (1, n, dim)


def _axis_interleaved(n: int, channels: int, temperature: float) -> torch.Tensor:
    f = 1.0 / (temperature ** (torch.arange(0, channels, 2).float() / channels))
    ang = torch.arange(n, dtype=torch.float32)[:, None] * f[None, :]           # (n, channels/2)
    return torch.stack((ang.sin(), ang.cos()), dim=-1).reshape(n, channels)    # sin,cos interleaved


def posemb_3d(t: int, h: int, w: int, dim: int, temperature: float = 10000.0) -> torch.Tensor:
    if dim % 6:
        raise ValueError("3-D sin-cos needs dim % 6 == 0")
    c = dim // 6 * 2
    out = torch.empty((t, h, w, 3 * c), dtype=torch.float32)
    out[..., :c] = _axis_interleaved(t, c, temperature)[:, None, None, :]
    out[..., c:2 * c] = _axis_interleaved(h, c, temperature)[None, :, None, :]
    out[..., 2 * c:] = _axis_interleaved(w, c, temperature)[None, None, :, :]
    return out.reshape(1, t * h * w, dim)


def build_pos_emb(m: Modality, dim: int) -> torch.Tensor:
    if m.kind == "video":
        return posemb_3d(*m.grid, dim)
    return posemb_1d(m.max_tokens, dim)
