"""Drop-in `EgoM2P` module surface over the HIP engine.

Same constructor keywords, `forward(mod_dict, num_encoder_tokens, num_decoder_tokens, loss_type,
return_logits)` signature, methods and `state_dict` key layout as the reference's
`egom2p/models/egom2p_model.py:57-734` and the registry / `create_model` API of
`egom2p/utils/timm/registry.py:25-50`, `model_builder.py:27-74`.  The module owns no math: its
parameters are views into the engine's flat fp32 buffer (`.grad` views into the flat gradient
buffer) and `forward` / `loss.backward()` run the hand-written HIP kernels.

Scope: the SwiGLU / bias-free / LayerNorm(no-bias) variants with token
modalities (`VideoToken*`, `GazeCamToken*` embeddings).  Other variants raise NotImplementedError.
"""
from __future__ import annotations

import random
from functools import partial
from typing import Any, Callable, Dict, Optional

import torch
from torch import nn

from . import _lib as L
from . import ops  # noqa: F401  (importing it fails loudly when the HIP library is missing: no CPU path)
from .config import MODALITIES, Modality, ModelCfg
from .engine import Engine

__all__ = ["EgoM2P", "LayerNorm", "create_model", "register_model", "list_models", "model_entrypoint",
           "VideoTokenEncoderEmbedding", "VideoTokenDecoderEmbedding", "GazeCamTokenEncoderEmbedding",
           "GazeCamTokenDecoderEmbedding", "MODALITY_INFO"]


# ------------------------------------------------------------------------------------------------
# embedding descriptors (same class names / ctor args as the reference; weights live in the engine)
# ------------------------------------------------------------------------------------------------
class _TokenEmbedding(nn.Module):
    kind = "seq1d"

    def __init__(self, vocab_size: int, dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, **kwargs):
        super().__init__()
        if not sincos_pos_emb:
            raise NotImplementedError("learned positional embeddings are outside the hot-path scope")
        self.vocab_size, self.dim_tokens = vocab_size, dim_tokens

    def init(self, dim_tokens: int = 768, init_std: float = 0.02):
        self.dim_tokens = dim_tokens

    @torch.jit.ignore
    def no_weight_decay(self):
        return set()


class GazeCamTokenEncoderEmbedding(_TokenEmbedding):
    """reference: egom2p/models/encoder_embeddings.py:125-210"""


class GazeCamTokenDecoderEmbedding(_TokenEmbedding):
    """reference: egom2p/models/decoder_embeddings.py:272-383"""

    def __init__(self, vocab_size: int, share_embedding: bool = True, **kw):
        super().__init__(vocab_size, **kw)
        self.share_embedding = share_embedding


class VideoTokenEncoderEmbedding(_TokenEmbedding):
    """reference: egom2p/models/encoder_embeddings.py:212-301"""
    kind = "video"

    def __init__(self, vocab_size: int = 64000, patch_size=(4, 8, 8), image_size=256, **kw):
        super().__init__(vocab_size, **kw)
        self.patch_size, self.image_size = patch_size, image_size


class VideoTokenDecoderEmbedding(VideoTokenEncoderEmbedding):
    """reference: egom2p/models/decoder_embeddings.py:385-500"""

    def __init__(self, vocab_size: int = 64000, share_embedding: bool = True, **kw):
        super().__init__(vocab_size, **kw)
        self.share_embedding = share_embedding


def _info(m: Modality, enc, dec):
    return {"vocab_size": m.vocab_size, "encoder_embedding": partial(enc, vocab_size=m.vocab_size),
            "decoder_embedding": partial(dec, vocab_size=m.vocab_size), "min_tokens": 0, "max_tokens": m.max_tokens,
            "type": m.type, "id": m.id, "pretokenized": True}


# the four mod4 entries of egom2p/data/modality_info.py:59-69,75-85,116-141
MODALITY_INFO: Dict[str, Dict[str, Any]] = {
    "tok_rgb": _info(MODALITIES["tok_rgb"], VideoTokenEncoderEmbedding, VideoTokenDecoderEmbedding),
    "tok_depth": _info(MODALITIES["tok_depth"], VideoTokenEncoderEmbedding, VideoTokenDecoderEmbedding),
    "tok_cam": _info(MODALITIES["tok_cam"], GazeCamTokenEncoderEmbedding, GazeCamTokenDecoderEmbedding),
    "tok_gaze": _info(MODALITIES["tok_gaze"], GazeCamTokenEncoderEmbedding, GazeCamTokenDecoderEmbedding),
}


class LayerNorm(nn.Module):
    """Marker with the reference's signature (egom2p/models/egom2p_utils.py:118-133); only used to carry
    `eps` / `bias` through `norm_layer=partial(LayerNorm, eps=1e-6, bias=False)`."""

    def __init__(self, normalized_shape: int, eps=1e-5, bias=True):
        super().__init__()
        self.eps, self.has_bias = eps, bias


class _LossFn(torch.autograd.Function):
    """Connects the engine to `loss.backward()`: backward runs the hand-written backward pass, which
    accumulates straight into the flat gradient buffer (the parameters' `.grad` views)."""

    @staticmethod
    def forward(ctx, anchor, model, loss):
        ctx.model = model
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        m = ctx.model
        m.engine.backward(gout, bucket_done=m._bucket_done if m._sync_grads else None)
        if m._after_backward is not None and m._sync_grads:
            m._after_backward()
        return None, None, None


class EgoM2P(nn.Module):
    def __init__(self,
                 encoder_embeddings: Dict[str, nn.Module],
                 decoder_embeddings: Dict[str, nn.Module],
                 modality_info: Dict[str, Any],
                 dim: int = 768, encoder_depth: int = 12, decoder_depth: int = 12, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, proj_bias: bool = True, mlp_bias: bool = True,
                 drop_path_rate_encoder: float = 0.0, drop_path_rate_decoder: float = 0.0, shared_drop_path: bool = False,
                 act_layer=nn.GELU, norm_layer=partial(LayerNorm, eps=1e-6), gated_mlp: bool = False, qk_norm: bool = False,
                 decoder_causal_mask: bool = False, decoder_sep_mask: bool = True, num_register_tokens: int = 0,
                 use_act_checkpoint: bool = False, share_modality_embeddings: bool = True,
                 device: Optional[str] = None):
        super().__init__()
        probe = norm_layer(dim)
        unsupported = []
        if qkv_bias or proj_bias or mlp_bias: unsupported.append("linear biases")
        if not gated_mlp or act_layer is not nn.SiLU: unsupported.append("non-SwiGLU MLP")
        if getattr(probe, "has_bias", True): unsupported.append("LayerNorm bias")
        if qk_norm: unsupported.append("qk_norm")
        if decoder_causal_mask or not decoder_sep_mask: unsupported.append("causal / non-separated decoder mask")
        if drop_path_rate_encoder or drop_path_rate_decoder: unsupported.append("drop path")
        if not share_modality_embeddings: unsupported.append("unshared modality embeddings")
        if set(encoder_embeddings) != set(decoder_embeddings): unsupported.append("different encoder/decoder modality sets")
        if unsupported:
            raise NotImplementedError("outside the MI355X hot-path scope (SURVEY.md section 8): " + ", ".join(unsupported))
        self.modality_info = modality_info
        self.dim, self.init_std = dim, 0.02
        self.decoder_causal_mask, self.decoder_sep_mask = decoder_causal_mask, decoder_sep_mask
        self.use_act_checkpoint, self.num_register_tokens = use_act_checkpoint, int(num_register_tokens)
        # (register_tokens: nn.Parameter (1, R, dim) attached with the other views when R > 0 - egom2p_model.py:170-174)
        for emb in list(encoder_embeddings.values()) + list(decoder_embeddings.values()):
            emb.init(dim_tokens=dim, init_std=self.init_std)
        self.encoder_modalities, self.decoder_modalities = set(encoder_embeddings), set(decoder_embeddings)
        share = all(getattr(e, "share_embedding", True) for e in decoder_embeddings.values())
        mods = []
        for name, emb in encoder_embeddings.items():
            info = modality_info[name]
            kind = "video" if isinstance(emb, VideoTokenEncoderEmbedding) else "seq1d"
            grid = (5, emb.image_size // emb.patch_size[1], emb.image_size // emb.patch_size[2]) if kind == "video" else (0, 0, 0)
            n_pos = grid[0] * grid[1] * grid[2] if kind == "video" else 30
            m = Modality(name, emb.vocab_size, n_pos, kind, info.get("type", "img"), grid)
            if info.get("id", m.id) != m.id:
                raise NotImplementedError(f"modality id of {name} does not follow sha256(name) % 2^15")
            mods.append(m)
        self._mods = mods
        self.cfg = ModelCfg("custom", dim, encoder_depth, decoder_depth, num_heads, mlp_ratio,
                            modalities=tuple(m.name for m in mods), share_embedding=share, eps=probe.eps,
                            num_register_tokens=int(num_register_tokens))
        # ModelCfg.mods looks names up in MODALITIES: custom vocab / positions go through a private table
        self._device = device or ("cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else None)
        if self._device is None:
            raise L.EgoHipError("egom2p_amd.EgoM2P needs a GPU: the hot path is hand-written HIP with no CPU fallback")
        for m in mods:
            ref = MODALITIES.get(m.name)
            if ref is None or (ref.vocab_size, ref.max_tokens, ref.kind) != (m.vocab_size, m.max_tokens, m.kind):
                raise NotImplementedError(f"modality {m.name} is not one of the mod4 token modalities")
        self.engine: Optional[Engine] = None
        self._engine_shape = (0, 0, 0)
        self._pending_sd: Optional[Dict[str, torch.Tensor]] = None
        self._anchor = torch.zeros((), device=self._device, requires_grad=True)
        self._bucket_done: Optional[Callable] = None
        self._after_backward: Optional[Callable] = None
        self._sync_grads = True
        self._build_engine(1, 1, 1)
        if not self.num_register_tokens:
            self.register_tokens = None                 # the reference's attribute when there are none (:174)
        self.init_weights()

    # ---- engine + parameter views ---------------------------------------------------------------
    def _build_engine(self, B, N, M):
        T = self.cfg.total_positions
        if self.engine is None:
            self.engine = Engine(self.cfg, self._device, max_batch=B, n_enc=min(N, T), n_dec=min(M, T))
            self._attach_views()
        else:                                # parameters (and the nn.Parameter objects) stay; only workspaces change
            self.engine.resize_workspaces(B, min(N, T), min(M, T))
        self._engine_shape = (B, N, M)

    def _attach_views(self):
        """(Re)create the nn.Parameter tree with the reference's key layout over the flat buffers."""
        for k in list(self._modules):
            del self._modules[k]
        for k in list(self._parameters):
            del self._parameters[k]
        eng, D = self.engine, self.dim
        sd = eng.state_dict()
        zeros = torch.zeros(D, device=eng.dev)

        def attach(root, dotted, tensor, grad=None, buffer=False):
            parts = dotted.split(".")
            mod = root
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            if buffer:
                mod.register_buffer(parts[-1], tensor)
            else:
                prm = nn.Parameter(tensor, requires_grad=True)
                prm.grad = grad
                mod.register_parameter(parts[-1], prm)
            return mod

        shared = {}
        for key, val in sd.items():
            if key.endswith("pos_emb") or (key.endswith(".bias") and "norm" in key):
                attach(self, key, val if key.endswith("pos_emb") else zeros, buffer=True)
                continue
            val, g = eng.param_views(key)                  # views of the flat buffers (see Engine.param_views for their shapes)
            ident = (val.data_ptr(), tuple(val.shape))
            if ident in shared:                      # tied tensors share ONE Parameter (mod_emb, to_logits)
                parts = key.split(".")
                mod = self
                for p in parts[:-1]:
                    if p not in mod._modules:
                        mod.add_module(p, nn.Module())
                    mod = mod._modules[p]
                mod.register_parameter(parts[-1], shared[ident])
                continue
            m = attach(self, key, val, g)
            shared[ident] = m._parameters[key.split(".")[-1]]

    def _ensure(self, B, N, M):
        b0, n0, m0 = self._engine_shape
        if B > b0 or N != n0 or M != m0:
            self._build_engine(max(B, b0), N, M)

    # ---- reference API --------------------------------------------------------------------------
    def init_weights(self):
        """MAE-style init of the reference (egom2p_model.py:185-222), drawn on the device.

        `named_modules()` of the reference visits `decoder_embeddings.X.token_emb` (Embedding: N(0, 0.02)) and then
        `decoder_embeddings.X.to_logits` (Linear: xavier-uniform on [V, D]); with `share_embedding` both are ONE
        tensor (decoder_embeddings.py:447-449), so the tied decoder table ends up xavier-uniform."""
        import math
        tied = self.cfg.share_embedding
        seen = set()
        for name, p in self.named_parameters(remove_duplicate=False):
            if id(p) in seen:
                continue
            seen.add(id(p))
            dec_table = name.startswith("decoder_embeddings") and name.endswith(("token_emb.weight", "to_logits.weight"))
            with torch.no_grad():
                if dec_table and (tied or name.endswith("to_logits.weight")):
                    a = math.sqrt(6.0 / float(p.shape[0] + p.shape[1]))
                    p.uniform_(-a, a)
                elif name.endswith("token_emb.weight") or name.endswith("mod_emb") or name in ("mask_token", "register_tokens"):
                    p.normal_(0, self.init_std)                     # (register_tokens: nn.init.normal_(std=init_std), :172)
                elif "norm" in name and name.endswith(".weight") and p.dim() == 1:
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                elif p.dim() >= 2:
                    fan_out, fan_in = self.engine._ref_shape(self.engine._canon_key(name))     # (a padded layout exposes per-head views)
                    if "qkv" in name: fan_out //= 3
                    elif "kv" in name: fan_out //= 2
                    a = math.sqrt(6.0 / float(fan_out + fan_in))
                    p.uniform_(-a, a)
        self.engine.weights_dirty = True

    def get_num_layers_encoder(self): return self.cfg.encoder_depth
    def get_num_layers_decoder(self): return self.cfg.decoder_depth
    def get_num_layers(self): return self.cfg.encoder_depth + self.cfg.decoder_depth

    @torch.jit.ignore
    def no_weight_decay(self):
        return set()

    def state_dict(self, *args, **kwargs):
        """Reference key layout and shapes.  With a padded storage layout (the registered ego-L / ego-XL: Engine docstring) the
        attention weights are exposed as per-head views as PARAMETERS; the state dict holds them in the reference's 2-D shape."""
        if self.engine is not None and self.engine.padded and not args and not kwargs.get("prefix"):
            return dict(self.engine.state_dict())
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self.engine.load_state_dict(state_dict)              # skips (and returns) keys it has no storage for
        own = set(self.state_dict().keys())
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _set_requires_grad(self, pred, flag):
        for n, p in self.named_parameters():
            if pred(n):
                p.requires_grad = flag

    def freeze_encoder(self, freeze_embeddings=True):
        self._set_requires_grad(lambda n: n.startswith(("encoder.", "encoder_norm")) or
                                (freeze_embeddings and n.startswith("encoder_embeddings")), False)

    def freeze_decoder(self, freeze_embeddings=True):
        self._set_requires_grad(lambda n: n.startswith(("decoder.", "decoder_norm")) or
                                (freeze_embeddings and n.startswith("decoder_embeddings")), False)

    def freeze_shared_params(self):
        self.freeze_encoder(False); self.freeze_decoder(False)

    def freeze_params_except_specific_embeddings(self, frozen_embedding_domain):
        doms = frozen_embedding_domain.split("-")
        self.freeze_encoder(False); self.freeze_decoder(False)
        self._set_requires_grad(lambda n: n.startswith(("encoder_embeddings", "decoder_embeddings")) and n.split(".")[1] in doms, False)

    def unfreeze_all(self):
        self._set_requires_grad(lambda n: True, True)

    def forward(self, mod_dict: Dict[str, Dict[str, torch.Tensor]], num_encoder_tokens: int, num_decoder_tokens: int,
                loss_type: str = "mod", return_logits: bool = False):
        if loss_type not in ("mod", "modality", "weighted_mod", "token"):
            raise ValueError("Invalid loss type")                        # egom2p_model.py:579
        names = [m.name for m in self._mods if m.name in mod_dict]
        if len(names) != len(self._mods):
            raise NotImplementedError("every configured modality must be present in mod_dict")
        B = mod_dict[names[0]]["tensor"].shape[0]
        self._ensure(B, num_encoder_tokens, num_decoder_tokens)
        md = {}
        for n in names:
            d = mod_dict[n]
            md[n] = {"tensor": d["tensor"].to(self._device, torch.int64),
                     "input_mask": d["input_mask"].to(self._device, torch.bool).contiguous(),
                     "target_mask": d["target_mask"].to(self._device, torch.bool).contiguous(),
                     "decoder_attention_mask": d["decoder_attention_mask"].to(self._device, torch.int32).contiguous()}
        # the reference shuffles the decoder modality order with python's global `random` (egom2p_model.py:312)
        order = [k for k, _ in random.sample(list(md.items()), len(md))]
        if return_logits:
            return self.engine.forward_logits(md, order)
        loss, mod_loss = self.engine.forward(md, dec_order=order, loss_type=loss_type)
        if torch.is_grad_enabled():
            loss = _LossFn.apply(self._anchor, self, loss)
        else:
            loss = loss.clone()
        return loss, {k: v.clone() for k, v in mod_loss.items()}


# ------------------------------------------------------------------------------------------------
# registry / builder (egom2p/utils/timm/registry.py:25-50, model_builder.py:27-74)
# ------------------------------------------------------------------------------------------------
_model_entrypoints: Dict[str, Callable] = {}


def register_model(fn):
    _model_entrypoints[fn.__name__] = fn
    return fn


def list_models(filter: str = ""):
    import fnmatch
    return sorted(n for n in _model_entrypoints if not filter or fnmatch.fnmatch(n, filter))


def model_entrypoint(name):
    return _model_entrypoints[name]


def create_model(model_name, pretrained=False, checkpoint_path="", **kwargs):
    if model_name not in _model_entrypoints:
        raise RuntimeError("Unknown model (%s)" % model_name)
    return _model_entrypoints[model_name](**kwargs)


def _swiglu_variant(dim, depth, heads):
    def fn(encoder_embeddings, decoder_embeddings, **kwargs):
        return EgoM2P(encoder_embeddings=encoder_embeddings, decoder_embeddings=decoder_embeddings,
                      encoder_depth=depth, decoder_depth=depth, dim=dim, num_heads=heads, mlp_ratio=4,
                      qkv_bias=False, proj_bias=False, mlp_bias=False,
                      norm_layer=partial(LayerNorm, eps=1e-6, bias=False), act_layer=nn.SiLU, gated_mlp=True, **kwargs)
    return fn


# large / xlarge (egom2p_model.py:1080-1118: dim 1020 = 15 heads of 68, dim 2046 = 31 heads of 66) run on padded storage: rows of
# 1024 / 2048, heads of 128, the ego_attn_*_hd kernels (engine.py) - parity configurations; the throughput ego-L is ego_L_1152
for _n, _a in {"egom2p_tiny_6e_6d_swiglu_nobias": (384, 6, 6), "egom2p_small_8e_8d_swiglu_nobias": (512, 8, 8),
               "egom2p_base_12e_12d_swiglu_nobias": (768, 12, 12), "egom2p_large_24e_24d_swiglu_nobias": (1020, 24, 15),
               "egom2p_xlarge_24e_24d_swiglu_nobias": (2046, 24, 31)}.items():
    _f = _swiglu_variant(*_a)
    _f.__name__ = _n
    register_model(_f)


def _unsupported(name, why):
    def fn(*a, **k):
        raise NotImplementedError(f"{name}: {why} (outside the MI355X hot-path scope, SURVEY.md section 8)")
    fn.__name__ = name
    return fn


for _n in ("egom2p_tiny_6e_6d_gelu", "egom2p_small_8e_8d_gelu", "egom2p_base_12e_12d_gelu", "egom2p_large_24e_24d_gelu",
           "egom2p_xlarge_24e_24d_gelu"):
    register_model(_unsupported(_n, "GELU / biased variant"))
for _n in ("egom2p_base_12e_12d_swiglu_nobias_causal", "egom2p_base_12e_12d_swiglu_qknorm_nobias",
           "egom2p_large_24e_24d_swiglu_qknorm_nobias", "egom2p_xlarge_24e_24d_swiglu_qknorm_nobias"):
    register_model(_unsupported(_n, "causal / qk-norm variant"))
