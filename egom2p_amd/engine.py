"""EgoM2P training/inference engine for one MI355X: owns the flat fp32 parameter / gradient buffers,
the bf16 weight copies, the activation workspaces, and issues the HIP kernels of one forward and one
hand-ordered backward pass (no autograd graph, no tracing compiler).

Dataflow = `EgoM2P.forward` of the reference (egom2p/models/egom2p_model.py:683-734) with the
autocast(bf16) numerics of `run_training_egom2p.py:725`: GEMM-class ops in bf16 with fp32 MFMA
accumulation, LayerNorm / softmax / cross-entropy / residual stream in fp32.

Layout decisions (MI355X-first, 288 GB HBM):
  * every parameter lives in ONE flat fp32 buffer in forward order; gradients in a twin buffer, so
    gradients become final from the tail backwards and per-layer slices are the all-reduce buckets;
  * every activation needed by the backward is kept (1.7 GB per clip at ego-b): no recompute;
  * decoder rows are written modality-grouped by the final LayerNorm so the per-modality logits / CE
    run on contiguous row ranges whose (offset, count) stay on the device (no host sync anywhere);
  * MLP hidden size is padded to a multiple of 128 in the masters themselves (pad entries are exactly
    zero and stay zero), so every GEMM dimension is tile-aligned.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops
from .config import ModelCfg, Modality
from .posemb import build_pos_emb

BF16 = torch.bfloat16
F32 = torch.float32
I32 = torch.int32


_PAIR_STREAMS: Dict[int, "torch.cuda.Stream"] = {}
_WARM_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def warmup_stream(dev: torch.device) -> "torch.cuda.Stream":
    """The side stream graph captures warm up on: one per device and process (a stream is a hardware queue)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _WARM_STREAMS:
        _WARM_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _WARM_STREAMS[key]


def head_layout(H: int, HD: int, min_pad: int = 0) -> Tuple[int, int]:
    """(elements between two stored heads, heads in storage) for H heads of dimension HD.

    Heads of 64 are stored as they are (the throughput attention kernels); A = H * 64 is the N / K extent of the qkv / proj
    GEMMs and the pitch of the [rows, A] buffers, which the kernels are validated for in multiples of 128 only: an odd head
    count (dim 960 = 15 x 64) is refused.  Other head dimensions are zero-padded to 96 or 128 (ego_attn_*_hd kernels) and, where
    that makes A a multiple of 128 with narrower rows, an all-zero PHANTOM head is appended: the registered ego-L (15 heads of
    68) is 16 x 96 = 1536 wide - the phantom head's q = k = v = 0 give a uniform softmax over zero values, output and every
    gradient through it exactly 0 - where 15 x 128 = 1920 cost 25 % more attention and qkv / proj GEMM work.
    min_pad (EGOM2P_HEAD_PAD) = 128 keeps the round-3 layout."""
    if HD == 64:
        if (H * 64) % 128:
            raise L.EgoHipError(f"heads of 64 need an even head count (heads={H}: A = {H * 64} is not a multiple of 128)")
        return 64, H
    fits = sorted((next(n for n in range(H, H + 9) if (n * h) % 128 == 0) * h, h) for h in (96, 128) if h >= HD and h >= min_pad)
    if not fits:
        raise L.EgoHipError(f"no storage layout for {H} heads of {HD}")
    return fits[0][1], fits[0][0] // fits[0][1]


def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


class _Lin:
    """One linear layer: master view [out, in] (possibly padded), bf16 W [out, in] and W^T [in, out]; with the fp8
    forward enabled also W as OCP e4m3 bytes with one fp32 scale per output channel."""
    __slots__ = ("name", "w", "g", "wb", "wt", "out_f", "in_f", "w8", "s8", "wt8", "st8")

    def __init__(self, name, w, g, dev, fp8=False, fp8_bwd=False):
        self.name, self.w, self.g = name, w, g
        self.out_f, self.in_f = w.shape
        self.wb = torch.zeros(self.out_f, self.in_f, device=dev, dtype=BF16)
        self.wt = torch.zeros(self.in_f, self.out_f, device=dev, dtype=BF16)
        ok8 = fp8 and self.in_f % 128 == 0 and self.in_f >= 256 and self.out_f % 128 == 0
        self.w8 = torch.zeros(self.out_f, self.in_f, device=dev, dtype=torch.uint8) if ok8 else None
        self.s8 = torch.ones(self.out_f, device=dev, dtype=F32) if ok8 else None
        # fp8 dgrad (dX = dY W): W^T [in, out] as e4m3 with one scale per input channel - the contraction runs over `out`
        okb = fp8_bwd and self.out_f % 128 == 0 and self.out_f >= 256 and self.in_f % 128 == 0
        self.wt8 = torch.zeros(self.in_f, self.out_f, device=dev, dtype=torch.uint8) if okb else None
        self.st8 = torch.ones(self.in_f, device=dev, dtype=F32) if okb else None


class Engine:
    def __init__(self, cfg: ModelCfg, device="cuda:0", max_batch: int = 1, n_enc: int = 2048, n_dec: int = 2048,
                 attn_o_residual: str = "cross", fp8_forward: bool = False, fp8_backward: bool = False):
        """attn_o_residual: which attention sites also keep the bf16 rounding residual of their output so that the
        backward's delta = rowsum(dO o O) is formed from O to ~16 bits ("cross": the cross-attention sites - where, with
        near-uniform attention over ~2000 context keys, the plain flash-style delta put 3.5 % error on the query-path
        gradients; "all"; "none").  Costs one more bf16 [rows, D] write + read per site (~0.5 % of a step for "cross")."""
        L.load()  # fail loudly if the HIP library is missing
        # fp8_forward (BASELINE config 5, "bf16 + fp8 MFMA GEMMs"): the forward linears (qkv / q / kv / proj / fc1||fc3 / fc2 /
        # context projection) run on e4m3 operands - activations quantised per row right before the GEMM, weights per
        # output channel once per optimiser step - with fp32 accumulation; the backward and the logits stay bf16.
        self.fp8_forward = bool(fp8_forward)
        # fp8_backward (round 5): the dgrad GEMMs dX = dY W on e4m3 operands too - dY quantised per row (the
        # contraction runs over the output channels, so a per-row scale of dY and a per-input-channel scale of W^T factor out
        # exactly like the forward's scales); e4m3 rather than e5m2 because the per-ROW scale already takes the gradient's
        # row-to-row dynamic range out and the 3-bit mantissa is then the limit.  Weight gradients stay bf16: their contraction
        # runs over the rows, where per-row scales do not factor out.  Measured gain / kill criterion: DESIGN section 4f.
        self.fp8_backward = bool(fp8_backward)
        if attn_o_residual not in ("cross", "all", "none"):
            raise ValueError("attn_o_residual must be 'cross', 'all' or 'none'")
        self.attn_o_residual = attn_o_residual
        # Storage geometry.  D = the row pitch of every [rows, dim] activation and of every weight's model-dim axis: cfg.dim
        # rounded up to 128 (GEMM K steps / tiles) with zero pad columns - equal to cfg.dim for ego-b and the 1152-wide
        # ego-L; the REGISTERED ego-L (egom2p_model.py:1080-1092: dim 1020, 15 heads of 68) lives in rows of 1024.  Heads
        # are stored HDP elements apart (64 -> 64: the throughput attention kernels; otherwise zero-padded to 96 or 128 for
        # the ego_attn_*_hd kernels), A = H * HDP is the width of the q / k / v / attention-output rows.  Pad columns and
        # pad weight rows are zero and stay zero: their gradients are exact zeros (zero operands), and AdamW maps (0, 0) to 0.
        self.cfg = cfg
        self.dev = torch.device(device)
        self.Dl, self.D, self.H = cfg.dim, _pad128(cfg.dim), cfg.num_heads
        self.HD = cfg.head_dim
        self.HDP, self.Hs = head_layout(cfg.num_heads, cfg.head_dim, int(os.environ.get("EGOM2P_HEAD_PAD", "0")))
        self.A = self.Hs * self.HDP
        self.padded = self.D != self.Dl or self.HDP != self.HD or self.Hs != self.H
        if self.padded and (fp8_forward or fp8_backward):
            raise L.EgoHipError("the fp8 GEMMs are built for the unpadded shapes (dim % 128 == 0, head_dim 64)")
        self.F, self.Fp = cfg.mlp_hidden, _pad128(cfg.mlp_hidden)
        self.mods: List[Modality] = cfg.mods
        self.n_mods = len(self.mods)
        self.N, self.M, self.Bmax = n_enc, n_dec, max_batch
        # register tokens (egom2p_model.py:170-171, 381-387): R learned rows in front of every sample's N kept encoder tokens - the
        # encoder, the context and the cross-attention keys have Ne = R + N rows per sample; the compaction keeps N
        self.R = int(getattr(cfg, "num_register_tokens", 0))
        self.Ne = self.N + self.R
        self.scale = cfg.head_dim ** -0.5
        self._build_params()
        self._alloc_workspaces()
        self.weights_dirty = True
        self._have_fwd = False
        self._ce_done, self._ce_grad = set(), None      # modalities whose CE backward ran inside the forward (loss_grad)
        self.max_graphs = 16                  # captured generation passes kept alive at once (LRU)
        self._iw, self._infer_key = None, None
        # optional: weight-gradient GEMMs on a side stream beside the dgrad / attention chain of the same layer, joined
        # at every bucket boundary.  Measured on MI355X: parity-clean but 5 % SLOWER than one stream (the co-running
        # kernels fight for LDS / L2), so it is off by default (EGOM2P_WGRAD_STREAM=1 enables it).
        self.side = torch.cuda.Stream(device=self.dev) if os.environ.get("EGOM2P_WGRAD_STREAM", "0") == "1" else None
        # optional (EGOM2P_WGRAD_PAIR=1): weight-gradient GEMMs paired with the LayerNorm backward launches (MFMA-bound beside
        # HBM-bound, never two MFMA kernels at once): _ln_bwd.  Measured on MI355X: 191.1 vs 190.5 clips/s (two runs each, one
        # box) - inside the noise: the LayerNorm backward already takes 5.4 TB/s and the GEMM's own 2.4 TB/s has nowhere to go.  Off.
        self._pend, self.pair_stream = None, None
        if self.side is None and os.environ.get("EGOM2P_WGRAD_PAIR", "0") == "1":
            # ONE second stream per device and process (every torch.cuda.Stream() is another hardware queue; a long-lived process
            # that builds many engines - the test suite - would otherwise oversubscribe the queues of a GPU it shares)
            key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
            if key not in _PAIR_STREAMS:
                _PAIR_STREAMS[key] = torch.cuda.Stream(device=self.dev)
            self.pair_stream = _PAIR_STREAMS[key]
            self._pair_ev, self._pair_i = [torch.cuda.Event() for _ in range(16)], 0
        # decoder self-attention launched by row groups (one interval per workgroup); 0: per-row interval launches (round 3)
        self.attn_groups = os.environ.get("EGOM2P_ATTN_GROUPS", "1") != "0"
        self.attn_split = os.environ.get("EGOM2P_ATTN_SPLIT", "1") != "0"       # generation path: split keys on under-filled grids
        self.cfg_pair = os.environ.get("EGOM2P_CFG_PAIR", "1") != "0"           # guided step: cond + uncond share one decoder pass
        self.gen_overlap = os.environ.get("EGOM2P_GEN_OVERLAP", "0") == "1"     # generation: independent launches on a second stream (_fork)
        # every decoder layer's context_norm normalises the SAME context tensor: one fused launch forward (x and its statistics
        # read once, one output per layer) and one backward (x once, one write of the context gradient) instead of one per layer
        self.ctx_ln_fused = (os.environ.get("EGOM2P_CTX_LN_FUSED", "1") != "0" and not self.fp8_forward and self.D <= 1536
                             and 0 < cfg.decoder_depth <= 32)

    def layout_tag(self) -> Tuple[int, ...]:
        """What the PHYSICAL flat buffers (P, G and the optimiser's m / v) depend on beyond the model: saved with optimiser
        checkpoints and checked on load (a resume under another EGOM2P_HEAD_PAD must not die in a bare copy_)."""
        return (2, int(self.n_flat), int(self.D), int(self.Hs), int(self.HDP), int(self.Fp))    # 2: context_norm weights in one group (round 5)

    # ---- generation path: independent launches of an under-filled pass beside each other (round 5) ----------------------------------
    # At batch 1 the rgb -> depth passes are latency-bound (3414 decoder rows fill a third of the chip; the unconditional
    # encoder group's self-attention is 324 workgroups for 768 slots), and several of their launches do not depend on each other:
    # the cross-attention kv projections of ALL decoder layers depend only on the context, the two halves of a guided step's
    # cross-attention and the two encoder groups' self-attention launches are independent.  They go to a second stream (fork / join
    # by stream waits, which a hipGraph capture turns into parallel branches); EGOM2P_GEN_OVERLAP=0 keeps one stream.
    def _fork(self, fn, which: int = 0):
        """run fn on side stream `which` (0: the launch beside the current one, 1: the decoder's kv side chain - a long-lived
        branch that must not sit in front of the short ones), ordered after everything issued on the current stream so far"""
        if not self.gen_overlap:
            fn()
            return None
        key = (self.dev.index if self.dev.index is not None else torch.cuda.current_device(), "gen", which)
        if key not in _PAIR_STREAMS:
            _PAIR_STREAMS[key] = torch.cuda.Stream(device=self.dev)      # (one per device and process: a stream is a hardware queue)
        side, main = _PAIR_STREAMS[key], torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
        return side

    def _join(self, side):
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def _ring_next(self):
        self._ring_i = (self._ring_i + 1) % len(self.ring_b)
        return self.ring_b[self._ring_i]

    def _join_side(self):
        if self._pend is not None:                 # paired mode: a weight gradient still held back goes out now
            self._pend()
            self._pend = None
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def _ln_bwd(self, *a, **kw):
        """ops.layernorm_bwd - an HBM-bound launch that leaves the MFMA pipes idle - with the weight-gradient GEMM that was held
        back (an MFMA-bound launch with no consumer before the next bucket boundary) beside it on a second stream: the GEMM
        starts when the LayerNorm does, and the main stream goes on only after both (two MFMA kernels never overlap: that was
        measured 5 - 26 % slower)."""
        run, self._pend = self._pend, None
        if run is None:
            return ops.layernorm_bwd(*a, **kw)
        main = torch.cuda.current_stream()
        e0, e1 = self._pair_ev[self._pair_i], self._pair_ev[self._pair_i + 1]
        self._pair_i = (self._pair_i + 2) % len(self._pair_ev)
        e0.record(main)
        with torch.cuda.stream(self.pair_stream):
            self.pair_stream.wait_event(e0)
            run()
            e1.record(self.pair_stream)
        ops.layernorm_bwd(*a, **kw)
        main.wait_event(e1)

    # ------------------------------------------------------------------------------------ parameters
    def _build_params(self):
        D, A, Fp, cfg = self.D, self.A, self.Fp, self.cfg
        spec: List[Tuple[str, Tuple[int, ...]]] = []
        groups: List[Tuple[str, int, int]] = []      # (group name, lo, hi) = all-reduce buckets, forward order

        def add(name, shape):
            spec.append((name, tuple(shape)))

        def layer(prefix, kind):
            # (a decoder layer's context_norm weight lives in the "ctx_norms" group in front of the decoder layers: with the fused
            #  context LayerNorm backward its gradient is only final after the LAST decoder layer's backward, and it must not
            #  hold the layer's 14 MB gradient bucket back from the data-parallel exchange until then - ADVICE r4)
            norms = ["norm1", "norm2"] if kind == "enc" else ["norm1", "query_norm", "norm2"]
            for n in norms:
                add(f"{prefix}.{n}.weight", (D,))
            if kind == "enc":
                add(f"{prefix}.attn.qkv.weight", (3 * A, D))
                add(f"{prefix}.attn.proj.weight", (D, A))
            else:
                add(f"{prefix}.self_attn.qkv.weight", (3 * A, D))
                add(f"{prefix}.self_attn.proj.weight", (D, A))
                add(f"{prefix}.cross_attn.q.weight", (A, D))
                add(f"{prefix}.cross_attn.kv.weight", (2 * A, D))
                add(f"{prefix}.cross_attn.proj.weight", (D, A))
            add(f"{prefix}.mlp.fc1.weight", (Fp, D))      # fc1 | fc3 adjacent: one fused [2Fp, D] GEMM operand
            add(f"{prefix}.mlp.fc3.weight", (Fp, D))
            add(f"{prefix}.mlp.fc2.weight", (D, Fp))

        marks = []
        for m in self.mods:
            marks.append((f"enc_table.{m.name}", len(spec)))
            add(f"encoder_embeddings.{m.name}.token_emb.weight", (m.vocab_size, D))
        marks.append(("mod_emb", len(spec)))
        for m in self.mods:
            add(f"encoder_embeddings.{m.name}.mod_emb", (D,))
        if getattr(cfg, "num_register_tokens", 0):
            add("register_tokens", (cfg.num_register_tokens, D))       # reference shape (1, R, dim); decayed like mod_emb / mask_token
        for i in range(cfg.encoder_depth):
            marks.append((f"encoder.{i}", len(spec)))
            layer(f"encoder.{i}", "enc")
        marks.append(("bridge", len(spec)))
        add("encoder_norm.weight", (D,))
        add("decoder_proj_context.bias", (D,))
        add("decoder_proj_context.weight", (D, D))
        if cfg.decoder_depth:
            marks.append(("ctx_norms", len(spec)))
            for i in range(cfg.decoder_depth):
                add(f"decoder.{i}.context_norm.weight", (D,))
        for i in range(cfg.decoder_depth):
            marks.append((f"decoder.{i}", len(spec)))
            layer(f"decoder.{i}", "dec")
        marks.append(("head", len(spec)))
        add("decoder_norm.weight", (D,))
        add("mask_token", (D,))
        for m in self.mods:
            marks.append((f"dec_table.{m.name}", len(spec)))
            add(f"decoder_embeddings.{m.name}.token_emb.weight", (m.vocab_size, D))
        if not cfg.share_embedding:
            for m in self.mods:
                marks.append((f"to_logits.{m.name}", len(spec)))
                add(f"decoder_embeddings.{m.name}.to_logits.weight", (m.vocab_size, D))

        offs, total = {}, 0
        for name, shape in spec:
            n = math.prod(shape)
            offs[name] = (total, n, shape)
            total += (n + 3) // 4 * 4
        self.n_flat = total
        self.P = torch.zeros(total, device=self.dev, dtype=F32)
        self.G = torch.zeros(total, device=self.dev, dtype=F32)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.offsets = offs
        for name, (o, n, shape) in offs.items():
            self.p[name] = self.P[o:o + n].view(shape)
            self.g[name] = self.G[o:o + n].view(shape)
        starts = [offs[spec[i][0]][0] for _, i in marks] + [total]
        self.buckets = [(marks[i][0], starts[i], starts[i + 1]) for i in range(len(marks))]
        # optimiser runs: contiguous ranges of equal weight-decay class (no_decay: norm weights and biases)
        # tensors no kernel ever writes a gradient for: with an untied head the decoder token tables are never read (the
        # decoder rows are the mask token, egom2p_model.py:328), the reference leaves their .grad None and AdamW skips them
        self.never_grad = set() if cfg.share_embedding else {f"decoder_embeddings.{m.name}.token_emb.weight" for m in self.mods}
        runs: List[List] = []
        for name, shape in spec:
            if name in self.never_grad:
                continue
            o, n, _ = offs[name]
            nd = ("norm." in name or ".norm" in name or name.endswith(".bias"))
            n4 = (n + 3) // 4 * 4
            if runs and runs[-1][2] == nd and runs[-1][1] == o:
                runs[-1][1] = o + n4
            else:
                runs.append([o, o + n4, nd])
        self.opt_runs = [(a, b, nd) for a, b, nd in runs]
        # fixed positional tables (buffers)
        self.pos = {}
        for m in self.mods:
            pe = torch.zeros(m.max_tokens, D, device=self.dev, dtype=F32)
            pe[:, :self.Dl] = build_pos_emb(m, self.Dl)[0].to(self.dev)
            self.pos[m.name] = pe

        # linear layers: bf16 copies
        self.lin: Dict[str, _Lin] = {}

        def lin(name):
            body = "token_emb" not in name and "to_logits" not in name
            self.lin[name] = _Lin(name, self.p[name], self.g[name], self.dev, fp8=self.fp8_forward and body, fp8_bwd=self.fp8_backward and body)

        for i in range(cfg.encoder_depth):
            for s in ("attn.qkv", "attn.proj", "mlp.fc2"):
                lin(f"encoder.{i}.{s}.weight")
            self._fuse13(f"encoder.{i}")
        for i in range(cfg.decoder_depth):
            for s in ("self_attn.qkv", "self_attn.proj", "cross_attn.q", "cross_attn.kv", "cross_attn.proj", "mlp.fc2"):
                lin(f"decoder.{i}.{s}.weight")
            self._fuse13(f"decoder.{i}")
        lin("decoder_proj_context.weight")
        for m in self.mods:
            key = (f"decoder_embeddings.{m.name}.token_emb.weight" if cfg.share_embedding
                   else f"decoder_embeddings.{m.name}.to_logits.weight")
            lin(key)
        self.logit_key = {m.name: (f"decoder_embeddings.{m.name}.token_emb.weight" if cfg.share_embedding
                                   else f"decoder_embeddings.{m.name}.to_logits.weight") for m in self.mods}

    def _fuse13(self, prefix):
        """fc1 and fc3 are adjacent in the flat buffer: expose them as one [2Fp, D] linear."""
        o1 = self.offsets[f"{prefix}.mlp.fc1.weight"][0]
        n = 2 * self.Fp * self.D
        w = self.P[o1:o1 + n].view(2 * self.Fp, self.D)
        g = self.G[o1:o1 + n].view(2 * self.Fp, self.D)
        self.lin[f"{prefix}.mlp.fc13"] = _Lin(f"{prefix}.mlp.fc13", w, g, self.dev, fp8=self.fp8_forward, fp8_bwd=self.fp8_backward)

    # reference key layout <-> engine storage ----------------------------------------------------
    def _view_for_key(self, key: str) -> Optional[torch.Tensor]:
        F, Fp = self.F, self.Fp
        if key.endswith("pos_emb") or (key.endswith(".bias") and "norm" in key):
            return None                                   # buffers: pos tables are rebuilt, LN bias is a zero buffer
        k = key.replace("decoder_embeddings", "DE").replace("encoder_embeddings", "EE")
        if k.startswith("DE") and k.endswith("mod_emb"):
            key = key.replace("decoder_embeddings", "encoder_embeddings")      # shared Parameter
        if key.endswith("to_logits.weight") and self.cfg.share_embedding:
            key = key.replace("to_logits.weight", "token_emb.weight")          # tied
        return self._logical(key, self.p[key])

    def _logical(self, key: str, t: torch.Tensor) -> torch.Tensor:
        """The part of a storage tensor (parameter or gradient) that is the reference's tensor `key`: a view without the
        zero padding.  Unpadded configurations: the tensor itself (fc1 / fc3 / fc2: without the F -> Fp pad).  Padded ones:
        the model-dim axes cut to cfg.dim and the head axis split so that every head loses its pad columns - a strided view
        of as many elements as the reference tensor, in its element order (`.reshape(reference shape)` gives the tensor)."""
        F, Dl, H, Hs, HD, HDP = self.F, self.Dl, self.H, self.Hs, self.HD, self.HDP
        if key.endswith("mlp.fc1.weight") or key.endswith("mlp.fc3.weight"):
            return t[:F, :Dl]
        if key.endswith("mlp.fc2.weight"):
            return t[:Dl, :F]
        if not self.padded:
            return t
        if t.dim() == 1:
            return t[:Dl]
        if key.endswith("qkv.weight"):
            return t.view(3, Hs, HDP, self.D)[:, :H, :HD, :Dl]
        if key.endswith("cross_attn.kv.weight"):
            return t.view(2, Hs, HDP, self.D)[:, :H, :HD, :Dl]
        if key.endswith("cross_attn.q.weight"):
            return t.view(Hs, HDP, self.D)[:H, :HD, :Dl]
        if key.endswith("proj.weight"):                       # attn.proj / self_attn.proj / cross_attn.proj: [D, A]
            return t.view(self.D, Hs, HDP)[:Dl, :H, :HD]
        if key == "decoder_proj_context.weight":
            return t[:Dl, :Dl]
        return t[:, :Dl]                                      # token tables, to_logits: [V, D]

    def _ref_shape(self, key: str) -> Tuple[int, ...]:
        """shape of the reference's tensor `key`"""
        v = self._view_for_key(key)
        if v.dim() == 4:
            return (v.shape[0] * v.shape[1] * v.shape[2], v.shape[3])
        if v.dim() == 3 and key.endswith("cross_attn.q.weight"):
            return (v.shape[0] * v.shape[1], v.shape[2])
        if v.dim() == 3:
            return (v.shape[0], v.shape[1] * v.shape[2])
        return tuple(v.shape)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> List[str]:
        """Copies every known entry into the flat buffer; returns the keys this engine has no storage for (the module's
        `load_state_dict` reports them as unexpected / raises under strict=True)."""
        unexpected = []
        for key, val in sd.items():
            if key.endswith("pos_emb") or (key.endswith(".bias") and "norm" in key):
                continue                                  # buffers: pos tables are rebuilt, LN bias is a zero buffer
            if self._canon_key(key) not in self.p:
                unexpected.append(key)
                continue
            v = self._view_for_key(key)
            v.copy_(val.to(self.dev, F32).reshape(v.shape))        # (padded layouts: v is a strided view without the pads)
        self.weights_dirty = True
        return unexpected

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        cfg, D = self.cfg, self.Dl

        def ref(name):           # the reference's tensor: a view of the storage, or (padded layouts) a copy without the pads
            return self._view_for_key(name).reshape(self._ref_shape(name))

        for m in self.mods:
            e, d = f"encoder_embeddings.{m.name}", f"decoder_embeddings.{m.name}"
            out[f"{e}.mod_emb"] = ref(f"{e}.mod_emb").view(1, 1, D)
            out[f"{e}.token_emb.weight"] = ref(f"{e}.token_emb.weight")
            out[f"{e}.pos_emb"] = self.pos[m.name][None, :, :D]
            out[f"{d}.mod_emb"] = out[f"{e}.mod_emb"]
            out[f"{d}.token_emb.weight"] = ref(f"{d}.token_emb.weight")
            out[f"{d}.to_logits.weight"] = ref(self.logit_key[m.name])
            out[f"{d}.pos_emb"] = self.pos[m.name][None, :, :D]
        zeros = torch.zeros(D, device=self.dev)
        for name in self.p:
            if name.startswith(("encoder.", "decoder.", "encoder_norm", "decoder_norm", "decoder_proj_context")):
                out[name] = ref(name)
                if name.endswith("norm.weight") or ".norm" in name or "norm1" in name or "norm2" in name:
                    out[name[:-len("weight")] + "bias"] = zeros
        out["mask_token"] = ref("mask_token").view(1, 1, D)
        if self.R:
            out["register_tokens"] = ref("register_tokens").view(1, self.R, D)
        return out

    def _canon_key(self, key: str) -> str:
        """reference state-dict key -> engine storage key (shared mod_emb, tied to_logits)."""
        if key.startswith("decoder_embeddings") and key.endswith("mod_emb"):
            key = key.replace("decoder_embeddings", "encoder_embeddings")
        if key.endswith("to_logits.weight") and self.cfg.share_embedding:
            key = key.replace("to_logits.weight", "token_emb.weight")
        return key

    def param_views(self, key: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """(parameter, gradient) of the reference's tensor `key` as VIEWS of the flat buffers - what the nn.Module wrapper
        registers as nn.Parameter / .grad.  They have the reference's shape (mod_emb / mask_token: (1, 1, dim)) except, on a
        padded layout, the attention weights: there a head's rows are not contiguous with the next head's, so the views keep
        the head axis ((3, H, hd, dim) for qkv, (2, H, hd, dim) for kv, (H, hd, dim) for q, (dim, H, hd) for proj); element
        order = the reference's (`.reshape(reference shape)` gives its tensor)."""
        ck = self._canon_key(key)
        out = []
        for store in (self.p, self.g):
            v = self._logical(ck, store[ck])
            if v.dim() <= 2:
                shape = ((1, 1, self.Dl) if (ck.endswith("mod_emb") or ck == "mask_token") else
                         (1, self.R, self.Dl) if ck == "register_tokens" else self._ref_shape(ck))
                v = v.view(shape)
            out.append(v)
        return out[0], out[1]

    def grad_of(self, key: str) -> torch.Tensor:
        key = self._canon_key(key)
        return self._logical(key, self.g[key]).reshape(self._ref_shape(key))

    @torch.no_grad()
    def init_random(self, seed: int = 0):
        """Random init in the reference's distributions (egom2p_model.py:185-222), drawn on the device:
        xavier-uniform linears (qkv / kv as 3 / 2 matrices), N(0, 0.02) encoder tables / mod_emb / mask_token,
        LayerNorm weight 1.  The `to_logits` operand - the tied decoder table when `share_embedding` is set
        (decoder_embeddings.py:447-449: token_emb and to_logits are one tensor and the Linear is initialised last) -
        is xavier-uniform on its [V, D] shape."""
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        self.P.zero_()
        logit_keys = set(self.logit_key.values())
        for name in self.p:
            v = self._view_for_key(name)
            if name in logit_keys:
                a = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
                v.copy_((torch.rand(v.shape, device=self.dev, generator=gen) * 2 - 1) * a)
            elif name.endswith("token_emb.weight") or name.endswith("mod_emb") or name in ("mask_token", "register_tokens"):
                v.copy_(torch.randn(v.shape, device=self.dev, generator=gen) * 0.02)
            elif name.endswith("norm.weight") or ".norm" in name:
                v.fill_(1.0)
            elif name.endswith(".bias"):
                v.zero_()
            else:
                fo, fi = self._ref_shape(name)
                if "qkv" in name: fo //= 3
                elif "kv" in name: fo //= 2
                a = math.sqrt(6.0 / (fo + fi))
                v.copy_((torch.rand(v.shape, device=self.dev, generator=gen) * 2 - 1) * a)
        self.weights_dirty = True

    def param_names(self) -> List[str]:
        return list(self.p.keys())

    def num_params(self) -> int:
        n = 0
        for name, t in self.p.items():
            v = self._view_for_key(name)
            n += v.numel()
        return n

    def refresh_weights(self):
        """fp32 masters -> bf16 W and W^T copies (once per optimiser step; autocast re-casts per forward)."""
        for name, l in self.lin.items():
            ops.cast_weight(l.w, l.wb, l.wt)
            if l.w8 is not None:
                ops.quant_fp8_rows(l.wb, l.w8, l.s8)
            if l.wt8 is not None:
                ops.quant_fp8_rows(l.wt, l.wt8, l.st8)
        self.weights_dirty = False

    # ------------------------------------------------------------------------------------ workspaces
    def _alloc_workspaces(self):
        B, N, M, D, A, Fp, H = self.Bmax, self.Ne, self.M, self.D, self.A, self.Fp, self.Hs      # N: encoder rows per sample (registers included)
        dev, cfg = self.dev, self.cfg
        RN, RM = B * N, B * M

        def e(*shape, dt=BF16):
            return torch.empty(*shape, device=dev, dtype=dt)

        def side(n_keep, n_rows):
            return dict(ids_keep=e(B, n_keep, dt=torch.int64), pad=e(B, n_keep, dt=torch.uint8),
                        mod_mask=e(B, n_keep, dt=torch.int16), slot=e(B, n_keep, dt=I32), local=e(B, n_keep, dt=I32),
                        tok=e(B, n_keep, dt=I32), ks=e(B, n_keep, dt=I32), ke=e(B, n_keep, dt=I32),
                        n_valid=e(B, dt=I32), seg=e(B, self.n_mods, 2, dt=I32), err=torch.zeros(1, device=dev, dtype=I32),
                        seg_bad=torch.zeros(B, device=dev, dtype=I32))

        self.ce, self.cd = side(N, RN), side(M, RM)
        self.zero_b = torch.zeros(B, device=dev, dtype=I32)
        self.emb_e = e(RN, D, dt=F32)
        self.perm = e(RM, dt=I32)
        self.tgt_perm = torch.zeros(RM, device=dev, dtype=I32)
        self.ranges = torch.zeros(self.n_mods, 2, device=dev, dtype=I32)
        self.perm_base = e(B, self.n_mods, dt=I32)
        self.canon = e(self.n_mods, dt=I32)

        def enc_layer():
            return dict(x=e(RN, D, dt=F32), xm=e(RN, D, dt=F32), ln1=e(RN, D), qkv=e(RN, 3 * A), ao=e(RN, A),
                        ao_lo=e(RN, A) if self.attn_o_residual == "all" else None,
                        lse=e(B, H, N, dt=F32), st1=e(2, RN, dt=F32), ln2=e(RN, D), ab=e(RN, 2 * Fp), h=e(RN, Fp),
                        st2=e(2, RN, dt=F32))

        def dec_layer():
            return dict(x=e(RM, D, dt=F32), x1=e(RM, D, dt=F32), x2=e(RM, D, dt=F32), ln1=e(RM, D), qkv=e(RM, 3 * A),
                        ao=e(RM, A), ao_lo=e(RM, A) if self.attn_o_residual == "all" else None, lse=e(B, H, M, dt=F32),
                        st1=e(2, RM, dt=F32), qn=e(RM, D), q=e(RM, A),
                        stq=e(2, RM, dt=F32), cn=e(RN, D), kv=e(RN, 2 * A), stc=e(2, RN, dt=F32), xo=e(RM, A),
                        xo_lo=e(RM, A) if self.attn_o_residual != "none" else None,
                        lse_x=e(B, H, M, dt=F32), ln2=e(RM, D), ab=e(RM, 2 * Fp), h=e(RM, Fp), st2=e(2, RM, dt=F32))

        self.enc = [enc_layer() for _ in range(cfg.encoder_depth)]
        self.dec = [dec_layer() for _ in range(cfg.decoder_depth)]
        self.x_enc_out = e(RN, D, dt=F32)          # residual stream after the last encoder block
        self.st_en = e(2, RN, dt=F32)
        self.xe = e(RN, D)                         # encoder_norm output (bf16)
        self.ctx = e(RN, D, dt=F32)
        self.y_out = e(RM, D, dt=F32)
        self.st_dn = e(2, RM, dt=F32)
        self.yn = torch.zeros(RM, D, device=dev, dtype=BF16)     # decoder_norm output, modality-grouped rows
        vs = sorted({m.vocab_size for m in self.mods})
        self.logits = {v: e(RM, v) for v in vs}
        self.lse_ce = e(RM, dt=F32)
        self.nll = torch.zeros(RM, device=dev, dtype=F32)
        self.loss_out = torch.zeros(1 + self.n_mods, device=dev, dtype=F32)
        self.loss_w = torch.zeros(self.n_mods, device=dev, dtype=F32)      # loss_type 'weighted_mod' / 'token': per-modality weights
        self.loss_ms = torch.ones(self.n_mods, device=dev, dtype=F32)
        # backward temporaries
        R = max(RN, RM)
        self.dres = e(RM, D, dt=F32)               # decoder residual-stream gradient
        self.dres_b = e(RM, D)
        self.ring_b = [e(R, D) for _ in range(4)]   # rotating bf16 copies of the residual-stream gradient (wgrad operands)
        self._ring_i = 0
        self.t_d3 = e(R, D)
        self.dctx = e(RN, D, dt=F32)
        self.dctx_b = e(RN, D)
        self.st_ctx = e(2, RN, dt=F32)              # statistics of the context rows (one LayerNorm input for all decoder layers)
        self.dcn = None                             # per-layer gradients of the context_norm outputs (fused backward): allocated on first use
        self.dxe = e(RN, D, dt=F32)                # encoder residual-stream gradient
        self.dxe_b = e(RN, D)
        self.t_d = e(R, D)                         # generic [rows, D] bf16 temp
        self.t_d2 = e(R, D)
        # [rows, A] temps (gradients of attention outputs / queries): the same memory as t_d / t_d2 when A == D
        self.t_a, self.t_a2 = (self.t_d, self.t_d2) if A == D else (e(R, A), e(R, A))
        self.t_3d = e(R, 3 * A)
        self.t_2d = e(RN, 2 * A)
        self.t_f = e(R, Fp)
        self.t_2f = e(R, 2 * Fp)
        self.dyn = torch.zeros(RM, D, device=dev, dtype=BF16)
        self.delta = e(B, H, max(N, M), dt=F32)
        self.slab = e(64 * 1024 * 1024 // 4, dt=F32)   # 64 MiB of split-K partials (256 workgroups x 256 KiB)
        if self.fp8_forward or self.fp8_backward:      # e4m3 copy + row scales of the current GEMM input
            self.q8 = e(R, max(D, 3 * A, 2 * Fp), dt=torch.uint8)
            self.qs = e(R, dt=F32)
        self.gscale = torch.ones(1, device=dev, dtype=F32)

    # ------------------------------------------------------------------------------------ small helpers
    def _ln(self, x, wname, y, st, out_row=None):
        """LayerNorm forward; with the fp8 forward on, the row also leaves as e4m3 (+ scale) for the GEMM that follows
        (no separate quantisation pass for the LayerNorm-fed linears)."""
        rows = x.shape[0]
        if self.fp8_forward and out_row is None and self.D % 128 == 0 and self.D >= 256:
            q = self._qbuf(rows, self.D)
            ops.layernorm_fwd(x, self.p[wname], y, st[0], st[1], eps=self.cfg.eps, q8=q, qscale=self.qs)
            self._q_of = (y.data_ptr(), rows, self.D)
        else:
            ops.layernorm_fwd(x, self.p[wname], y, st[0], st[1], out_row=out_row, eps=self.cfg.eps, width=self.Dl)

    def _qbuf(self, rows, K):
        if getattr(self, "q8", None) is None or self.q8.shape[0] < rows or self.q8.shape[1] < K:
            self.q8 = torch.empty(rows, max(self.D, 3 * self.A, 2 * self.Fp), device=self.dev, dtype=torch.uint8)
            self.qs = torch.empty(rows, device=self.dev, dtype=F32)
        self._q_of = None
        return self.q8[:, :K]

    def _quant(self, A, rows, K):
        """e4m3 copy of the first `rows` rows of A (row scales in self.qs); the buffers are reused by the next GEMM input"""
        if getattr(self, "_q_of", None) == (A.data_ptr(), rows, K):      # the LayerNorm that made A already left its e4m3 copy
            self._q_of = None
            return self.q8[:, :K]
        q = self._qbuf(rows, K)
        ops.quant_fp8_rows(A, q, self.qs, rows=rows, K=K)
        return q

    def _lin_fwd(self, name, A, C, rows, epi=L.EPI_BF16, R=None, bias=None):
        l = self.lin[name]
        if l.w8 is not None and rows > 0:
            q = self._quant(A, rows, l.in_f)
            ops.gemm_nt_fp8(q, self.qs, l.w8, l.s8, C, rows, l.out_f, l.in_f, epi, R=R, bias=bias)
            return
        ops.gemm_nt(A, l.wb, C, rows, l.out_f, l.in_f, epi, R=R, bias=bias, lda=A.shape[-1], ldb=l.in_f, ldc=C.shape[-1],
                    ldr=None if R is None else R.shape[-1])

    def _mlp_gate_fwd(self, pre, xn, ab, h, rows):
        """ab = fc1||fc3(xn), h = silu(a) * b - one launch where the shape allows it"""
        l = self.lin[f"{pre}.mlp.fc13"]
        if l.w8 is not None and rows > 0:
            q = self._quant(xn, rows, l.in_f)
            ops.gemm_nt_swiglu_fwd_fp8(q, self.qs, l.w8, l.s8, ab, h, rows, self.Fp, l.in_f)
            return
        # (fused launch from 3000 rows on: at 3414 rows - the paired decoder passes of a guided generation step - 29 us against 35 for
        #  GEMM + gate pass; at 1707 rows the two launches win, 22 against 27: profiles/r05_gen_gemm_sweep_after.log)
        if ops.swiglu_fwd_fusable(self.Fp, l.in_f) and rows >= 3000:
            ops.gemm_nt_swiglu_fwd(xn, l.wb, ab, h, rows, self.Fp, l.in_f, ldx=xn.shape[-1], ldw=l.in_f)
        else:
            self._lin_fwd(f"{pre}.mlp.fc13", xn, ab, rows)
            ops.swiglu_fwd(ab, h, rows, self.Fp)

    def _lin_bwd(self, name, dY, X, dX, rows):
        """dX(bf16) = dY @ W ; dW += dY^T @ X."""
        l = self.lin[name]
        if dX is not None and l.wt8 is not None and rows > 0:
            q = self._quant(dY, rows, l.out_f)               # e4m3 rows of dY + per-row scales (one pass over dY)
            ops.gemm_nt_fp8(q, self.qs, l.wt8, l.st8, dX, rows, l.in_f, l.out_f, L.EPI_BF16)
        elif dX is not None:
            ops.gemm_nt(dY, l.wt, dX, rows, l.in_f, l.out_f, L.EPI_BF16, lda=dY.shape[-1], ldb=l.out_f, ldc=dX.shape[-1])
        self._wgrad(l.g, dY, X, l.out_f, l.in_f, rows, ldp=dY.shape[-1], ldq=X.shape[-1])

    def _wgrad(self, G, dY, X, Ni, Nj, rows, ldp, ldq, m_range=None):
        """G[Ni,Nj] += dY^T X on the side stream (operands must stay untouched until the next _join_side)."""
        splits = ops.tn_splits(Ni, Nj, rows, self.slab.numel(), ranged=m_range is not None, ldp=ldp, ldq=ldq)

        def run():
            ops.gemm_tn(dY, X, G, Ni, Nj, rows, m_range=m_range, splits=splits, slab=self.slab if splits > 1 else None,
                        ldp=ldp, ldq=ldq, ldc=Nj)
        if self.pair_stream is not None:
            # paired mode: at most one weight-gradient GEMM is held back, to run beside the next LayerNorm backward (_ln_bwd);
            # an older one goes out now, in launch order
            if self._pend is not None:
                self._pend()
            self._pend = run
        elif self.side is None:
            run()
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                run()

    def _attn(self, q_t, q_off, q_rs, kv_t, k_off, v_off, kv_rs, o_t, lse, ks, ke, r_bs, r_rs, B, Nq, Nk, o_lo=None, seg=None, seg_bad=None):
        A = self.A
        ops.attn_fwd(q_t.data_ptr() + 2 * q_off, Nq * q_rs, q_rs, kv_t.data_ptr() + 2 * k_off, Nk * kv_rs, kv_rs,
                     kv_t.data_ptr() + 2 * v_off, Nk * kv_rs, kv_rs, o_t.data_ptr(), Nq * A, A, lse, ks, ke, r_bs, r_rs,
                     B, self.Hs, Nq, Nk, self.scale, o_lo=None if o_lo is None else o_lo.data_ptr(), hd_pad=self.HDP,
                     seg=seg, seg_bad=seg_bad, hd=self.HD)

    # generation path: under-filled attention grids (1707 decoder rows x 12 heads = 168 workgroups on 256 CUs, each walking every
    # key tile serially: 26 - 47 us per launch) get their keys cut into runs (ego_attn_fwd_d64_split) until ~640 workgroups exist
    SPLIT_TARGET_WGS = 640

    def _kv_splits(self, B, Nq, Nk):
        """Key runs per query tile of a generation-path attention launch (ego_attn_fwd_d64_split), from the measured table
        profiles/r05_gen_attn_split_sweep_before.log (one sample, 12 heads; us per launch unsplit / best split):
          168 workgroups (1707 queries): 1707 keys 28 / - (unsplit wins), 3414 keys 47 / 38 (3 runs), 8534 keys 105 / 67 (3 runs);
          324 (3414 x 3414): 66 / 62 (2 runs);  480, 648 (5120, 6827 rows): unsplit wins (99, 173 us: one full round of 768 slots);
          804 (8534 rows: 36 workgroups more than the 768 slots of one round): 299 / 275 (3 runs even out the tail).
        Round 4 split whenever fewer than 640 workgroups existed (1707 x 1707: 32 us against 28 unsplit)."""
        base = B * self.Hs * ((Nq + 127) // 128)
        if self.HDP != 64 or base <= 0 or not self.attn_split or Nk < 2048:
            return 1
        slots = 768                                           # three 4-wave workgroups per CU
        if base <= 200:
            s = 3
        elif base <= 400:
            s = 2
        elif slots < base <= slots + slots // 2:              # a short second round: cut every tile's keys so the rounds even out
            s = 3
        else:
            s = 1
        return max(1, min(s, ((Nk + 63) // 64) // 4))         # at least four 64-key tiles per run

    def _attn_infer(self, w, q_t, q_off, q_rs, kv_t, k_off, v_off, kv_rs, o_t, ks, ke, B, Nq, Nk, lane=0):
        """`_attn` of the generation passes: one interval per sample, no LSE consumer, split keys when the grid is small.
        lane = 1: a launch that runs BESIDE another attention launch (_fork): its own split scratch and LSE rows"""
        A = self.A
        sp = self._kv_splits(B, Nq, Nk)
        wkey, lkey = ("att_ws", "lse") if lane == 0 else ("att_ws2", "lse2")
        ws = w.get(wkey)
        if sp > 1 and ws is not None and ws.numel() < ops.attn_fwd_split_floats(B, self.Hs, Nq, sp) and not torch.cuda.is_current_stream_capturing():
            ws = w[wkey] = torch.empty(ops.attn_fwd_split_floats(B, self.Hs, Nq, sp) + 4096, device=self.dev, dtype=F32)
        if sp > 1 and ws is not None and ws.numel() >= ops.attn_fwd_split_floats(B, self.Hs, Nq, sp):
            ops.attn_fwd_split(q_t.data_ptr() + 2 * q_off, Nq * q_rs, q_rs, kv_t.data_ptr() + 2 * k_off, Nk * kv_rs, kv_rs,
                               kv_t.data_ptr() + 2 * v_off, Nk * kv_rs, kv_rs, o_t.data_ptr(), Nq * A, A, w[lkey], ks, ke, 1, 0,
                               B, self.Hs, Nq, Nk, self.scale, sp, ws)
        else:
            self._attn(q_t, q_off, q_rs, kv_t, k_off, v_off, kv_rs, o_t, w[lkey], ks, ke, 1, 0, B, Nq, Nk)

    def _dec_groups(self):
        """Row groups of the decoder's block-diagonal self-attention mask for the attention kernels (head dim 64): the
        compaction's (start, count) per modality slot + its per-sample "not the contract's mask" flags.  With them every
        attention workgroup sees one interval; EGOM2P_ATTN_GROUPS=0 keeps the per-row launches."""
        if self.HDP != 64 or not self.attn_groups:
            return {}
        return dict(seg=self.cd["seg"][:self.B], seg_bad=self.cd["seg_bad"][:self.B])

    def _attn_bwd(self, q_t, q_off, q_rs, kv_t, k_off, v_off, kv_rs, o_t, do_t, lse, dq_t, dkv_t, ks, ke, r_bs, r_rs, B, Nq, Nk,
                  o_lo=None, seg=None, seg_bad=None):
        A = self.A
        ops.attn_bwd(q_t.data_ptr() + 2 * q_off, Nq * q_rs, q_rs, kv_t.data_ptr() + 2 * k_off, Nk * kv_rs, kv_rs,
                     kv_t.data_ptr() + 2 * v_off, Nk * kv_rs, kv_rs, o_t.data_ptr(), Nq * A, A, do_t.data_ptr(), Nq * A, A,
                     lse, self.delta, dq_t.data_ptr() + 2 * q_off, Nq * q_rs, q_rs, dkv_t.data_ptr() + 2 * k_off, Nk * kv_rs,
                     kv_rs, dkv_t.data_ptr() + 2 * v_off, Nk * kv_rs, kv_rs, ks, ke, r_bs, r_rs, B, self.Hs, Nq, Nk, self.scale,
                     o_lo=None if o_lo is None else o_lo.data_ptr(), hd_pad=self.HDP, seg=seg, seg_bad=seg_bad, hd=self.HD)

    # ------------------------------------------------------------------------------------ forward
    def forward(self, mod_dict: Dict[str, Dict[str, torch.Tensor]], dec_order: Optional[Sequence[str]] = None,
                need_loss: bool = True, group_rows: bool = True, loss_grad=None, loss_type: str = "mod"):
        """Forward of one micro-batch (tensors already on the device).  Returns (loss, {mod: loss}) as
        views of a device buffer: reading them is the only host sync.
        loss_grad (float or 1-element device tensor, optional): the upstream d loss the following `backward` will be
        called with.  A training step knows it when the loss is formed, and the cross-entropy then runs forward and
        backward in ONE pass over the logits (`ego_ce_fwd_bwd`: bitwise the two-call result, the 16.5 GB of logits of a
        64-clip micro-batch are read once instead of twice); `backward` must then be given the same value.
        loss_type: 'mod' (forward_mod_loss, egom2p_model.py:614-644), 'weighted_mod' (:583-612) or 'token' (:646-681); the two
        others weight the modalities' mean cross-entropies by device-side weights (`ego_loss_weights`), forward and backward."""
        if loss_type not in ops.LOSS_MODES:
            raise ValueError("Invalid loss type")                  # the reference's message (egom2p_model.py:579)
        self._loss_mode = ops.LOSS_MODES[loss_type]
        if self.weights_dirty:
            self.refresh_weights()
        cfg, D, A, Fp, N, M = self.cfg, self.D, self.A, self.Fp, self.Ne, self.M      # N: encoder rows per sample (registers included)
        mods = self.mods
        B = mod_dict[mods[0].name]["input_mask"].shape[0]
        if B > self.Bmax:
            raise L.EgoHipError(f"batch {B} exceeds the engine's workspace batch {self.Bmax}")
        self.B = B
        RN, RM = B * N, B * M
        byname = {m.name: m for m in mods}
        dmods = [byname[n] for n in (dec_order or [m.name for m in mods])]
        self.dmods = dmods

        def flat_ids(m):
            t = mod_dict[m.name]["tensor"]
            return t.reshape(B, -1).contiguous()

        # ---- compaction + embeddings (egom2p_model.py:706-718, 723)
        ce, cd = self.ce, self.cd
        ops.compact([mod_dict[m.name]["input_mask"] for m in mods], [flat_ids(m) for m in mods], None,
                    [m.max_tokens for m in mods], [m.id for m in mods], self.N, False, ce, B, n_reg=self.R)
        ops.compact([mod_dict[m.name]["target_mask"] for m in dmods], [flat_ids(m) for m in dmods],
                    [mod_dict[m.name]["decoder_attention_mask"] for m in dmods],
                    [m.max_tokens for m in dmods], [m.id for m in dmods], M, True, cd, B)
        x0 = self.enc[0]["x"] if cfg.encoder_depth else self.x_enc_out
        ops.embed_fwd([self.p[f"encoder_embeddings.{m.name}.token_emb.weight"] for m in mods],
                      [self.pos[m.name] for m in mods], [self.p[f"encoder_embeddings.{m.name}.mod_emb"] for m in mods],
                      None, ce["slot"], ce["local"], ce["tok"], x0, self.emb_e, RN, D, reg=self.p["register_tokens"] if self.R else None)
        y0 = self.dec[0]["x"] if cfg.decoder_depth else self.y_out
        ops.embed_fwd(None, [self.pos[m.name] for m in dmods], [self.p[f"encoder_embeddings.{m.name}.mod_emb"] for m in dmods],
                      self.p["mask_token"], cd["slot"], cd["local"], cd["tok"], y0, None, RM, D)
        ckey = tuple(m.name for m in dmods)
        cache = self.__dict__.setdefault("_canon_dev", {})
        if ckey not in cache:                        # one small H2D copy per decoder order, ever (never inside a graph capture)
            cache[ckey] = torch.tensor([mods.index(m) for m in dmods], dtype=I32).to(self.dev)
        self.canon.copy_(cache[ckey])
        ops.loss_perm(cd["seg"], self.canon, cd["slot"], cd["tok"], B, M, self.n_mods, self.perm, self.tgt_perm,
                      self.ranges, self.perm_base)

        # ---- encoder (egom2p_model.py:496-499; Block.forward egom2p_utils.py:356-359)
        for i, w in enumerate(self.enc):
            pre = f"encoder.{i}"
            nxt = self.enc[i + 1]["x"] if i + 1 < cfg.encoder_depth else self.x_enc_out
            self._ln(w["x"][:RN], f"{pre}.norm1.weight", w["ln1"], w["st1"])
            self._lin_fwd(f"{pre}.attn.qkv.weight", w["ln1"], w["qkv"], RN)
            # key-padding mask = one interval [0, n_valid) per SAMPLE (the per-row copies ce["ks"/"ke"] hold the same
            # numbers): the uniform form lets the attention kernels walk one (batch, head) pair per XCD (L2-resident K / V)
            self._attn(w["qkv"], 0, 3 * A, w["qkv"], A, 2 * A, 3 * A, w["ao"], w["lse"], self.zero_b, ce["n_valid"], 1, 0, B, N, N,
                       o_lo=w["ao_lo"])
            self._lin_fwd(f"{pre}.attn.proj.weight", w["ao"], w["xm"], RN, L.EPI_RESID, R=w["x"])
            self._ln(w["xm"][:RN], f"{pre}.norm2.weight", w["ln2"], w["st2"])
            self._mlp_gate_fwd(pre, w["ln2"], w["ab"], w["h"], RN)
            self._lin_fwd(f"{pre}.mlp.fc2.weight", w["h"], nxt, RN, L.EPI_RESID, R=w["xm"])
        self._ln(self.x_enc_out[:RN], "encoder_norm.weight", self.xe, self.st_en)
        # context = decoder_proj_context(x) + encoder_emb   (egom2p_model.py:722)
        self._lin_fwd("decoder_proj_context.weight", self.xe, self.ctx, RN, L.EPI_BIAS_RESID, R=self.emb_e,
                      bias=self.p["decoder_proj_context.bias"])

        # ---- decoder (egom2p_model.py:520-523; DecoderBlock.forward egom2p_utils.py:387-391)
        if self.ctx_ln_fused:
            ops.layernorm_fwd_multi(self.ctx[:RN], [self.p[f"decoder.{i}.context_norm.weight"] for i in range(cfg.decoder_depth)],
                                    [w["cn"] for w in self.dec], self.st_ctx[0], self.st_ctx[1], eps=cfg.eps, width=self.Dl)
        for i, w in enumerate(self.dec):
            pre = f"decoder.{i}"
            nxt = self.dec[i + 1]["x"] if i + 1 < cfg.decoder_depth else self.y_out
            self._ln(w["x"][:RM], f"{pre}.norm1.weight", w["ln1"], w["st1"])
            self._lin_fwd(f"{pre}.self_attn.qkv.weight", w["ln1"], w["qkv"], RM)
            self._attn(w["qkv"], 0, 3 * A, w["qkv"], A, 2 * A, 3 * A, w["ao"], w["lse"], cd["ks"], cd["ke"], M, 1, B, M, M,
                       o_lo=w["ao_lo"], **self._dec_groups())
            self._lin_fwd(f"{pre}.self_attn.proj.weight", w["ao"], w["x1"], RM, L.EPI_RESID, R=w["x"])
            self._ln(w["x1"][:RM], f"{pre}.query_norm.weight", w["qn"], w["stq"])
            self._lin_fwd(f"{pre}.cross_attn.q.weight", w["qn"], w["q"], RM)
            if not self.ctx_ln_fused:
                self._ln(self.ctx[:RN], f"{pre}.context_norm.weight", w["cn"], w["stc"])
            self._lin_fwd(f"{pre}.cross_attn.kv.weight", w["cn"], w["kv"], RN)
            self._attn(w["q"], 0, A, w["kv"], 0, A, 2 * A, w["xo"], w["lse_x"], self.zero_b, ce["n_valid"], 1, 0, B, M, N,
                       o_lo=w["xo_lo"])
            self._lin_fwd(f"{pre}.cross_attn.proj.weight", w["xo"], w["x2"], RM, L.EPI_RESID, R=w["x1"])
            self._ln(w["x2"][:RM], f"{pre}.norm2.weight", w["ln2"], w["st2"])
            self._mlp_gate_fwd(pre, w["ln2"], w["ab"], w["h"], RM)
            self._lin_fwd(f"{pre}.mlp.fc2.weight", w["h"], nxt, RM, L.EPI_RESID, R=w["x2"])
        # decoder_norm, rows written modality-grouped (the row order of y[decoder_mod_mask == id], :633)
        ops.layernorm_fwd(self.y_out[:RM], self.p["decoder_norm.weight"], self.yn, self.st_dn[0], self.st_dn[1],
                          out_row=self.perm if group_rows else None, eps=cfg.eps, width=self.Dl)
        self._have_fwd = True
        if not need_loss:
            return None

        # ---- per-modality logits + CE (forward_mod_loss, egom2p_model.py:614-644)
        self._ce_done, self._ce_grad = set(), loss_grad
        if loss_grad is not None:
            self._set_gscale(loss_grad)
        lw = ms = None
        if self._loss_mode:
            lw, ms = self.loss_w, self.loss_ms
            ops.loss_weights(self.ranges, [m.vocab_size for m in mods], self._loss_mode, lw, ms)
        for c, m in enumerate(mods):
            l = self.lin[self.logit_key[m.name]]
            ub = min(RM, B * m.max_tokens)
            lg = self.logits[m.vocab_size]
            ops.gemm_nt(self.yn, l.wb, lg, ub, m.vocab_size, D, L.EPI_BF16, m_range=self.ranges[c], lda=D, ldb=D, ldc=m.vocab_size)
            if loss_grad is not None and ops.ce_fusable(m.vocab_size):
                ops.ce_fwd_bwd(lg, m.vocab_size, m.vocab_size, self.tgt_perm, self.ranges[c], ub, self.lse_ce, self.nll, self.gscale,
                               self.n_mods, loss_w=None if lw is None else lw[c:c + 1])
                self._ce_done.add(c)
            else:
                ops.ce_fwd(lg, m.vocab_size, m.vocab_size, self.tgt_perm, self.ranges[c], ub, self.lse_ce, self.nll)
        # a decoder_attention_mask that is not one interval per row (compaction flag) turns the loss into NaN
        ops.loss_finalize(self.nll, self.ranges, self.n_mods, self.loss_out, err=cd["err"], loss_w=lw, mod_scale=ms)
        return self.loss_out[0], {m.name: self.loss_out[1 + c] for c, m in enumerate(mods)}

    # ------------------------------------------------------------------------------------ backward
    def _mlp_bwd(self, pre, w, dres, dres_b, rows, xin):
        """residual-stream gradient through x + fc2(swiglu(fc13(LN2(x)))); xin = LN2 input (saved)."""
        Fp = self.Fp
        dh, dab, dln = self.t_f, self.t_2f, self.t_d
        l2 = self.lin[f"{pre}.mlp.fc2.weight"]
        if ops.swiglu_bwd_fusable(Fp, l2.out_f) and rows >= 4096:
            # fc2 dgrad and the gate backward in one launch (dh never reaches HBM), then the fc2 wgrad
            ops.gemm_nt_swiglu_bwd(dres_b, l2.wt, w["ab"], dab, rows, Fp, l2.out_f, ldy=dres_b.shape[-1], ldw=l2.out_f)
            self._wgrad(l2.g, dres_b, w["h"], l2.out_f, l2.in_f, rows, ldp=dres_b.shape[-1], ldq=w["h"].shape[-1])
        else:
            self._lin_bwd(f"{pre}.mlp.fc2.weight", dres_b, w["h"], dh, rows)
            ops.swiglu_bwd(w["ab"], dh, dab, rows, Fp)
        self._lin_bwd(f"{pre}.mlp.fc13", dab, w["ln2"], dln, rows)
        nb = self._ring_next()
        self._ln_bwd(dln, xin[:rows], w["st2"][0], w["st2"][1], self.p[f"{pre}.norm2.weight"], dres, self.g[f"{pre}.norm2.weight"],
                          dx_in=dres, dx_bf16=nb, width=self.Dl)
        return nb

    def _self_attn_bwd(self, pre, attn_name, w, dres, dres_b, rows, Nq, ks, ke, r_bs=None, r_rs=1, groups=None):
        A, B = self.A, self.B
        r_bs = Nq if r_bs is None else r_bs
        dao, dqkv, dln = self.t_a, self.t_3d, self.t_d2
        self._lin_bwd(f"{pre}.{attn_name}.proj.weight", dres_b, w["ao"], dao, rows)
        self._attn_bwd(w["qkv"], 0, 3 * A, w["qkv"], A, 2 * A, 3 * A, w["ao"], dao, w["lse"], dqkv, dqkv, ks, ke, r_bs, r_rs, B, Nq, Nq,
                       o_lo=w["ao_lo"], **(groups or {}))
        self._lin_bwd(f"{pre}.{attn_name}.qkv.weight", dqkv, w["ln1"], dln, rows)
        nb = self._ring_next()
        self._ln_bwd(dln, w["x"][:rows], w["st1"][0], w["st1"][1], self.p[f"{pre}.norm1.weight"], dres, self.g[f"{pre}.norm1.weight"],
                          dx_in=dres, dx_bf16=nb, width=self.Dl)
        return nb

    def _set_gscale(self, gscale):
        if isinstance(gscale, torch.Tensor):
            self.gscale.copy_(gscale.reshape(1).to(F32))
        else:
            self.gscale.fill_(float(gscale))

    def backward(self, gscale=1.0, bucket_done: Optional[Callable[[str, int, int], None]] = None):
        """Backward of the last forward; gradients are ACCUMULATED into the flat grad buffer scaled by
        `gscale` (float or 1-element device tensor: the upstream d loss).  `bucket_done(name, lo, hi)` is
        called (in launch order) as soon as every kernel writing G[lo:hi] has been enqueued."""
        assert self._have_fwd, "backward() needs a forward()"
        cfg, D, A, N, M, B = self.cfg, self.D, self.A, self.Ne, self.M, self.B       # N: encoder rows per sample (registers included)
        RN, RM = B * N, B * M
        mods, ce, cd = self.mods, self.ce, self.cd
        if getattr(self, "_ce_done", None):
            # the forward already turned the logits into d logits with the upstream gradient it was promised
            same = (gscale is self._ce_grad) or (not isinstance(gscale, torch.Tensor) and not isinstance(self._ce_grad, torch.Tensor)
                                                 and float(gscale) == float(self._ce_grad))
            if not same:
                raise L.EgoHipError("backward(gscale) differs from the loss_grad the forward was given")
        else:
            self._set_gscale(gscale)
        bmap = {n: (lo, hi) for n, lo, hi in self.buckets}

        def done(name):
            self._join_side()          # the bucket's weight gradients come from the side stream
            if bucket_done is not None:
                lo, hi = bmap[name]
                bucket_done(name, lo, hi)

        # ---- loss head: d logits in place, d yn, d table (tied to_logits/token_emb)
        for c, m in enumerate(mods):
            l = self.lin[self.logit_key[m.name]]
            ub = min(RM, B * m.max_tokens)
            V = m.vocab_size
            lg = self.logits[V]
            if c not in self._ce_done:
                ops.ce_bwd(lg, V, V, self.tgt_perm, self.ranges[c], ub, self.lse_ce, self.gscale, self.n_mods,
                           loss_w=self.loss_w[c:c + 1] if getattr(self, "_loss_mode", 0) else None)
            ops.gemm_nt(lg, l.wt, self.dyn, ub, D, V, L.EPI_BF16, m_range=self.ranges[c], lda=V, ldb=V, ldc=D)
            self._wgrad(l.g, lg, self.yn, V, D, ub, ldp=V, ldq=D, m_range=self.ranges[c])
        for m in reversed(mods):
            done(f"dec_table.{m.name}" if cfg.share_embedding else f"to_logits.{m.name}")
        dres, dres_b = self.dres, self._ring_next()
        ops.layernorm_bwd(self.dyn, self.y_out[:RM], self.st_dn[0], self.st_dn[1], self.p["decoder_norm.weight"], dres,
                          self.g["decoder_norm.weight"], dx_in=None, dx_bf16=dres_b, dy_row=self.perm, width=self.Dl)

        # ---- decoder layers
        first_ctx = True
        fused = self.ctx_ln_fused
        if fused and self.dcn is None:
            self.dcn = [torch.empty(self.Bmax * N, D, device=self.dev, dtype=BF16) for _ in range(cfg.decoder_depth)]
        for i in reversed(range(cfg.decoder_depth)):
            w, pre = self.dec[i], f"decoder.{i}"
            dres_b = self._mlp_bwd(pre, w, dres, dres_b, RM, w["x2"])
            # cross attention
            dxo, dq, dkv, dln = self.t_a, self.t_a2, self.t_2d, self.t_d
            self._lin_bwd(f"{pre}.cross_attn.proj.weight", dres_b, w["xo"], dxo, RM)
            self._attn_bwd(w["q"], 0, A, w["kv"], 0, A, 2 * A, w["xo"], dxo, w["lse_x"], dq, dkv, self.zero_b, ce["n_valid"],
                           1, 0, B, M, N, o_lo=w["xo_lo"])
            self._lin_bwd(f"{pre}.cross_attn.q.weight", dq, w["qn"], dln, RM)
            nb = self._ring_next()
            self._ln_bwd(dln, w["x1"][:RM], w["stq"][0], w["stq"][1], self.p[f"{pre}.query_norm.weight"], dres,
                              self.g[f"{pre}.query_norm.weight"], dx_in=dres, dx_bf16=nb, width=self.Dl)
            dres_b = nb
            dcn = self.dcn[i] if fused else self.t_d3    # not t_d2: dq is still being read by the q-projection wgrad on the side stream
            self._lin_bwd(f"{pre}.cross_attn.kv.weight", dkv, w["cn"], dcn, RN)
            if not fused:
                self._ln_bwd(dcn, self.ctx[:RN], w["stc"][0], w["stc"][1], self.p[f"{pre}.context_norm.weight"], self.dctx,
                                  self.g[f"{pre}.context_norm.weight"], dx_in=None if first_ctx else self.dctx, width=self.Dl)
            first_ctx = False
            dres_b = self._self_attn_bwd(pre, "self_attn", w, dres, dres_b, RM, M, cd["ks"], cd["ke"], groups=self._dec_groups())
            done(pre)                 # (the layer's context_norm weight is not in this bucket: "ctx_norms")
        if cfg.decoder_depth == 0:
            self.dctx[:RN].zero_()
        elif fused:
            # all layers' context_norm backward in one launch: the context and its statistics are read once, the context gradient
            # is written once (fp32 + the bf16 copy the context projection's backward reads) - summed in the chained launches'
            # order, bit for bit their result; the layers' own gradient buckets went out layer by layer, only the small
            # "ctx_norms" bucket (one weight vector per layer) is completed here
            Ld = cfg.decoder_depth
            ops.layernorm_bwd_multi([self.dcn[i][:RN] for i in range(Ld)], self.ctx[:RN], self.st_ctx[0], self.st_ctx[1],
                                    [self.p[f"decoder.{i}.context_norm.weight"] for i in range(Ld)], self.dctx[:RN],
                                    [self.g[f"decoder.{i}.context_norm.weight"] for i in range(Ld)], dx_bf16=self.dctx_b[:RN], width=self.Dl)
        if cfg.decoder_depth:
            done("ctx_norms")
        # decoder input embeddings: mask token + (pos + mod_emb)
        dmods = self.dmods
        ops.embed_bwd(None, [self.g[f"encoder_embeddings.{m.name}.mod_emb"] for m in dmods], self.g["mask_token"],
                      dres, None, cd["slot"], cd["tok"], RM, D)
        done("head")

        # ---- context projection + encoder_norm
        if not (fused and cfg.decoder_depth > 0):
            ops.cast_f32_bf16(self.dctx[:RN], self.dctx_b)
        ops.bias_grad(self.dctx_b, RN, D, self.g["decoder_proj_context.bias"])
        dxe_n = self.t_d
        self._lin_bwd("decoder_proj_context.weight", self.dctx_b, self.xe, dxe_n, RN)
        dxe, dxe_b = self.dxe, self._ring_next()
        self._ln_bwd(dxe_n, self.x_enc_out[:RN], self.st_en[0], self.st_en[1], self.p["encoder_norm.weight"], dxe,
                          self.g["encoder_norm.weight"], dx_in=None, dx_bf16=dxe_b, width=self.Dl)
        done("bridge")

        # ---- encoder layers
        for i in reversed(range(cfg.encoder_depth)):
            w, pre = self.enc[i], f"encoder.{i}"
            dxe_b = self._mlp_bwd(pre, w, dxe, dxe_b, RN, w["xm"])
            dxe_b = self._self_attn_bwd(pre, "attn", w, dxe, dxe_b, RN, N, self.zero_b, ce["n_valid"], r_bs=1, r_rs=0)
            done(pre)
        # encoder input embeddings: token rows, mod_emb (emb is used twice: x = tok + emb and context += emb)
        ops.embed_bwd([self.g[f"encoder_embeddings.{m.name}.token_emb.weight"] for m in mods],
                      [self.g[f"encoder_embeddings.{m.name}.mod_emb"] for m in mods], None, dxe, self.dctx,
                      ce["slot"], ce["tok"], RN, D, touched=getattr(self, "touched", None))
        if self.R:        # d register_tokens = the encoder input gradient of the register rows, summed over the batch (emb is 0 there)
            ops.reg_grad(dxe, B, N, self.R, D, self.g["register_tokens"])
        done("mod_emb")
        for m in reversed(mods):
            done(f"enc_table.{m.name}")
        self._join_side()
        self._have_fwd = False
        self._ce_done = set()

    def zero_grad(self):
        self.G.zero_()

    def track_touched_table_rows(self, on: bool = True):
        """Let the embedding backward flag every encoder-table row that receives a gradient (uint8 [V] per table,
        accumulated over micro-batches, consumed and cleared by dp.SparseTableExchange).  Returns
        [(gradient view [V, D], flags)] for the encoder tables, in modality order."""
        if not on:
            self.touched = None
            return []
        self.touched = [torch.zeros(m.vocab_size, device=self.dev, dtype=torch.uint8) for m in self.mods]
        return [(self.g[f"encoder_embeddings.{m.name}.token_emb.weight"], self.touched[i]) for i, m in enumerate(self.mods)]

    def resize_workspaces(self, max_batch: int, n_enc: int, n_dec: int):
        """Re-allocate the activation workspaces for another (batch, N, M); parameters are untouched."""
        self.Bmax, self.N, self.M = max_batch, n_enc, n_dec
        self.Ne = self.N + self.R
        self._alloc_workspaces()
        self._have_fwd = False
        self.drop_graphs()

    def drop_graphs(self):
        """Forget every captured generation graph (and the workspace it owns).  A hipGraph bakes device addresses in: it
        is only valid while the buffers it was captured against are alive, so anything that re-allocates engine memory
        calls this.  (Round 1 hit exactly that: a graph captured against a shared inference workspace replayed after a
        larger pass had re-allocated it -> GPU memory access fault; since then each graph owns its workspace.)"""
        self.__dict__.pop("_graphs", None)
        self._iw, self._infer_key = None, None

    # ------------------------------------------------------------------------------------ generation (config 4)
    def _alloc_infer(self, B: int, Nmax: int, Mmax: int, fresh: bool = False, groups: int = 1):
        """One set of buffers (nothing is saved for a backward) for encoder-decoder passes of the ROAR / CFG
        generation path, sized for up to Nmax encoder rows and Mmax decoder rows per sample.  groups = 2: the conditional
        and the unconditional pass of one guided step share ONE decoder pass (`infer_logits_cfg`) - decoder buffers hold
        2 B samples, context buffers the two contexts one after the other."""
        G = int(groups)
        key = (G, B, Nmax, Mmax)
        if not fresh and self._infer_key is not None and self._infer_key[0] >= G and all(a >= b for a, b in zip(self._infer_key[1:], key[1:])):
            return self._iw
        D, A, Fp, H, dev = self.D, self.A, self.Fp, self.Hs, self.dev
        RE, RD, RC = G * B * Nmax, G * B * Mmax, G * B * Nmax     # encoder rows (all groups, back to back), decoder rows, context rows
        R = max(RE, RD)

        def e(*shape, dt=BF16):
            return torch.empty(*shape, device=dev, dtype=dt)

        def side():
            return dict(ids_keep=e(B, Nmax, dt=torch.int64), pad=e(B, Nmax, dt=torch.uint8), mod_mask=e(B, Nmax, dt=torch.int16),
                        slot=e(B, Nmax, dt=I32), local=e(B, Nmax, dt=I32), tok=e(B, Nmax, dt=I32), ks=e(B, Nmax, dt=I32),
                        ke=e(B, Nmax, dt=I32), n_valid=torch.zeros(B, device=dev, dtype=I32), seg=e(B, self.n_mods, 2, dt=I32),
                        err=torch.zeros(1, device=dev, dtype=I32))

        w = dict(
            groups=G, sides=[side() for _ in range(G)],
            xa=e(RE, D, dt=F32), xb=e(RE, D, dt=F32), emb=e(RE, D, dt=F32), ctx=e(RC, D, dt=F32),
            ya=e(RD, D, dt=F32), yb=e(RD, D, dt=F32),
            ln=e(max(R, RC), D), qkv=e(R, 3 * A), ao=e(R, A), ab=e(R, 2 * Fp), h=e(R, Fp), q=e(RD, A), cn=e(RC, D),
            kv=e(RC, 2 * A), st=e(2, max(R, RC), dt=F32), lse=e(G * B, H, max(Nmax, Mmax), dt=F32),
            zero_b=torch.zeros(G * B, device=dev, dtype=I32), full_m=torch.zeros(G * B, device=dev, dtype=I32),
            dslot=torch.zeros(RD, device=dev, dtype=I32), dtok=torch.zeros(RD, device=dev, dtype=I32),
        )
        w["side"] = w["sides"][0]
        # every decoder layer's context_norm output from one pass over the context (ego_layernorm_fwd_multi)
        w["cns"] = [e(RC, D) for _ in range(self.cfg.decoder_depth)] if self.ctx_ln_fused else None
        # scratch of the split-key attention launches: splits x B x H x ceil(Nq / 128) <= SPLIT_TARGET_WGS bounds it
        # (sized for the largest split launch this workspace can see: an encoder group's self-attention, the decoder's self-attention
        #  over all groups, one group's cross-attention - whatever _kv_splits would split; shorter passes through the same buffers
        #  split less or not at all)
        need = 0
        if self.HDP == 64 and self.attn_split:
            for b_, nq, nk in ((B, Nmax, Nmax), (G * B, Mmax, Mmax), (B, Mmax, Nmax)):
                for nq_ in {nq, max(1, nq // 2), max(1, nq // 4)}:          # (the split rule is not monotonic in the row count)
                    sp_ = max(self._kv_splits(b_, nq_, nk), self._kv_splits(b_, nq_, max(1, nk // 2)))
                    if sp_ > 1:
                        need = max(need, int(ops.attn_fwd_split_floats(b_, H, nq_, sp_)))
        w["att_ws"] = e(need + 4096, dt=F32) if (self.HDP == 64 and self.attn_split) else None
        # a second attention launch beside the first (_fork): its own scratch and LSE rows; one kv buffer per decoder layer when the
        # kv projections run as a side chain
        w["att_ws2"] = (e(need + 4096, dt=F32) if (self.HDP == 64 and self.attn_split and G > 1 and self.gen_overlap) else None)
        w["lse2"] = e(G * B, H, max(Nmax, Mmax), dt=F32) if (G > 1 and self.gen_overlap) else None
        w["st2"] = e(2, RC, dt=F32) if self.gen_overlap else None
        w["kvs"] = ([e(RC, 2 * A) for _ in range(self.cfg.decoder_depth)] if (self.gen_overlap and self.ctx_ln_fused) else None)
        if not fresh:
            self._iw, self._infer_key = w, key
        return w

    def _infer_context(self, w, parts, B: int):
        """Encoder half of a generation pass: compaction of the unmasked inputs, embeddings, encoder blocks,
        decoder_proj_context (+ embeddings) -> w["ctx"] fp32 (forward_mask_encoder_generation + encoder,
        egom2p/models/generate.py:747-757).  parts = [(side, enc_inputs, N, first row)]: one entry per group of B samples
        (the conditional and the unconditional inputs of a guided step); the groups' rows lie back to back, every row-wise
        launch (LayerNorm, GEMMs, gate) covers all of them - the unconditional pass's 1707 / 3414 rows alone leave most CUs
        idle - compaction, embedding and self-attention run per group (own row counts)."""
        cfg, D, A = self.cfg, self.D, self.A
        parts = [p for p in parts if p[2] > 0]
        R = sum(B * n for _, _, n, _ in parts)
        x, xn = w["xa"], w["xb"]
        for side, enc_inputs, N, r0 in parts:          # N = rows per sample, the engine's register tokens included
            mods = [m for m in self.mods if m.name in enc_inputs]
            ops.compact([enc_inputs[m.name][1].contiguous() for m in mods], [enc_inputs[m.name][0].reshape(B, -1).contiguous() for m in mods],
                        None, [m.max_tokens for m in mods], [m.id for m in mods], N - self.R, False, side, B, n_reg=self.R)
            ops.embed_fwd([self.p[f"encoder_embeddings.{m.name}.token_emb.weight"] for m in mods], [self.pos[m.name] for m in mods],
                          [self.p[f"encoder_embeddings.{m.name}.mod_emb"] for m in mods], None, side["slot"], side["local"], side["tok"],
                          x[r0:], w["emb"][r0:], B * N, D, reg=self.p["register_tokens"] if self.R else None)
        for i in range(cfg.encoder_depth):
            pre = f"encoder.{i}"
            self._ln(x[:R], f"{pre}.norm1.weight", w["ln"], w["st"])
            self._lin_fwd(f"{pre}.attn.qkv.weight", w["ln"], w["qkv"], R)
            def enc_attn(gi):
                side, _, N, r0 = parts[gi]
                self._attn_infer(w, w["qkv"], r0 * 3 * A, 3 * A, w["qkv"], r0 * 3 * A + A, r0 * 3 * A + 2 * A, 3 * A, w["ao"][r0:], w["zero_b"],
                                 side["n_valid"], B, N, N, lane=1 if (gi == 1 and w.get("lse2") is not None) else 0)
            # the second group's self-attention (the unconditional inputs: 324 workgroups at 3414 rows) goes out first, on the second
            # stream, and runs beside the first group's launch
            forked = self._fork(lambda: enc_attn(1)) if (len(parts) == 2 and w.get("lse2") is not None) else None
            enc_attn(0)
            if len(parts) == 2 and w.get("lse2") is None:
                enc_attn(1)
            self._join(forked)
            self._lin_fwd(f"{pre}.attn.proj.weight", w["ao"], xn, R, L.EPI_RESID, R=x)
            self._ln(xn[:R], f"{pre}.norm2.weight", w["ln"], w["st"])
            self._mlp_gate_fwd(pre, w["ln"], w["ab"], w["h"], R)
            self._lin_fwd(f"{pre}.mlp.fc2.weight", w["h"], x, R, L.EPI_RESID, R=xn)
        self._ln(x[:R], "encoder_norm.weight", w["ln"], w["st"])
        self._lin_fwd("decoder_proj_context.weight", w["ln"], w["ctx"], R, L.EPI_BIAS_RESID, R=w["emb"],
                      bias=self.p["decoder_proj_context.bias"])

    def _infer_decode(self, w, parts, target: str, dec_pos: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
        """Decoder half of a generation pass over len(parts) groups of B samples that decode the SAME positions against
        different contexts (parts[g] = (side, N_g, first context row); groups with an empty context last).  Everything
        but the cross-attention itself is one launch for all groups.  Returns bf16 logits [groups * B * M, V]."""
        cfg, D, A = self.cfg, self.D, self.A
        tm = {m.name: m for m in self.mods}[target]
        B, M = dec_pos.shape
        G = len(parts)
        RM, RD = B * M, G * B * M
        RC = sum(B * n for _, n, _ in parts)
        n_ctx = sum(1 for _, n, _ in parts if n > 0)
        if any(n > 0 for _, n, _ in parts[n_ctx:]):
            raise ValueError("groups with an empty context come last")
        RQ = n_ctx * RM                                     # decoder rows that have a context to attend to
        # decoder rows: mask token + positional + modality embedding of the selected target positions (:481-516)
        y, yn = w["ya"], w["yb"]
        local = dec_pos.to(self.dev, I32)
        local = (local.unsqueeze(0).expand(G, B, M) if G > 1 else local).contiguous()
        ops.embed_fwd(None, [self.pos[tm.name]], [self.p[f"encoder_embeddings.{tm.name}.mod_emb"]], self.p["mask_token"],
                      w["dslot"], local, w["dtok"], y, None, RD, D)
        w["full_m"].fill_(M)
        fused = self.ctx_ln_fused and w.get("cns") is not None and RC > 0
        kv_side, kv_evs = None, None
        if fused and w.get("kvs") is not None and RQ > 0:
            # side chain: every layer's context LayerNorm output (one fused launch) and kv projection depend on the context only -
            # they run on the second stream beside the decoder's self-attention blocks; layer i's cross-attention waits for kv i
            kv_evs = [torch.cuda.Event() for _ in range(cfg.decoder_depth)]

            def chain():
                ops.layernorm_fwd_multi(w["ctx"][:RC], [self.p[f"decoder.{i}.context_norm.weight"] for i in range(cfg.decoder_depth)],
                                        w["cns"], w["st2"][0], w["st2"][1], eps=cfg.eps, width=self.Dl)
                for i in range(cfg.decoder_depth):
                    self._lin_fwd(f"decoder.{i}.cross_attn.kv.weight", w["cns"][i], w["kvs"][i], RC)
                    kv_evs[i].record(torch.cuda.current_stream())
            kv_side = self._fork(chain, which=1)
            if kv_side is None:
                kv_evs = None
        elif fused:
            ops.layernorm_fwd_multi(w["ctx"][:RC], [self.p[f"decoder.{i}.context_norm.weight"] for i in range(cfg.decoder_depth)],
                                    w["cns"], w["st"][0], w["st"][1], eps=cfg.eps, width=self.Dl)
        for i in range(cfg.decoder_depth):
            pre = f"decoder.{i}"
            self._ln(y[:RD], f"{pre}.norm1.weight", w["ln"], w["st"])
            self._lin_fwd(f"{pre}.self_attn.qkv.weight", w["ln"], w["qkv"], RD)
            self._attn_infer(w, w["qkv"], 0, 3 * A, w["qkv"], A, 2 * A, 3 * A, w["ao"], w["zero_b"], w["full_m"], G * B, M, M)
            self._lin_fwd(f"{pre}.self_attn.proj.weight", w["ao"], yn, RD, L.EPI_RESID, R=y)
            if RQ > 0:
                self._ln(yn[:RQ], f"{pre}.query_norm.weight", w["ln"], w["st"])
                self._lin_fwd(f"{pre}.cross_attn.q.weight", w["ln"], w["q"], RQ)
                cn = w["cns"][i] if fused else w["cn"]
                kvb = w["kv"]
                if kv_side is not None or (fused and w.get("kvs") is not None and RQ > 0):
                    kvb = w["kvs"][i]                          # made by the side chain (or, one stream: in line, at the top)
                    if kv_evs is not None:
                        torch.cuda.current_stream().wait_event(kv_evs[i])
                else:
                    if not fused:
                        self._ln(w["ctx"][:RC], f"{pre}.context_norm.weight", cn, w["st"])
                    self._lin_fwd(f"{pre}.cross_attn.kv.weight", cn, kvb, RC)

                def cross(g):
                    side, n, c0 = parts[g]
                    self._attn_infer(w, w["q"], g * RM * A, A, kvb, c0 * 2 * A, c0 * 2 * A + A, 2 * A, w["ao"][g * RM:], w["zero_b"],
                                     side["n_valid"], B, M, n, lane=1 if (g == 1 and w.get("lse2") is not None) else 0)
                # the two halves of a guided step attend different contexts: the second half beside the first
                forked = self._fork(lambda: cross(1)) if (n_ctx == 2 and w.get("lse2") is not None) else None
                cross(0)
                if n_ctx == 2 and w.get("lse2") is None:
                    cross(1)
                for g in range(2, n_ctx):
                    cross(g)
                self._join(forked)
                self._lin_fwd(f"{pre}.cross_attn.proj.weight", w["ao"], y, RQ, L.EPI_RESID, R=yn)
                if RQ < RD:
                    y[RQ:RD].copy_(yn[RQ:RD])
            else:
                # empty context: softmax over zero keys contributes nothing (attn @ v over an empty axis = 0) and the
                # bias-free proj keeps it 0, so the cross-attention residual is the identity
                y, yn = yn, y
            self._ln(y[:RD], f"{pre}.norm2.weight", w["ln"], w["st"])
            self._mlp_gate_fwd(pre, w["ln"], w["ab"], w["h"], RD)
            self._lin_fwd(f"{pre}.mlp.fc2.weight", w["h"], yn, RD, L.EPI_RESID, R=y)
            y, yn = yn, y
        self._join(kv_side)                                  # (every kv event has been waited for; a capture wants the branch joined)
        self._ln(y[:RD], "decoder_norm.weight", w["ln"], w["st"])
        l = self.lin[self.logit_key[tm.name]]
        logits = out if out is not None else torch.empty(RD, tm.vocab_size, device=self.dev, dtype=BF16)
        ops.gemm_nt(w["ln"], l.wb, logits, RD, tm.vocab_size, D, L.EPI_BF16, lda=D, ldb=D, ldc=tm.vocab_size)
        return logits

    @torch.no_grad()
    def infer_logits(self, enc_inputs: Dict[str, Tuple[torch.Tensor, torch.Tensor]], n_enc: int, target: str,
                     dec_pos: torch.Tensor, out: Optional[torch.Tensor] = None, ws: Optional[dict] = None) -> torch.Tensor:
        """One encoder-decoder pass of `forward_enc_dec_roar_batched` (egom2p/models/generate.py:747-766).

        enc_inputs: modality -> (ids int64 [B, n], input_mask bool [B, n]); n_enc = rows kept per sample (the
        reference takes the max unmasked count over the batch, :413-415; 0 = unconditional pass with an empty
        context).  dec_pos: int [B, M] positions of the decoded target tokens inside `target` (their order is the
        ROAR order; decoder self-attention is unmasked, sa_mask=None at :761).  Returns bf16 logits [B, M, V]."""
        if self.weights_dirty:
            self.refresh_weights()
        B, M = dec_pos.shape
        N = int(n_enc) + self.R                  # register tokens ride in front of the kept inputs: the context is never empty then
        w = ws if ws is not None else self._alloc_infer(B, max(N, 1), M)
        side = w["sides"][0]
        if N > 0:
            self._infer_context(w, [(side, enc_inputs, N, 0)], B)
        logits = self._infer_decode(w, [(side, N, 0)], target, dec_pos, out)
        return logits.view(B, M, -1)

    @torch.no_grad()
    def infer_logits_cfg(self, enc_cond, n_cond: int, enc_uncond, n_uncond: int, target: str, dec_pos: torch.Tensor,
                         out: Optional[torch.Tensor] = None, ws: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The conditional and the unconditional pass of one classifier-free-guidance step (generate.py:881-905 runs
        forward_enc_dec_roar_batched twice) as ONE pass: the encoder runs the two input sets as two groups of rows back to
        back (row-wise launches cover both, self-attention per group); both passes decode the same positions from the same
        mask-token rows, only the cross-attention context differs - the decoder runs 2 B samples per launch (the launches
        of this path are latency-bound: twice the rows cost the same time) and cross-attends per half.
        Returns (logits_cond, logits_uncond), each bf16 [B, M, V]."""
        if self.weights_dirty:
            self.refresh_weights()
        B, M = dec_pos.shape
        Nc, Nu = int(n_cond), int(n_uncond)
        if Nc <= 0:
            raise ValueError("the conditional pass of a guided step has a context")
        Nc, Nu = Nc + self.R, Nu + self.R        # register tokens ride in front of the kept inputs of either pass
        w = ws if ws is not None else self._alloc_infer(B, max(Nc, Nu, 1), M, groups=2)
        if w["groups"] < 2:
            raise ValueError("workspace allocated for single passes (groups=1)")
        sc, su = w["sides"]
        self._infer_context(w, [(sc, enc_cond, Nc, 0), (su, enc_uncond, Nu, B * Nc)], B)
        logits = self._infer_decode(w, [(sc, Nc, 0), (su, Nu, B * Nc)], target, dec_pos, out)
        V = logits.shape[-1]
        return logits[:B * M].view(B, M, V), logits[B * M:2 * B * M].view(B, M, V)

    def _graphed(self, key, make_state, run):
        """Capture-once / replay cache of the generation passes: `make_state()` builds the static buffers (a captured
        graph bakes in device addresses: it owns its workspace for as long as it lives), `run(st)` issues the launches."""
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.pop(key, None)
        if g is None:
            # every graph owns a workspace of ~30 KB per row: bound the cache (a ROAR schedule needs 2 x steps shapes;
            # the least recently used graph and its buffers are released first)
            while len(graphs) >= self.max_graphs:
                graphs.pop(next(iter(graphs)))
            st = make_state()
            side = warmup_stream(self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                           # warm-up (lazy inits happen here, not in the capture)
                run(st)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run(st)
            g = (graph, st)
        graphs[key] = g                                            # most recently used last
        return g

    def infer_logits_graphed(self, enc_inputs, n_enc: int, target: str, dec_pos: torch.Tensor) -> torch.Tensor:
        """`infer_logits` replayed from a captured hipGraph (BASELINE config 4: "hipGraph-captured decode").
        One graph per pass shape (batch, modalities, N, M, target): inputs are copied into static buffers, the
        ~300 kernel launches of a pass are replayed with one host call, the logits land in a static buffer
        that stays valid until the next replay of the same graph."""
        if self.weights_dirty:
            self.refresh_weights()
        B, M = dec_pos.shape
        names = tuple(m.name for m in self.mods if m.name in enc_inputs)
        V = {m.name: m for m in self.mods}[target].vocab_size

        def make_state():
            return {"ws": self._alloc_infer(B, max(int(n_enc) + self.R, 1), M, fresh=True),
                    "ids": {n: enc_inputs[n][0].reshape(B, -1).to(self.dev, torch.int64).clone() for n in names},
                    "mask": {n: enc_inputs[n][1].reshape(B, -1).to(self.dev, torch.bool).clone() for n in names},
                    "pos": dec_pos.to(self.dev, I32).clone(),
                    "out": torch.empty(B * M, V, device=self.dev, dtype=BF16)}

        graph, st = self._graphed((B, names, int(n_enc), M, target), make_state,
                                  lambda st: self.infer_logits({n: (st["ids"][n], st["mask"][n]) for n in names}, n_enc, target, st["pos"],
                                                               out=st["out"], ws=st["ws"]))
        for n in names:
            st["ids"][n].copy_(enc_inputs[n][0].reshape(B, -1))
            st["mask"][n].copy_(enc_inputs[n][1].reshape(B, -1))
        st["pos"].copy_(dec_pos)
        graph.replay()
        return st["out"].view(B, M, V)

    def infer_logits_cfg_graphed(self, enc_cond, n_cond: int, enc_uncond, n_uncond: int, target: str, dec_pos: torch.Tensor):
        """`infer_logits_cfg` (both passes of a guided step, one decoder pass) replayed from a captured hipGraph"""
        if self.weights_dirty:
            self.refresh_weights()
        B, M = dec_pos.shape
        names = tuple(m.name for m in self.mods if m.name in enc_cond)
        V = {m.name: m for m in self.mods}[target].vocab_size
        sets = (enc_cond, enc_uncond)

        def make_state():
            return {"ws": self._alloc_infer(B, max(int(n_cond), int(n_uncond), 1) + self.R, M, fresh=True, groups=2),
                    "ids": [{n: e[n][0].reshape(B, -1).to(self.dev, torch.int64).clone() for n in names} for e in sets],
                    "mask": [{n: e[n][1].reshape(B, -1).to(self.dev, torch.bool).clone() for n in names} for e in sets],
                    "pos": dec_pos.to(self.dev, I32).clone(),
                    "out": torch.empty(2 * B * M, V, device=self.dev, dtype=BF16)}

        graph, st = self._graphed(("cfg", B, names, int(n_cond), int(n_uncond), M, target), make_state,
                                  lambda st: self.infer_logits_cfg({n: (st["ids"][0][n], st["mask"][0][n]) for n in names}, n_cond,
                                                                   {n: (st["ids"][1][n], st["mask"][1][n]) for n in names}, n_uncond,
                                                                   target, st["pos"], out=st["out"], ws=st["ws"]))
        for k, e in enumerate(sets):
            for n in names:
                st["ids"][k][n].copy_(e[n][0].reshape(B, -1))
                st["mask"][k][n].copy_(e[n][1].reshape(B, -1))
        st["pos"].copy_(dec_pos)
        graph.replay()
        return st["out"][:B * M].view(B, M, V), st["out"][B * M:].view(B, M, V)

    @torch.no_grad()
    def forward_logits(self, mod_dict, dec_order=None) -> Dict[str, torch.Tensor]:
        """`return_logits=True` path of EgoM2P.forward (egom2p_model.py:727-729, 546-547): logits of every
        decoder row for every modality, (B, M, V) bf16."""
        self.forward(mod_dict, dec_order=dec_order, need_loss=False, group_rows=False)
        if int(self.cd["err"].item()):              # inference API: the caller reads the logits on the host anyway
            self.cd["err"].zero_()
            raise L.EgoHipError("decoder_attention_mask is not one key interval per row (egom2p_model.py:446-481 would "
                                "produce a mask this engine cannot express)")
        B, M, D = self.B, self.M, self.D
        out = {}
        for m in self.mods:
            l = self.lin[self.logit_key[m.name]]
            lg = torch.empty(B * M, m.vocab_size, device=self.dev, dtype=BF16)
            ops.gemm_nt(self.yn, l.wb, lg, B * M, m.vocab_size, D, L.EPI_BF16, lda=D, ldb=D, ldc=m.vocab_size)
            out[m.name] = lg.view(B, M, m.vocab_size)
        self._have_fwd = False
        return out
