"""ctypes binding of libegom2p_hip.so (the C-ABI declared in include/egom2p_hip.h).

The product path fails loudly when the HIP library is missing: there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EGOM2P_HIP_LIB", os.path.join(_HERE, "libegom2p_hip.so"))   # override: kernel experiments
MAX_MODS = 8

ABI_VERSION = 6          # == EGO_ABI_VERSION of include/egom2p_hip.h (tests/test_cabi_exports.py holds the two together)
EPI_BF16, EPI_F32, EPI_RESID, EPI_BIAS_RESID = 0, 1, 2, 3

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float


class CompactDesc(C.Structure):
    _fields_ = [
        ("n_mods", i32), ("n_keep", i32), ("is_decoder", i32),
        ("mask", vp * MAX_MODS), ("ids", vp * MAX_MODS), ("dam", vp * MAX_MODS),
        ("n_pos", i32 * MAX_MODS), ("mod_id", i32 * MAX_MODS),
        ("ids_keep", vp), ("pad", vp), ("mod_mask", vp), ("slot", vp), ("local", vp), ("tok", vp),
        ("ks", vp), ("ke", vp), ("n_valid", vp), ("seg", vp), ("err", vp), ("seg_bad", vp), ("n_reg", i32),
    ]


MAX_MIX = 8


class BudgetDesc(C.Structure):
    _fields_ = [
        ("n_mods", i32), ("n_mix", i32),
        ("in_alpha", (f32 * MAX_MODS) * MAX_MIX), ("tgt_alpha", (f32 * MAX_MODS) * MAX_MIX), ("mix_weight", f32 * MAX_MIX),
        ("max_tokens", i32 * MAX_MODS), ("min_tokens", i32 * MAX_MODS), ("not_seq", i32 * MAX_MODS),
        ("n_in_lo", i32), ("n_in_hi", i32), ("n_tgt_lo", i32), ("n_tgt_hi", i32), ("max_tries", i32),
    ]


class EmbedDesc(C.Structure):
    _fields_ = [
        ("table", vp * MAX_MODS), ("pos", vp * MAX_MODS), ("mod", vp * MAX_MODS), ("base_vec", vp),
        ("slot", vp), ("local", vp), ("tok", vp), ("x", vp), ("emb", vp), ("rows", i64), ("D", i32), ("reg", vp),
    ]


class EmbedBwdDesc(C.Structure):
    _fields_ = [
        ("dtable", vp * MAX_MODS), ("dmod", vp * MAX_MODS), ("dbase", vp), ("dx", vp), ("d2", vp),
        ("slot", vp), ("tok", vp), ("rows", i64), ("D", i32), ("n_mods", i32), ("touched", vp * MAX_MODS),
        ("work", vp), ("work_floats", i64), ("vocab", i32 * MAX_MODS),
    ]


_SIGS = {
    "ego_abi_version": [],
    "ego_gemm_kernel_mode": [i32, i32],
    "ego_gemm_small_tiles": [i32],
    "ego_gemm_tune": [i32, i32],
    "ego_attn_tune": [i32, i32],
    "ego_compact": [C.POINTER(CompactDesc), i32, vp],
    "ego_embed_fwd": [C.POINTER(EmbedDesc), vp],
    "ego_embed_bwd_work_floats": [i64, i32, i32],
    "ego_embed_bwd": [C.POINTER(EmbedBwdDesc), vp],
    "ego_reg_grad": [vp, i32, i64, i32, i32, vp, vp],
    "ego_rows_compact": [vp, i32, i32, vp, vp, vp],
    "ego_rows_gather": [vp, vp, vp, i32, i32, vp, vp],
    "ego_rows_scatter": [vp, vp, vp, i32, i32, vp, i32, vp],
    "ego_loss_perm": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "ego_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i64, f32, vp, i64, vp, vp],
    "ego_layernorm_bwd_work_floats": [i32, i32],
    "ego_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, vp],
    "ego_layernorm_fwd_multi": [vp, i32, vp, vp, vp, vp, i32, i32, i64, f32, vp],
    "ego_layernorm_bwd_multi_work_floats": [i32, i32, i32],
    "ego_layernorm_bwd_multi": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, vp],
    "ego_gemm_nt_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp],
    "ego_gemm_tn_bf16": [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, vp, i32, i32, i32, i32, vp, vp],
    "ego_quant_fp8_rows": [vp, i64, i64, i32, vp, i64, vp, vp],
    "ego_gemm_nt_fp8": [vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, vp, i32, i32, i32, i32, vp],
    "ego_gemm_nt_swiglu_fwd_fp8": [vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, vp],
    "ego_gemm_tn_plan": [i32, i32, i32, i64, i64, i64, i32],
    "ego_attn_fwd_d64": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, i64, i64,
                         i32, i32, i32, i32, f32, vp],
    "ego_attn_bwd_d64": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, vp, vp,
                         vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, f32, vp],
    "ego_attn_fwd_d64_seg": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, i64, i64, vp, i32, vp,
                             i32, i32, i32, i32, f32, vp],
    "ego_attn_bwd_d64_seg": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, vp, vp,
                             vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, vp, i32, vp, i32, i32, i32, i32, f32, vp],
    "ego_attn_fwd_split_floats": [i32, i32, i32, i32],
    "ego_attn_fwd_d64_split": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64, i64,
                               i32, i32, i32, i32, f32, i32, vp, i64, vp],
    "ego_attn_fwd_hd": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, i64, i64,
                        i32, i32, i32, i32, i32, f32, vp],
    "ego_attn_bwd_hd": [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, vp, vp,
                        vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, i32, f32, vp],
    "ego_budget_dirichlet": [C.POINTER(BudgetDesc), vp, i32, vp, vp, vp],
    "ego_clip_synth": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "ego_swiglu_fwd": [vp, vp, i64, i32, vp],
    "ego_swiglu_bwd": [vp, vp, vp, i64, i32, vp],
    "ego_gemm_nt_swiglu_bwd": [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, vp],
    "ego_gemm_nt_swiglu_fwd": [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "ego_ce_fwd": [vp, i64, i32, vp, vp, i32, vp, vp, vp],
    "ego_ce_bwd": [vp, i64, i32, vp, vp, i32, vp, vp, i32, vp, vp],
    "ego_ce_fwd_bwd": [vp, i64, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp],
    "ego_loss_weights": [vp, C.POINTER(i32), i32, i32, vp, vp, vp],
    "ego_loss_finalize": [vp, vp, i32, vp, vp, vp, vp, vp],
    "ego_cast_weight": [vp, i32, i32, i64, vp, i64, vp, i64, i32, vp],
    "ego_cast_f32_bf16": [vp, vp, i64, vp],
    "ego_bias_grad_work_floats": [i64, i32],
    "ego_bias_grad": [vp, i64, i32, vp, vp, i64, vp],
    "ego_grad_sqnorm": [vp, i64, vp, vp, vp],
    "ego_adamw_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, f32, vp, i32, vp],
    "ego_sample_cfg_topp": [vp, vp, i64, i32, f32, f32, i32, f32, vp, vp, vp, i32, vp],
    "ego_dp_unique_id": [vp],
    "ego_dp_comm_create": [vp, i32, i32, C.POINTER(vp)],
    "ego_dp_comm_destroy": [vp],
    "ego_dp_allreduce_begin": [vp, vp, i64, i32, vp, vp],
    "ego_dp_wait": [vp, vp, vp],
}

EXPORTS = tuple(_SIGS)
_lib = None


class EgoHipError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EgoHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C egom2p_amd/csrc`). There is no CPU fallback for the product path.")
        # PyTorch-ROCm bundles its own libamdhip64; it must be in the process BEFORE our library is
        # dlopen'ed so both share one HIP runtime (streams and device pointers are per-runtime).
        import torch
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = i64 if name.endswith(("_work_floats", "_split_floats")) else i32
        if lib.ego_abi_version() != ABI_VERSION:
            raise EgoHipError(f"libegom2p_hip.so ABI version {lib.ego_abi_version()} != binding {ABI_VERSION}: rebuild (make -C egom2p_amd/csrc)")
        # kernel experiments: EGO_GEMM_NT256 / EGO_GEMM_TN256 = 0 | 1 | 2 pick the GEMM tile family (the library itself
        # reads no environment; this is the same call tests make through ops.gemm_kernel_mode)
        if "EGO_GEMM_NT256" in os.environ or "EGO_GEMM_TN256" in os.environ:
            lib.ego_gemm_kernel_mode(int(os.environ.get("EGO_GEMM_NT256", "1")), int(os.environ.get("EGO_GEMM_TN256", "1")))
        if "EGO_GEMM_SMALL_TILES" in os.environ:
            lib.ego_gemm_small_tiles(int(os.environ["EGO_GEMM_SMALL_TILES"]))
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise EgoHipError(f"{what} failed: {'bad arguments' if rc == 1 else 'kernel launch failed'} (rc={rc})")

