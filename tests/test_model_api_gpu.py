"""The drop-in module surface on the GPU: `EgoM2P` / `create_model` / `create_optimizer` /
`NativeScalerWithGradNormCount` / `DataParallel` used the way run_training_egom2p.py uses the reference's
(`:381-387, 514-518, 723-741`), checked against the reference goldens."""
import random
import types
from functools import partial

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

from conftest import load_golden, rel_l2  # noqa: E402
from egom2p_amd import synth  # noqa: E402
from egom2p_amd.config import MODEL_CFGS  # noqa: E402
from egom2p_amd.dp import DataParallel  # noqa: E402
from egom2p_amd.model import (MODALITY_INFO, EgoM2P, LayerNorm, create_model, list_models)  # noqa: E402
from egom2p_amd.optim import NativeScalerWithGradNormCount, create_optimizer  # noqa: E402


def _tiny_model():
    mods = ["tok_cam", "tok_gaze"]
    enc = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}
    return EgoM2P(enc, dec, {m: MODALITY_INFO[m] for m in mods}, dim=128, encoder_depth=2, decoder_depth=2, num_heads=2,
                  mlp_ratio=4, qkv_bias=False, proj_bias=False, mlp_bias=False,
                  norm_layer=partial(LayerNorm, eps=1e-6, bias=False), act_layer=nn.SiLU, gated_mlp=True)


def test_module_forward_backward_step_matches_reference():
    g, meta = load_golden("tiny_pad")
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    model = _tiny_model()
    # state-dict key layout is the reference's
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd)
    for k, v in model.state_dict().items():
        assert torch.equal(v.float().cpu().reshape(sd[k].shape), sd[k]), k
    ddp = DataParallel(model)                           # single process: passthrough with .module / no_sync()
    args = types.SimpleNamespace(opt="adamw", lr=meta["lr"], weight_decay=meta["wd"], opt_betas=(0.9, 0.95), opt_eps=1e-8)
    opt = create_optimizer(args, ddp.module)
    scaler = NativeScalerWithGradNormCount(enabled=False)

    random.seed(meta["py_seed"])                        # same python-random state as the golden run -> same decoder order
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    loss, mod_loss = ddp(mdg, num_encoder_tokens=meta["n_enc"], num_decoder_tokens=meta["n_dec"], loss_type="mod")
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"])
    for m in cfg.mods:
        assert abs(mod_loss[m.name].item() - float(g[f"mod_loss.{m.name}"])) < 1e-3 * max(1.0, float(g[f"mod_loss.{m.name}"]))
    opt.zero_grad()
    norm = scaler(loss, opt, clip_grad=1.0, parameters=ddp.parameters(), update_grad=True)
    assert abs(norm.item() - float(g["clip_total_norm"])) < 2e-2 * float(g["clip_total_norm"])
    new = model.state_dict()
    for key in g.files:
        if key.startswith("adamw."):
            n = key.split(".", 1)[1]
            assert np.abs(new[n].float().cpu().numpy().reshape(g[key].shape) - g[key]).max() <= 2.2 * meta["lr"], n
    # the fused step zeroed the gradients (zero_grad folded in)
    assert float(model.engine.G.abs().max().item()) == 0.0


def test_gradients_visible_through_param_grad_and_no_sync_accumulates():
    g, meta = load_golden("tiny")
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    model = _tiny_model()
    model.load_state_dict(sd)
    ddp = DataParallel(model)
    named = dict(model.named_parameters())
    for rep in range(2):
        random.seed(meta["py_seed"])
        with ddp.no_sync():
            loss, _ = ddp(mdg, meta["n_enc"], meta["n_dec"])
            (loss / 2).backward()
    coef = min(1.0, 1.0 / (float(g["clip_total_norm"]) + 1e-6))
    for key in g.files:
        if key.startswith("grad."):
            n = key[5:]
            got = named[n].grad.float().cpu().numpy().reshape(g[key].shape) * coef      # two half-weighted passes = one
            assert rel_l2(got, g[key]) < 4e-2, (n, rel_l2(got, g[key]))


def test_registered_ego_l_geometry_through_the_module_surface():
    """dim 1020, 15 heads of 68 (egom2p_large_24e_24d_swiglu_nobias, egom2p_model.py:1080-1092; here at 2 + 2 layers, the
    depth of tests/golden/L1020.npz): the module keeps the reference's key layout and state-dict shapes on the padded
    storage, loss / clipped norm / AdamW step follow the reference's, and the storage pads stay exact zeros."""
    assert {"egom2p_large_24e_24d_swiglu_nobias", "egom2p_xlarge_24e_24d_swiglu_nobias"} <= set(list_models())
    g, meta = load_golden("L1020")
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    mods = ["tok_rgb", "tok_depth", "tok_cam", "tok_gaze"]
    enc = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}
    model = EgoM2P(enc, dec, {m: MODALITY_INFO[m] for m in mods}, dim=1020, encoder_depth=2, decoder_depth=2, num_heads=15,
                   mlp_ratio=4, qkv_bias=False, proj_bias=False, mlp_bias=False,
                   norm_layer=partial(LayerNorm, eps=1e-6, bias=False), act_layer=nn.SiLU, gated_mlp=True)
    eng = model.engine
    assert eng.padded and (eng.D, eng.HDP, eng.Hs, eng.A) == (1024, 96, 16, 1536)      # 15 heads of 68 + one phantom head, zero
    assert sum(p.numel() for p in model.parameters()) == eng.num_params() == sum(v.numel() for k, v in sd.items()
                                                                                 if not k.endswith("pos_emb") and not (k.endswith(".bias") and "norm" in k)
                                                                                 and not (k.startswith("decoder_embeddings") and k.endswith(("mod_emb", "to_logits.weight"))))
    # the constructor's init left the pads at zero and drew the reference's xavier bound for the (3 x [1020, 1020]) qkv
    rest = eng.P.clone()
    for n, (o, cnt, shape) in eng.offsets.items():
        eng._logical(n, rest[o:o + cnt].view(shape)).zero_()
    assert not bool(rest.any())
    qkv = dict(model.named_parameters())["encoder.0.attn.qkv.weight"]
    assert tuple(qkv.shape) == (3, 15, 68, 1020) and abs(float(qkv.detach().abs().max()) - (6.0 / 2040) ** 0.5) < 1e-3
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd)
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape) and torch.equal(v.float().cpu(), sd[k]), k
    args = types.SimpleNamespace(opt="adamw", lr=meta["lr"], weight_decay=meta["wd"], opt_betas=(0.9, 0.95), opt_eps=1e-8)
    opt = create_optimizer(args, model)
    scaler = NativeScalerWithGradNormCount(enabled=False)
    random.seed(meta["py_seed"])
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    loss, mod_loss = model(mdg, num_encoder_tokens=meta["n_enc"], num_decoder_tokens=meta["n_dec"], loss_type="mod")
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"])
    opt.zero_grad()
    norm = scaler(loss, opt, clip_grad=1.0, parameters=model.parameters(), update_grad=True)
    assert abs(norm.item() - float(g["clip_total_norm"])) < 2e-2 * float(g["clip_total_norm"])
    new = model.state_dict()
    for key in g.files:
        if key.startswith("adamw."):
            n = key.split(".", 1)[1]
            assert np.abs(new[n].float().cpu().numpy().reshape(g[key].shape) - g[key]).max() <= 2.2 * meta["lr"], n
    rest = eng.P.clone()
    for n, (o, cnt, shape) in eng.offsets.items():
        eng._logical(n, rest[o:o + cnt].view(shape)).zero_()
    assert not bool(rest.any())


def test_registry_and_scope_errors():
    assert "egom2p_base_12e_12d_swiglu_nobias" in list_models()
    mods = ["tok_rgb", "tok_depth", "tok_cam", "tok_gaze"]
    enc = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}
    model = create_model("egom2p_tiny_6e_6d_swiglu_nobias", encoder_embeddings=enc, decoder_embeddings=dec,
                         modality_info=MODALITY_INFO, num_register_tokens=0)
    n = sum(p.numel() for p in model.parameters())
    assert n == model.engine.num_params()
    with pytest.raises(NotImplementedError):
        create_model("egom2p_base_12e_12d_gelu", encoder_embeddings=enc, decoder_embeddings=dec, modality_info=MODALITY_INFO)
    # eval / return_logits path
    md = synth.make_clip_batch(MODEL_CFGS["ego_b_2e_2d"], 1, None, seed=9)
    with torch.no_grad():
        logits = model({k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}, 256, 256, return_logits=True)
    assert logits["tok_rgb"].shape == (1, 256, 64000) and logits["tok_cam"].shape == (1, 256, 256)
    assert torch.isfinite(logits["tok_rgb"].float()).all()
    # register tokens (`create_model(..., num_register_tokens=args.num_register_tokens)`, run_training_egom2p.py:381-387): a learned
    # (1, R, dim) parameter under the reference's name, counted, trained (the engine-level parity is tests/golden/b2_reg4.npz)
    assert model.register_tokens is None and "register_tokens" not in model.state_dict()
    enc4 = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods}
    dec4 = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}
    m4 = create_model("egom2p_tiny_6e_6d_swiglu_nobias", encoder_embeddings=enc4, decoder_embeddings=dec4,
                      modality_info=MODALITY_INFO, num_register_tokens=4)
    assert m4.num_register_tokens == 4 and tuple(m4.register_tokens.shape) == (1, 4, 384)
    assert tuple(m4.state_dict()["register_tokens"].shape) == (1, 4, 384)
    assert sum(p.numel() for p in m4.parameters()) == n + 4 * 384 == m4.engine.num_params()
    assert abs(float(m4.register_tokens.detach().std()) - 0.02) < 4e-3            # nn.init.normal_(std=init_std), egom2p_model.py:172
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    loss, _ = m4(mdg, 256, 256)
    loss.backward()
    assert torch.isfinite(loss) and float(m4.register_tokens.grad.abs().sum()) > 0


def _ego_b_2e_2d_model():
    mods = ["tok_rgb", "tok_depth", "tok_cam", "tok_gaze"]
    enc = {m: MODALITY_INFO[m]["encoder_embedding"]() for m in mods}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"]() for m in mods}
    return EgoM2P(enc, dec, {m: MODALITY_INFO[m] for m in mods}, dim=768, encoder_depth=2, decoder_depth=2, num_heads=12,
                  mlp_ratio=4, qkv_bias=False, proj_bias=False, mlp_bias=False,
                  norm_layer=partial(LayerNorm, eps=1e-6, bias=False), act_layer=nn.SiLU, gated_mlp=True)


def _check_init(named_tensors, g, tag):
    """every parameter against the statistics of the reference's own constructor init (tests/golden/init_stats.npz,
    made by oracle/make_goldens.py:init_stats from egom2p_model.py:185-222): same family, same scale."""
    for n, (mean, std, mn, mx, numel) in zip([str(x) for x in g[f"{tag}.names"]], g[f"{tag}.stats"]):
        t = named_tensors[n].double().flatten()
        assert t.numel() == int(numel), n
        if std == 0.0:                                          # LayerNorm weight 1, Linear bias 0
            assert float((t - mean).abs().max()) == 0.0, n
            continue
        bound = max(abs(mn), abs(mx))
        s, amax = float(t.std()), float(t.abs().max())
        rel = 4.0 / np.sqrt(2.0 * numel) + 2e-3                 # 4 sigma of a std estimate from numel samples
        assert abs(s - std) < rel * std, (n, s, std)
        assert abs(float(t.mean()) - mean) < 6.0 * std / np.sqrt(numel) + 1e-7, (n, float(t.mean()), mean)
        if bound / std < 1.8:                                   # uniform(-a, a): a / std = sqrt(3); normal: >= 3
            assert amax <= bound * (1 + 1e-3) and amax >= bound * (1 - 1e-2), (n, amax, bound)
        else:
            assert amax / s > 2.5, (n, amax / s)


def test_init_weights_match_reference_distributions():
    """SURVEY section 8 row a14 / ADVICE r1: xavier-uniform linears with qkv / kv as 3 / 2 matrices, N(0, 0.02) encoder
    tables / mod_emb / mask_token, and the tied decoder table xavier-uniform on [V, D] (the `to_logits` Linear is
    initialised after the Embedding it shares its weight with)."""
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN_DIR, "init_stats.npz"), allow_pickle=False)
    torch.manual_seed(5)
    model = _ego_b_2e_2d_model()
    _check_init(model.state_dict(), g, "tied")
    # the engine's own device-side initialiser (bench.py / run_training_egom2p.py --init random), tied and untied
    from egom2p_amd.engine import Engine
    for tag, cfg_name in (("tied", "ego_b_2e_2d"), ("untied", "ego_b_2e_2d_untied")):
        eng = Engine(MODEL_CFGS[cfg_name], "cuda:0", max_batch=1, n_enc=64, n_dec=64)
        eng.init_random(seed=3)
        _check_init(eng.state_dict(), g, tag)
        del eng


def test_frozen_encoder_steps_match_torch_adamw_on_the_unfrozen_subset():
    """with frozen tensors (`freeze_encoder`) the clip norm covers only tensors that have a gradient, the
    frozen tensors stay untouched, and nothing accumulates in the flat gradient buffer from step to step."""
    g, meta = load_golden("tiny")
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    model = _tiny_model()
    model.load_state_dict(sd)
    model.freeze_encoder()
    args = types.SimpleNamespace(opt="adamw", lr=1e-2, weight_decay=0.05, opt_betas=(0.9, 0.95), opt_eps=1e-8)
    opt = create_optimizer(args, model)
    scaler = NativeScalerWithGradNormCount(enabled=False)
    named = dict(model.named_parameters())
    live = [n for n, p in named.items() if p.requires_grad]
    frozen = [n for n, p in named.items() if not p.requires_grad]
    assert frozen and live
    frozen0 = {n: named[n].detach().clone() for n in frozen}
    # torch reference on clones of the unfrozen tensors, fed with the engine's own gradients
    ref = {n: named[n].detach().clone().requires_grad_(True) for n in live}
    nd = lambda n: ("norm." in n or ".norm" in n or n.endswith(".bias"))
    topt = torch.optim.AdamW([{"params": [ref[n] for n in live if not nd(n)], "weight_decay": 0.05},
                              {"params": [ref[n] for n in live if nd(n)], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    clip = 0.05                                                   # below the gradient norm: the clip coefficient matters
    for step in range(3):
        random.seed(meta["py_seed"] + step)
        loss, _ = model(mdg, meta["n_enc"], meta["n_dec"])
        loss.backward()
        for n in live:
            ref[n].grad = named[n].grad.detach().clone()
        tnorm = torch.nn.utils.clip_grad_norm_([ref[n] for n in live], clip)
        topt.step()
        norm = opt.step(clip_grad=clip)
        assert tnorm.item() > clip
        assert abs(norm.item() - tnorm.item()) < 1e-5 * tnorm.item(), (step, norm.item(), tnorm.item())
        for n in live:
            assert rel_l2(named[n].detach().float().cpu().numpy(), ref[n].detach().float().cpu().numpy()) < 2e-6, (step, n)
        for n in frozen:
            assert torch.equal(named[n].detach(), frozen0[n]), n
        assert float(model.engine.G.abs().max().item()) == 0.0    # frozen ranges are cleared too


def test_unfrozen_tensors_start_their_adamw_step_count_at_one():
    """torch.optim.AdamW keeps state['step'] per parameter and passes parameters without a gradient by, so the
    blocks that --frozen_model_epochs unfreezes after k steps get bias correction 1 - beta^1 on their first update, not
    1 - beta^(k + 1) (with betas (0.9, 0.95) the latter makes the first dozens of updates ~0.45 x the reference's).  The SAME
    torch optimiser sees requires_grad flip, exactly like the reference loop (run_training_egom2p.py:686-693); the optimiser
    state also round-trips through state_dict() with the per-tensor offsets and the storage-layout tag."""
    g, meta = load_golden("tiny")
    cfg = MODEL_CFGS[meta["cfg"]]
    sd = synth.build_state_dict(cfg, meta["seed"])
    md = synth.make_clip_batch(cfg, meta["batch"], meta["budgets"], meta["seed"])
    mdg = {k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in md.items()}
    model = _tiny_model()
    model.load_state_dict(sd)
    args = types.SimpleNamespace(opt="adamw", lr=1e-2, weight_decay=0.05, opt_betas=(0.9, 0.95), opt_eps=1e-8)
    opt = create_optimizer(args, model)
    named = dict(model.named_parameters())
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in named.items()}
    nd = lambda n: ("norm." in n or ".norm" in n or n.endswith(".bias"))
    topt = torch.optim.AdamW([{"params": [ref[n] for n in named if not nd(n)], "weight_decay": 0.05},
                              {"params": [ref[n] for n in named if nd(n)], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    shared = None
    for step in range(5):
        if step == 0:
            model.freeze_shared_params()
        elif step == 3:
            model.unfreeze_all()
        live = [n for n, p in named.items() if p.requires_grad]
        if step == 0:
            shared = [n for n in named if n not in live]
            assert shared and live
        random.seed(meta["py_seed"] + step)
        loss, _ = model(mdg, meta["n_enc"], meta["n_dec"])
        loss.backward()
        for n in named:
            ref[n].grad = named[n].grad.detach().clone() if n in live else None      # torch skips parameters without a gradient
        before = {n: named[n].detach().clone() for n in shared}
        topt.step()
        opt.step(clip_grad=None)
        for n in named:
            assert rel_l2(named[n].detach().float().cpu().numpy(), ref[n].detach().float().cpu().numpy()) < 2e-6, (step, n)
        if step == 3:
            # first update of a just-unfrozen tensor: Adam's first step moves every element with a gradient by ~lr (bias
            # correction 1 - beta^1); with the global counter (t = 4) it was ~0.45 lr for betas (0.9, 0.95)
            n = next(x for x in shared if x.endswith("attn.qkv.weight"))
            d = (named[n].detach() - before[n] * (1 - 1e-2 * 0.05)).abs()
            assert float(d.max()) > 0.9e-2 and float(d.max()) < 1.01e-2, float(d.max())
            assert topt.state[ref[n]]["step"].item() == 1 and opt.skipped[model.engine._canon_key(n)] == 3
    # state round trip: a second optimiser restored from state_dict() takes the same next step; a foreign layout is refused
    st = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()}
    opt2 = create_optimizer(args, model)
    opt2.load_state_dict(st)
    assert opt2.t == opt.t and opt2.skipped == opt.skipped and [r[3] for r in opt2._active_runs()] == [r[3] for r in opt._active_runs()]
    bad = dict(st, layout=(1, 2, 3, 4, 5))
    with pytest.raises(RuntimeError, match="storage layout"):
        create_optimizer(args, model).load_state_dict(bad)
    untagged = {k: v for k, v in st.items() if k != "layout"}          # a checkpoint written before round 5: same sizes, another element order
    with pytest.raises(RuntimeError, match="storage-layout tag"):
        create_optimizer(args, model).load_state_dict(untagged)


def test_training_script_resumes_where_it_stopped(tmp_path):
    """run_training_egom2p.py --auto_resume (the reference's auto_load_model, utils/checkpoint.py:123-157): a run that is
    interrupted after its first epoch and restarted ends with the parameters and AdamW state of the uninterrupted run
    (same clips, same cosine schedule position) - BIT FOR BIT: the training step has no float atomics, the
    checkpoint holds the fp32 masters and moments, the bf16 weight copies are a function of the masters."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_training_egom2p_amd", os.path.join(root, "run_training_egom2p.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    common = ["--model", "egom2p_tiny_6e_6d_swiglu_nobias", "--in_domains", "tok_cam-tok_gaze", "--out_domains", "tok_cam-tok_gaze",
              "--num_input_tokens", "32", "--num_target_tokens", "32", "--batch_size", "4", "--epochs", "2", "--epoch_size", "8",
              "--clip_grad", "1.0", "--blr", "1e-3", "--print_freq", "100", "--seed", "3"]
    env_keys = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        a_dir, b_dir = str(tmp_path / "a"), str(tmp_path / "b")
        R.main(R.get_args(common + ["--output_dir", a_dir]))
        # interrupted run: the second epoch never starts
        real = R.train_one_epoch

        def one_epoch_then_die(model, loader, optimizer, scaler, args, epoch, *rest):
            if epoch >= 1:
                raise KeyboardInterrupt
            return real(model, loader, optimizer, scaler, args, epoch, *rest)

        R.train_one_epoch = one_epoch_then_die
        with pytest.raises(KeyboardInterrupt):
            R.main(R.get_args(common + ["--output_dir", b_dir]))
        R.train_one_epoch = real
        assert os.path.exists(os.path.join(b_dir, "checkpoint-0.pth")) and not os.path.exists(os.path.join(b_dir, "checkpoint-1.pth"))
        R.main(R.get_args(common + ["--output_dir", b_dir, "--auto_resume"]))
    finally:
        os.environ.update(env_keys)
    ca = torch.load(os.path.join(a_dir, "checkpoint-1.pth"), map_location="cpu", weights_only=True)
    cb = torch.load(os.path.join(b_dir, "checkpoint-1.pth"), map_location="cpu", weights_only=True)
    assert ca["optimizer"]["t"] == cb["optimizer"]["t"] == 4
    for k, v in ca["model"].items():
        assert torch.equal(v, cb["model"][k]), (k, (v.float() - cb["model"][k].float()).abs().max().item())
    for k in ("m", "v"):
        assert torch.equal(ca["optimizer"][k], cb["optimizer"][k]), (k, (ca["optimizer"][k] - cb["optimizer"][k]).abs().max().item())


def test_training_script_frozen_phase_schedules_and_evaluate(tmp_path):
    """run_training_egom2p.py, the host side of the reference's loop: (a) the frozen-model phase (run_training_egom2p.py:117-138,
    524-527, 686-693): during --frozen_model_epochs only the embeddings (and what freeze_shared_params leaves trainable)
    move, at the constant frozen lr; (b) lr AND weight decay set per loader step from the schedule arrays (:707-713), here
    inverse_sqrt with a cool-down and a weight-decay end value; (c) evaluate (:800-835): a no-grad pass over held-out clips
    whose averages land in the epoch's stats and in log.txt, leave every parameter and the AdamW state untouched, and repeat
    bit for bit when run again."""
    import importlib.util
    import json
    import os
    from egom2p_amd.scheduler import build_schedules
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_training_egom2p_amd2", os.path.join(root, "run_training_egom2p.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    out = str(tmp_path / "run")
    argv = ["--model", "egom2p_tiny_6e_6d_swiglu_nobias", "--in_domains", "tok_cam-tok_gaze", "--out_domains", "tok_cam-tok_gaze",
            "--num_input_tokens", "32", "--num_target_tokens", "32", "--batch_size", "4", "--epochs", "3", "--epoch_size", "12",
            "--clip_grad", "1.0", "--blr", "1e-3", "--min_blr", "1e-5", "--print_freq", "100", "--seed", "5", "--output_dir", out,
            "--frozen_model_epochs", "1", "--frozen_model_blr", "5e-4", "--scheduler", "inverse_sqrt-20", "--warmup_steps", "2",
            "--cooldown_steps", "2", "--weight_decay", "0.05", "--weight_decay_end", "0.01", "--eval_steps", "2", "--eval_freq", "1",
            "--loss_type", "token"]
    snaps, groups, evals = [], [], []
    real_train, real_eval = R.train_one_epoch, R.evaluate

    def spy_train(model, loader, optimizer, scaler, args, epoch, *rest):
        if epoch == 0:
            snaps.append({k: v.detach().clone() for k, v in model.module.state_dict().items()})
        r = real_train(model, loader, optimizer, scaler, args, epoch, *rest)
        snaps.append({k: v.detach().clone() for k, v in model.module.state_dict().items()})
        groups.append([(g["lr"], g["weight_decay"]) for g in optimizer.param_groups])
        return r

    def spy_eval(model, loader, device, args, prefix="[Eval] "):
        before = model.module.engine.P.clone()
        a = real_eval(model, loader, device, args, prefix)
        b = real_eval(model, loader, device, args, prefix)           # the same held-out clips again (fresh decoder-order stream)
        assert torch.equal(before, model.module.engine.P)
        evals.append((a, b))
        return a

    env_keys = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    R.train_one_epoch, R.evaluate = spy_train, spy_eval
    try:
        args = R.get_args(argv)
        R.main(args)
    finally:
        R.train_one_epoch, R.evaluate = real_train, real_eval
        os.environ.update(env_keys)
    shared = lambda k: k.startswith(("encoder.", "decoder.", "encoder_norm", "decoder_norm"))
    init, e0, e1 = snaps[0], snaps[1], snaps[2]
    moved = lambda a, b, k: not torch.equal(a[k], b[k])
    trainable = [k for k in init if not (k.endswith("pos_emb") or (k.endswith(".bias") and "norm" in k))]
    assert all(not moved(init, e0, k) for k in trainable if shared(k))                        # frozen epoch: shared blocks untouched
    assert all(moved(init, e0, k) for k in trainable if not shared(k))                        # embeddings, mask token, context projection train
    assert all(moved(e0, e1, k) for k in trainable)                                           # afterwards everything trains
    # schedules: 3 loader steps per epoch; the values of an epoch's last step are still in the groups when it returns
    lr, wd = build_schedules(args, 3)
    assert len(lr) == 9 and lr[0] == lr[2] == 5e-4 * 4 / 256 and wd[0] == 0.05
    for ep, grp in enumerate(groups):
        it = 3 * ep + 2
        assert grp[0] == (float(lr[it]), float(wd[it])) and grp[1] == (float(lr[it]), 0.0), (ep, grp, lr[it], wd[it])
    assert wd[-1] == 0.01 and abs(lr[-1] - 1e-5 * 4 / 256) < 1e-12
    # evaluate: repeatable bit for bit, finite, logged
    assert len(evals) == 3
    for a, b in evals:
        assert a == b and set(a) == {"[Eval] loss", "[Eval] tok_cam_loss", "[Eval] tok_gaze_loss"}
        assert all(np.isfinite(v) and 3.0 < v < 8.0 for v in a.values())
    lines = [json.loads(x) for x in open(os.path.join(out, "log.txt"))]
    assert [x["epoch"] for x in lines] == [0, 1, 2] and all("[Eval] loss" in x for x in lines)


def test_training_script_reads_reference_token_shards(tmp_path):
    """`run_training_egom2p.py --data_path`: token shards in the reference's on-disk layout (README_DATA.md: one tar per
    modality and shard, '<key>.npz' members) are read by egom2p_amd.data.TokenShards, masked on the device by
    egom2p_amd.masking.UnifiedMasking and trained on."""
    import importlib.util
    import io
    import json
    import os
    import tarfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_training_egom2p_amd2", os.path.join(root, "run_training_egom2p.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    rng = np.random.default_rng(1)
    for f in ("cam", "gaze"):
        d = tmp_path / f / "toy" / "token"
        os.makedirs(d)
        for s in range(2):
            with tarfile.open(d / f"shard-{s:06d}.tar", "w") as tar:
                for i in range(8):
                    b = io.BytesIO()
                    np.savez(b, rng.integers(0, 256, size=30).astype(np.int32))
                    info = tarfile.TarInfo(f"clip{s}_{i:02d}.npz")
                    info.size = b.getbuffer().nbytes
                    b.seek(0)
                    tar.addfile(info, b)
    out_dir = str(tmp_path / "out")
    argv = ["--model", "egom2p_tiny_6e_6d_swiglu_nobias", "--in_domains", "tok_cam-tok_gaze", "--out_domains", "tok_cam-tok_gaze",
            "--num_input_tokens", "32", "--num_target_tokens", "32", "--batch_size", "4", "--epochs", "1", "--epoch_size", "16",
            "--clip_grad", "1.0", "--blr", "1e-3", "--print_freq", "100", "--output_dir", out_dir,
            "--data_path", f"{tmp_path}/[cam,gaze]/toy/token/shard-{{000000..000001}}.tar"]
    env_keys = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        R.main(R.get_args(argv))
    finally:
        os.environ.update(env_keys)
    ck = torch.load(os.path.join(out_dir, "checkpoint-0.pth"), map_location="cpu", weights_only=True)
    assert ck["optimizer"]["t"] == 4                                          # 16 clips / batch 4
    assert all(torch.isfinite(v.float()).all() for v in ck["model"].values())
