"""ORACLE tooling - fixtures for the reference's other two loss types (`loss_type='token'` / `'weighted_mod'`,
egom2p/models/egom2p_model.py:583-612, 646-681), made by running the REAL reference model end to end.

Runs only in the build container (needs /root/reference; see oracle/make_goldens.py for how the four hot-path files are
imported).  Same generator-made weights / clips / python-random seeds as the `tiny_pad`, `tiny8` and `b2_ragged` cases of
make_goldens.py, so the decoder order of `tests/golden/<case>.npz` applies.  Stores per (case, loss type): the loss, the
per-modality losses, the squared gradient norm of every parameter and the total gradient norm -> tests/golden/loss_types.npz.

    python oracle/make_goldens_loss_types.py
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import make_goldens as MG                          # noqa: E402
from egom2p_amd import synth                       # noqa: E402
from egom2p_amd.config import MODEL_CFGS           # noqa: E402


def main():
    enc, dec, model = MG.load_reference()
    torch.set_num_threads(8)
    gold = {}
    for case in ("tiny_pad", "tiny8", "b2_ragged"):
        kw = dict(MG.CASES[case])
        cfg = MODEL_CFGS[kw["cfg_name"]]
        if kw["budgets"] == "dirichlet":
            kw["budgets"] = synth.dirichlet_budgets(cfg, kw["batch"], kw["n_enc"], kw["n_dec"], kw["seed"])
        torch.manual_seed(0)
        net = MG.build_reference_model(cfg, enc, dec, model)
        net.load_state_dict(synth.build_state_dict(cfg, kw["seed"]), strict=True)
        net.train()
        mod_dict = synth.make_clip_batch(cfg, kw["batch"], kw["budgets"], kw["seed"])
        for lt in ("mod", "weighted_mod", "token"):
            random.seed(kw["py_seed"])
            md = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in mod_dict.items()}
            net.zero_grad()
            loss, mod_loss = net(md, kw["n_enc"], kw["n_dec"], lt)
            loss.backward()
            named = dict(net.named_parameters())
            pre = f"{case}.{lt}"
            gold[f"{pre}.loss"] = np.array(loss.item(), dtype=np.float64)
            gold[f"{pre}.mod_names"] = np.array(list(mod_loss.keys()))
            gold[f"{pre}.mod_loss"] = np.array([v.item() for v in mod_loss.values()], dtype=np.float64)
            gold[f"{pre}.grad_names"] = np.array(list(named.keys()))
            gold[f"{pre}.grad_sqnorm_all"] = np.array([p.grad.double().pow(2).sum().item() if p.grad is not None else -1.0
                                                       for p in named.values()])
            gold[f"{pre}.grad_total_norm"] = np.array(sum(p.grad.double().pow(2).sum().item() for p in named.values()
                                                          if p.grad is not None) ** 0.5)
            print(f"[goldens] {pre}: loss {loss.item():.6f}  mod {[round(v.item(), 4) for v in mod_loss.values()]}")
    gold["meta"] = np.array(repr(dict(what="reference EgoM2P.forward(loss_type=...) on the tiny_pad / tiny8 / b2_ragged inputs")))
    path = os.path.join(ROOT, "tests", "golden", "loss_types.npz")
    np.savez_compressed(path, **gold)
    print(f"[goldens] -> {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


if __name__ == "__main__":
    main()
