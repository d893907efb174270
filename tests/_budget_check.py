"""Shared checker: a budget sampler against the draws of the REAL reference's `UnifiedMasking` budget sampler
(tests/golden/budget_stats.npz, made by oracle/make_goldens_masking.py from egom2p/data/masking.py:181-234, 530-541).

Two independent random streams can only be compared in law.  With n draws on each side the bars are about four standard
errors of the two-sample difference: means 4 * sqrt(2) * std / sqrt(n) (+ 0.5 token), standard deviations 3 % (+ 0.5),
the three probabilities (budget 0, budget at the modality's cap, >= 95 % of the clip on one modality) 0.012, and the
two-sample Kolmogorov-Smirnov distance of every modality's budget distribution 0.025 (critical value at 1e-4 for
n = 24,000: 0.020)."""
import numpy as np


def stats(arr, cap):
    tot = arr.sum(0).clip(min=1)
    return np.stack([arr.mean(1), arr.std(1), (arr == 0).mean(1), (arr == cap[:, None]).mean(1),
                     (arr >= 0.95 * tot[None, :]).mean(1)], axis=1)


def ks(a, b):
    grid = np.union1d(a, b)
    fa = np.searchsorted(np.sort(a), grid, side="right") / a.size
    fb = np.searchsorted(np.sort(b), grid, side="right") / b.size
    return float(np.abs(fa - fb).max())


def check_against_reference(g, case, k_in, k_tg, n_in=None, n_tg=None):
    """k_in / k_tg: int arrays [n_mods, draws] from the sampler under test, same ranges as the fixture's `case`."""
    cap = g["max_tokens"].astype(np.int64)
    ref_in, ref_tg = g[f"{case}.k_in"].astype(np.int64), g[f"{case}.k_tgt"].astype(np.int64)
    k_in, k_tg = np.asarray(k_in, np.int64), np.asarray(k_tg, np.int64)
    n = min(k_in.shape[1], ref_in.shape[1])
    assert n >= 20000, n
    # per-draw invariants that hold for every reference draw hold for ours
    (lo_i, hi_i), (lo_t, hi_t) = g[f"{case}.range"]
    for a, t, tag in ((ref_in, ref_tg, "reference"), (k_in, k_tg, "ours")):
        assert (a >= 0).all() and (t >= 0).all(), tag
        assert (a <= cap[:, None]).all() and (a + t <= cap[:, None]).all(), tag       # targets take what the inputs left
        assert (a.sum(0) <= hi_i).all() and (t.sum(0) <= hi_t).all(), tag
        short = a.sum(0) < lo_i                                                       # fewer tokens than drawn: a clamp took some
        assert ((a == cap[:, None]).any(0) | ~short).all(), tag
    report = {}
    for side, ours, ref in (("in", k_in, ref_in), ("tgt", k_tg, ref_tg)):
        so, sr = stats(ours, cap), g[f"{case}.stats.{side}"]
        assert np.allclose(sr, stats(ref, cap))                                       # the fixture's stats are those of its draws
        se_mean = 4.0 * np.sqrt(2.0) * sr[:, 1] / np.sqrt(n) + 0.5
        assert np.all(np.abs(so[:, 0] - sr[:, 0]) < se_mean), (case, side, "mean", so[:, 0], sr[:, 0], se_mean)
        assert np.all(np.abs(so[:, 1] - sr[:, 1]) < 0.03 * sr[:, 1] + 0.5), (case, side, "std", so[:, 1], sr[:, 1])
        assert np.all(np.abs(so[:, 2:] - sr[:, 2:]) < 0.012), (case, side, "P(0) / P(cap) / one-hot", so[:, 2:], sr[:, 2:])
        d = [ks(ours[j], ref[j]) for j in range(ours.shape[0])]
        assert max(d) < 0.025, (case, side, "KS", d)
        report[side] = dict(mean=so[:, 0], ref_mean=sr[:, 0], ks=d)
    # joint law of (inputs, targets): the total kept per clip
    assert ks((k_in + k_tg).sum(0), (ref_in + ref_tg).sum(0)) < 0.025
    return report
