"""The token-shard reader (egom2p_amd/data.py) on the reference's on-disk layout (README_DATA.md:7-60): shards written
here with numpy + tarfile in that layout, read back aligned, in order, dealt to ranks round-robin.  CPU only."""
import io
import os
import tarfile

import numpy as np
import pytest
import torch

from egom2p_amd.data import TokenShards, expand_data_path


def _write(root, folders, n_shards, per_shard, shape_of):
    rng = np.random.default_rng(0)
    truth = {}
    for s in range(n_shards):
        keys = [f"clip_{s:02d}_{i:03d}" for i in range(per_shard)]
        for f in folders:
            d = os.path.join(root, f, "setA", "token")
            os.makedirs(d, exist_ok=True)
            with tarfile.open(os.path.join(d, f"shard-{s:06d}.tar"), "w") as tar:
                for k in keys:
                    arr = rng.integers(0, 64000 if f in ("rgb", "depth") else 256, size=shape_of[f]).astype(np.int32)
                    truth[(f, k)] = arr
                    b = io.BytesIO()
                    np.savez(b, arr)
                    info = tarfile.TarInfo(f"{k}.npz")
                    info.size = b.getbuffer().nbytes
                    b.seek(0)
                    tar.addfile(info, b)
    return truth


def test_path_notation():
    p = expand_data_path("/d/[rgb,cam]/holo/token/shard-{000008..000011}.tar")
    assert list(p) == ["rgb", "cam"] and p["cam"][0] == "/d/cam/holo/token/shard-000008.tar" and len(p["rgb"]) == 4
    assert p["rgb"][-1].endswith("shard-000011.tar")


def test_shards_are_read_aligned_and_dealt_to_ranks(tmp_path):
    folders = ["rgb", "cam", "gaze"]
    shape = {"rgb": (5, 32, 32), "cam": (30,), "gaze": (30,)}
    truth = _write(str(tmp_path), folders, n_shards=4, per_shard=5, shape_of=shape)
    path = f"{tmp_path}/[rgb,cam,gaze]/setA/token/shard-{{000000..000003}}.tar"
    ds = TokenShards(path, batch_size=4, pin_memory=False)
    batches = list(ds)
    assert len(batches) == 5 and set(batches[0]) == {"tok_rgb", "tok_cam", "tok_gaze"}            # 20 samples, drop_last
    assert batches[0]["tok_rgb"].shape == (4, 5, 32, 32) and batches[0]["tok_rgb"].dtype == torch.int64
    # first batch = the first four samples of shard 0, the same key in every modality
    for i in range(4):
        for f in folders:
            assert np.array_equal(batches[0][f"tok_{f}"][i].numpy(), truth[(f, f"clip_00_{i:03d}")])
    # two ranks: shards 0, 2 and 1, 3 - together every sample exactly once
    seen = []
    for r in range(2):
        for b in TokenShards(path, batch_size=5, rank=r, world=2, pin_memory=False):
            seen.extend(int(x.sum()) for x in b["tok_cam"])
    want = sorted(int(v.sum()) for (f, k), v in truth.items() if f == "cam")
    assert sorted(seen) == want
    # shuffled shard order is a permutation, reproducible per (seed, epoch)
    a = [int(b["tok_cam"].sum()) for b in TokenShards(path, batch_size=5, shuffle_seed=3, pin_memory=False)]
    b2 = [int(b["tok_cam"].sum()) for b in TokenShards(path, batch_size=5, shuffle_seed=3, pin_memory=False)]
    assert a == b2 and sorted(a) == sorted(int(b["tok_cam"].sum()) for b in TokenShards(path, batch_size=5, pin_memory=False))


def test_misaligned_shards_are_refused(tmp_path):
    _write(str(tmp_path), ["cam"], 1, 3, {"cam": (30,)})
    # a gaze shard with different keys
    d = os.path.join(str(tmp_path), "gaze", "setA", "token")
    os.makedirs(d)
    with tarfile.open(os.path.join(d, "shard-000000.tar"), "w") as tar:
        for k in ("other_0", "other_1", "other_2"):
            b = io.BytesIO()
            np.savez(b, np.zeros(30, np.int32))
            info = tarfile.TarInfo(f"{k}.npz")
            info.size = b.getbuffer().nbytes
            b.seek(0)
            tar.addfile(info, b)
    with pytest.raises(AssertionError):
        list(TokenShards(f"{tmp_path}/[cam,gaze]/setA/token/shard-{{000000..000000}}.tar", batch_size=1, pin_memory=False))


def test_directory_layout(tmp_path):
    for f in ("cam", "gaze"):
        os.makedirs(tmp_path / f / "setB")
        for i in range(6):
            np.savez(tmp_path / f / "setB" / f"s{i:02d}.npz", np.full(30, i, np.int32))
    got = list(TokenShards(str(tmp_path), batch_size=3, modalities=["cam", "gaze"], pin_memory=False))
    assert len(got) == 2 and torch.equal(got[1]["tok_gaze"][:, 0], torch.tensor([3, 4, 5]))


def test_two_ranks_uneven_shards_are_disjoint_and_step_matched(tmp_path):
    """with a rank-independent shuffle seed the two ranks' shards are a partition of the set (5 shards: 3 + 2),
    and `batches(steps)` yields the same number of batches on both ranks although their shares differ."""
    folders = ["rgb", "cam"]
    shape = {"rgb": (5, 32, 32), "cam": (30,)}
    truth = _write(str(tmp_path), folders, n_shards=5, per_shard=4, shape_of=shape)
    path = f"{tmp_path}/[rgb,cam]/setA/token/shard-{{000000..000004}}.tar"
    per_rank = []
    for r in range(2):
        ds = TokenShards(path, batch_size=2, rank=r, world=2, shuffle_seed=7, pin_memory=False, vocab={"tok_rgb": 64000, "tok_cam": 256})
        ds.set_epoch(3)
        one_pass = [int(x.sum()) for b in ds for x in b["tok_cam"]]
        per_rank.append(one_pass)
        assert len(list(ds.batches(9))) == 9                        # more steps than the smaller share holds (2 shards = 4 batches)
        assert ds.epoch == 3
    assert len(per_rank[0]) == 12 and len(per_rank[1]) == 8         # 3 shards vs 2 shards of 4 samples
    want = sorted(int(v.sum()) for (f, k), v in truth.items() if f == "cam")
    assert sorted(per_rank[0] + per_rank[1]) == want                # disjoint and complete
    # different seeds per rank (what run_training passed before) would NOT be a partition - the reason for the rule
    a = [int(x.sum()) for b in TokenShards(path, 2, rank=0, world=2, shuffle_seed=1, pin_memory=False) for x in b["tok_cam"]]
    b = [int(x.sum()) for b in TokenShards(path, 2, rank=1, world=2, shuffle_seed=2, pin_memory=False) for x in b["tok_cam"]]
    assert sorted(a + b) != want


def test_out_of_range_and_16_bit_token_ids(tmp_path):
    """Token ids index device tables directly: ids outside [0, vocab) raise on the host (the reference's nn.Embedding raises);
    int16 files are read as unsigned 16-bit (ids up to 63999 do not fit int16)."""
    root = str(tmp_path)
    for f, arr in (("rgb", np.full((5, 32, 32), 63999, np.uint16).view(np.int16)), ("cam", np.arange(30, dtype=np.int16))):
        os.makedirs(os.path.join(root, f), exist_ok=True)
        for i in range(2):
            np.savez(os.path.join(root, f, f"k{i}.npz"), arr)
    b = next(iter(TokenShards(root, 2, modalities=["rgb", "cam"], pin_memory=False, vocab={"tok_rgb": 64000, "tok_cam": 256})))
    assert int(b["tok_rgb"].max()) == 63999 and int(b["tok_rgb"].min()) == 63999
    with pytest.raises(ValueError, match="tok_cam"):
        next(iter(TokenShards(root, 2, modalities=["rgb", "cam"], pin_memory=False, vocab={"tok_rgb": 64000, "tok_cam": 16})))
    np.savez(os.path.join(root, "cam", "k1.npz"), np.full(30, -3, np.int32))
    with pytest.raises(ValueError, match="tok_cam"):
        next(iter(TokenShards(root, 2, modalities=["rgb", "cam"], pin_memory=False)))
    # without a `vocab` argument the registry's vocabulary applies: a corrupt -1 in a 16-bit rgb shard reads as 65535 and must not
    # reach the 64000-row device table
    np.savez(os.path.join(root, "cam", "k1.npz"), np.arange(30, dtype=np.int16))
    np.savez(os.path.join(root, "rgb", "k1.npz"), np.full((5, 32, 32), -1, np.int16))
    with pytest.raises(ValueError, match="tok_rgb"):
        next(iter(TokenShards(root, 2, modalities=["rgb", "cam"], pin_memory=False)))
