"""Reader for the reference's on-disk token format (README_DATA.md:7-60; egom2p/data/unified_datasets.py:162-222).

Pre-computed tokens live in a "modified WebDataset" layout: one tar shard per modality holding the same sample keys,

    root/<modality folder>/<dataset>/shard-000000.tar          members  <key>.npz  (array 'arr_0': the token ids)
    data_path = 'root/[rgb,depth,cam,gaze]/holoassist/token/shard-{000000..000195}.tar'      (the reference's notation)

or, for small sets, plain directories `root/<modality>/<dataset>/<key>.npz`.  The reference streams them with the
`webdataset` package through CPU workers and masks each sample there; here a shard reader built on `tarfile` yields
aligned BATCHES `{tok_<modality>: int64 [B, ...]}` (pinned host memory), which go to the GPU as they are and are masked
there (`egom2p_amd.masking.UnifiedMasking`).  Shards are dealt to ranks round-robin (what `wds.split_by_node` does).
Token files are decoded with `numpy.load(allow_pickle=False)` only.
"""
from __future__ import annotations

import io
import os
import random
import re
import tarfile
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch


def _brace_expand(s: str) -> List[str]:
    """'{000..012}' numeric ranges (zero-padded like the bounds) and '{a,b}' lists, left to right."""
    m = re.search(r"\{([^{}]*)\}", s)
    if not m:
        return [s]
    body, out = m.group(1), []
    r = re.fullmatch(r"(\d+)\.\.(\d+)", body)
    if r:
        lo, hi, w = int(r.group(1)), int(r.group(2)), len(r.group(1))
        items = [str(i).zfill(w) for i in range(lo, hi + 1)]
    else:
        items = body.split(",")
    for it in items:
        out.extend(_brace_expand(s[:m.start()] + it + s[m.end():]))
    return out


def expand_data_path(data_path: str) -> Dict[str, List[str]]:
    """'root/[rgb,depth]/set/shard-{00..03}.tar' -> {'rgb': [4 paths], 'depth': [4 paths]} (bracket = modality folders)."""
    m = re.search(r"\[([^\[\]]*)\]", data_path)
    folders = m.group(1).split(",") if m else [None]
    out = {}
    for f in folders:
        p = data_path if f is None else data_path[:m.start()] + f + data_path[m.end():]
        out[f if f is not None else ""] = _brace_expand(p)
    return out


def _decode(raw: bytes) -> np.ndarray:
    with np.load(io.BytesIO(raw), allow_pickle=False) as z:
        return np.asarray(z["arr_0"] if "arr_0" in z.files else z[z.files[0]])


class TokenShards:
    """Iterable over aligned token batches.

    data_path: the reference's notation (brackets = modality folders, braces = shard range), or a directory root with
        `modalities` given (simple hierarchical format: root/<modality>/**/<key>.npz).
    rename: modality folder -> model modality name (default 'tok_<folder>', the reference's `rename_modalities`)."""

    def __init__(self, data_path: str, batch_size: int, rename: Optional[Dict[str, str]] = None, rank: int = 0, world: int = 1,
                 shuffle_seed: Optional[int] = None, drop_last: bool = True, modalities: Optional[Sequence[str]] = None,
                 pin_memory: bool = True, vocab: Optional[Dict[str, int]] = None):
        """shuffle_seed must be the SAME on every rank: the shard list is shuffled with (seed, epoch) and then dealt
        `order[rank::world]`, which is a partition only if every rank shuffles alike.  vocab: model modality name -> vocabulary
        size; every batch is range-checked against it on the host (ids index embedding tables on the device)."""
        self.batch, self.rank, self.world, self.seed, self.drop_last = int(batch_size), rank, world, shuffle_seed, drop_last
        self.vocab = dict(vocab or {})
        self.pin = pin_memory and torch.cuda.is_available()
        self.epoch = 0
        if os.path.isdir(data_path):
            assert modalities, "a directory root needs the modality folder names"
            self.kind = "dir"
            self.files = {f: sorted(os.path.join(dp, n) for dp, _, ns in os.walk(os.path.join(data_path, f)) for n in ns if n.endswith(".npz"))
                          for f in modalities}
            counts = {f: len(v) for f, v in self.files.items()}
            assert len(set(counts.values())) == 1 and next(iter(counts.values())) > 0, f"unaligned or empty modalities: {counts}"
        else:
            self.kind = "tar"
            self.files = expand_data_path(data_path)
            counts = {f: len(v) for f, v in self.files.items()}
            assert "" not in self.files and len(set(counts.values())) == 1, f"need '[mod_a,mod_b]' folders with equal shard counts: {counts}"
        self.folders = list(self.files)
        self.names = {f: (rename or {}).get(f, f if f.startswith("tok_") else f"tok_{f}") for f in self.folders}

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    # ---- samples: (key, {folder: ndarray}) in shard order, this rank's shards only
    def _samples(self) -> Iterator[Tuple[str, Dict[str, np.ndarray]]]:
        if self.kind == "dir":
            idx = list(range(len(self.files[self.folders[0]])))
            if self.seed is not None:
                random.Random(self.seed * 1000003 + self.epoch).shuffle(idx)
            for i in idx[self.rank::self.world]:
                keys = {f: os.path.splitext(os.path.basename(self.files[f][i]))[0] for f in self.folders}
                assert len(set(keys.values())) == 1, f"file names differ across modalities: {keys}"
                yield keys[self.folders[0]], {f: _decode(open(self.files[f][i], "rb").read()) for f in self.folders}
            return
        order = list(range(len(self.files[self.folders[0]])))
        if self.seed is not None:
            random.Random(self.seed * 1000003 + self.epoch).shuffle(order)
        for si in order[self.rank::self.world]:
            tars = {f: tarfile.open(self.files[f][si], "r") for f in self.folders}
            try:
                its = {f: (m for m in tars[f] if m.isfile()) for f in self.folders}
                while True:
                    members = {f: next(its[f], None) for f in self.folders}
                    if all(m is None for m in members.values()):
                        break
                    assert all(m is not None for m in members.values()), f"shard {si}: modalities hold different sample counts"
                    keys = {f: os.path.splitext(os.path.basename(m.name))[0] for f, m in members.items()}
                    assert len(set(keys.values())) == 1, f"shard {si}: sample keys differ across modalities: {keys}"
                    yield keys[self.folders[0]], {f: _decode(tars[f].extractfile(members[f]).read()) for f in self.folders}
            finally:
                for t in tars.values():
                    t.close()

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        buf: List[Dict[str, np.ndarray]] = []
        for _, sample in self._samples():
            buf.append(sample)
            if len(buf) == self.batch:
                yield self._collate(buf)
                buf = []
        if buf and not self.drop_last:
            yield self._collate(buf)

    def batches(self, steps: int) -> Iterator[Dict[str, torch.Tensor]]:
        """Exactly `steps` batches on EVERY rank, whatever its share of the shards holds: when the rank's shards are
        exhausted the pass restarts on the next epoch's shuffle (the reference's `wds.ResampledShards(...).with_epoch(n)`
        is likewise an endless stream cut to a fixed step count, unified_datasets.py:430-470).  Ranks that issue different
        numbers of steps would issue different numbers of gradient all-reduces and hang."""
        done, epoch0 = 0, self.epoch
        while done < steps:
            got = 0
            for b in self:
                yield b
                got += 1
                done += 1
                if done == steps:
                    break
            if got == 0:
                self.epoch = epoch0
                raise RuntimeError(f"TokenShards: rank {self.rank} of {self.world} cannot form one batch of {self.batch} samples "
                                   f"from its share of the shards")
            self.epoch += 1
        self.epoch = epoch0

    @staticmethod
    def _registry_vocab(name: str) -> int:
        from .model import MODALITY_INFO
        info = MODALITY_INFO.get(name)
        return int(info["vocab_size"]) if info is not None and "vocab_size" in info else 65536

    def _collate(self, buf) -> Dict[str, torch.Tensor]:
        out = {}
        for f in self.folders:
            arr = np.stack([b[f] for b in buf])
            if arr.dtype == np.int16:            # 16-bit files hold ids up to 65535 (vocabulary 64000): reinterpret, do not sign-extend
                arr = arr.view(np.uint16)
            arr = arr.astype(np.int64)           # `tok_to_int64`, unified_datasets.py:218-222
            name = self.names[f]
            lo, hi = int(arr.min()), int(arr.max())
            V = self.vocab.get(name)
            if V is None:
                # no vocabulary given for this modality: fall back to the registry's (a corrupt -1 in a 16-bit shard reads as
                # 65535 and would index a 64000-row device table out of bounds), else to the 16-bit key range of the tables
                V = self._registry_vocab(name)
            if lo < 0 or hi >= V:
                # nn.Embedding raises an index error in the reference; here ids index device tables directly
                raise ValueError(f"TokenShards: modality {name}: token ids in [{lo}, {hi}] outside [0, {V})")
            t = torch.from_numpy(arr)
            out[name] = t.pin_memory() if self.pin else t
        return out
