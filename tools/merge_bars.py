"""Fold the lines a GPU run wrote with EGOM2P_RECORD_BARS=<file> into tests/golden/parity_bars.json (the recorded parity
baselines `tests/conftest.py:bar` checks against).  A name measured more than once keeps its largest value.
usage: python tools/merge_bars.py gpurun_out/bars.jsonl [--reset]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "parity_bars.json")
bars = {} if "--reset" in sys.argv or not os.path.exists(path) else json.load(open(path))
new = {}
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    new[r["name"]] = max(new.get(r["name"], 0.0), r["value"])
bars.update(new)
json.dump(dict(sorted(bars.items())), open(path, "w"), indent=0)
print(f"{len(new)} measured, {len(bars)} baselines in {path}")
