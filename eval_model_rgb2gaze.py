#!/usr/bin/env python
"""rgb2gaze generation on MI355X: the flow of the reference's `eval_model_rgb2gaze.py` on the HIP engine (see
egom2p_amd/eval_generation.py for the shared body and what stays outside the hot-path scope).

    python eval_model_rgb2gaze.py [--ckpt checkpoint-main.pth] [--tokens clip.npz] [--out tokens.npz] [--batch B] [--bench 5]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from egom2p_amd.eval_generation import main  # noqa: E402

if __name__ == "__main__":
    main("rgb2gaze")
