"""Generates tests/golden/slices.npz: the first pairs of every synthetic config with the
oracle's outputs, so the GPU box can check generator + kernels with no reference code present.

Run from the repo root:  python tests/golden/make_fixtures.py
Inputs come from stringwars_amd.generate_pairs (seed 42); expected outputs from oracle/ (Wagner-Fischer /
Gotoh restatements, cross-checked against the Hyyro bit-parallel implementation before writing).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import stringwars_amd as sw  # noqa: E402

SLICES = {"words16": 256, "tokens64": 256, "utf8_lines": 32, "protein4k": 6, "short_words": 256, "bytes4k": 4}
out = {}
for name, count in SLICES.items():
    a, b = sw.generate_pairs(name, count, seed=42)
    out[f"{name}.a_data"], out[f"{name}.a_offsets"] = a.data, a.offsets
    out[f"{name}.b_data"], out[f"{name}.b_offsets"] = b.data, b.offsets
    wf = oracle.levenshtein_pairs(a, b)
    assert (wf == oracle.levenshtein_pairs(a, b, algo="hyyro")).all()
    out[f"{name}.lev_bytes"] = wf
    if name == "utf8_lines":
        out[f"{name}.lev_utf8"] = oracle.levenshtein_pairs(a, b, utf8=True)
    if name in ("protein4k", "bytes4k"):
        alphabet = sw.synth.AMINO_ACIDS if name == "protein4k" else None
        matrix = sw.substitution_matrix(42, alphabet)
        out[f"{name}.matrix"] = matrix
        out[f"{name}.nw_linear_m4"] = oracle.nw_pairs(a, b, matrix, -4, -4)
        out[f"{name}.nw_affine_m11_m1"] = oracle.nw_pairs(a, b, matrix, -11, -1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "slices.npz"), **out)
print({k: v.shape for k, v in out.items() if not k.endswith("data")})
