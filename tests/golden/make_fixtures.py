"""Generates tests/golden/slices.npz: the first 256 pairs of every synthetic config with the
oracle's outputs, so the GPU box can check generator + kernels with no reference code present.

Run from the repo root:  python tests/golden/make_fixtures.py      (about two minutes of CPU)

Inputs come from stringwars_amd.generate_pairs (seed 42); expected outputs from oracle/ (Wagner-Fischer and
Gotoh restatements; Levenshtein cross-checked against the Hyyro bit-parallel implementation, alignment scores of
the first pairs against the general-gap table, before anything is written).

Layout. The word-sized workloads (words16, tokens64, short_words) carry their 256 input pairs verbatim. The
KB-sized ones (utf8_lines ~1 KB, protein4k / bytes4k ~4 KB per string: 0.5-2 MB of incompressible text per
workload) carry the inputs of a short prefix verbatim (`<name>.a_data` ...: 32 / 6 / 4 pairs, as in round 1) and,
for the full 256 pairs, a SHA-256 over the generated tapes (`<name>.n256.sha256`) next to the 256 expected outputs
(`<name>.n256.*`): a test regenerates the 256 pairs, checks the digest -- which pins the generator -- and then
compares the kernels' outputs with the stored vectors.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import stringwars_amd as sw  # noqa: E402

FULL = 256
VERBATIM = {"words16": 256, "tokens64": 256, "utf8_lines": 32, "protein4k": 6, "short_words": 256, "bytes4k": 4}


def tape_digest(a, b) -> str:
    h = hashlib.sha256()
    for array in (a.data, a.offsets, b.data, b.offsets):
        h.update(np.ascontiguousarray(array).tobytes())
    return h.hexdigest()


def multilingual_words():
    """tests/golden/uwords.npz: word-sized tokens of four scripts (1 .. 3-byte sequences, ~5 code points each) with a few long ones among
    them -- 24 queries x 40 candidates -- and what the reference's engines would report for their cross-product: Levenshtein over bytes and
    over code points (`LevenshteinDistances` / `LevenshteinDistancesUtf8`, bench.rs:382-399), NW / SW with `unary_class_costs(2, -1)`, linear
    -2 / -2 and affine -5 / -1 (bench.rs:640, :655, :966). The shapes this round's routes were built for: code-point cross-products of
    words, tokens of up to 64 bytes, a batch whose few longer tokens are scored apart."""
    rng = np.random.default_rng(4242)
    cps = np.array([0x61, 0x65, 0x6F, 0x74, 0xE9, 0xFC, 0x430, 0x435, 0x43E, 0x442, 0x4E2D, 0x6587, 0x65E5, 0x672C], dtype=np.uint32)

    def token(n):
        return "".join(chr(int(c)) for c in cps[rng.integers(0, len(cps), int(n))]).encode()

    queries = [token(n) for n in list(np.clip(rng.poisson(4.0, 21) + 1, 1, 14)) + [0, 28, 61]]
    candidates = [token(n) for n in list(np.clip(rng.poisson(4.0, 36) + 1, 1, 14)) + [0, 22, 30, 140]]
    candidates[5] = queries[3] + token(2)
    q, c = sw.Strs(queries), sw.Strs(candidates)
    out = {"q_data": q.data, "q_offsets": q.offsets, "c_data": c.data, "c_offsets": c.offsets}
    out["lev_bytes"] = np.array([[oracle.levenshtein(x, y) for y in candidates] for x in queries], dtype=np.int64)
    assert (out["lev_bytes"] == np.array([[oracle.levenshtein(x, y, algo="hyyro") for y in candidates] for x in queries])).all()
    out["lev_utf8"] = np.array([[oracle.levenshtein_utf8(x, y) for y in candidates] for x in queries], dtype=np.int64)
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    unary = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
    for tag, (open_, extend) in {"linear_m2": (-2, -2), "affine_m5_m1": (-5, -1)}.items():
        for kind, local in (("nw", False), ("sw", True)):
            scores = np.array([[oracle.nw_score(x, y, unary, open_, extend, local=local) for y in candidates] for x in queries], dtype=np.int64)
            for i in range(0, 24, 5):   # the cubic second implementation on a few rows
                for j in range(0, 40, 7):
                    assert scores[i, j] == oracle.align_score_general(queries[i], candidates[j], unary, open_, extend, local=local)
            out[f"{kind}_unary_{tag}"] = scores
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "uwords.npz"), **out)
    print({k: v.shape for k, v in out.items()})


def script_lines():
    """tests/golden/script_lines.npz: the first 64 pairs of the `script_lines` workload -- unrelated article lines of 700 ... 1300 code points,
    one script each: what the reference's cross-product of XLSum lines pairs up for `LevenshteinDistancesUtf8` (bench.rs:386-399,
    similarities/README.md:18, :39-40) -- as a SHA-256 over the generated tapes, the first 8 pairs verbatim, and the oracle's distances
    over code points and over bytes (Wagner-Fischer, cross-checked against the Hyyro restatement before anything is written)."""
    count, verbatim = 64, 8
    a, b = sw.generate_pairs("script_lines", count, seed=42)
    head_a, head_b = sw.generate_pairs("script_lines", verbatim, seed=42)
    out = {"a_data": head_a.data, "a_offsets": head_a.offsets, "b_data": head_b.data, "b_offsets": head_b.offsets,
           "n64.sha256": np.frombuffer(tape_digest(a, b).encode(), dtype=np.uint8)}
    out["n64.lev_utf8"] = oracle.levenshtein_pairs(a, b, utf8=True)
    assert (out["n64.lev_utf8"] == oracle.levenshtein_pairs(a, b, utf8=True, algo="hyyro")).all()
    out["n64.lev_bytes"] = oracle.levenshtein_pairs(a, b)
    assert (out["n64.lev_bytes"] == oracle.levenshtein_pairs(a, b, algo="hyyro")).all()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "script_lines.npz"), **out)
    print({k: v.shape for k, v in out.items()})


def main():
    if sys.argv[1:] == ["script_lines"]:   # (this fixture alone: the others are not rewritten)
        return script_lines()
    multilingual_words()
    script_lines()
    out = {}
    for name, verbatim in VERBATIM.items():
        a, b = sw.generate_pairs(name, FULL, seed=42)
        head_a, head_b = sw.generate_pairs(name, verbatim, seed=42)
        out[f"{name}.a_data"], out[f"{name}.a_offsets"] = head_a.data, head_a.offsets
        out[f"{name}.b_data"], out[f"{name}.b_offsets"] = head_b.data, head_b.offsets
        wf = oracle.levenshtein_pairs(a, b)
        assert (wf == oracle.levenshtein_pairs(a, b, algo="hyyro")).all()
        out[f"{name}.lev_bytes"] = wf[:verbatim]
        out[f"{name}.n256.sha256"] = np.frombuffer(tape_digest(a, b).encode(), dtype=np.uint8)
        out[f"{name}.n256.lev_bytes"] = wf
        if name == "utf8_lines":
            cp = oracle.levenshtein_pairs(a, b, utf8=True)
            out[f"{name}.lev_utf8"] = cp[:verbatim]
            out[f"{name}.n256.lev_utf8"] = cp
        if name == "words16":
            # The reference's alignment rows on its default `words` tokens (bench.rs:271): unary_class_costs(2, -1) folded into the
            # 32-class table (bench.rs:98-108, :655), linear gaps -2 / -2 (bench.rs:640) and affine -5 / -1 (bench.rs:966), global and
            # local -- pairwise over the 256 pairs, and the 16 x 16 cross-product of the first 16 a-strings with the first 16 b-strings
            byte_to_class, class_costs = sw.unary_class_costs(2, -1)
            unary = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
            for tag, (open_, extend) in {"linear_m2": (-2, -2), "affine_m5_m1": (-5, -1)}.items():
                for kind, local in (("nw", False), ("sw", True)):
                    pairs = np.array([oracle.nw_score(a[i], b[i], unary, open_, extend, local=local) for i in range(FULL)], dtype=np.int64)
                    if not local:
                        assert (pairs == oracle.nw_pairs(a, b, unary, open_, extend)).all()
                    for i in range(64):   # the cubic second implementation
                        assert pairs[i] == oracle.align_score_general(a[i], b[i], unary, open_, extend, local=local)
                    out[f"{name}.n256.{kind}_unary_{tag}"] = pairs
                    out[f"{name}.cross16.{kind}_unary_{tag}"] = np.array(
                        [[oracle.nw_score(a[i], b[j], unary, open_, extend, local=local) for j in range(16)] for i in range(16)], dtype=np.int64)
        if name in ("protein4k", "bytes4k"):
            alphabet = sw.synth.AMINO_ACIDS if name == "protein4k" else None
            matrix = sw.substitution_matrix(42, alphabet)
            out[f"{name}.matrix"] = matrix
            for tag, (open_, extend) in {"linear_m4": (-4, -4), "affine_m11_m1": (-11, -1)}.items():
                nw = oracle.nw_pairs(a, b, matrix, open_, extend)
                out[f"{name}.nw_{tag}"] = nw[:verbatim]
                out[f"{name}.n256.nw_{tag}"] = nw
                out[f"{name}.n256.sw_{tag}"] = np.array(
                    [oracle.nw_score(a[i], b[i], matrix, open_, extend, local=True) for i in range(FULL)], dtype=np.int64)
                # the cubic second implementation on prefixes of the first pairs (whole 4 KB strings would take hours)
                for i in range(4):
                    pa, pb = a[i][:300], b[i][:280]
                    assert oracle.nw_score(pa, pb, matrix, open_, extend) == oracle.align_score_general(pa, pb, matrix, open_, extend)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "slices.npz"), **out)
    print({k: v.shape for k, v in out.items() if not k.endswith("data")})


if __name__ == "__main__":
    main()
