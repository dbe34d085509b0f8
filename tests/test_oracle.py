"""CPU tests of the oracle itself: known answers, two independent algorithms, metric properties.

The reference has no tests for this path, so these pin the oracle instead: the
textbook KAT table (tests/golden/kat.json), Wagner-Fischer == Hyyro bit-parallel across the 64-bit
word boundaries, and hypothesis properties of a metric.
"""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLDEN, "kat.json"), encoding="utf-8"))


def unary_matrix(match, mismatch):
    m = np.full((256, 256), mismatch, np.int8)
    np.fill_diagonal(m, match)
    return m


@pytest.mark.parametrize("a,b,codepoints,nbytes", KAT["levenshtein"])
def test_kat_levenshtein(orc, a, b, codepoints, nbytes):
    assert orc.levenshtein_utf8(a, b) == codepoints
    assert orc.levenshtein(a, b) == nbytes
    assert orc.levenshtein(a, b, "hyyro") == nbytes


@pytest.mark.parametrize("a,b,k,expected", KAT["bounded"])
def test_kat_bounded(orc, sw, a, b, k, expected):
    ta, tb = sw.Strs([a]), sw.Strs([b])
    assert orc.levenshtein_pairs(ta, tb, bound=k)[0] == expected


@pytest.mark.parametrize("a,b,linear,affine", KAT["nw_unary_2_m1"]["cases"])
def test_kat_nw(orc, a, b, linear, affine):
    m = unary_matrix(2, -1)
    assert orc.nw_score(a, b, m, -2, -2) == linear
    assert orc.nw_score(a, b, m, -5, -1) == affine


@pytest.mark.parametrize("a,b,twin_linear,twin_affine", KAT["nw_bio_twins"]["cases"])
def test_kat_nw_bio_convention(orc, a, b, twin_linear, twin_affine):
    m = unary_matrix(2, -1)
    assert orc.nw_score(a, b, m, -4, -2) == twin_linear
    assert orc.nw_score(a, b, m, -6, -1) == twin_affine


def test_kat_nw_classic(orc):
    a, b, match, mismatch, open_, extend, expected = KAT["nw_classic"][0]
    assert orc.nw_score(a, b, unary_matrix(match, mismatch), open_, extend) == expected


@pytest.mark.parametrize("a,b,linear,affine", KAT["sw_unary_2_m1"]["cases"])
def test_kat_sw(orc, a, b, linear, affine):
    m = unary_matrix(2, -1)
    assert orc.nw_score(a, b, m, -2, -2, local=True) == linear
    assert orc.nw_score(a, b, m, -5, -1, local=True) == affine


@pytest.mark.parametrize("alphabet", [2, 4, 26, 256])
def test_two_implementations_agree(orc, sw, alphabet):
    rng = np.random.default_rng(alphabet)
    lengths = list(range(0, 20)) + [31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 300]
    items_a, items_b = [], []
    for _ in range(6):
        for la in lengths:
            lb = int(rng.choice(lengths))
            a = rng.integers(0, alphabet, la, dtype=np.uint8).tobytes()
            b = bytearray(a[:lb] if rng.random() < 0.5 else rng.integers(0, alphabet, lb, dtype=np.uint8).tobytes())
            for _ in range(int(rng.integers(0, 5))):
                if b:
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, alphabet))
            items_a.append(a)
            items_b.append(bytes(b))
    ta, tb = sw.Strs(items_a), sw.Strs(items_b)
    assert (orc.levenshtein_pairs(ta, tb) == orc.levenshtein_pairs(ta, tb, algo="hyyro")).all()


@pytest.mark.parametrize("alphabet", [2, 4, 26, 256])
def test_wagner_fischer_vs_hyyro_one_million_pairs(orc, alphabet):
    """two independent CPU implementations agree on >= 1e6 random pairs (4 x 250,000), alphabets
    {2, 4, 26, 256}, lengths 0-300 incl. the 63/64/65, 127/128/129 ... word boundaries; both argument orders; every
    16th pair also against the anti-diagonal full table. The loop runs in C (oracle.c: orc_selfcheck_levenshtein)."""
    bad, first, cells = orc.selfcheck("levenshtein", 20260101, 250_000, alphabet, 300)
    assert bad == 0, f"first disagreement at case {first}"
    assert cells > 1_000_000_000


@pytest.mark.parametrize("alphabet", [2, 4, 20, 256])
def test_gotoh_vs_general_gap_table(orc, alphabet):
    """The NW / SW oracle (Gotoh, rolling rows) against a second implementation that shares only the definition:
    the Waterman-Smith-Beyer general-gap table (cubic, every gap length spelled out). 4 x 30,000 = 120,000 random
    cases: symmetric and asymmetric i8 matrices, gaps with |open| >= |extend|, lengths 0-40, global and local."""
    bad, first, cells = orc.selfcheck("alignment", 20260102, 30_000, alphabet, 40)
    assert bad == 0, f"first disagreement at case {first}"
    assert cells > 10_000_000


def _enumerate_alignments(a, b, matrix, open_, extend):
    """Best score over ALL alignments of a and b, by brute force: every monotone path of diagonal / down / right
    moves, maximal runs of down (right) moves scored as one gap of that length. No dynamic programming."""
    best = [None]

    def walk(i, j, score, run_kind, run_len):
        def close(sc):
            return sc + (open_ + (run_len - 1) * extend if run_len else 0)
        if i == len(a) and j == len(b):
            total = close(score)
            if best[0] is None or total > best[0]:
                best[0] = total
            return
        if i < len(a) and j < len(b):
            walk(i + 1, j + 1, close(score) + int(matrix[a[i], b[j]]), 0, 0)
        if i < len(a):
            if run_kind == 1:
                walk(i + 1, j, score, 1, run_len + 1)
            else:
                walk(i + 1, j, close(score), 1, 1)
        if j < len(b):
            if run_kind == 2:
                walk(i, j + 1, score, 2, run_len + 1)
            else:
                walk(i, j + 1, close(score), 2, 1)

    walk(0, 0, 0, 0, 0)
    return best[0]


def test_alignment_scores_by_exhaustive_enumeration(orc):
    """Third anchor for NW / SW: exhaustive enumeration of every alignment of strings of up to 5 (global) / 4 (local:
    every pair of substrings) symbols over a 3-letter alphabet, against both oracle implementations."""
    rng = np.random.default_rng(11)
    for case in range(160):
        matrix = np.zeros((256, 256), np.int8)
        sub = rng.integers(-6, 7, (3, 3))
        if case % 2 == 0:
            sub = np.minimum(sub, sub.T)
        matrix[:3, :3] = sub
        extend = -int(rng.integers(0, 4))
        open_ = extend - int(rng.integers(0, 6))
        a = rng.integers(0, 3, int(rng.integers(0, 6)), dtype=np.uint8)
        b = rng.integers(0, 3, int(rng.integers(0, 6)), dtype=np.uint8)
        want = _enumerate_alignments(a, b, matrix, open_, extend)
        assert orc.nw_score(a, b, matrix, open_, extend) == want
        assert orc.align_score_general(a, b, matrix, open_, extend) == want
        if case % 4 == 0:
            a, b = a[:4], b[:4]
            local = 0
            for i0 in range(len(a) + 1):
                for i1 in range(i0, len(a) + 1):
                    for j0 in range(len(b) + 1):
                        for j1 in range(j0, len(b) + 1):
                            local = max(local, _enumerate_alignments(a[i0:i1], b[j0:j1], matrix, open_, extend))
            assert orc.nw_score(a, b, matrix, open_, extend, local=True) == local
            assert orc.align_score_general(a, b, matrix, open_, extend, local=True) == local


small = st.binary(max_size=24)


@settings(max_examples=200, deadline=None)
@given(small, small, small)
def test_metric_properties(orc, a, b, c):
    d = orc.levenshtein
    assert d(a, a) == 0
    assert d(a, b) == d(b, a)
    assert abs(len(a) - len(b)) <= d(a, b) <= max(len(a), len(b))
    assert d(a, c) <= d(a, b) + d(b, c)
    assert d(a, a + c) == len(c)
    assert d(a, b) == d(a, b, "hyyro") == orc.levenshtein_antidiagonal(a, b)


@settings(max_examples=100, deadline=None)
@given(small, small)
def test_nw_relates_to_levenshtein(orc, a, b):
    # NW with (0, -1) substitution and -1 gaps is the negated distance; general-cost DP with unit costs too.
    assert orc.nw_score(a, b, unary_matrix(0, -1), -1, -1) == -orc.levenshtein(a, b)
    assert orc.levenshtein_costs(a, b, 0, 1, 1, 1) == orc.levenshtein(a, b)
    m = unary_matrix(2, -1)
    assert orc.nw_score(a, b, m, -5, -1) == orc.nw_score(b, a, m, -5, -1)


@settings(max_examples=100, deadline=None)
@given(st.text(max_size=16), st.text(max_size=16))
def test_utf8_code_points(orc, a, b):
    assert list(orc.utf8_decode(a)) == [ord(ch) for ch in a]
    if a.isascii() and b.isascii():
        assert orc.levenshtein_utf8(a, b) == orc.levenshtein(a, b)


@pytest.mark.parametrize("bad", [b"\xff", b"\xc0\x80", b"\xe0\x80\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xe2\x82", b"\x80"])
def test_invalid_utf8_rejected(orc, bad):
    with pytest.raises(ValueError):
        orc.utf8_decode(bad)


def _digest(a, b) -> str:
    import hashlib
    h = hashlib.sha256()
    for array in (a.data, a.offsets, b.data, b.offsets):
        h.update(np.ascontiguousarray(array).tobytes())
    return h.hexdigest()


def test_golden_script_lines_reproducible(orc, sw):
    """tests/golden/script_lines.npz = generator(seed 42) + oracle: the digest pins the `script_lines` generator (unrelated article lines, one
    script each), the stored distances are the oracle's over code points and over bytes, by both of its Levenshtein routines."""
    z = np.load(os.path.join(GOLDEN, "script_lines.npz"))
    a, b = sw.generate_pairs("script_lines", 64, seed=42)
    assert _digest(a, b) == bytes(z["n64.sha256"]).decode()
    head = sw.Strs(data=z["a_data"], offsets=z["a_offsets"])
    assert all(head[i] == a[i] for i in range(8))
    for utf8, key in ((True, "n64.lev_utf8"), (False, "n64.lev_bytes")):
        assert (orc.levenshtein_pairs(a, b, utf8=utf8, count=16) == z[key][:16]).all()
        assert (orc.levenshtein_pairs(a, b, utf8=utf8, algo="hyyro") == z[key]).all()
    lines = [a[i].decode() for i in range(64)] + [b[i].decode() for i in range(64)]
    assert all(700 <= len(line) <= 1300 for line in lines)
    assert max(len(set(line)) for line in lines) < 120        # one script and the common ASCII: what a 251-slot dictionary takes with room to spare


def test_golden_slices_reproducible(orc, sw):
    """The committed fixture = generator(seed 42) + oracle; regenerate and compare (catches drift in either).
    All six workloads: the SHA-256 of the first 256 generated pairs; oracle outputs for the word-sized ones at 256
    pairs and for the KB-sized ones on their verbatim prefix (the 4 KB alignments at 256 pairs take minutes on a CPU:
    they are compared with the kernels' outputs in the -m gpu suite)."""
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    for name in ("words16", "tokens64", "utf8_lines", "protein4k", "short_words", "bytes4k"):
        a, b = sw.generate_pairs(name, 256, seed=42)
        assert _digest(a, b) == bytes(z[f"{name}.n256.sha256"]).decode()
        prefix = len(z[f"{name}.a_offsets"]) - 1
        head = sw.Strs(data=z[f"{name}.a_data"], offsets=z[f"{name}.a_offsets"])
        assert all(head[i] == a[i] for i in range(prefix))
        assert (z[f"{name}.n256.lev_bytes"][:prefix] == z[f"{name}.lev_bytes"]).all()
    for name in ("words16", "tokens64", "short_words"):
        a, b = sw.generate_pairs(name, 256, seed=42)
        assert (orc.levenshtein_pairs(a, b) == z[f"{name}.n256.lev_bytes"]).all()
    a, b = sw.generate_pairs("utf8_lines", 32, seed=42)
    assert (orc.levenshtein_pairs(a, b, utf8=True) == z["utf8_lines.n256.lev_utf8"][:32]).all()
    # the reference's alignment rows on word-sized tokens: unary_class_costs(2, -1), linear -2 / -2 and affine -5 / -1, NW and SW
    a, b = sw.generate_pairs("words16", 256, seed=42)
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    unary = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
    for tag, gaps in (("linear_m2", (-2, -2)), ("affine_m5_m1", (-5, -1))):
        assert (orc.nw_pairs(a, b, unary, *gaps) == z[f"words16.n256.nw_unary_{tag}"]).all()
        for kind, local in (("nw", False), ("sw", True)):
            assert [orc.nw_score(a[i], b[i], unary, *gaps, local=local) for i in range(0, 256, 5)] == z[f"words16.n256.{kind}_unary_{tag}"][::5].tolist()
            assert [[orc.nw_score(a[i], b[j], unary, *gaps, local=local) for j in range(16)] for i in range(16)] == z[f"words16.cross16.{kind}_unary_{tag}"].tolist()
    a, b = sw.generate_pairs("protein4k", 2, seed=42)
    assert (orc.nw_pairs(a, b, z["protein4k.matrix"], -4, -4) == z["protein4k.n256.nw_linear_m4"][:2]).all()
    assert orc.nw_score(a[0], b[0], z["protein4k.matrix"], -11, -1, local=True) == z["protein4k.n256.sw_affine_m11_m1"][0]


def test_golden_multilingual_words_reproducible(orc, sw):
    """tests/golden/uwords.npz against the oracle: every matrix of the committed fixture recomputed from its committed inputs (Levenshtein
    over bytes by both CPU routines and over code points; NW / SW with the reference's unary class costs, linear and affine; a few
    entries also by the cubic general-gap table)."""
    z = np.load(os.path.join(GOLDEN, "uwords.npz"))
    q = sw.Strs(data=z["q_data"], offsets=z["q_offsets"])
    c = sw.Strs(data=z["c_data"], offsets=z["c_offsets"])
    queries, candidates = [q[i] for i in range(len(q))], [c[j] for j in range(len(c))]
    assert any(len(x) > 64 for x in queries + candidates) and any(b > 0x7F for x in queries for b in x)   # long tokens, several scripts
    assert [[orc.levenshtein(x, y) for y in candidates] for x in queries] == z["lev_bytes"].tolist()
    assert [[orc.levenshtein(x, y, algo="hyyro") for y in candidates] for x in queries] == z["lev_bytes"].tolist()
    assert [[orc.levenshtein_utf8(x, y) for y in candidates] for x in queries] == z["lev_utf8"].tolist()
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    unary = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
    for tag, gaps in (("linear_m2", (-2, -2)), ("affine_m5_m1", (-5, -1))):
        for kind, local in (("nw", False), ("sw", True)):
            assert [[orc.nw_score(x, y, unary, *gaps, local=local) for y in candidates] for x in queries] == z[f"{kind}_unary_{tag}"].tolist()
            assert orc.align_score_general(queries[2], candidates[7], unary, *gaps, local=local) == z[f"{kind}_unary_{tag}"][2, 7]


def test_generator_is_sliceable(sw):
    """Pair i depends only on (workload, seed, i): rank shards reproduce the global stream."""
    whole_a, whole_b = sw.generate_pairs("tokens64", 3000, seed=7)
    part_a, part_b = sw.generate_pairs("tokens64", 1000, seed=7, first=2000)
    for i in (0, 1, 500, 999):
        assert part_a[i] == whole_a[2000 + i] and part_b[i] == whole_b[2000 + i]
    single_a, _ = sw.generate_pairs("tokens64", 3000, seed=7, threads=1)
    assert (single_a.data == whole_a.data).all()
