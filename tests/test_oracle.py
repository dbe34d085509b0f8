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
    assert d(a, b) == d(a, b, "hyyro")


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


def test_golden_slices_reproducible(orc, sw):
    """The committed fixture = generator(seed 42) + oracle; regenerate and compare (catches drift in either)."""
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    for name, count in {"words16": 256, "tokens64": 256, "utf8_lines": 32, "short_words": 256}.items():
        a, b = sw.generate_pairs(name, count, seed=42)
        assert (a.data == z[f"{name}.a_data"]).all() and (a.offsets == z[f"{name}.a_offsets"]).all()
        assert (b.data == z[f"{name}.b_data"]).all() and (b.offsets == z[f"{name}.b_offsets"]).all()
        assert (orc.levenshtein_pairs(a, b) == z[f"{name}.lev_bytes"]).all()
    a, b = sw.generate_pairs("utf8_lines", 32, seed=42)
    assert (orc.levenshtein_pairs(a, b, utf8=True) == z["utf8_lines.lev_utf8"]).all()


def test_generator_is_sliceable(sw):
    """Pair i depends only on (workload, seed, i): rank shards reproduce the global stream."""
    whole_a, whole_b = sw.generate_pairs("tokens64", 3000, seed=7)
    part_a, part_b = sw.generate_pairs("tokens64", 1000, seed=7, first=2000)
    for i in (0, 1, 500, 999):
        assert part_a[i] == whole_a[2000 + i] and part_b[i] == whole_b[2000 + i]
    single_a, _ = sw.generate_pairs("tokens64", 3000, seed=7, threads=1)
    assert (single_a.data == whole_a.data).all()
