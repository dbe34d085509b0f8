"""GPU parity tests: every call goes through the C ABI (libstringwars_amd.so) and is compared with the
CPU oracle on the same inputs, bit-exact. Run on a real MI355X: `pytest tests -m gpu`."""
import json
import os

import numpy as np
import pytest

from conftest import TEST_LIBRARY_ENV, child_pythonpath, run_in_child

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLDEN, "kat.json"), encoding="utf-8"))
ALGORITHMS = ["auto", "wavefront", "bitparallel", "tiled"]


def unary_matrix(match, mismatch):
    m = np.full((256, 256), mismatch, np.int8)
    np.fill_diagonal(m, match)
    return m


def random_pairs(rng, count, lengths, alphabet, related=0.5):
    items_a, items_b = [], []
    for _ in range(count):
        la, lb = int(rng.choice(lengths)), int(rng.choice(lengths))
        a = rng.integers(0, alphabet, la, dtype=np.uint8)
        if rng.random() < related and la:
            b = bytearray(a.tobytes())
            for _ in range(int(rng.integers(0, 6))):
                op = int(rng.integers(0, 3))
                pos = int(rng.integers(0, len(b) + (op == 1))) if len(b) + (op == 1) else 0
                if op == 0 and b:
                    b[pos] = int(rng.integers(0, alphabet))
                elif op == 1:
                    b.insert(pos, int(rng.integers(0, alphabet)))
                elif len(b) > 1:
                    del b[pos]
            b = bytes(b)
        else:
            b = rng.integers(0, alphabet, lb, dtype=np.uint8).tobytes()
        items_a.append(a.tobytes())
        items_b.append(b)
    return items_a, items_b


# ----------------------------------------------------------------------------------------------------
# known answers
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algorithm", ALGORITHMS)
def test_kat_levenshtein(sw, scope, algorithm):
    a = sw.Strs([row[0] for row in KAT["levenshtein"]])
    b = sw.Strs([row[1] for row in KAT["levenshtein"]])
    got_bytes = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(a, b, scope)
    got_utf8 = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm=algorithm).pairs(a, b, scope)
    assert got_bytes.tolist() == [row[3] for row in KAT["levenshtein"]]
    assert got_utf8.tolist() == [row[2] for row in KAT["levenshtein"]]


@pytest.mark.parametrize("algorithm", ALGORITHMS)
def test_kat_bounded(sw, scope, algorithm):
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)
    for a, b, k, expected in KAT["bounded"]:
        assert engine.pairs([a], [b], scope, bound=k).tolist() == [expected]


def test_kat_needleman_wunsch(sw, scope):
    cases = KAT["nw_unary_2_m1"]["cases"]
    a, b = sw.Strs([c[0] for c in cases]), sw.Strs([c[1] for c in cases])
    classes, costs = sw.unary_class_costs(2, -1)
    linear = sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope).pairs(a, b, scope)
    affine = sw.NeedlemanWunschScores(classes, costs, open=-5, extend=-1, capabilities=scope).pairs(a, b, scope)
    assert linear.tolist() == [c[2] for c in cases]
    assert affine.tolist() == [c[3] for c in cases]
    twins = KAT["nw_bio_twins"]["cases"]
    a, b = sw.Strs([c[0] for c in twins]), sw.Strs([c[1] for c in twins])
    m = unary_matrix(2, -1)
    assert sw.NeedlemanWunschScores(substitution_matrix=m, open=-4, extend=-2, capabilities=scope).pairs(a, b, scope).tolist() == [c[2] for c in twins]
    assert sw.NeedlemanWunschScores(substitution_matrix=m, open=-6, extend=-1, capabilities=scope).pairs(a, b, scope).tolist() == [c[3] for c in twins]
    x, y, match, mismatch, open_, extend, expected = KAT["nw_classic"][0]
    engine = sw.NeedlemanWunschScores(substitution_matrix=unary_matrix(match, mismatch), open=open_, extend=extend, capabilities=scope)
    assert engine.pairs([x], [y], scope).tolist() == [expected]


# ----------------------------------------------------------------------------------------------------
# committed golden slices of every synthetic config (generator + kernel self-check, no oracle needed)
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["words16", "tokens64", "utf8_lines", "protein4k", "short_words", "bytes4k"])
@pytest.mark.parametrize("algorithm", ALGORITHMS)
def test_golden_slices_levenshtein(sw, scope, name, algorithm):
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    a = sw.Strs(data=z[f"{name}.a_data"], offsets=z[f"{name}.a_offsets"])
    b = sw.Strs(data=z[f"{name}.b_data"], offsets=z[f"{name}.b_offsets"])
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)
    assert (engine.pairs(a, b, scope) == z[f"{name}.lev_bytes"]).all()
    if name == "utf8_lines":
        utf8 = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm=algorithm)
        expected = z["utf8_lines.lev_utf8"]
        assert (utf8.pairs(a, b, scope) == expected).all()
        for k in (0, 7, 32):
            assert (utf8.pairs(a, b, scope, bound=k) == np.minimum(expected, k + 1)).all()


@pytest.mark.parametrize("name", ["protein4k", "bytes4k"])
def test_golden_slices_needleman_wunsch(sw, scope, name):
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    a = sw.Strs(data=z[f"{name}.a_data"], offsets=z[f"{name}.a_offsets"])
    b = sw.Strs(data=z[f"{name}.b_data"], offsets=z[f"{name}.b_offsets"])
    matrix = z[f"{name}.matrix"]
    linear = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=-4, extend=-4, capabilities=scope)
    affine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=-11, extend=-1, capabilities=scope)
    assert (linear.pairs(a, b, scope) == z[f"{name}.nw_linear_m4"]).all()
    assert (affine.pairs(a, b, scope) == z[f"{name}.nw_affine_m11_m1"]).all()


def test_golden_script_lines(sw, scope):
    """tests/golden/script_lines.npz: 64 pairs of unrelated article lines, one script each (the reference's cross-product of XLSum lines,
    similarities/README.md:18, :39-40) -- the committed oracle distances over code points (`LevenshteinDistancesUtf8`: the dense-alphabet
    items of bp_dense.hpp) and over bytes, raw and prepared tapes, every algorithm switch, bounded."""
    import hashlib
    z = np.load(os.path.join(GOLDEN, "script_lines.npz"))
    a, b = sw.generate_pairs("script_lines", 64, seed=42)
    h = hashlib.sha256()
    for array in (a.data, a.offsets, b.data, b.offsets):
        h.update(np.ascontiguousarray(array).tobytes())
    assert h.hexdigest() == bytes(z["n64.sha256"]).decode()
    for algorithm in ALGORITHMS:
        assert (sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm=algorithm).pairs(a, b, scope) == z["n64.lev_utf8"]).all(), algorithm
        assert (sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(a, b, scope) == z["n64.lev_bytes"]).all(), algorithm
    utf8 = sw.LevenshteinDistancesUTF8(capabilities=scope)
    pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)
    for _ in range(2):
        assert (utf8.pairs(pa, pb, scope) == z["n64.lev_utf8"]).all()
    for bound in (32, 100, 700):
        assert (utf8.pairs(a, b, scope, bound=bound) == np.minimum(z["n64.lev_utf8"], bound + 1)).all(), bound


@pytest.mark.parametrize("name", ["words16", "tokens64", "utf8_lines", "protein4k", "short_words", "bytes4k"])
def test_golden_256_pairs_of_every_config(sw, scope, name):
    """the first 256 pairs of each synthetic config against the committed oracle outputs. The KB-sized
    workloads are regenerated here (their SHA-256 is part of the fixture, tests/golden/make_fixtures.py)."""
    import hashlib
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    a, b = sw.generate_pairs(name, 256, seed=42)
    h = hashlib.sha256()
    for array in (a.data, a.offsets, b.data, b.offsets):
        h.update(np.ascontiguousarray(array).tobytes())
    assert h.hexdigest() == bytes(z[f"{name}.n256.sha256"]).decode()
    want = z[f"{name}.n256.lev_bytes"]
    for algorithm in ALGORITHMS:
        engine = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)
        assert (engine.pairs(a, b, scope) == want).all()
    assert (sw.LevenshteinDistances(capabilities=scope).pairs(a, b, scope, bound=32) == np.minimum(want, 33)).all()
    if name == "utf8_lines":
        utf8 = sw.LevenshteinDistancesUTF8(capabilities=scope)
        want = z["utf8_lines.n256.lev_utf8"]
        assert (utf8.pairs(a, b, scope) == want).all()
        for k in (0, 7, 32, 63):
            assert (utf8.pairs(a, b, scope, bound=k) == np.minimum(want, k + 1)).all()
    if name in ("protein4k", "bytes4k"):
        matrix = z[f"{name}.matrix"]
        for tag, (open_, extend) in {"linear_m4": (-4, -4), "affine_m11_m1": (-11, -1)}.items():
            nw = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=open_, extend=extend, capabilities=scope)
            assert (nw.pairs(a, b, scope) == z[f"{name}.n256.nw_{tag}"]).all()
            local = sw.SmithWatermanScores(substitution_matrix=matrix, open=open_, extend=extend, capabilities=scope)
            assert (local.pairs(a, b, scope) == z[f"{name}.n256.sw_{tag}"]).all()


# ----------------------------------------------------------------------------------------------------
# random sweeps against the oracle
# ----------------------------------------------------------------------------------------------------
LENGTHS_SHORT = list(range(0, 40)) + [47, 48, 49, 63, 64, 65, 95, 96, 97, 127, 128, 129, 130, 160, 200, 255, 256, 257]
LENGTHS_LONG = [1, 31, 100, 511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 2500, 3000]


@pytest.mark.parametrize("algorithm", ALGORITHMS)
@pytest.mark.parametrize("alphabet", [2, 4, 26, 256])
def test_random_levenshtein_short(sw, orc, scope, algorithm, alphabet):
    rng = np.random.default_rng(100 + alphabet)
    items_a, items_b = random_pairs(rng, 6000, LENGTHS_SHORT, alphabet)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(a, b, scope)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])


@pytest.mark.parametrize("utf8", [False, True])
def test_small_tapes_and_strings_at_tape_edges(sw, orc, scope, utf8):
    """Tapes of 0..80 bytes: the kernels read strings with 4- and 16-byte loads clamped into the tape, so every tape
    size around those widths, with the first and last strings touching the tape's ends, has to come out right --
    both through the planned kernels and through the direct kernel for short pairs (taken on the second call, once
    the scope has seen that the batch is mostly short)."""
    rng = np.random.default_rng(20260101)
    cls = sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances
    alphabet = [chr(c) for c in range(0x61, 0x67)] + (["\u00e9", "\u0416", "\u4e2d", "\U0001f600"] if utf8 else [])
    for total in list(range(0, 40)) + [47, 48, 49, 63, 64, 65, 80]:
        for trial in range(3):
            count = int(rng.integers(1, 5))
            cuts_a = np.sort(rng.integers(0, total + 1, count - 1)) if count > 1 else np.array([], np.int64)
            cuts_b = np.sort(rng.integers(0, total + 1, count - 1)) if count > 1 else np.array([], np.int64)
            la = np.diff(np.concatenate([[0], cuts_a, [total]]))
            lb = np.diff(np.concatenate([[0], cuts_b, [total]]))
            items_a = ["".join(rng.choice(alphabet, int(n))) for n in la]
            items_b = ["".join(rng.choice(alphabet, int(n))) for n in lb]
            a, b = sw.Strs([x.encode("utf-8") for x in items_a]), sw.Strs([x.encode("utf-8") for x in items_b])
            engine = cls(capabilities=scope)
            want = orc.levenshtein_pairs(a, b, utf8=utf8)
            for call in range(2):
                got = engine.pairs(a, b, scope)
                assert got.tolist() == want.tolist(), (total, trial, call, items_a, items_b)


@pytest.mark.parametrize("algorithm", ALGORITHMS)
def test_random_levenshtein_long(sw, orc, scope, algorithm):
    rng = np.random.default_rng(7)
    items_a, items_b = random_pairs(rng, 300, LENGTHS_LONG, 20)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(a, b, scope)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])


def test_more_long_pairs_than_waves(sw, orc, scope):
    """k_bitparallel_long runs one pair per wave on at most 4096 waves; with more pairs than that the waves draw them from a
    ticket. 4500 pairs of 2.1-2.4 K symbols: symmetry and identity over the whole batch, 40 pairs against the oracle; and
    the same through STRINGWARS_AMD_LONG_TICKET=0's round-robin list in a process of its own."""
    rng = np.random.default_rng(4500)
    count = 4500
    la, lb = rng.integers(2100, 2400, count), rng.integers(2100, 2400, count)
    items_a = [rng.integers(97, 101, n).astype(np.uint8) for n in la]
    items_b = []
    for i, n in enumerate(lb):
        if i % 9 == 0:
            items_b.append(items_a[i].copy())                 # identical
        elif i % 3 == 0:
            edited = items_a[i].copy()                        # related: substitutions, then cut or padded to length n
            where = rng.integers(0, edited.size, 60)
            edited[where] = rng.integers(97, 101, 60)
            items_b.append(np.concatenate([edited, rng.integers(97, 101, max(0, n - edited.size)).astype(np.uint8)])[:max(n, 1)])
        else:
            items_b.append(rng.integers(97, 101, n).astype(np.uint8))
    a, b = sw.Strs([x.tobytes() for x in items_a]), sw.Strs([x.tobytes() for x in items_b])
    engine = sw.LevenshteinDistances(capabilities=scope)
    got, swapped = engine.pairs(a, b, scope), engine.pairs(b, a, scope)
    assert (got == swapped).all()
    assert (got[::9] == 0).all() and (got <= np.maximum(a.lengths, b.lengths)).all()
    assert (got >= np.abs(a.lengths.astype(np.int64) - b.lengths.astype(np.int64))).all()
    sample = np.sort(rng.choice(count, 40, replace=False))
    sa, sb = sw.Strs([items_a[i].tobytes() for i in sample]), sw.Strs([items_b[i].tobytes() for i in sample])
    assert got[sample].tolist() == orc.levenshtein_pairs(sa, sb, algo="hyyro").tolist()
    assert engine.pairs(a, b, scope).tolist() == got.tolist()   # the ticket starts from zero again


@pytest.mark.parametrize("utf8", [False, True])
def test_patterns_longer_than_64_blocks(sw, orc, scope, utf8):
    """Both strings > 2048 symbols: the bit-parallel kernel walks the text once per pass of 64 blocks and hands the
    horizontal deltas from pass to pass (2 to 5 passes here, with partial last passes, block-aligned and unaligned
    lengths, texts shorter and longer than the carry-word granularity, mixed with ordinary pairs)."""
    rng = np.random.default_rng(77)
    lengths = [(2049, 2049), (2048 + 32, 2100), (4096, 4096), (4097, 5000), (3000, 9000), (6143, 6145), (8200, 2300),
               (2050, 2080), (10, 5000), (5000, 10), (2500, 2500), (70, 90), (0, 3000), (4095, 4095),
               (20_000, 24_000) if utf8 else (50_000, 60_000)]   # tens of passes, carry words far beyond one cache line
    alphabet = np.array([ord(c) for c in "ACGT"], np.uint32) if not utf8 else np.array([0x41, 0xE9, 0x416, 0x4E2D, 0x1F600], np.uint32)
    items_a, items_b = [], []
    for la, lb in lengths:
        a = alphabet[rng.integers(0, len(alphabet), la)]
        if la and lb and rng.random() < 0.7:     # related strings: a few hundred edits
            b = list(a)
            for _ in range(int(rng.integers(1, 300))):
                op, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(len(b), 1)))
                if op == 0 and b:
                    b[pos] = int(alphabet[rng.integers(0, len(alphabet))])
                elif op == 1:
                    b.insert(pos, int(alphabet[rng.integers(0, len(alphabet))]))
                elif len(b) > 1:
                    del b[pos]
            b = np.array(b[:max(lb, 1)] if len(b) > lb else b, np.uint32)
        else:
            b = alphabet[rng.integers(0, len(alphabet), lb)]
        enc = (lambda x: "".join(map(chr, x)).encode("utf-8")) if utf8 else (lambda x: bytes(x.astype(np.uint8)))
        items_a.append(enc(a))
        items_b.append(enc(b))
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    cls = sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances
    want = orc.levenshtein_pairs(a, b, utf8=utf8, algo="hyyro" if not utf8 else "wf")
    for algorithm in ("auto", "bitparallel"):
        engine = cls(capabilities=scope, algorithm=algorithm)
        got = engine.pairs(a, b, scope)
        assert got.tolist() == want.tolist(), (algorithm, got.tolist(), want.tolist())
        timing_names = scope.last_timing()["dominant_name"] if hasattr(scope, "last_timing") else ""
        got_bounded = engine.pairs(a, b, scope, bound=100)
        assert got_bounded.tolist() == np.minimum(want, 101).tolist(), algorithm
    # 32-bit offsets, device-resident tapes, and the same strings as a cross-product (rows x columns) job
    engine = cls(capabilities=scope)
    a32, b32 = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    assert engine.pairs(a32, b32, scope).tolist() == want.tolist()
    assert engine.pairs(a32.to_device(scope), b32.to_device(scope), scope).tolist() == want.tolist()
    rows, cols = a.subview(0, 5), b.subview(0, 4)
    matrix = engine(rows, cols, scope)
    for i in range(5):
        for j in range(4):
            one = orc.levenshtein_pairs(a.subview(i, i + 1), b.subview(j, j + 1), utf8=utf8, algo="wf")
            assert int(matrix[i, j]) == int(one[0]), (i, j)


@pytest.mark.parametrize("algorithm", ["wavefront", "bitparallel"])
def test_very_long_pair(sw, orc, scope, algorithm):
    """Beyond 64 blocks (bit-parallel hands over to the wavefront) and beyond one wavefront pass (6144 columns)."""
    rng = np.random.default_rng(3)
    items_a, items_b = random_pairs(rng, 6, [6500, 7000, 9000], 4, related=1.0)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(a, b, scope)
    assert (got == orc.levenshtein_pairs(a, b, algo="hyyro")).all()


def test_exhaustive_binary_alphabet_crossproduct(sw, orc, scope):
    """All 255 strings over {a,b} of length 0..7 against each other: 65,025 pairs through the
    cross-product entry point (the reference's `compute_into` shape, bench.rs:478-486)."""
    strings = [b""] + [bytes(97 + ((v >> i) & 1) for i in range(n)) for n in range(1, 8) for v in range(1 << n)]
    tape = sw.Strs(strings)
    for algorithm in ("auto", "wavefront"):
        matrix = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)(tape, tape, scope)
        assert matrix.shape == (255, 255) and matrix.dtype == np.uint64
        rows = sw.Strs([s for s in strings for _ in strings])
        cols = sw.Strs([t for _ in strings for t in strings])
        want = orc.levenshtein_pairs(rows, cols).reshape(255, 255)
        assert (matrix == want).all()
    symmetric = sw.LevenshteinDistances(capabilities=scope)(tape, None, scope)
    assert (symmetric == want).all()


@pytest.mark.parametrize("algorithm", ALGORITHMS)
def test_bounded_is_min_of_distance_and_bound_plus_one(sw, orc, scope, algorithm):
    rng = np.random.default_rng(11)
    items_a, items_b = random_pairs(rng, 3000, LENGTHS_SHORT, 8)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    full = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)
    for k in (0, 1, 2, 5, 32, 1000):
        assert (engine.pairs(a, b, scope, bound=k) == np.minimum(full, k + 1)).all(), k
        assert (orc.levenshtein_pairs(a, b, bound=k) == np.minimum(full, k + 1)).all()


@pytest.mark.parametrize("utf8", [False, True])
def test_banded_window_kernel(sw, orc, scope, utf8):
    """Bounded calls with k <= 63 take the sliding 64-bit band (banded.hip): tight Ukkonen band, k+1 diagonals."""
    rng = np.random.default_rng(31 + utf8)
    items_a, items_b = [], []
    alphabet = [chr(c) for c in range(0x61, 0x7B)] + (["é", "я", "語", "😀"] if utf8 else [])
    for _ in range(1500):
        n = int(rng.integers(1, 700))
        a = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), n)]
        b = list(a)
        edits = int(rng.choice([0, 1, 2, 5, 8, 16, 31, 32, 33, 40, 63, 64, 70, 100]))
        for _ in range(edits):
            op = int(rng.integers(0, 3))
            if op == 0 and b:
                b[int(rng.integers(0, len(b)))] = alphabet[int(rng.integers(0, len(alphabet)))]
            elif op == 1:
                b.insert(int(rng.integers(0, len(b) + 1)), alphabet[int(rng.integers(0, len(alphabet)))])
            elif len(b) > 1:
                del b[int(rng.integers(0, len(b)))]
        if rng.random() < 0.1:  # pure insertions / deletions: length difference == distance
            b = a[: max(1, n - int(rng.integers(0, 64)))]
        items_a.append("".join(a))
        items_b.append("".join(b))
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    full = orc.levenshtein_pairs(a, b, utf8=utf8)
    engine = (sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances)(capabilities=scope)
    for k in (0, 1, 3, 7, 8, 15, 16, 31, 32, 33, 35, 36, 50, 63):
        got = engine.pairs(a, b, scope, bound=k)
        want = np.minimum(full, k + 1)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (k, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
        assert (engine.pairs(b, a, scope, bound=k) == want).all(), k


def test_banded_window_kernel_large_batch(sw, orc, scope):
    """Above ~131 K pairs the banded kernel switches from 32-pair to 64-pair wave items: same results."""
    rng = np.random.default_rng(4242)
    items_a, items_b = random_pairs(rng, 140_000, list(range(90, 131)), 26, related=0.8)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    engine = sw.LevenshteinDistances(capabilities=scope)
    for bound in (8, 20):
        got = engine.pairs(a, b, scope, bound=bound)
        want = orc.levenshtein_pairs(a, b, algo="hyyro", bound=bound)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bound, bad[:5], got[bad[:5]], want[bad[:5]])
    assert scope.last_timing()["cells"] > 0


def test_high_bytes_take_the_8bit_table(sw, orc, scope):
    """Bytes >= 0x80 overflow the 7-bit match table and are deferred to the 8-bit kernel."""
    rng = np.random.default_rng(5)
    items_a, items_b = random_pairs(rng, 4000, LENGTHS_SHORT, 256)
    for i in range(0, 4000, 3):  # a third of the pairs stay pure ASCII
        items_a[i] = bytes(c & 0x7F for c in items_a[i])
        items_b[i] = bytes(c & 0x7F for c in items_b[i])
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistances(capabilities=scope, algorithm="bitparallel").pairs(a, b, scope)
    assert (got == orc.levenshtein_pairs(a, b, algo="hyyro")).all()


def test_utf8_random_scripts(sw, orc, scope):
    rng = np.random.default_rng(13)
    pools = [range(0x20, 0x7F), range(0x400, 0x500), range(0x4E00, 0x4F00), range(0x1F600, 0x1F650)]
    def text(n):
        return "".join(chr(int(rng.choice(pools[int(rng.integers(0, 4))]))) for _ in range(n))
    items_a = [text(int(rng.integers(0, 90))) for _ in range(1500)]
    items_b = [s[: int(rng.integers(0, len(s) + 1))] + text(int(rng.integers(0, 8))) if rng.random() < 0.6 else text(int(rng.integers(0, 90))) for s in items_a]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistancesUTF8(capabilities=scope).pairs(a, b, scope)
    assert (got == orc.levenshtein_pairs(a, b, utf8=True)).all()
    assert (sw.edit_distance(a, b, scope) == got).all()


@pytest.mark.parametrize("costs", [(0, 2, 3, 3), (0, 1, 2, 1), (1, 3, 4, 2), (0, 1, 1, 1)])
def test_general_cost_levenshtein(sw, orc, scope, costs):
    """`LevenshteinDistances::new(&scope, match, mismatch, open, extend)` with non-unit costs (bench.rs:382)."""
    rng = np.random.default_rng(17)
    items_a, items_b = random_pairs(rng, 1500, list(range(0, 70)) + [130, 200, 300], 6)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    got = sw.LevenshteinDistances(*costs, capabilities=scope).pairs(a, b, scope)
    want = orc.levenshtein_costs_pairs(a, b, *costs)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (costs, bad[:5], got[bad[:5]], want[bad[:5]])


@pytest.mark.parametrize("costs", [(0, 2, 3, 3), (0, 1, 2, 1), (1, 3, 4, 2)])
def test_general_cost_levenshtein_bounded(sw, orc, scope, costs):
    """The cutoff out = min(d, bound + 1) holds for every cost model, empty sides included (pairs with an
    empty side used to skip the clamp unless the costs were (0,1,1,1))."""
    engine = sw.LevenshteinDistances(*costs, capabilities=scope)
    assert engine.pairs([b""], [b"abcdef"], scope, bound=4).tolist() == [5]
    assert engine.pairs([b"abcdef", b"", b"ab"], [b"", b"", b"ab"], scope, bound=4).tolist() == [5, 0, min(2 * costs[0], 5)]
    rng = np.random.default_rng(29)
    items_a, items_b = random_pairs(rng, 800, list(range(0, 50)) + [130, 200], 5)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    full = orc.levenshtein_costs_pairs(a, b, *costs)
    for bound in (0, 4, 17, 300):
        assert (engine.pairs(a, b, scope, bound=bound) == np.minimum(full, bound + 1)).all()


@pytest.mark.parametrize("costs", [(0, 1, 2, 1), (1, 3, 4, 2), (0, 2, 3, 3), (0, 1, 1, 1)])
def test_general_cost_levenshtein_over_code_points(sw, orc, scope, costs):
    """`LevenshteinDistancesUtf8::new(&scope, match, mismatch, open, extend)` (bench.rs:386-389) with non-unit and affine
    costs: the distance depends on the symbols only through equality, so the byte oracle scores the same strings with
    every distinct code point renamed to a byte."""
    rng = np.random.default_rng(41)
    alphabet = [chr(c) for c in (0x41, 0x7A, 0xE9, 0x416, 0x4E2D, 0x1F600, 0x10FFFF, 0x800, 0x7FF)]
    def text(n):
        return "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n))
    items_a = [text(int(rng.integers(0, 60))) for _ in range(700)] + [text(150), text(300), ""]
    items_b = [s[: int(rng.integers(0, len(s) + 1))] + text(int(rng.integers(0, 6))) if rng.random() < 0.6 else text(int(rng.integers(0, 60)))
               for s in items_a]
    rename = {ch: bytes([65 + i]) for i, ch in enumerate(alphabet)}
    as_bytes = lambda items: sw.Strs([b"".join(rename[ch] for ch in s) for s in items])
    want = orc.levenshtein_costs_pairs(as_bytes(items_a), as_bytes(items_b), *costs)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    engine = sw.LevenshteinDistancesUTF8(*costs, capabilities=scope)
    assert (engine.pairs(a, b, scope) == want).all()
    assert (engine.pairs(a, b, scope, bound=6) == np.minimum(want, 7)).all()
    pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)
    assert (engine.pairs(pa, pb, scope) == want).all()
    grid = engine(a.subview(0, 20), b.subview(0, 25), scope)
    wide = orc.levenshtein_costs_pairs(as_bytes([x for x in items_a[:20] for _ in range(25)]), as_bytes(items_b[:25] * 20), *costs)
    assert (grid.reshape(-1) == wide).all()


def test_scores_beyond_int32_are_refused(sw, scope):
    """Gap costs at the accepted limit times long strings leave the wavefront kernels' 32-bit score range: the call is
    refused (unsupported_length) instead of returning wrapped scores."""
    engine = sw.NeedlemanWunschScores(substitution_matrix=unary_matrix(2, -1), open=-4096, extend=-4096, capabilities=scope)
    assert engine.pairs([b"ACGT" * 50], [b"ACGA" * 50], scope).size == 1
    long_a, long_b = [b"AC" * 20000], [b"GT" * 20000]
    with pytest.raises(sw.StringWarsError) as info:
        engine.pairs(long_a, long_b, scope)
    assert info.value.status == "unsupported_length"


@pytest.mark.parametrize("gaps", [(-4, -4), (-11, -1), (-2, -2), (-5, -1), (0, 0)])
@pytest.mark.parametrize("symmetric", [True, False])
def test_random_needleman_wunsch(sw, orc, scope, gaps, symmetric):
    rng = np.random.default_rng(23)
    matrix = rng.integers(-8, 12, (256, 256)).astype(np.int8)
    if symmetric:
        matrix = np.minimum(matrix, matrix.T)
    lengths = list(range(0, 40)) + [100, 129, 200, 400, 700, 1100, 1600, 2100]
    items_a, items_b = random_pairs(rng, 500, lengths, 24)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    engine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
    got = engine.pairs(a, b, scope)
    want = orc.nw_pairs(a, b, matrix, *gaps)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (gaps, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])


@pytest.mark.parametrize("gaps", [(-4, -4), (-11, -1), (-1, -1)])
@pytest.mark.parametrize("classes", [2, 21, 32])
def test_class_table_needleman_wunsch(sw, orc, scope, gaps, classes):
    """Matrices with <= 32 symbol classes run on the register cost-row model (v_perm selects), both when given as
    `byte_to_class` + 32x32 costs (bench.rs:658-662) and when a 256x256 matrix merely happens to have few classes."""
    rng = np.random.default_rng(classes * 7 + gaps[0])
    byte_to_class = rng.integers(0, classes, 256).astype(np.uint8)
    costs = np.zeros((32, 32), dtype=np.int8)
    costs[:classes, :classes] = rng.integers(-9, 12, (classes, classes))          # asymmetric on purpose
    full = costs[byte_to_class][:, byte_to_class].astype(np.int8)                  # the expanded 256x256 table
    lengths = list(range(0, 40)) + [63, 64, 65, 100, 129, 200, 257, 400, 513, 700, 1100, 1600, 2100, 3100]
    items_a, items_b = random_pairs(rng, 350, lengths, 256)
    items_a += [bytes(rng.integers(0, 256, 7000, dtype=np.uint8)), bytes(rng.integers(0, 256, 5000, dtype=np.uint8))]
    items_b += [bytes(rng.integers(0, 256, 6600, dtype=np.uint8)), items_a[-1][100:4900]]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    want = orc.nw_pairs(a, b, full, *gaps)
    by_classes = sw.NeedlemanWunschScores(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=scope)
    by_matrix = sw.NeedlemanWunschScores(substitution_matrix=full, open=gaps[0], extend=gaps[1], capabilities=scope)
    for engine in (by_classes, by_matrix):
        got = engine.pairs(a, b, scope)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (classes, gaps, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])


@pytest.mark.parametrize("gaps", [(-4, -4), (-11, -1)])
@pytest.mark.parametrize("classes", [33, 52, 64, 100, 128, 129])
def test_wide_class_table_needleman_wunsch(sw, orc, scope, gaps, classes):
    """33 .. 128 symbol classes -- mixed-case text, IUPAC codes with case: what a rust-bio style scoring closure distinguishes
    (bench.rs:746-752 is a closure over ANY alphabet) -- are too many for the register cost rows of the 32-class model, but the
    column-profile kernel only needs one cost row per class while it builds a pass's profile: such matrices get a table of
    128-byte rows (Scoring::wide_table) and their long pairs run on k_nwprofile with strips of eight (<= 56 classes) or four columns,
    global and local, linear and affine (52 letters: 5.3 TCUPS against 3.8 for the LDS gather); 129 classes and up keep the 256 x 256 matrix in LDS. Scores are the oracle's throughout:
    strings on both sides of the profile kernel's 384 columns, several passes, rows beyond a ring block, empty strings."""
    rng = np.random.default_rng(classes * 11 - gaps[1])
    byte_to_class = rng.integers(0, classes, 256).astype(np.int64)
    byte_to_class[rng.permutation(256)[:classes]] = np.arange(classes)              # every class in use
    costs = rng.integers(-9, 12, (classes, classes)).astype(np.int8)               # asymmetric on purpose
    costs[np.arange(classes), np.arange(classes)] = rng.integers(4, 12, classes)
    costs[0, 1] = -9; costs[1, 0] = 11                                              # no two classes alike
    full = costs[byte_to_class][:, byte_to_class].astype(np.int8)
    lengths = [0, 1, 7, 40, 129, 383, 384, 385, 511, 512, 513, 700, 1023, 1024, 1025, 1100, 1600, 2049, 3100]
    items_a, items_b = random_pairs(rng, 150, lengths, 256)
    items_a += [bytes(rng.integers(0, 256, 7000, dtype=np.uint8)), bytes(rng.integers(0, 256, 5000, dtype=np.uint8))]
    items_b += [bytes(rng.integers(0, 256, 6600, dtype=np.uint8)), items_a[-1][100:4900]]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    for local in (False, True):
        Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
        engine = Engine(substitution_matrix=full, open=gaps[0], extend=gaps[1], capabilities=scope)
        want = np.array([orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for x, y in zip(items_a, items_b)])
        scope.set_profiling(True)
        got = engine.pairs(a, b, scope)
        name = scope.last_timing()["dominant_name"]
        scope.set_profiling(False)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (classes, gaps, local, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
        if classes <= 128:
            assert name.startswith("nwprofile") and name.endswith("_w8" if classes <= 56 else "_w4"), name
        else:
            assert name.startswith("wavefront"), name
        pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
        assert (engine.pairs(pa, pb, scope) == want).all()


@pytest.mark.parametrize("gaps", [(-4, -4), (-11, -1)])
@pytest.mark.parametrize("classes", [5, 21, 32])
def test_column_profile_kernel_every_strip_shape(sw, orc, scope, gaps, classes):
    """nwprofile.hip (global alignment on a class table, pairs of more than 384 columns): every strip width the kernel
    switches on (4 / 8 / 12 / 16 columns per lane, capped by the class count), column counts just below / at / above a multiple
    of 64 x the strip and of the pass width, one to five passes, rows shorter and longer than the 64-step ring blocks,
    symmetric (columns = shorter string) and asymmetric matrices -- and the same batch on the wavefront kernels
    (`STRINGWARS_AMD_NW=classic` is read once per process, so the comparison runs in a child)."""
    rng = np.random.default_rng(1000 + classes + gaps[1])
    byte_to_class = rng.integers(0, classes, 256).astype(np.uint8)
    costs = np.zeros((32, 32), dtype=np.int8)
    costs[:classes, :classes] = rng.integers(-9, 12, (classes, classes))
    strip = 16 if classes <= 16 else (12 if classes <= 24 else 8)
    cols = sorted({385, 448, 511, 512, 513, 640, 767, 768, 769, 1023, 1024, 1025, 64 * strip - 1, 64 * strip, 64 * strip + 1,
                   128 * strip, 128 * strip + 1, 192 * strip + 7, 256 * strip - 5, 256 * strip + 3, 3000, 4100})
    items_a, items_b = [], []
    for n_cols in cols:
        for n_rows in (n_cols, n_cols + 1, n_cols + 61, n_cols + 64, n_cols + 130, 2 * n_cols + 3):
            text = bytes(rng.integers(0, 256, n_rows, dtype=np.uint8))
            other = bytearray(text[:n_cols]) if rng.random() < 0.5 else bytearray(rng.integers(0, 256, n_cols, dtype=np.uint8).tobytes())
            for at in rng.integers(0, n_cols, n_cols // 7):
                other[at] = int(rng.integers(0, 256))
            if rng.random() < 0.5:
                items_a.append(text); items_b.append(bytes(other))
            else:
                items_a.append(bytes(other)); items_b.append(text)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    for symmetric in (False, True):
        table = np.minimum(costs, costs.T) if symmetric else costs
        full = table[byte_to_class][:, byte_to_class].astype(np.int8)
        want = orc.nw_pairs(a, b, full, *gaps)
        engine = sw.NeedlemanWunschScores(byte_to_class, table, open=gaps[0], extend=gaps[1], capabilities=scope)
        scope.set_profiling(True)
        got = engine.pairs(a, b, scope)
        assert scope.last_timing()["dominant_name"].startswith("nwprofile"), scope.last_timing()
        scope.set_profiling(False)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (classes, gaps, symmetric, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
        cross = engine(sw.Strs(items_a[:3]), sw.Strs(items_b[:4]), scope)      # the cross-product entry point takes the same route
        assert (cross == np.array([[orc.nw_score(x, y, full, *gaps) for y in items_b[:4]] for x in items_a[:3]])).all()
        # Smith-Waterman on the same kernel (zero floor, running maximum carried over the passes)
        local = sw.SmithWatermanScores(byte_to_class, table, open=gaps[0], extend=gaps[1], capabilities=scope)
        scope.set_profiling(True)
        got = local.pairs(a, b, scope)
        assert scope.last_timing()["dominant_name"].startswith("nwprofile_local"), scope.last_timing()
        scope.set_profiling(False)
        want_local = np.array([orc.nw_score(x, y, full, gaps[0], gaps[1], local=True) for x, y in zip(items_a, items_b)])
        bad = np.nonzero(got != want_local)[0]
        assert bad.size == 0, ("local", classes, gaps, symmetric, bad[:5], got[bad[:5]], want_local[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])


def test_gotoh_narrow_strips_follow_the_scores(sw, orc, scope):
    """nwprofile.hip, affine gaps: the strips hold 16-bit distances to a wave-wide `shift` that is moved every 64 steps
    (`kNarrow`). Pairs whose scores run away from the all-gaps baseline as fast as the costs allow, upwards (identical
    strings, +11 a symbol) and downwards (nothing in common), over one to six passes; a pair that changes its mind half way;
    and a matrix whose costs are too wide for 16 bits, which must take the 32-bit kernel."""
    rng = np.random.default_rng(77)
    byte_to_class = (np.arange(256) % 21).astype(np.uint8)
    costs = np.zeros((32, 32), dtype=np.int8)
    costs[:21, :21] = -9
    costs[np.arange(21), np.arange(21)] = 11
    items_a, items_b = [], []
    for n in (400, 769, 1500, 3000, 5200):
        same = bytes(rng.integers(0, 21, n, dtype=np.uint8))
        low, high = bytes(rng.integers(0, 10, n, dtype=np.uint8)), bytes(rng.integers(10, 21, n + 37, dtype=np.uint8))
        items_a += [same, low, same + low, low + same, same]
        items_b += [same, high, same + high, high[: n // 2] + same, same[: n // 2] + high]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    full = costs[byte_to_class][:, byte_to_class].astype(np.int8)
    engine = sw.NeedlemanWunschScores(byte_to_class, costs, open=-11, extend=-1, capabilities=scope)
    scope.set_profiling(True)
    got = engine.pairs(a, b, scope)
    assert "narrow" in scope.last_timing()["dominant_name"], scope.last_timing()
    want = orc.nw_pairs(a, b, full, -11, -1)
    assert (got == want).all(), np.nonzero(got != want)[0][:8]
    assert (engine.pairs(b, a, scope) == want).all()
    # at the edge of what the strips may hold: max |cost| + |open| + |extend| = 29, times the 1024 cells a wave spans = 29 696 of the
    # 30 000 allowed; strings made of long runs, so that one half of a wave's columns climbs 17 a cell while the other half falls
    steep = costs.copy()
    steep[:21, :21] = -17
    steep[np.arange(21), np.arange(21)] = 17
    run_a, run_b = [], []
    for n in (900, 1600, 3100, 4700):
        for period in (64, 300, 768, n // 2):
            x = bytes((np.arange(n) // period % 2).astype(np.uint8))                 # 000..111..000..
            y = bytes(((np.arange(n + 11) // (period + 5)) % 3 == 0).astype(np.uint8))
            run_a += [x, x, bytes(n), x[: n // 2] + bytes([2]) * (n // 2)]
            run_b += [y, x[::-1], bytes([1]) * n, x]
    ra, rb = sw.Strs(run_a), sw.Strs(run_b)
    engine = sw.NeedlemanWunschScores(byte_to_class, steep, open=-11, extend=-1, capabilities=scope)
    got = engine.pairs(ra, rb, scope)
    assert "narrow" in scope.last_timing()["dominant_name"], scope.last_timing()
    want = orc.nw_pairs(ra, rb, steep[byte_to_class][:, byte_to_class].astype(np.int8), -11, -1)
    assert (got == want).all(), np.nonzero(got != want)[0][:8]
    steep[0, 0] = 18                                                                   # one more: 30 x 1024 > 30 000
    engine = sw.NeedlemanWunschScores(byte_to_class, steep, open=-11, extend=-1, capabilities=scope)
    got = engine.pairs(ra, rb, scope)
    assert "narrow" not in scope.last_timing()["dominant_name"], scope.last_timing()
    assert (got == orc.nw_pairs(ra, rb, steep[byte_to_class][:, byte_to_class].astype(np.int8), -11, -1)).all()
    wide = costs.copy()
    wide[:21, :21] = -60
    wide[np.arange(21), np.arange(21)] = 60
    engine = sw.NeedlemanWunschScores(byte_to_class, wide, open=-11, extend=-1, capabilities=scope)
    got = engine.pairs(a, b, scope)
    name = scope.last_timing()["dominant_name"]
    scope.set_profiling(False)
    assert name.startswith("nwprofile_affine") and "narrow" not in name, name
    assert (got == orc.nw_pairs(a, b, wide[byte_to_class][:, byte_to_class].astype(np.int8), -11, -1)).all()
    # the comparison knob (read once per process): the same scores from the 32-bit strips
    import subprocess
    import sys
    code = ("import numpy as np, stringwars_amd as sw\n"
            "scope = sw.DeviceScope(gpu_device=0)\n"
            "pa, pb = sw.generate_pairs('protein4k', 8, seed=5)\n"
            "engine = sw.NeedlemanWunschScores(substitution_matrix=sw.substitution_matrix(5), open=-11, extend=-1, capabilities=scope)\n"
            "scope.set_profiling(True)\n"
            "got = engine.pairs(pa, pb, scope)\n"
            "print(scope.last_timing()['dominant_name'], ' '.join(str(int(v)) for v in got))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for narrow in ("1", "0"):
        done = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_NWP_NARROW=narrow, PYTHONPATH=child_pythonpath()),
                              capture_output=True, text=True, timeout=300)
        assert done.returncode == 0, done.stderr[-2000:]
        lines.append(done.stdout.strip().splitlines()[-1].split(" ", 1))
    assert "narrow" in lines[0][0] and "narrow" not in lines[1][0] and lines[1][0].startswith("nwprofile_affine"), lines
    assert lines[0][1] == lines[1][1]
    pa, pb = sw.generate_pairs("protein4k", 8, seed=5)
    assert lines[0][1] == " ".join(str(int(v)) for v in orc.nw_pairs(pa, pb, sw.substitution_matrix(5), -11, -1))


def test_smith_waterman(sw, orc, scope):
    """`SmithWatermanScores` (bench.rs:882-963): KATs of SURVEY 8c, random matrices, multi-pass, cross-product."""
    cases = KAT["sw_unary_2_m1"]["cases"]
    a, b = sw.Strs([c[0] for c in cases]), sw.Strs([c[1] for c in cases])
    classes, costs = sw.unary_class_costs(2, -1)
    linear = sw.SmithWatermanScores(classes, costs, open=-2, extend=-2, capabilities=scope)
    affine = sw.SmithWatermanScores(classes, costs, open=-5, extend=-1, capabilities=scope)
    assert linear.pairs(a, b, scope).tolist() == [c[2] for c in cases]
    assert affine.pairs(a, b, scope).tolist() == [c[3] for c in cases]
    rng = np.random.default_rng(41)
    lengths = list(range(0, 40)) + [100, 129, 200, 400, 700, 1100, 1600, 2100]
    items_a, items_b = random_pairs(rng, 400, lengths, 24)
    items_a += [bytes(rng.integers(65, 85, 7000, dtype=np.uint8)), bytes(rng.integers(65, 85, 3300, dtype=np.uint8))]
    items_b += [items_a[-2][500:6800], bytes(rng.integers(65, 85, 7000, dtype=np.uint8))]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    for symmetric in (True, False):
        matrix = rng.integers(-8, 12, (256, 256)).astype(np.int8)
        if symmetric:
            matrix = np.minimum(matrix, matrix.T)
        for gaps in ((-4, -4), (-11, -1), (-2, -2)):
            engine = sw.SmithWatermanScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
            got = engine.pairs(a, b, scope)
            want = np.array([orc.nw_score(x, y, matrix, gaps[0], gaps[1], local=True) for x, y in zip(items_a, items_b)])
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (symmetric, gaps, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
    q = sw.Strs(items_a[:19])
    c = sw.Strs(items_b[:11])
    engine = sw.SmithWatermanScores(classes, costs, open=-2, extend=-2, capabilities=scope)
    got = engine(q, c, scope)
    m32 = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    want = np.array([[orc.nw_score(q[i], c[j], m32, -2, -2, local=True) for j in range(11)] for i in range(19)])
    assert (got == want).all()


def test_needleman_wunsch_multipass_and_cross(sw, orc, scope):
    rng = np.random.default_rng(29)
    matrix = rng.integers(-5, 9, (256, 256)).astype(np.int8)  # asymmetric: columns cannot be swapped
    items_a, items_b = random_pairs(rng, 4, [3300, 7000], 20, related=1.0)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    for gaps in ((-4, -4), (-11, -1)):
        engine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
        assert (engine.pairs(a, b, scope) == orc.nw_pairs(a, b, matrix, *gaps)).all(), gaps
    q = sw.Strs([bytes(rng.integers(65, 85, int(n), dtype=np.uint8)) for n in rng.integers(0, 60, 37)])
    c = sw.Strs([bytes(rng.integers(65, 85, int(n), dtype=np.uint8)) for n in rng.integers(0, 60, 23)])
    classes, costs = sw.unary_class_costs(2, -1)
    engine = sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope)
    got = engine(q, c, scope)
    rows = sw.Strs([q[i] for i in range(len(q)) for _ in range(len(c))])
    cols = sw.Strs([c[j] for _ in range(len(q)) for j in range(len(c))])
    m32 = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    assert got.dtype == np.int64 and (got == orc.nw_pairs(rows, cols, m32, -2, -2).reshape(len(q), len(c))).all()


# ----------------------------------------------------------------------------------------------------
# boundary behaviour: tape widths, residency, strides, errors
# ----------------------------------------------------------------------------------------------------
def test_tape_widths_residency_and_strides(sw, orc, scope):
    import ctypes as C
    from stringwars_amd import _native as N
    a, b = sw.generate_pairs("tokens64", 5000, seed=3)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope)
    assert (engine.pairs(a, b, scope) == want).all()                                   # u64, host
    a32, b32 = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    assert (engine.pairs(a32, b32, scope) == want).all()                               # u32, host
    da, db = a32.to_device(scope), b32.to_device(scope)
    assert (engine.pairs(da, db, scope) == want).all()                                 # u32, device, host out
    strided = np.full((5000, 3), 0xDEADBEEF, dtype=np.uint32)
    engine.pairs(da, db, scope, out=strided[:, 1])                                     # stride 12 bytes
    assert (strided[:, 1] == want).all() and (strided[:, 0] == 0xDEADBEEF).all() and (strided[:, 2] == 0xDEADBEEF).all()
    sub = engine.pairs(a.subview(100, 350), b.subview(100, 350), scope)                # zero-copy sub-views
    assert (sub == want[100:350]).all()
    import torch
    ta = sw.DeviceTape.from_torch(torch.from_numpy(a.data).cuda(), torch.from_numpy(a.offsets.astype(np.int64)).cuda())
    tb = sw.DeviceTape.from_torch(torch.from_numpy(b.data).cuda(), torch.from_numpy(b.offsets.astype(np.int64)).cuda())
    out = torch.zeros(5000, dtype=torch.int32, device="cuda")
    torch_scope = sw.DeviceScope(gpu_device=0, stream=torch.cuda.current_stream().cuda_stream)
    engine2 = sw.LevenshteinDistances(capabilities=torch_scope)
    engine2.pairs(ta, tb, torch_scope, out=out)                                        # all device, torch stream
    assert (out.cpu().numpy().astype(np.uint32) == want).all()
    pointer, err = C.c_void_p(), C.c_char_p()
    N.check(N.lib.swh_unified_alloc(scope.handle, 5000 * 4, C.byref(pointer), C.byref(err)), err)
    unified = np.ctypeslib.as_array(C.cast(pointer, C.POINTER(C.c_uint32)), shape=(5000,))
    engine.pairs(a, b, scope, out=unified)                                             # UnifiedAlloc output
    assert (unified == want).all()
    N.lib.swh_unified_free(scope.handle, pointer)


def test_results_are_out_when_a_synchronous_call_returns(sw, orc):
    """A synchronous call with its results in device memory returns when the kernel's summary has landed in host-mapped memory,
    not when the stream reports the kernel complete (api.hip: wait_for_summary); the results are written through and acknowledged
    before that word goes out (common.hpp: store_out, report_call_summary). So they must be readable by ANYBODY on return: here by
    a copy on another stream (torch's; the scope's own stream is non-blocking, nothing orders the two), call after call with
    fresh inputs and a poisoned output -- words (k_direct_short / k_short_tiled), tokens (the tiled kernel), a cross-product of
    words, alignment scores of words."""
    import torch
    rng = np.random.default_rng(77)
    scope = sw.DeviceScope(gpu_device=0)
    lev = sw.LevenshteinDistances(capabilities=scope)
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    matrix = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
    nw = sw.NeedlemanWunschScores(byte_to_class, class_costs, open=-2, extend=-2, capabilities=scope)
    copier = torch.cuda.Stream()

    def strs(count, lo, hi):
        lens = rng.integers(lo, hi + 1, count)
        offsets = np.zeros(count + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        return sw.Strs(data=rng.integers(97, 101, int(offsets[-1]), dtype=np.uint8), offsets=offsets)

    for round_no in range(30):
        count = int(rng.integers(300, 70000)) if round_no % 3 else int(rng.integers(300, 3000))
        lo, hi = ((1, 16) if round_no % 2 else (20, 90)) if round_no % 3 else (1, 12)
        a, b = strs(count, lo, hi), strs(count, lo, hi)
        da, db = a.to_device(scope), b.to_device(scope)
        if round_no % 3:
            out = torch.full((count,), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            (nw if round_no % 6 == 1 and hi <= 16 else lev).pairs(da, db, scope, out=out)
            with torch.cuda.stream(copier):
                got = out.to("cpu", non_blocking=False).numpy()
            if round_no % 6 == 1 and hi <= 16:
                want = orc.nw_pairs(a, b, matrix, -2, -2)
            else:
                want = orc.levenshtein_pairs(a, b, algo="hyyro")
            assert (got == want).all(), (round_no, count, np.nonzero(got != want)[0][:5])
        else:
            side = int(count ** 0.5)
            q, c = a.subview(0, side).to_device(scope), b.subview(0, side).to_device(scope)
            out = torch.full((side, side), -7, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            lev(q, c, scope, out=out)
            with torch.cuda.stream(copier):
                got = out.to("cpu", non_blocking=False).numpy()
            want = np.array([[orc.levenshtein(a[i], b[j]) for j in range(side)] for i in range(side)])
            assert (got == want).all(), (round_no, side)
            q.free(); c.free()
        da.free(); db.free()


def test_believed_tape_sizes_are_checked_on_the_device(sw, orc):
    """A UTF-8 call on raw device tapes believes the byte totals it read for the same tapes (pointers, count) last time instead of
    fetching offsets[count] again; k_utf8_finish compares and the call is redone with fresh totals when the tapes were rewritten
    in place -- shorter, longer, and back -- between calls (api.hip: size_belief; STRINGWARS_AMD_SIZE_BELIEF=0 always fetches)."""
    import torch
    rng = np.random.default_rng(91)
    scope = sw.DeviceScope(gpu_device=0)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    count = 3000

    def batch(lo, hi):
        cps = [0x41, 0x62, 0xE9, 0x416, 0x4E2D, 0x1F600]
        strings = ["".join(chr(cps[int(c)]) for c in rng.integers(0, len(cps), int(n))).encode() for n in rng.integers(lo, hi, count)]
        return sw.Strs(strings)

    versions = [(batch(20, 60), batch(20, 60)), (batch(5, 25), batch(30, 80)), (batch(40, 90), batch(1, 10)), (batch(20, 60), batch(20, 60))]
    room = max(max(len(a.data), len(b.data)) for a, b in versions) + 64
    data_a, data_b = torch.zeros(room, dtype=torch.uint8, device="cuda"), torch.zeros(room, dtype=torch.uint8, device="cuda")
    offs_a, offs_b = torch.zeros(count + 1, dtype=torch.int64, device="cuda"), torch.zeros(count + 1, dtype=torch.int64, device="cuda")
    ta, tb = sw.DeviceTape.from_torch(data_a, offs_a), sw.DeviceTape.from_torch(data_b, offs_b)
    for a, b in versions:
        data_a[:len(a.data)] = torch.from_numpy(a.data).cuda(); offs_a.copy_(torch.from_numpy(a.offsets.astype(np.int64)))
        data_b[:len(b.data)] = torch.from_numpy(b.data).cuda(); offs_b.copy_(torch.from_numpy(b.offsets.astype(np.int64)))
        torch.cuda.synchronize()
        want = orc.levenshtein_pairs(a, b, utf8=True)
        for _ in range(3):   # the first call on rewritten tapes meets the stale belief, the next ones the fresh one
            assert (engine.pairs(ta, tb, scope) == want).all()
        assert (engine.pairs(ta, tb, scope, bound=7) == np.minimum(want, 8)).all()


def test_ascii_tapes_through_the_utf8_engine_run_on_their_bytes(sw, orc):
    """Raw UTF-8 device tapes that held only ASCII the last time a scope staged them are scored on their bytes the next time (code
    points of ASCII text are its bytes: the word-sized and cross-product byte kernels instead of staging + code-point kernels), behind
    a kernel that checks every byte; a tape rewritten in place with text beyond ASCII is caught by that check and the call is done
    again the long way (api.hip: k_ascii_check). Pairs and cross-products, words and longer strings."""
    import torch
    rng = np.random.default_rng(123)
    scope = sw.DeviceScope(gpu_device=0)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    count = 2000

    def batch(lo, hi, cps):
        strings = ["".join(chr(cps[int(c)]) for c in rng.integers(0, len(cps), int(n))).encode() for n in rng.integers(lo, hi, count)]
        return sw.Strs(strings)

    ascii_cps, mixed_cps = list(range(97, 123)), [0x41, 0x62, 0xE9, 0x416, 0x4E2D, 0x1F600]
    for lo, hi in ((1, 12), (30, 90)):
        versions = [(batch(lo, hi, ascii_cps), batch(lo, hi, ascii_cps)), (batch(lo, hi, ascii_cps), batch(lo, hi, mixed_cps)),
                    (batch(lo, hi, ascii_cps), batch(lo, hi, ascii_cps))]
        # (a fourth version for the words: LONGER tapes whose bytes beyond the believed totals are not ASCII -- the check reads the totals itself)
        if hi <= 12:
            grown_b = sw.Strs([bytes(versions[2][1][i]) + ("\u0416\u4e2d".encode() if i >= count - 50 else b"") for i in range(count)])
            versions.append((versions[2][0], grown_b))
        room = max(max(len(a.data), len(b.data)) for a, b in versions) + 64
        data_a, data_b = torch.zeros(room, dtype=torch.uint8, device="cuda"), torch.zeros(room, dtype=torch.uint8, device="cuda")
        offs_a, offs_b = torch.zeros(count + 1, dtype=torch.int64, device="cuda"), torch.zeros(count + 1, dtype=torch.int64, device="cuda")
        ta, tb = sw.DeviceTape.from_torch(data_a, offs_a), sw.DeviceTape.from_torch(data_b, offs_b)
        names = []
        for a, b in versions:
            data_a[:len(a.data)] = torch.from_numpy(a.data).cuda(); offs_a.copy_(torch.from_numpy(a.offsets.astype(np.int64)))
            data_b[:len(b.data)] = torch.from_numpy(b.data).cuda(); offs_b.copy_(torch.from_numpy(b.offsets.astype(np.int64)))
            torch.cuda.synchronize()
            want = orc.levenshtein_pairs(a, b, utf8=True)
            for _ in range(3):
                scope.set_profiling(True)
                got = engine.pairs(ta, tb, scope)
                names.append(scope.last_timing()["dominant_name"])
                scope.set_profiling(False)
                assert (got == want).all(), names
            side = 40
            cross = np.array([[orc.levenshtein_utf8(a[i], b[j]) for j in range(side)] for i in range(side)])
            for _ in range(2):
                assert (engine(ta.subview(0, side) if hasattr(ta, "subview") else ta, tb.subview(0, side) if hasattr(tb, "subview") else tb, scope)[:side, :side] == cross).all()
        # token-sized strings name their kernels by symbol width -- ASCII version: staged once (code points), then on the byte kernel;
        # mixed version: code points throughout (the first call meets the stale belief and is redone); ASCII again: back on bytes
        if lo >= 30:
            assert names[0] == "bitparallel_tiled_u32" and names[1] == names[2] == "bitparallel_tiled", names
            assert names[3] == names[4] == names[5] == "bitparallel_tiled_u32", names
            assert names[6] == "bitparallel_tiled_u32" and names[7] == names[8] == "bitparallel_tiled", names


@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("gaps", [(-2, -2), (-5, -1)])
def test_alignment_of_tokens_up_to_64_bytes(sw, orc, local, gaps):
    """k_align_short with a register row of 64 cells: tokens of up to 64 bytes over ANY alphabet (multilingual words: a few of them reach
    past 32 bytes) stay on the lane-per-pair kernel instead of sending the whole batch to the planned path -- pairwise and cross-product,
    prepared tapes (lengths known) and raw ones (second call), lengths around 32 / 33 / 63 / 64, a 65-byte token falls back."""
    rng = np.random.default_rng(64 + local)
    scope = sw.DeviceScope(gpu_device=0)
    Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
    byte_to_class, costs = sw.unary_class_costs(2, -1)
    full = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    engine = Engine(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=scope)

    def token(n):
        return bytes(rng.integers(0, 256, int(n), dtype=np.uint8))

    lens = [0, 1, 31, 32, 33, 63, 64] + list(rng.integers(1, 65, 300))
    xs, ys = [token(n) for n in lens], [token(n) for n in rng.permutation(lens)]
    ys[10] = xs[10][:40] + token(5)
    want = np.array([orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for x, y in zip(xs, ys)])
    pa, pb = sw.PreparedTape(scope, sw.Strs(xs)), sw.PreparedTape(scope, sw.Strs(ys))
    scope.set_profiling(True)
    got = engine.pairs(pa, pb, scope)
    name = scope.last_timing()["dominant_name"]
    scope.set_profiling(False)
    assert name.startswith("align_short") and name.endswith("w64"), name
    assert (got == want).all(), np.nonzero(got != want)[0][:5]
    a, b = sw.Strs(xs), sw.Strs(ys)
    for _ in range(2):
        assert (engine.pairs(a, b, scope) == want).all()
    qs, cs = xs[:40], ys[:90]
    cross = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in cs] for x in qs])
    pq, pc = sw.PreparedTape(scope, sw.Strs(qs)), sw.PreparedTape(scope, sw.Strs(cs))
    for _ in range(2):   # (the first call may try the small-alphabet kernels: their condition fails on 256 byte values, the redo lands here)
        scope.set_profiling(True)
        got = engine(pq, pc, scope)
        name = scope.last_timing()["dominant_name"]
        scope.set_profiling(False)
        assert (got == cross).all()
    assert name.startswith("align_short") and name.endswith("w64"), name
    xs2, ys2 = xs + [token(3)], ys + [token(65)]
    want2 = np.array([orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for x, y in zip(xs2, ys2)])
    assert (engine.pairs(sw.PreparedTape(scope, sw.Strs(xs2)), sw.PreparedTape(scope, sw.Strs(ys2)), scope) == want2).all()


@pytest.mark.parametrize("local", [False, True])
def test_word_batches_with_a_few_long_tokens(sw, orc, local):
    """Word tokens with a few long ones among them (a URL, a sentence of a script that writes no spaces): k_align_short scores every pair
    of two strings that fit its 64 cells and reports that some did not; the redo plans ONLY the pairs with a longer string
    (PrepassArgs::skip_upto) and must leave the others' scores alone. Cross-product and pairs, prepared and raw tapes (second call: the
    first one's statistics), every score against the oracle; the second call's dominant kernel is the lane kernel's."""
    rng = np.random.default_rng(72 + local)
    scope = sw.DeviceScope(gpu_device=0)
    Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
    byte_to_class, costs = sw.unary_class_costs(2, -1)
    full = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    engine = Engine(byte_to_class, costs, open=-5, extend=-1, capabilities=scope)

    def token(n):
        return bytes(rng.integers(0, 256, int(n), dtype=np.uint8))

    qs = [token(n) for n in list(rng.integers(1, 20, 60)) + [70, 130]]
    cs = [token(n) for n in list(rng.integers(1, 24, 150)) + [65, 300, 64]]
    want = np.array([[orc.nw_score(x, y, full, -5, -1, local=local) for y in cs] for x in qs])
    pq, pc = sw.PreparedTape(scope, sw.Strs(qs)), sw.PreparedTape(scope, sw.Strs(cs))
    for _ in range(2):
        out = engine(pq, pc, scope)
        assert (out == want).all(), np.argwhere(out != want)[:5]
    raw_q, raw_c = sw.Strs(qs), sw.Strs(cs)
    for _ in range(3):
        assert (engine(raw_q, raw_c, scope) == want).all()
    xs = [token(n) for n in list(rng.integers(1, 20, 500)) + [90]]
    ys = [token(n) for n in list(rng.integers(1, 20, 500)) + [5]]
    want_pairs = np.array([orc.nw_score(x, y, full, -5, -1, local=local) for x, y in zip(xs, ys)])
    pa, pb = sw.PreparedTape(scope, sw.Strs(xs)), sw.PreparedTape(scope, sw.Strs(ys))
    for _ in range(2):
        assert (engine.pairs(pa, pb, scope) == want_pairs).all()


def test_cross_product_of_word_sized_code_points(sw, orc, scope):
    """k_cross_short_cp (cross.hip): queries x candidates of up to 32 CODE POINTS each -- `LevenshteinDistancesUtf8` on word-sized tokens of
    several scripts (1 .. 4-byte sequences, symbols that collide in single groups of their three-bit group tables), empty strings,
    lengths up to exactly 32 -- against the oracle; the second call on the same scope (lengths known from the first) must run on
    `cross_short_u32`, a candidate of 33 code points takes the general path and scores the same."""
    rng = np.random.default_rng(55)
    cps = [0x61, 0x69, 0x71, 0xE9, 0xE1, 0x430, 0x438, 0x4E2D, 0x4E25, 0x1F600, 0x1F608, 0x10FFFF, 0x7F, 0x80, 0x7FF, 0x800, 0xFFFF, 0x10000]

    def word(n):
        return "".join(chr(cps[int(c)]) for c in rng.integers(0, len(cps), int(n))).encode()

    queries = [word(n) for n in [0, 1, 2, 31, 32] + list(rng.integers(1, 12, 95))]
    candidates = [word(n) for n in [0, 1, 32, 32, 5] + list(rng.integers(1, 14, 200))]
    candidates[7] = queries[9] + word(2)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    want = np.array([[orc.levenshtein_utf8(x, y) for y in candidates] for x in queries])
    q, c = sw.Strs(queries).to_device(scope), sw.Strs(candidates).to_device(scope)
    assert (engine(q, c, scope) == want).all()
    scope.set_profiling(True)
    got = engine(q, c, scope)
    name = scope.last_timing()["dominant_name"]
    scope.set_profiling(False)
    assert name == "cross_short_u32", name
    assert (got == want).all(), np.argwhere(got != want)[:5]
    longer = candidates + [word(33)]
    want2 = np.array([[orc.levenshtein_utf8(x, y) for y in longer] for x in queries])
    c2 = sw.Strs(longer).to_device(scope)
    for _ in range(2):
        assert (engine(q, c2, scope) == want2).all()
    q.free(); c.free(); c2.free()


def test_pipelined_scope_lanes(sw, orc):
    """Pipelined mode alternates calls between two internal lanes; results must be complete after synchronize()
    (or, on the scope's own stream, after join()) and identical to the synchronous path."""
    import torch
    torch_scope = sw.DeviceScope(gpu_device=0, stream=torch.cuda.current_stream().cuda_stream)
    engine = sw.LevenshteinDistances(capabilities=torch_scope)
    batches = []
    for seed, workload, count in ((1, "tokens64", 30000), (2, "words16", 50000), (3, "tokens64", 7000), (4, "short_words", 90000),
                                  (5, "tokens64", 30000), (6, "utf8_lines", 300)):
        a, b = sw.generate_pairs(workload, count, seed=seed)
        batches.append((a.to_device(torch_scope), b.to_device(torch_scope), orc.levenshtein_pairs(a, b, algo="hyyro"),
                        torch.zeros(count, dtype=torch.int32, device="cuda")))
    torch_scope.set_async(True)
    torch_scope.set_pipelined(True)
    for _ in range(3):
        for da, db, _, out in batches:
            out.zero_()
        for da, db, _, out in batches:
            engine.pairs(da, db, torch_scope, out=out)
        torch_scope.join()                       # torch's stream now waits for the last call only
        last = batches[-1][3].cpu().numpy().astype(np.uint32)
        assert (last == batches[-1][2]).all()
        torch_scope.synchronize()
        for _, _, want, out in batches:
            assert (out.cpu().numpy().astype(np.uint32) == want).all()
    torch_scope.set_pipelined(False)
    torch_scope.set_async(False)
    da, db, want, out = batches[0]
    engine.pairs(da, db, torch_scope, out=out)
    assert (out.cpu().numpy().astype(np.uint32) == want).all()


# ----------------------------------------------------------------------------------------------------
# prepared tapes and the plan-free kernels (tiled bit-parallel, direct-short on its own)
# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("workload,count", [("words16", 10_000), ("tokens64", 60_000), ("short_words", 150_000), ("utf8_lines", 600)])
def test_prepared_byte_tapes(sw, orc, scope, workload, count):
    """`swh_tape_prepare_*` + `*_pairs_prepared`: same distances as the raw-tape call, whole tapes and sub-views
    (bench.rs:134-139), bounded and unbounded, host-resident and device-resident sources."""
    a, b = sw.generate_pairs(workload, count, seed=7)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope)
    pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
    info = pa.info
    assert info["count"] == count and info["bytes"] == int(a.offsets[-1]) and info["longest"] == int(a.lengths.max()) and not info["utf8"]
    assert (engine.pairs(pa, pb, scope) == want).all()
    for bound in (0, 3, 40):
        assert (engine.pairs(pa, pb, scope, bound=bound) == np.minimum(want, bound + 1)).all()
    lo, hi = count // 3, count // 3 + count // 2
    assert (engine.pairs(pa[lo:hi], pb[lo:hi], scope) == want[lo:hi]).all()
    assert engine.pairs(pa[5:5], pb[9:9], scope).size == 0
    da, db = a.to_device(scope), b.to_device(scope)                         # device tapes are used in place
    qa, qb = sw.PreparedTape(scope, da), sw.PreparedTape(scope, db)
    assert (engine.pairs(qa, qb, scope) == want).all()
    with pytest.raises(TypeError):
        engine.pairs(pa, b, scope)                                          # both prepared, or neither
    u32a = sw.PreparedTape(scope, a.with_offsets(np.uint32))
    with pytest.raises(sw.StringWarsError):
        engine.pairs(u32a, pb, scope)                                       # one offset width per call
    u32b = sw.PreparedTape(scope, b.with_offsets(np.uint32))
    assert (engine.pairs(u32a, u32b, scope) == want).all()


def test_prepared_utf8_tapes(sw, orc, scope):
    """UTF-8 is validated and decoded once, at prepare time (`CharsTapeView::try_from`, bench.rs:303-306)."""
    a, b = sw.generate_pairs("utf8_lines", 1500, seed=11)
    want = orc.levenshtein_pairs(a, b, utf8=True)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)
    info = pa.info
    assert info["utf8"] and not info["ascii"] and info["symbols"] == sum(len(a[i].decode()) for i in range(1500))
    assert info["longest"] == max(len(a[i].decode()) for i in range(1500))
    assert (engine.pairs(pa, pb, scope) == want).all()
    for bound in (0, 7, 32, 63, 500):
        assert (engine.pairs(pa, pb, scope, bound=bound) == np.minimum(want, bound + 1)).all()
    assert (engine.pairs(pa[100:900], pb[100:900], scope, bound=32) == np.minimum(want[100:900], 33)).all()
    for algorithm in ("wavefront", "bitparallel", "tiled"):
        other = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm=algorithm)
        assert (other.pairs(pa[:300], pb[:300], scope) == want[:300]).all()
    with pytest.raises(ValueError):
        sw.LevenshteinDistances(capabilities=scope).pairs(pa, pb, scope)    # byte engine, code-point tapes
    with pytest.raises(sw.StringWarsError) as info:
        sw.PreparedTape(scope, sw.Strs([b"ok", b"\xe2\x82", b"fine"]), utf8=True)
    assert info.value.status == "invalid_utf8"
    # pure ASCII prepared as UTF-8: code points are bytes, the byte kernels run
    ta, tb = sw.generate_pairs("tokens64", 20_000, seed=3)
    xa, xb = sw.PreparedTape(scope, ta, utf8=True), sw.PreparedTape(scope, tb, utf8=True)
    assert xa.info["ascii"] and xa.info["symbols"] == xa.info["bytes"]
    assert (engine.pairs(xa, xb, scope) == orc.levenshtein_pairs(ta, tb, algo="hyyro")).all()
    mixed = engine.pairs(xa[:1500], pb, scope)                              # ASCII against non-ASCII: code points
    assert (mixed == orc.levenshtein_pairs(ta.subview(0, 1500), b, utf8=True)).all()
    # whether a call is accepted does not depend on what the tapes hold: u32 offsets on one side, u64 on the other work for
    # ASCII tapes (which otherwise take the byte kernels' shortcut) exactly as for non-ASCII ones
    narrow = sw.PreparedTape(scope, ta.with_offsets(np.uint32), utf8=True)
    assert (engine.pairs(narrow, xb, scope) == orc.levenshtein_pairs(ta, tb, algo="hyyro")).all()
    assert (engine.pairs(sw.PreparedTape(scope, a.with_offsets(np.uint32), utf8=True), pb, scope, bound=32) == np.minimum(want, 33)).all()


def test_prepared_cross_product_and_alignment(sw, orc, scope):
    rng = np.random.default_rng(5)
    items_a, items_b = random_pairs(rng, 70, list(range(0, 50)) + [90, 130], 20)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
    engine = sw.LevenshteinDistances(capabilities=scope)
    want = np.array([[orc.levenshtein(x, y) for y in items_b] for x in items_a], dtype=np.uint64)
    assert (engine(pa, pb, scope) == want).all()
    assert (engine(pa[10:30], pb[40:70], scope) == want[10:30, 40:70]).all()
    self_product = engine(pa, None, scope)
    assert (self_product == np.array([[orc.levenshtein(x, y) for y in items_a] for x in items_a], dtype=np.uint64)).all()
    matrix = np.minimum(rng.integers(-6, 9, (256, 256)), rng.integers(-6, 9, (256, 256)).T).astype(np.int8)
    for cls, local in ((sw.NeedlemanWunschScores, False), (sw.SmithWatermanScores, True)):
        for gaps in ((-3, -3), (-7, -1)):
            scorer = cls(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
            expected = np.array([orc.nw_score(x, y, matrix, gaps[0], gaps[1], local=local) for x, y in zip(items_a, items_b)])
            assert (scorer.pairs(pa, pb, scope) == expected).all()
            assert (scorer.pairs(a, b, scope) == expected).all()
            grid = scorer(pa[:8], pb[:9], scope)
            assert grid.tolist() == [[orc.nw_score(x, y, matrix, gaps[0], gaps[1], local=local) for y in items_b[:9]] for x in items_a[:8]]


def test_cross_product_of_words_shares_the_query_table(sw, orc, scope):
    """The reference's own call shape on word-sized strings (`compute_into(queries, candidates, &mut matrix)`,
    bench.rs:478-486): k_cross_short keeps 64 candidates in registers and builds each query's match table once per wave.
    Lengths 0..32 on both sides, ragged counts, u64 matrix with a row stride, prepared and raw tapes; a string of more
    than 32 bytes on either side sends the call back to the general path."""
    rng = np.random.default_rng(43)
    def words(count, longest):
        return [rng.integers(97, 101, int(rng.integers(0, longest + 1)), dtype=np.uint8).tobytes() for _ in range(count)]
    queries, candidates = words(130, 32), words(333, 32)
    queries[7], candidates[11] = b"", b""
    queries[9], candidates[12] = b"a" * 32, b"a" * 31 + b"b"
    q, c = sw.Strs(queries), sw.Strs(candidates)
    flat_q = sw.Strs([x for x in queries for _ in candidates])
    flat_c = sw.Strs(candidates * len(queries))
    want = orc.levenshtein_pairs(flat_q, flat_c, algo="hyyro").reshape(len(queries), len(candidates)).astype(np.uint64)
    engine = sw.LevenshteinDistances(capabilities=scope)
    pq, pc = sw.PreparedTape(scope, q), sw.PreparedTape(scope, c)
    scope.set_profiling(True)
    got = engine(pq, pc, scope)
    timing = scope.last_timing()
    scope.set_profiling(False)
    assert (got == want).all()
    assert timing["dominant_name"] == "cross_short" and timing["cells"] == int(q.lengths.sum()) * int(c.lengths.sum())
    assert (engine(pq[10:77], pc[100:333], scope) == want[10:77, 100:333]).all()
    assert (engine(pc, None, scope) == orc.levenshtein_pairs(sw.Strs([x for x in candidates for _ in candidates]), sw.Strs(candidates * len(candidates)),
                                                             algo="hyyro").reshape(len(candidates), len(candidates))).all()
    dq, dc = q.to_device(scope), c.to_device(scope)
    for _ in range(2):                                                       # raw tapes: the second call rides on the first one's lengths
        assert (engine(dq, dc, scope) == want).all()
    # a long string on either side: redone on the general path, same answers
    long_q = sw.Strs(queries[:20] + [b"ab" * 40])
    long_c = sw.Strs(candidates[:50] + [b"ba" * 300])
    want_long = orc.levenshtein_pairs(sw.Strs([long_q[i] for i in range(len(long_q)) for _ in range(len(long_c))]),
                                      sw.Strs([long_c[j] for _ in range(len(long_q)) for j in range(len(long_c))]), algo="hyyro")
    assert (engine(long_q.to_device(scope), long_c.to_device(scope), scope).reshape(-1) == want_long).all()
    assert (engine(sw.PreparedTape(scope, long_q), sw.PreparedTape(scope, long_c), scope).reshape(-1) == want_long).all()


def test_tiled_kernel_every_block_count(sw, orc, scope):
    """The tiled kernel forced onto strings of 1 .. 2048 symbols (classes G = 1 .. 64 inside one tile, left-overs moving
    up a class, partly filled items), bytes and code points, bounded and not."""
    rng = np.random.default_rng(31)
    lengths = list(range(1, 70)) + [95, 96, 97, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1024, 1025, 1500, 2047, 2048]
    items_a, items_b = random_pairs(rng, 2600, lengths, 5)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm="tiled")
    assert (engine.pairs(a, b, scope) == want).all()
    assert (engine.pairs(b, a, scope) == want).all()
    pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
    assert (engine.pairs(pa, pb, scope) == want).all()
    for bound in (0, 5, 33, 200):
        assert (engine.pairs(pa, pb, scope, bound=bound) == np.minimum(want, bound + 1)).all()
    chars = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm="tiled")
    ua, ub = sw.generate_pairs("utf8_lines", 700, seed=13)
    assert (chars.pairs(ua, ub, scope) == orc.levenshtein_pairs(ua, ub, utf8=True)).all()
    # code points over the same 1 .. 2048 block counts: the ten-wave workgroups (640 threads) once scanned only 640 of the
    # 1024 key-counter words -- classes >= 40 (patterns beyond 1280 code points) came out wrong
    script = "\u00e9\u4e2d\U0001f600z\u0416"
    ca = sw.Strs(["".join(script[v] for v in s) for s in items_a])
    cb = sw.Strs(["".join(script[v] for v in s) for s in items_b])
    assert (chars.pairs(ca, cb, scope) == want).all()
    assert (chars.pairs(cb, ca, scope, bound=33) == np.minimum(want, 34)).all()
    pca, pcb = sw.PreparedTape(scope, ca, utf8=True), sw.PreparedTape(scope, cb, utf8=True)
    assert (chars.pairs(pca, pcb, scope) == want).all()


def test_tiled_kernel_full_tiles_of_code_points(sw, orc, scope):
    """More code-point pairs than 640 per tile and workgroup slot (tiles hold up to 1024 pairs, ten-wave workgroups have
    640 threads: each thread classifies two pairs of a full tile), raw and prepared."""
    a, b = sw.generate_pairs("words16", 400_000, seed=29)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    def cyrillic(tape):   # every byte becomes one two-byte code point: same distances
        data = np.empty((tape.data.size, 2), np.uint8)
        data[:, 0] = 0xD0
        data[:, 1] = 0x90 + (tape.data & 31)
        return sw.Strs(data=data.reshape(-1), offsets=tape.offsets * 2)
    assert len(set(int(v) & 31 for v in b"abcdefghijklmnopqrstuvwxyz")) == 26
    ua, ub = cyrillic(a), cyrillic(b)
    chars = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm="tiled")
    assert (chars.pairs(ua, ub, scope) == want).all()
    pa, pb = sw.PreparedTape(scope, ua, utf8=True), sw.PreparedTape(scope, ub, utf8=True)
    assert (chars.pairs(pa, pb, scope) == want).all()
    assert (chars.pairs(pa, pb, scope, bound=7) == np.minimum(want, 8)).all()


def test_pairs_with_common_affixes(sw, orc, scope):
    """Kernels may cut what a pair shares at both ends before the DP (k_short_tiled does, up to 8 bytes each): distances
    must not notice. Pairs built around every interesting shape -- shared prefix / suffix of 0..70 symbols, identical
    strings, one string a prefix / suffix / infix of the other, repeats that let prefix and suffix compete for the same
    symbols -- as bytes and as code points, bounded and not, raw and prepared."""
    rng = np.random.default_rng(47)
    def word(n, alphabet="ab"):
        return "".join(rng.choice(list(alphabet), n))
    items_a, items_b = [], []
    for pre in (0, 1, 3, 4, 5, 7, 8, 15, 16, 17, 47, 48, 49, 70):
        for suf in (0, 1, 3, 4, 5, 8, 16, 47, 48, 49, 70):
            head, tail = word(pre, "abcd"), word(suf, "abcd")
            items_a.append(head + word(int(rng.integers(0, 20))) + tail)
            items_b.append(head + word(int(rng.integers(0, 20))) + tail)
    for n in (1, 2, 5, 31, 32, 33, 64, 100):
        s = word(n, "abc")
        items_a += [s, s, s + "x", "x" + s, s, s * 2, "a" * n, "a" * n, "ab" * n]
        items_b += [s, s + word(7), s, s, s[: n // 2], s, "a" * (n + 3), "a" * max(n - 1, 0), "ba" * n]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm="tiled")
    for tapes in ((a, b), (b, a), (sw.PreparedTape(scope, a), sw.PreparedTape(scope, b))):
        assert (engine.pairs(tapes[0], tapes[1], scope) == want).all()
        for bound in (0, 2, 9):
            assert (engine.pairs(tapes[0], tapes[1], scope, bound=bound) == np.minimum(want, bound + 1)).all()
    accents = str.maketrans({"a": "\u00e9", "b": "\u4e2d", "c": "\U0001f600", "d": "z"})
    ua, ub = sw.Strs([s.translate(accents) for s in items_a]), sw.Strs([s.translate(accents) for s in items_b])
    chars = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm="tiled")
    assert (chars.pairs(ua, ub, scope) == want).all()
    assert (chars.pairs(sw.PreparedTape(scope, ua, utf8=True), sw.PreparedTape(scope, ub, utf8=True), scope, bound=4) == np.minimum(want, 5)).all()
    grid = engine(a.subview(0, 40), b.subview(100, 150), scope)             # cross-product: an affix per (query, candidate)
    flat = orc.levenshtein_pairs(sw.Strs([items_a[i] for i in range(40) for _ in range(50)]), sw.Strs(items_b[100:150] * 40), algo="hyyro")
    assert (grid.reshape(-1) == flat).all()


def test_word_sized_batches_on_the_chunked_kernel(sw, orc, request):
    """k_short_tiled (strings <= 16 bytes, pairwise): chunks of the tapes staged in LDS, common affixes cut, pairs sorted by
    what remains. Every shape that steers it: affixes of every length around every residue, the 8-byte compare window,
    empty strings, 16-byte strings back to back (segments beyond the LDS capacity: the chunk is halved), chunk and tile
    edges, both offset widths, strided outputs, bounds, sub-views that start in the middle of a tape, and a
    batch that stops being word-sized (the kernel reports it, the call is redone on another route)."""
    # batches below 64 K pairs take k_direct_short otherwise (the threshold is read once per process: a child process with it set)
    if not run_in_child(request, env={"STRINGWARS_AMD_SHORT_MIN_PAIRS": "1"}, test_library=True):
        return
    scope = sw.DeviceScope(gpu_device=0)
    rng = np.random.default_rng(53)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    def word(n, k=26):
        return letters[rng.integers(0, k, n)].tobytes()
    items_a, items_b = [], []
    for pre in range(0, 12):
        for suf in range(0, 12):
            for ra, rb in ((0, 0), (1, 0), (0, 2), (1, 1), (2, 3), (4, 4), (3, 1)):
                if pre + suf + max(ra, rb) > 16:
                    continue
                head, tail = word(pre, 3), word(suf, 3)
                items_a.append(head + word(ra, 2) + tail)
                items_b.append(head + word(rb, 2) + tail)
    for _ in range(3000):                       # unrelated words, every length 0..16, tiny alphabets (accidental affixes)
        items_a.append(word(int(rng.integers(0, 17)), int(rng.choice([1, 2, 26]))))
        items_b.append(word(int(rng.integers(0, 17)), int(rng.choice([1, 2, 26]))))
    items_a += [b"", b"", b"a", b"abcdefghijklmnop", b"abcdefghijklmnop", b"abcdefghijklmnop", b"aaaaaaaaaaaaaaaa"]
    items_b += [b"", b"abc", b"", b"abcdefghijklmnop", b"bcdefghijklmnopq", b"", b"aaaaaaaaaaaaaaa"]
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    engine = sw.LevenshteinDistances(capabilities=scope)
    for offsets in (np.uint64, np.uint32):
        pa, pb = sw.PreparedTape(scope, a.with_offsets(offsets)), sw.PreparedTape(scope, b.with_offsets(offsets))
        assert pa.info["longest"] <= 16
        assert (engine.pairs(pa, pb, scope) == want).all()
        assert (engine.pairs(pb, pa, scope) == want).all()
        for bound in (0, 1, 3, 15, 16, 40):
            assert (engine.pairs(pa, pb, scope, bound=bound) == np.minimum(want, bound + 1)).all()
        for lo, hi in ((1, 2), (7, 1031), (1000, 1001 + 1024), (len(items_a) - 3, len(items_a))):
            assert (engine.pairs(pa[lo:hi], pb[lo:hi], scope) == want[lo:hi]).all()
    strided = np.full((len(items_a), 3), 0xDEADBEEF, dtype=np.uint32)
    engine.pairs(pa, pb, scope, out=strided[:, 1])
    assert (strided[:, 1] == want).all() and (strided[:, 0] == 0xDEADBEEF).all() and (strided[:, 2] == 0xDEADBEEF).all()
    # raw tapes: the scope learns the lengths from its first call, the second call takes the chunked kernel
    fresh = sw.DeviceScope(gpu_device=0)
    da, db = a.to_device(fresh), b.to_device(fresh)
    for _ in range(2):
        assert (engine.pairs(da, db, fresh) == want).all()
    # one match table or two: a chunk whose bytes share their upper three bits (lower-case words; upper-case words; digits) takes
    # the single 32-entry table, any other chunk the nibble tables -- regions of each kind in one batch, symbols that collide
    # in their low five bits ('a' / 'A' / '!' / 0x81), and a single foreign byte in an otherwise lower-case region
    def region(count, alphabet):
        pool = np.frombuffer(bytes(alphabet), np.uint8)
        out_a, out_b = [], []
        for _ in range(count):
            x = pool[rng.integers(0, len(pool), int(rng.integers(0, 17)))].tobytes()
            y = bytearray(x)
            for _ in range(int(rng.integers(0, 4))):
                if y and rng.random() < 0.7:
                    y[int(rng.integers(0, len(y)))] = int(pool[rng.integers(0, len(pool))])
                elif len(y) < 16:
                    y.insert(int(rng.integers(0, len(y) + 1)), int(pool[rng.integers(0, len(pool))]))
            out_a.append(x); out_b.append(bytes(y) if rng.random() < 0.7 else pool[rng.integers(0, len(pool), int(rng.integers(0, 17)))].tobytes())
        return out_a, out_b
    mixed_a, mixed_b = [], []
    for count, alphabet in ((2500, range(97, 123)), (2500, range(0, 256)), (1500, range(65, 91)), (1500, b"aA!\x81\xa1\xc1\xe1bB\""),
                            (1500, range(48, 58)), (2500, range(97, 123)), (1200, list(range(97, 123)) + [32])):
        ra, rb = region(count, alphabet)
        mixed_a += ra; mixed_b += rb
    mixed_a[9000] = b"wordWord"; mixed_b[9000] = b"wordword"                 # one capital in the last-but-one lower-case region
    ma, mb = sw.Strs(mixed_a), sw.Strs(mixed_b)
    mwant = orc.levenshtein_pairs(ma, mb, algo="hyyro")
    for offsets in (np.uint64, np.uint32):
        mpa, mpb = sw.PreparedTape(scope, ma.with_offsets(offsets)), sw.PreparedTape(scope, mb.with_offsets(offsets))
        assert (engine.pairs(mpa, mpb, scope) == mwant).all() and (engine.pairs(mpb, mpa, scope) == mwant).all()
        assert (engine.pairs(mpa[2000:9500], mpb[2000:9500], scope, bound=2) == np.minimum(mwant[2000:9500], 3)).all()
    # 16-byte strings only: 1024 pairs are 16 KB per tape, twice what a chunk may hold
    full_a, full_b = sw.Strs([word(16, 4) for _ in range(5000)]), sw.Strs([word(16, 4) for _ in range(5000)])
    fa, fb = sw.PreparedTape(scope, full_a), sw.PreparedTape(scope, full_b)
    assert (engine.pairs(fa, fb, scope) == orc.levenshtein_pairs(full_a, full_b, algo="hyyro")).all()
    # tapes shorter than one 16-byte load
    ta, tb = sw.Strs([b"ab", b"", b"xyz"]), sw.Strs([b"b", b"q", b"xyz"])
    assert engine.pairs(sw.PreparedTape(scope, ta), sw.PreparedTape(scope, tb), scope).tolist() == [1, 1, 0]
    # the belief breaks: words, then a batch with a 40-byte string in the middle of the words
    for _ in range(2):
        assert (engine.pairs(da, db, fresh) == want).all()
    longer_a = sw.Strs(items_a[:700] + [word(40, 5)] + items_a[700:1500])
    longer_b = sw.Strs(items_b[:700] + [word(33, 5)] + items_b[700:1500])
    assert (engine.pairs(longer_a.to_device(fresh), longer_b.to_device(fresh), fresh) == orc.levenshtein_pairs(longer_a, longer_b, algo="hyyro")).all()
    # full-size shapes: every pair of two workloads against the oracle
    for workload, count in (("short_words", 300_000), ("words16", 100_000)):
        wa, wb = sw.generate_pairs(workload, count, seed=11)
        wpa, wpb = sw.PreparedTape(scope, wa.with_offsets(np.uint32)), sw.PreparedTape(scope, wb.with_offsets(np.uint32))
        full = orc.levenshtein_pairs(wa, wb, algo="hyyro")
        scope.set_profiling(True)
        assert (engine.pairs(wpa, wpb, scope) == full).all()
        timing = scope.last_timing()
        scope.set_profiling(False)
        assert timing["dominant_name"] == "short_tiled" and timing["cells"] == int((wa.lengths.astype(np.int64) * wb.lengths.astype(np.int64)).sum())
        assert (engine.pairs(wpa, wpb, scope, bound=2) == np.minimum(full, 3)).all()


def test_plan_free_route_falls_back_when_lengths_grow(sw, orc):
    """Raw tapes: the scope believes the next batch looks like the last one and skips the pre-pass; the kernels verify
    the belief per pair, and a batch that breaks it (both strings beyond 2048 symbols, or words turned into lines) is
    redone on the planned path -- same results either way."""
    scope = sw.DeviceScope(gpu_device=0)
    engine = sw.LevenshteinDistances(capabilities=scope)
    rng = np.random.default_rng(37)
    words_a, words_b = sw.generate_pairs("short_words", 40_000, seed=2)
    tokens_a, tokens_b = sw.generate_pairs("tokens64", 30_000, seed=2)
    long_a, long_b = random_pairs(rng, 40, [5, 60, 2100, 2500, 3000, 5000], 4)
    long_a, long_b = sw.Strs(long_a), sw.Strs(long_b)
    batches = [(words_a, words_b), (words_a, words_b), (tokens_a, tokens_b), (tokens_a, tokens_b), (long_a, long_b),
               (long_a, long_b), (words_a, words_b), (tokens_a, tokens_b), (long_a, long_b), (words_a, words_b)]
    for a, b in batches:
        assert (engine.pairs(a, b, scope) == orc.levenshtein_pairs(a, b, algo="hyyro")).all()
        da, db = a.to_device(scope), b.to_device(scope)
        assert (engine.pairs(da, db, scope, bound=9) == np.minimum(orc.levenshtein_pairs(a, b, algo="hyyro"), 10)).all()
    # the same for code points: word-sized batches, then lines of 700 .. 2000 code points on the strength of that belief
    chars = sw.LevenshteinDistancesUTF8(capabilities=scope)
    script = [0x41, 0x416, 0x4E2D, 0x1F600]
    def line(n):
        return "".join(chr(script[i]) for i in rng.integers(0, 4, n))
    short_a, short_b = sw.Strs([line(int(n)) for n in rng.integers(1, 20, 500)]), sw.Strs([line(int(n)) for n in rng.integers(1, 20, 500)])
    for low, high in ((700, 1400), (1000, 2000), (2000, 2600)):
        lines_a = sw.Strs([line(int(n)) for n in rng.integers(low, high, 50)])
        lines_b = sw.Strs([line(int(n)) for n in rng.integers(low, high, 50)])
        for a, b in ((short_a, short_b), (lines_a, lines_b), (short_a, short_b), (lines_a, lines_b)):
            assert (chars.pairs(a, b, scope) == orc.levenshtein_pairs(a, b, utf8=True)).all()
    # forced tiled with a pair beyond its reach: flagged, redone
    tiled = sw.LevenshteinDistances(capabilities=scope, algorithm="tiled")
    assert (tiled.pairs(long_a, long_b, scope) == orc.levenshtein_pairs(long_a, long_b, algo="hyyro")).all()


def test_prepared_tapes_in_asynchronous_and_pipelined_scopes(sw, orc):
    import torch
    scope = sw.DeviceScope(gpu_device=0, stream=torch.cuda.current_stream().cuda_stream)
    engine = sw.LevenshteinDistances(capabilities=scope)
    work = []
    for seed, workload, count in ((1, "tokens64", 50_000), (2, "words16", 30_000), (3, "short_words", 200_000), (4, "tokens64", 9_000)):
        a, b = sw.generate_pairs(workload, count, seed=seed)
        work.append((sw.PreparedTape(scope, a), sw.PreparedTape(scope, b), orc.levenshtein_pairs(a, b, algo="hyyro"),
                     torch.zeros(count, dtype=torch.int32, device="cuda")))
    for pipelined in (False, True):
        scope.set_async(True)
        scope.set_pipelined(pipelined)
        for _ in range(3):
            for pa, pb, _, out in work:
                out.zero_()
                engine.pairs(pa, pb, scope, out=out)
            scope.synchronize()
            for _, _, want, out in work:
                assert (out.cpu().numpy().astype(np.uint32) == want).all()
        scope.set_pipelined(False)
        scope.set_async(False)
    # A prepared tape is measured once; its device memory is the caller's. If the caller changes it afterwards -- here: every
    # first string grows beyond what the plan-free kernel can score -- an asynchronous call cannot be redone
    # behind the caller's back: the next synchronisation reports it instead of leaving stale distances unnoticed.
    a, b = sw.generate_pairs("tokens64", 40_000, seed=9)
    da, db = a.to_device(scope), b.to_device(scope)
    pa, pb = sw.PreparedTape(scope, da), sw.PreparedTape(scope, db)
    out = torch.zeros(40_000, dtype=torch.int32, device="cuda")
    assert (engine.pairs(pa, pb, scope, out=out).cpu().numpy().astype(np.uint32) == orc.levenshtein_pairs(a, b, algo="hyyro")).all()
    import ctypes as C
    from stringwars_amd import _native as N
    for strs, tape in ((a, da), (b, db)):
        grown = strs.offsets[:62].copy()
        grown[1:61] = grown[61]        # the first string now spans 61 tokens (~3.9 KB: more than the 64 blocks of 32 rows the plan-free
                                       # kernel can take as a pattern -- on both sides, so neither can be the text), sixty are empty
        N.check(N.lib.swh_copy_to_device(scope.handle, C.c_void_p(tape.offsets_ptr), grown.ctypes.data, grown.nbytes, None), C.c_char_p())
    scope.set_async(True)
    engine.pairs(pa, pb, scope, out=out)
    with pytest.raises(sw.StringWarsError) as info:
        scope.synchronize()
    assert info.value.status == "invalid_argument" and "not scored" in str(info.value)
    scope.synchronize()                                           # reported once
    # ... and it is reported whichever call it happened in: the summary of an asynchronous call is overwritten by the next call's
    # and is only read when the host synchronises right behind it, so the faulty call is followed here by calls on INTACT tapes
    # (their summaries are clean) -- plain asynchronous, then on the two pipeline lanes, where a lane carries two calls per sync.
    ga, gb = sw.generate_pairs("tokens64", 30_000, seed=10)
    good_a, good_b = sw.PreparedTape(scope, ga.to_device(scope)), sw.PreparedTape(scope, gb.to_device(scope))
    good_out = torch.zeros(30_000, dtype=torch.int32, device="cuda")
    want_good = orc.levenshtein_pairs(ga, gb, algo="hyyro")
    for pipelined, calls_after in ((False, 1), (False, 3), (True, 2), (True, 5)):
        scope.set_pipelined(pipelined)
        engine.pairs(pa, pb, scope, out=out)                     # the call that cannot score its first pair
        for _ in range(calls_after):
            engine.pairs(good_a, good_b, scope, out=good_out)
        with pytest.raises(sw.StringWarsError) as info:
            scope.synchronize()
        assert info.value.status == "invalid_argument" and "not scored" in str(info.value), (pipelined, calls_after)
        assert (good_out.cpu().numpy().astype(np.uint32) == want_good).all()
        for _ in range(3):                                       # nothing left over: clean calls synchronise cleanly afterwards
            engine.pairs(good_a, good_b, scope, out=good_out)
        scope.synchronize()
    scope.set_pipelined(False)
    scope.set_async(False)
    # a SYNCHRONOUS call on the damaged tapes is redone on the planned path on the spot (and must leave nothing behind for a later synchronize)
    got = engine.pairs(pa, pb, scope, out=out).cpu().numpy().astype(np.uint32)
    scope.synchronize()
    assert (got[61:] == orc.levenshtein_pairs(a, b, algo="hyyro")[61:]).all()


def test_multi_device_scope_on_one_gpu(sw, orc):
    """`swh_scope_init_gpus` + `swh_sharded_prepare_*` + `swh_levenshtein_pairs_sharded` with the members sharing device
    0 (N scopes on one GPU: the whole sharding / per-member scoring / gather-in-place path except RCCL itself, which
    needs distinct GPUs), and a one-member scope (communicator-free by construction)."""
    a, b = sw.generate_pairs("tokens64", 60_000, seed=21)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    for devices in ([0], [0, 0], [0, 0, 0, 0, 0]):
        scope = sw.DeviceScope(gpu_devices=devices)
        assert scope.device_count == len(devices) and scope.compute_units == 256
        engine = sw.LevenshteinDistances(capabilities=scope)
        batch = sw.ShardedPairs(scope, a, b)
        cuts = batch.cuts
        assert cuts == sw.shard_cuts(a, b, len(devices))
        assert (engine.pairs_sharded(batch, scope) == want).all()
        assert (engine.pairs_sharded(batch, scope, bound=5) == np.minimum(want, 6)).all()
        timing = scope.shard_timing()
        assert timing["pairs"] == 60_000 and timing["cells"] == int((a.lengths * b.lengths).sum()) and timing["compute_ms"] > 0
        assert (engine.pairs(a, b, scope) == want).all()                      # ordinary calls run on the first device
        import torch
        out = torch.zeros(60_000, dtype=torch.int32, device="cuda")           # first-device memory as the destination
        engine.pairs_sharded(batch, scope, out=out)
        assert (out.cpu().numpy().astype(np.uint32) == want).all()
        batch.free()
    ua, ub = sw.generate_pairs("utf8_lines", 900, seed=21)
    scope = sw.DeviceScope(gpu_devices=[0, 0, 0])
    chars = sw.LevenshteinDistancesUTF8(capabilities=scope)
    batch = sw.ShardedPairs(scope, ua, ub, utf8=True)
    assert (chars.pairs_sharded(batch, scope, bound=32) == orc.levenshtein_pairs(ua, ub, utf8=True, bound=32)).all()
    ragged = sw.Strs([b"x" * 3000, b"", b"ab", b"c"] + [b"w"] * 50)       # one pair holds nearly all the cells
    other = sw.Strs([b"y" * 2900, b"q", b"", b"c"] + [b"w"] * 50)
    batch = sw.ShardedPairs(scope, ragged, other)
    assert (sw.LevenshteinDistances(capabilities=scope).pairs_sharded(batch, scope) == orc.levenshtein_pairs(ragged, other)).all()
    # alignment engines on a multi-device scope: ordinary calls run on its first device, sharded calls on per-member clones
    nw = sw.NeedlemanWunschScores(*sw.unary_class_costs(2, -1), open=-2, extend=-2, capabilities=scope)
    assert nw.pairs(sw.Strs([b"GATTACA"]), sw.Strs([b"GCATGCU"]), scope).tolist() == [2]
    pa, pb = sw.generate_pairs("protein4k", 12, seed=8)
    ta, tb = sw.generate_pairs("tokens64", 20_000, seed=8)
    matrix = sw.substitution_matrix(8)
    for tapes, count in (((pa, pb), 12), ((ta, tb), 500)):
        batch = sw.ShardedPairs(scope, *tapes)
        for gaps in ((-4, -4), (-11, -1)):
            glob = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
            local = sw.SmithWatermanScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
            got = glob.pairs_sharded(batch, scope)
            assert got.dtype == np.int32 and (got[:count] == orc.nw_pairs(*tapes, matrix, *gaps, count=count)).all()
            assert (got == glob.pairs(*tapes, scope)).all()
            assert (local.pairs_sharded(batch, scope) == local.pairs(*tapes, scope)).all()
        with pytest.raises(sw.StringWarsError):
            sw.LevenshteinDistances(capabilities=scope)._handle and nw.__class__.pairs_sharded(sw.SmithWatermanScores(*sw.unary_class_costs(2, -1), capabilities=scope), batch, sw.DeviceScope(gpu_device=0))
        batch.free()
    with pytest.raises(sw.StringWarsError) as info:                          # ... but a single-device scope does not shard
        sw.ShardedPairs(sw.DeviceScope(gpu_device=0), a, b)
    assert info.value.status == "invalid_argument"


def test_sharded_cross_product_on_one_gpu(sw, orc):
    """`swh_sharded_cross_prepare_u64tape` + `swh_levenshtein_cross_sharded` / `swh_nw_cross_sharded` / `swh_sw_cross_sharded`:
    the reference's own call shape (`compute_into(queries, candidates, &mut matrix)`, bench.rs:478-486) over the members of a
    multi-device scope -- here sharing device 0 --, every member filling its rows of a host matrix, a strided host matrix and a
    matrix in device memory; bytes and code points; ragged query lengths (row blocks balance symbols, not counts)."""
    import torch
    rng = np.random.default_rng(77)
    words = [bytes(rng.integers(97, 123, int(n), dtype=np.uint8)) for n in rng.integers(0, 40, 301)] + [b"x" * 900, b"", b"yy"]
    cands = [bytes(rng.integers(97, 123, int(n), dtype=np.uint8)) for n in rng.integers(0, 40, 157)]
    q, c = sw.Strs(words), sw.Strs(cands)
    single = sw.DeviceScope(gpu_device=0)
    want = sw.LevenshteinDistances(capabilities=single)(q, c, single)
    rows = sw.Strs([q[i] for i in (0, 7, 301, 302, 303)])
    assert (want[[0, 7, 301, 302, 303]] == np.array([[orc.levenshtein(rows[i], c[j]) for j in range(len(cands))] for i in range(5)])).all()
    for devices in ([0], [0, 0, 0]):
        scope = sw.DeviceScope(gpu_devices=devices)
        engine = sw.LevenshteinDistances(capabilities=scope)
        product = sw.ShardedCross(scope, q, c)
        assert (engine.cross_sharded(product, scope) == want).all()
        wide = np.full((len(words), len(cands) + 3), 7, dtype=np.uint64)
        engine.cross_sharded(product, scope, out=wide[:, :len(cands)])
        assert (wide[:, :len(cands)] == want).all() and (wide[:, len(cands):] == 7).all()
        on_device = torch.zeros((len(words), len(cands)), dtype=torch.int64, device="cuda")
        engine.cross_sharded(product, scope, out=on_device)
        assert (on_device.cpu().numpy().astype(np.uint64) == want).all()
        assert scope.shard_timing()["pairs"] == len(words) * len(cands)
        classes, costs = sw.unary_class_costs(2, -1)
        for cls in (sw.NeedlemanWunschScores, sw.SmithWatermanScores):
            aligner = cls(classes, costs, open=-5, extend=-1, capabilities=scope)
            assert (aligner.cross_sharded(product, scope) == cls(classes, costs, open=-5, extend=-1, capabilities=single)(q, c, single)).all()
        product.free()
        uq = sw.Strs(["na\u00efve", "\u0416\u4e2d", "", "abc\U0001F600"] * 9)
        uc = sw.Strs(["naive", "\u4e2d\u0416", "\U0001F600"] * 5)
        chars = sw.LevenshteinDistancesUTF8(capabilities=scope)
        uproduct = sw.ShardedCross(scope, uq, uc, utf8=True)
        assert (chars.cross_sharded(uproduct, scope) == sw.LevenshteinDistancesUTF8(capabilities=single)(uq, uc, single)).all()
        with pytest.raises(ValueError):
            engine.cross_sharded(uproduct, scope)
    with pytest.raises(sw.StringWarsError):
        sw.ShardedCross(single, q, c)                                       # a single-device scope does not shard


def test_sharded_call_pieces_self_check_and_error_paths(sw, orc, monkeypatch, request):
    """`swh_levenshtein_pairs_sharded` beyond the happy path: shards large enough to be scored in four pieces (the send of a piece
    enqueued behind its kernel), the gather's self-check (per-shard checksums computed on the shard's device and again over the
    gathered vector: first call of a scope, every call with STRINGWARS_AMD_SHARD_CHECK=1) catching a damaged distance, HIP and
    RCCL failures surfacing as statuses -- an unknown device, RCCL unavailable, a communicator RCCL refuses -- and RCCL's own
    entry points (init of a one-rank communicator, an empty group, destroy) on this box's single GPU."""
    # (STRINGWARS_AMD_SHARD_FAULT / STRINGWARS_AMD_RCCL are test hooks: only the test build of the library reads them)
    if not run_in_child(request, test_library=True):
        return
    a, b = sw.generate_pairs("short_words", 1_200_000, seed=33)
    scope = sw.DeviceScope(gpu_devices=[0, 0, 0])
    engine = sw.LevenshteinDistances(capabilities=scope)
    single = sw.DeviceScope(gpu_device=0)
    want = sw.LevenshteinDistances(capabilities=single).pairs(a, b, single)
    assert (want[:50_000] == orc.levenshtein_pairs(a, b, algo="hyyro", count=50_000)).all()
    batch = sw.ShardedPairs(scope, a, b)
    cuts = batch.cuts
    assert min(hi - lo for lo, hi in zip(cuts, cuts[1:])) >= 4 * 65536          # every shard runs as four pieces
    for _ in range(2):                                                           # first call: self-checked; second: not
        assert (engine.pairs_sharded(batch, scope) == want).all()
    monkeypatch.setenv("STRINGWARS_AMD_SHARD_FAULT", "1")
    assert (engine.pairs_sharded(batch, scope) != want).sum() == 1              # unchecked: the damaged word gets through
    fresh = sw.DeviceScope(gpu_devices=[0, 0])                                   # a scope's first call checks itself
    fresh_batch = sw.ShardedPairs(fresh, a, b)
    with pytest.raises(sw.StringWarsError) as info:
        sw.LevenshteinDistances(capabilities=fresh).pairs_sharded(fresh_batch, fresh)
    assert info.value.status == "device_error" and "checksum mismatch" in str(info.value)
    monkeypatch.delenv("STRINGWARS_AMD_SHARD_FAULT")
    assert (sw.LevenshteinDistances(capabilities=fresh).pairs_sharded(fresh_batch, fresh) == want).all()
    # failures come back as statuses, and nothing is left behind
    with pytest.raises(sw.StringWarsError) as info:
        sw.DeviceScope(gpu_devices=[0, 977])
    assert info.value.status in ("no_device", "device_error", "invalid_argument")
    monkeypatch.setenv("STRINGWARS_AMD_RCCL", "off")
    with pytest.raises(sw.StringWarsError) as info:
        sw.DeviceScope(gpu_devices=[0, 0])
    assert info.value.status == "rccl_error" and "could not be loaded" in str(info.value)
    monkeypatch.setenv("STRINGWARS_AMD_RCCL", "force")
    with pytest.raises(sw.StringWarsError) as info:                             # one GPU twice in a communicator: RCCL says no
        sw.DeviceScope(gpu_devices=[0, 0])
    assert info.value.status == "rccl_error" and "ncclCommInitAll failed" in str(info.value)
    alone = sw.DeviceScope(gpu_devices=[0])                                      # a one-rank communicator: RCCL's init / group / destroy run
    monkeypatch.delenv("STRINGWARS_AMD_RCCL")
    one = sw.ShardedPairs(alone, a.subview(0, 300_000), b.subview(0, 300_000))
    assert (sw.LevenshteinDistances(capabilities=alone).pairs_sharded(one, alone) == want[:300_000]).all()
    del one, alone


BAD_UTF8 = [b"\xff", b"\xc0\x80", b"\xc1\xbf", b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf",
            b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"\xf8\x88\x80\x80\x80", b"\xe2\x82", b"\xf0\x9f\x98",
            b"\xc3", b"\x80", b"\xbf\xbf", b"\xc3\x28", b"\xe2\x28\xa1", b"\xe2\x82\x28", b"\xf0\x28\x8c\xbc", b"\xf0\x90\x28\xbc",
            b"\xf0\x90\x8c\x28"]
GOOD_UTF8 = ["", "a", "\u007f", "\u0080", "\u07ff", "\u0800", "\ud7ff", "\ue000", "\uffff", "\U00010000", "\U0010ffff",
             "na\u00efve \u0416\u4e2d\U0001f600"]


def test_utf8_validation_matches_the_oracle(sw, orc, scope):
    """Strict UTF-8 (RFC 3629): every malformed sequence is rejected wherever it sits -- alone, inside a longer tape,
    at the very end of the tape, across the decoder's 256 B / 1 KB / 8 KB tile edges, or split over two strings --
    and the boundary code points of every sequence length decode to the oracle's code points."""
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    good = [g.encode("utf-8") for g in GOOD_UTF8]
    want = orc.levenshtein_pairs(sw.Strs(good), sw.Strs(list(reversed(good))), utf8=True)
    assert engine.pairs(good, list(reversed(good)), scope).tolist() == want.tolist()
    filler = "x\u00e9\u4e2d\U0001f600".encode("utf-8")   # 1 + 2 + 3 + 4 = 10 bytes
    for bad in BAD_UTF8:
        with pytest.raises(ValueError):
            orc.utf8_decode(bad)
        for lead in (0, 250, 255, 256, 1020, 1023, 1024, 8185, 8190, 8192, 8193):
            pad = (filler * (lead // len(filler) + 1))[: lead - lead % 1] if lead else b""
            pad = pad[: len(pad) - 0]
            # cut the filler on a sequence boundary so that only `bad` is malformed
            while pad and (pad[-1] & 0xC0) == 0x80 or pad and pad[-1] >= 0xC0:
                pad = pad[:-1]
            pad = pad + b"y" * (lead - len(pad))
            for tail in (b"", b"zz", filler * 30):
                for side in (0, 1):
                    strings = [b"ok", pad + bad + tail, b"fine"]
                    other = [b"ok", b"ok", b"ok"]
                    with pytest.raises(sw.StringWarsError) as info:
                        engine.pairs(*((strings, other) if side == 0 else (other, strings)), scope)
                    assert info.value.status == "invalid_utf8", (bad, lead, tail[:4], side)
    # a valid sequence cut in two by a string boundary is invalid in both strings
    euro = "\u20ac".encode("utf-8")
    with pytest.raises(sw.StringWarsError):
        engine.pairs([euro[:1], euro[1:]], [b"a", b"b"], scope)
    with pytest.raises(sw.StringWarsError):
        engine.pairs([b"a" * 1023 + euro[:2], euro[2:] + b"b"], [b"a", b"b"], scope)
    # ... while the same bytes inside one string, straddling every tile edge, are fine
    for lead in (254, 255, 1022, 1023, 8190, 8191):
        s = b"a" * lead + euro + b"b"
        assert engine.pairs([s], [b"a" * lead + b"b"], scope).tolist() == [1]


def test_utf8_validation_fuzz(sw, orc, scope):
    """Random byte strings over the bytes that matter to a UTF-8 decoder (the edges of every lead / continuation range),
    planted in valid text at random offsets around the decoder's tile edges: the library rejects exactly the tapes the oracle's
    decoder rejects, and scores the others like the oracle."""
    rng = np.random.default_rng(99)
    edge_bytes = np.array([0x00, 0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xEC, 0xED, 0xEE, 0xEF,
                           0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xF8, 0xFF], dtype=np.uint8)
    edge_points = [0x00, 0x41, 0x7F, 0x80, 0x7FF, 0x800, 0xFFF, 0x1000, 0xCFFF, 0xD000, 0xD7FF, 0xE000, 0xFFFD, 0xFFFF, 0x10000, 0x3FFFF,
                   0x40000, 0xFFFFF, 0x100000, 0x10FFFF]
    filler = "x\u00e9\u4e2d\U0001f600y".encode("utf-8")
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    rejected = accepted = 0
    for _ in range(1200):
        # boundary code points of every sequence length, then (mostly) one byte replaced, inserted or dropped
        junk = bytearray("".join(chr(edge_points[i]) for i in rng.integers(0, len(edge_points), int(rng.integers(1, 4)))).encode("utf-8"))
        if rng.random() < 0.6:
            at, kind = int(rng.integers(0, len(junk))), int(rng.integers(0, 3))
            if kind == 0: junk[at] = int(edge_bytes[rng.integers(0, edge_bytes.size)])
            elif kind == 1: junk.insert(at, int(edge_bytes[rng.integers(0, edge_bytes.size)]))
            else: del junk[at]
        junk = bytes(junk) or b"\x80"
        lead = int(rng.choice([0, 3, 250, 1017, 8180, 8189, 16379])) + int(rng.integers(0, 8))
        head = (filler * (lead // len(filler) + 1))[:lead]
        while head and ((head[-1] & 0xC0) == 0x80 or head[-1] >= 0xC0):   # cut the filler on a sequence boundary
            head = head[:-1]
        text = head + junk + filler * int(rng.integers(0, 3))
        try:
            orc.utf8_decode(text)
            valid = True
        except ValueError:
            valid = False
        other = "reference \u00e9".encode("utf-8")
        if valid:
            accepted += 1
            want = orc.levenshtein_pairs(sw.Strs([text, other]), sw.Strs([other, text]), utf8=True)
            assert engine.pairs([text, other], [other, text], scope).tolist() == want.tolist(), junk
        else:
            rejected += 1
            with pytest.raises(sw.StringWarsError) as info:
                engine.pairs([other, text], [other, other], scope)
            assert info.value.status == "invalid_utf8", junk
    assert rejected > 200 and accepted > 200, (rejected, accepted)


def test_utf8_look_back_epoch_wraps(orc):
    """The one-pass staging tags its look-back words with a 16-bit call epoch instead of clearing them; every 65 535 calls the
    words ARE cleared and the epoch starts over. STRINGWARS_AMD_UTF8_EPOCH (a test hook, read once per process) starts a fresh
    buffer near the end: calls across the wrap, small tapes (one launch pair for both) and large ones (a stream per tape),
    growing and shrinking so that words of older epochs sit where the newer calls look."""
    import subprocess
    import sys
    code = (
        "import numpy as np, stringwars_amd as sw, oracle\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "engine = sw.LevenshteinDistancesUTF8(capabilities=scope)\n"
        "big = sw.generate_pairs('utf8_lines', 9000, seed=5)       # ~9 MB a tape: a stream per tape\n"
        "small = sw.generate_pairs('utf8_lines', 300, seed=6)\n"
        "tiny = (sw.Strs(['h\u00e9llo', '\u4e2d\u6587', 'abc']), sw.Strs(['hello', '\u4e2d', 'abd']))\n"
        "want = {id(t): oracle.levenshtein_pairs(t[0], t[1], utf8=True, bound=40) for t in (big, small, tiny)}\n"
        "for round in range(12):\n"
        "    for t in (small, tiny, big, tiny, small):\n"
        "        assert (engine.pairs(t[0], t[1], scope, bound=40) == want[id(t)]).all(), round\n"
        "print('epochs ok')\n")
    # (tapes of more than 4 MB together take the stream-per-tape path here; the default is 48 MB)
    env = dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_UTF8_EPOCH="65520", STRINGWARS_AMD_UTF8_MERGED_MB="4", STRINGWARS_AMD_UTF8_STAGING="tiles",
               PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0 and "epochs ok" in done.stdout, done.stderr[-2000:]


def test_doubling_schedule_for_bounds_beyond_one_band_word(orc):
    """A unit-cost call with no bound, or one beyond the 64-bit band window, runs in two stages when its strings are long enough: the
    one-word band at k = 63 over everything the band pays for, then only the pairs that came back 64 under the call's own bound
    (api.hip: doubling; rapidfuzz's own score-hint schedule). Similar lines (C3's: half within 32 edits), unrelated lines, words and
    empty strings in one batch; no bound and bounds on both sides of every kernel switch; code points (raw and prepared) and bytes;
    32- and 64-bit results; a cross-product. STRINGWARS_AMD_STAMPS shows the two stages (two plans, the band first), and a scope
    whose first stage settles too little stops trying for a while."""
    import subprocess
    import sys
    code = (
        "import numpy as np, stringwars_amd as sw, oracle\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "scope.set_profiling(True)\n"
        "rng = np.random.default_rng(8)\n"
        "a, b = sw.generate_pairs('utf8_lines', 13000, seed=5)\n"
        "la, lb = [bytes(x) for x in a], [bytes(x) for x in b]\n"
        "for i in range(0, 13000, 9): lb[i] = la[(i + 4001) % 13000]            # unrelated lines: the second stage's work\n"
        "for i in range(0, 13000, 31): la[i] = la[i][:40].decode('utf-8', 'ignore').encode('utf-8')   # words among the lines\n"
        "la[7] = b''; lb[11] = b''; la[12] = b''; lb[12] = b''\n"
        "a, b = sw.Strs(la), sw.Strs(lb)\n"
        "full = oracle.levenshtein_pairs(a, b, utf8=True)\n"
        "assert (full > 200).sum() > 1000 and (full <= 63).sum() > 5000\n"
        "engine = sw.LevenshteinDistancesUTF8(capabilities=scope)\n"
        "pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)\n"
        "da, db = a.to_device(scope), b.to_device(scope)\n"
        "for bound in (None, 64, 65, 100, 127, 128, 255, 5000, 63, 32):\n"
        "    want = full if bound is None else np.minimum(full, bound + 1)\n"
        "    assert (engine.pairs(pa, pb, scope, bound=bound) == want).all(), ('prepared', bound)\n"
        "    assert (engine.pairs(da, db, scope, bound=bound) == want).all(), ('raw', bound)\n"
        "out64 = np.zeros(13000, dtype=np.uint64)\n"
        "engine.pairs(pa, pb, scope, out=out64)\n"
        "assert (out64 == full).all()\n"
        "print('STAGES-BEGIN', flush=True)\n"
        "import sys; sys.stderr.write('STAGES-BEGIN\\n'); sys.stderr.flush()\n"
        "assert (engine.pairs(pa, pb, scope) == full).all()\n"
        "sys.stderr.write('STAGES-END\\n'); sys.stderr.flush()\n"
        "# the same tapes as bytes\n"
        "bytes_engine = sw.LevenshteinDistances(capabilities=scope)\n"
        "full_bytes = oracle.levenshtein_pairs(a, b, algo='hyyro')\n"
        "ba, bb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)\n"
        "for bound in (None, 64, 127, 300):\n"
        "    want = full_bytes if bound is None else np.minimum(full_bytes, bound + 1)\n"
        "    assert (bytes_engine.pairs(ba, bb, scope, bound=bound) == want).all(), ('bytes', bound)\n"
        "# a cross-product large enough for two stages: 120 x 120 lines\n"
        "q, c = sw.Strs(la[100:220]), sw.Strs(lb[100:220])\n"
        "wide = oracle.levenshtein_pairs(sw.Strs([x for x in la[100:220] for _ in range(120)]), sw.Strs(lb[100:220] * 120), utf8=True)\n"
        "assert (engine(sw.PreparedTape(scope, q, utf8=True), sw.PreparedTape(scope, c, utf8=True), scope).reshape(-1) == wide).all()\n"
        "# unrelated lines only: the first stage settles nothing, the scope sits the next calls out\n"
        "ua, ub = sw.Strs(la[:12000]), sw.Strs([la[(i + 4001) % 13000] for i in range(12000)])\n"
        "fresh = sw.DeviceScope(gpu_device=0); fresh.set_profiling(True)\n"
        "fa, fb = sw.PreparedTape(fresh, ua, utf8=True), sw.PreparedTape(fresh, ub, utf8=True)\n"
        "other = sw.LevenshteinDistancesUTF8(capabilities=fresh)\n"
        "want = oracle.levenshtein_pairs(ua, ub, utf8=True)\n"
        "sys.stderr.write('UNRELATED-BEGIN\\n'); sys.stderr.flush()\n"
        "for _ in range(3): assert (other.pairs(fa, fb, fresh) == want).all()\n"
        "sys.stderr.write('UNRELATED-END\\n'); sys.stderr.flush()\n"
        "print('doubling ok')\n")
    env = dict(os.environ, STRINGWARS_AMD_STAMPS="1", PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert done.returncode == 0 and "doubling ok" in done.stdout, done.stderr[-3000:]
    stages = done.stderr.split("STAGES-BEGIN")[1].split("STAGES-END")[0]
    assert stages.count("stamp plan_") == 2 and stages.count("stamp banded") == 1 and "bitparallel_u32" in stages, stages
    unrelated = done.stderr.split("UNRELATED-BEGIN")[1].split("UNRELATED-END")[0]
    assert unrelated.count("stamp banded") == 1 and unrelated.count("stamp plan_") == 4, unrelated    # tried once, then sat out
    # the comparison knob: one stage
    env.update(TEST_LIBRARY_ENV, STRINGWARS_AMD_DOUBLING="0")      # (A / B switches are test hooks since round 6: the test library reads them)
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert done.returncode != 0 or "doubling ok" in done.stdout
    stages = done.stderr.split("STAGES-BEGIN")[1].split("STAGES-END")[0]
    assert stages.count("stamp plan_") == 1 and "stamp banded" not in stages, stages


def test_utf8_lines_are_staged_string_by_string(orc):
    """Raw UTF-8 tapes of lines (a mean string of >= 192 bytes) are staged by k_utf8_strings -- a wave per string, the code points where
    the string's bytes were, (first, end) extents instead of abutting offsets (TapeRef::gap) -- and every kernel behind it reads those
    extents: the banded kernel (bounds of one and two words), the planned bit-parallel kernels, the tiled kernel, general costs on
    the wavefront kernels, a cross-product, pairwise and with one tape against itself; u32 and u64 offsets, host and device tapes, strings
    of every length around the 1 KB rounds (1023, 1024, 1025 ... bytes, ending in sequences of every length) and empty ones among them.
    STRINGWARS_AMD_STAMPS shows which staging kernel ran (read once per process, hence the subprocess)."""
    import subprocess
    import sys
    code = (
        "import numpy as np, stringwars_amd as sw, oracle\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "scope.set_profiling(True)\n"
        "engine = sw.LevenshteinDistancesUTF8(capabilities=scope)\n"
        "a, b = sw.generate_pairs('utf8_lines', 3000, seed=5)\n"
        "for bound in (None, 0, 7, 32, 63, 64, 100, 127, 200):\n"
        "    want = oracle.levenshtein_pairs(a, b, utf8=True, bound=bound)\n"
        "    for ta, tb in ((a, b), (a.with_offsets(np.uint64).to_device(scope), b.with_offsets(np.uint64).to_device(scope))):\n"
        "        assert (engine.pairs(ta, tb, scope, bound=bound) == want).all(), bound\n"
        "assert (engine.pairs(a, a, scope) == 0).all()\n"
        "tiled = sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm='tiled')\n"
        "sa, sb = sw.generate_pairs('utf8_lines', 400, seed=9)\n"
        "short_a = sw.Strs([bytes(x)[:200].decode('utf-8', 'ignore') for x in sa]); short_b = sw.Strs([bytes(x)[:230].decode('utf-8', 'ignore') for x in sb])\n"
        "assert (tiled.pairs(short_a, short_b, scope) == oracle.levenshtein_pairs(short_a, short_b, utf8=True)).all()\n"
        "q, c = [bytes(x) for x in list(sa)[:24]], [bytes(x) for x in list(sb)[:40]]\n"
        "wide = oracle.levenshtein_pairs(sw.Strs([x for x in q for _ in c]), sw.Strs(c * len(q)), utf8=True)\n"
        "assert (engine(sw.Strs(q), sw.Strs(c), scope).reshape(-1) == wide).all()\n"
        "self_wide = oracle.levenshtein_pairs(sw.Strs([x for x in q for _ in q]), sw.Strs(q * len(q)), utf8=True)\n"
        "assert (engine(sw.Strs(q), None, scope).reshape(-1) == self_wide).all()\n"
        "# strings around the rounds of 1024 bytes, ending in sequences of 1 .. 4 bytes, empty strings among them\n"
        "tails = ['x', '\u00e9', '\u4e2d', '\U0001f600']\n"
        "edge = []\n"
        "for n in (0, 1, 15, 16, 17, 1019, 1020, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 1028, 2047, 2048, 2049, 3071, 5000):\n"
        "    for t in tails:\n"
        "        body = ('ab\u00e9\u4e2d\U0001f600' * (n // 10 + 1)).encode('utf-8')[:max(n - len(t.encode('utf-8')), 0)]\n"
        "        while body and ((body[-1] & 0xC0) == 0x80 or body[-1] >= 0xC0): body = body[:-1]\n"
        "        body = body + b'y' * max(n - len(t.encode('utf-8')) - len(body), 0)\n"
        "        edge.append(body + (t.encode('utf-8') if n else b''))\n"
        "ea = sw.Strs(edge); eb = sw.Strs([e.decode('utf-8')[1:].encode('utf-8') + b'zz' for e in edge])\n"
        "for bound in (None, 32):\n"
        "    assert (engine.pairs(ea, eb, scope, bound=bound) == oracle.levenshtein_pairs(ea, eb, utf8=True, bound=bound)).all()\n"
        "print('strings ok')\n")
    env = dict(os.environ, STRINGWARS_AMD_STAMPS="1", PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0 and "strings ok" in done.stdout, done.stderr[-3000:]
    assert "utf8_strings" in done.stderr and done.stderr.count("utf8_strings") >= 20, done.stderr[-2000:]
    # the comparison knob keeps the flat kernel
    env.update(TEST_LIBRARY_ENV, STRINGWARS_AMD_UTF8_STAGING="tiles")
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0 and "strings ok" in done.stdout, done.stderr[-3000:]
    assert "utf8_strings" not in done.stderr and "utf8_tile_decode" in done.stderr


def test_utf8_string_by_string_staging_validates(orc):
    """The UTF-8 validation and fuzz tests, the random scripts, the tapes at their edges and the size beliefs once more with EVERY raw
    UTF-8 call on the planned / tiled routes staged string by string (STRINGWARS_AMD_UTF8_STAGING=strings: words, empty strings, strings cut inside a
    sequence, malformed bytes at every offset): the per-string validation rejects exactly what the flat one rejects."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_UTF8_STAGING="strings", PYTHONPATH=child_pythonpath())
    picks = ("test_utf8_validation_matches_the_oracle or test_utf8_validation_fuzz or test_utf8_random_scripts or test_small_tapes_and_strings_at_tape_edges "
             "or test_believed_tape_sizes_are_checked_on_the_device or test_kat_levenshtein or test_golden_multilingual_words or test_config3_bounded_utf8 "
             "or test_general_cost_levenshtein_over_code_points or test_patterns_longer_than_64_blocks or test_banded_window_kernel")
    done = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", picks],
                          env=env, capture_output=True, text=True, timeout=1800)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    assert " passed" in done.stdout and "failed" not in done.stdout, done.stdout[-1000:]


def test_utf8_string_too_long_for_a_wave_of_its_own(sw, orc, scope):
    """A string beyond 64 KB would keep one wave of k_utf8_strings busy for milliseconds: the kernel says so, the call is redone with the
    flat staging (and the scope's next calls are too), the distances are the oracle's either way."""
    rng = np.random.default_rng(5)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    base = "caf\u00e9 \u4e2d\u6587 \U0001f600 "
    long_one = (base * 9000).encode("utf-8")             # ~170 KB
    a = sw.Strs([long_one, (base * 40).encode("utf-8"), b"x" * 700])
    b = sw.Strs([(base * 8998).encode("utf-8") + b"tail", (base * 39).encode("utf-8"), b"x" * 650 + b"y" * 10])
    want = orc.levenshtein_pairs(a, b, utf8=True, bound=60)
    for _ in range(3):
        assert engine.pairs(a, b, scope, bound=60).tolist() == want.tolist()
    la, lb = sw.generate_pairs("utf8_lines", 300, seed=2)
    assert (engine.pairs(la, lb, scope, bound=32) == orc.levenshtein_pairs(la, lb, utf8=True, bound=32)).all()


@pytest.mark.parametrize("mode", ["split", "scan"])
def test_utf8_three_kernel_scan_path(orc, mode):
    """The decoder before the one-pass kernel (count, scan, write): STRINGWARS_AMD_UTF8_SCAN=scan runs it with the one-launch
    scan, =split with the three-kernel scan that tapes beyond 0.5 GB took (read once per process, hence the subprocess).
    The default -- one kernel per tape, tile prefixes by decoupled look-back over up to hundreds of tiles -- is what every
    other UTF-8 test runs."""
    import subprocess
    import sys
    code = (
        "import numpy as np, stringwars_amd as sw, oracle\n"
        "a, b = sw.generate_pairs('utf8_lines', 3000, seed=5)\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "got = sw.LevenshteinDistancesUTF8(capabilities=scope).pairs(a, b, scope, bound=40)\n"
        "want = oracle.levenshtein_pairs(a, b, utf8=True, bound=40)\n"
        "assert (got == want).all()\n"
        "print('split-scan ok')\n")
    env = dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_UTF8_SCAN=mode, STRINGWARS_AMD_UTF8_STAGING="tiles", PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0 and "split-scan ok" in done.stdout, done.stderr[-2000:]


@pytest.mark.parametrize("waves", ["4", "8", "16"])
def test_tiled_kernel_workgroup_shapes(orc, waves):
    """k_bitparallel_tiled runs byte strings as sixteen-wave workgroups with tiles of up to 4096 pairs when a batch fills the device, as
    eight-wave workgroups with tiles of up to 2048 otherwise (STRINGWARS_AMD_TILED_WAVES forces one: read once per process, hence the
    subprocess). Every shape on the same batches: tokens with shared affixes in full tiles (9000 pairs: tiles of ~4096 / 2048 and a
    short last one), strings over every block count up to 64 blocks, empty strings, a bound."""
    import subprocess
    import sys
    code = (
        "import numpy as np, stringwars_amd as sw, oracle\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "engine = sw.LevenshteinDistances(capabilities=scope, algorithm='tiled')\n"
        "a, b = sw.generate_pairs('tokens64', 9000, seed=11)\n"
        "assert (engine.pairs(a, b, scope) == oracle.levenshtein_pairs(a, b, algo='hyyro')).all()\n"
        "assert (engine.pairs(a, b, scope, bound=5) == oracle.levenshtein_pairs(a, b, algo='hyyro', bound=5)).all()\n"
        "rng = np.random.default_rng(3)\n"
        "xs = [bytes(rng.integers(97, 101, int(n), dtype=np.uint8)) for n in list(range(0, 2049, 37)) * 3]\n"
        "ys = [bytes(rng.integers(97, 101, int(n), dtype=np.uint8)) for n in rng.integers(0, 2049, len(xs))]\n"
        "sa, sb = sw.Strs(xs), sw.Strs(ys)\n"
        "scope.set_profiling(True)\n"
        "got = engine.pairs(sa, sb, scope)\n"
        "name = scope.last_timing()['dominant_name']\n"
        "assert name == 'bitparallel_tiled', name\n"
        "assert (got == oracle.levenshtein_pairs(sa, sb, algo='hyyro')).all()\n"
        "print('tiled shapes ok')\n")
    env = dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_TILED_WAVES=waves, PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0 and "tiled shapes ok" in done.stdout, (done.stdout[-500:], done.stderr[-2000:])


def test_edge_cases_and_errors(sw, orc, scope):
    engine = sw.LevenshteinDistances(capabilities=scope)
    assert engine.pairs([], [], scope).size == 0
    assert engine.pairs([b""], [b""], scope).tolist() == [0]
    assert engine.pairs([b"", b"abc", b""], [b"xy", b"", b""], scope).tolist() == [2, 3, 0]
    assert engine.pairs([b"a"], [b"b"], scope).tolist() == [1]            # tapes shorter than one dword
    assert engine.pairs([b"ab"], [b"b"], scope, bound=0).tolist() == [1]
    with pytest.raises(ValueError):
        engine.pairs([b"a"], [b"a", b"b"], scope)
    utf8 = sw.LevenshteinDistancesUTF8(capabilities=scope)
    with pytest.raises(sw.StringWarsError) as info:
        utf8.pairs([b"ok", b"\xff\xfe"], [b"ok", b"ok"], scope)
    assert info.value.status == "invalid_utf8"
    with pytest.raises(sw.StringWarsError):
        sw.LevenshteinDistances(0, -1, 1, 1, capabilities=scope)
    with pytest.raises(sw.StringWarsError):
        sw.NeedlemanWunschScores(substitution_matrix=np.zeros((256, 256), np.int8), open=3, extend=1, capabilities=scope)
    nw = sw.NeedlemanWunschScores(*sw.unary_class_costs(2, -1), open=-2, extend=-2, capabilities=scope)
    assert nw.pairs([b"", b"ACGT"], [b"", b""], scope).tolist() == [0, -8]
    assert scope.compute_units == 256


# ----------------------------------------------------------------------------------------------------
# full-size runs: size-independent properties + sampled oracle
# ----------------------------------------------------------------------------------------------------
def test_config2_full_size_properties(sw, orc, scope):
    """BASELINE config C2 at full size: 1M ASCII token pairs, unbounded."""
    a, b = sw.generate_pairs("tokens64", 1_000_000, seed=42)
    da, db = a.to_device(scope), b.to_device(scope)
    engine = sw.LevenshteinDistances(capabilities=scope)
    scope.set_profiling(True)
    forward = engine.pairs(da, db, scope)
    timing = scope.last_timing()
    scope.set_profiling(False)
    assert timing["cells"] == int((a.lengths * b.lengths).sum())          # the reference's CUPS numerator
    backward = engine.pairs(db, da, scope)
    assert (forward == backward).all()                                    # symmetry
    assert (engine.pairs(da, da, scope) == 0).all()                       # identity
    la, lb = a.lengths, b.lengths
    assert (forward >= np.abs(la - lb)).all() and (forward <= np.maximum(la, lb)).all()
    sample = np.arange(0, 1_000_000, 50)                                   # 2% against the oracle
    want = orc.levenshtein_pairs(a, b, algo="hyyro", first=0, count=20_000)
    assert (forward[:20_000] == want).all()
    for i in sample[:2000]:
        assert forward[i] == orc.levenshtein(a[int(i)], b[int(i)], "hyyro")
    engine.set_algorithm("wavefront")
    head_a, head_b = a.subview(0, 100_000), b.subview(0, 100_000)
    assert (engine.pairs(head_a, head_b, scope) == forward[:100_000]).all()  # the two algorithms agree
    assert int(forward.astype(np.uint64).sum()) == int(forward[::-1].astype(np.uint64).sum())


def _pick(sw, tape, index):
    """Host sub-tape holding strings `index` of `tape` (for oracle samples spread over a full-size batch)."""
    lengths = tape.lengths[index]
    offsets = np.zeros(len(index) + 1, np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    starts = tape.offsets[index].astype(np.int64)
    data = np.concatenate([tape.data[s:s + n] for s, n in zip(starts, lengths)]) if len(index) else np.zeros(0, np.uint8)
    return sw.Strs(data=data, offsets=offsets)


def _code_point_lengths(tape):
    leads = np.concatenate([[0], np.cumsum((tape.data & 0xC0) != 0x80)])
    return np.diff(leads[tape.offsets.astype(np.int64)])


def test_config3_bounded_utf8(sw, orc, scope):
    """Config C3 (reduced count for the oracle): ~1 KB UTF-8 lines, bound k = 32."""
    a, b = sw.generate_pairs("utf8_lines", 2000, seed=42)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    got = engine.pairs(a, b, scope, bound=32)
    want = orc.levenshtein_pairs(a, b, utf8=True, bound=32, count=300)
    assert (got[:300] == want).all()
    assert got.max() <= 33 and (got == 33).any() and (got < 33).any()
    full = engine.pairs(a, b, scope)
    assert (np.minimum(full, 33) == got).all()


def test_config3_full_size_properties(sw, orc, scope):
    """BASELINE config C3 at full size: 100 K UTF-8 line pairs of ~1 KB, bounded k = 32 -- the production shape of the
    banded kernel (32-pair items below 131 K pairs) and of the two-tape decode. Whole batch: work units, symmetry,
    identity, bounds, bounded == min(unbounded, k + 1); 2 % of the pairs, spread over the batch, against the oracle."""
    pairs, k = 100_000, 32
    a, b = sw.generate_pairs("utf8_lines", pairs, seed=42)
    da, db = a.to_device(scope), b.to_device(scope)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    scope.set_profiling(True)
    bounded = engine.pairs(da, db, scope, bound=k)
    timing = scope.last_timing()
    scope.set_profiling(False)
    cpa, cpb = _code_point_lengths(a), _code_point_lengths(b)
    assert timing["cells"] == int((cpa * cpb).sum())                      # cells counted in code points (bench.rs:434)
    assert (engine.pairs(db, da, scope, bound=k) == bounded).all()        # symmetry
    assert (engine.pairs(da, da, scope, bound=k) == 0).all()              # identity
    assert bounded.max() == k + 1 and (bounded >= np.minimum(np.abs(cpa - cpb), k + 1)).all()
    share = float((bounded == k + 1).mean())
    assert 0.2 < share < 0.8                                              # about half the pairs exceed k
    unbounded = engine.pairs(da, db, scope)
    assert (np.minimum(unbounded, k + 1) == bounded).all()
    assert (unbounded <= np.maximum(cpa, cpb)).all()
    for other in (0, 7, 63):                                              # the other window widths of the banded kernel
        assert (engine.pairs(da, db, scope, bound=other) == np.minimum(unbounded, other + 1)).all()
    sample = np.arange(17, pairs, 50)                                     # 2,000 pairs
    want = orc.levenshtein_pairs(_pick(sw, a, sample), _pick(sw, b, sample), utf8=True)
    assert (unbounded[sample] == want).all()
    assert (bounded[sample] == np.minimum(want, k + 1)).all()
    # the same tapes as bytes (banded<u8> and the byte bit-parallel kernel at G ~ 32 blocks)
    bytes_engine = sw.LevenshteinDistances(capabilities=scope)
    bytes_bounded = bytes_engine.pairs(da, db, scope, bound=k)
    bytes_full = bytes_engine.pairs(da, db, scope)
    assert (np.minimum(bytes_full, k + 1) == bytes_bounded).all()
    head = np.arange(0, pairs, 200)
    assert (bytes_full[head] == orc.levenshtein_pairs(_pick(sw, a, head), _pick(sw, b, head), algo="hyyro")).all()


@pytest.mark.parametrize("gaps", [(-4, -4), (-11, -1)], ids=["linear", "affine"])
def test_config4_full_size_properties(sw, orc, scope, gaps):
    """BASELINE config C4 at full size: 10 K pairs of ~4 KB amino-acid sequences, 256x256 i8 matrix, linear and
    affine gaps -- the production shape of the class-table wavefront kernels on two streams. Whole batch: work units,
    symmetry (the matrix is symmetric), the self-alignment score, bounds, local >= global; 1 % against the oracle."""
    pairs = 10_000
    open_, extend = gaps
    a, b = sw.generate_pairs("protein4k", pairs, seed=42)
    da, db = a.to_device(scope), b.to_device(scope)
    matrix = sw.substitution_matrix(42)
    assert (matrix == matrix.T).all()
    engine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=open_, extend=extend, capabilities=scope)
    scope.set_profiling(True)
    forward = engine.pairs(da, db, scope)
    timing = scope.last_timing()
    scope.set_profiling(False)
    la, lb = a.lengths, b.lengths
    assert timing["cells"] == int((la * lb).sum())
    assert (engine.pairs(db, da, scope) == forward).all()                 # symmetry
    diagonal = matrix[np.arange(256), np.arange(256)].astype(np.int64)
    def self_score(tape):
        prefix = np.concatenate([[0], np.cumsum(diagonal[tape.data])])
        return np.diff(prefix[tape.offsets.astype(np.int64)])
    sa, sb = self_score(a), self_score(b)
    assert (engine.pairs(da, da, scope) == sa).all()                      # identity: every symbol against itself
    # diagonal >= 4 > off-diagonal: an aligned column (x, y) scores at most (m[x][x] + m[y][y]) / 2, gaps are negative
    assert (2 * forward.astype(np.int64) <= sa + sb).all()
    def gap(n):
        return np.where(n > 0, open_ + (n - 1) * extend, 0)
    assert (forward >= gap(la) + gap(lb)).all()                           # all-gaps alignment
    local = sw.SmithWatermanScores(substitution_matrix=matrix, open=open_, extend=extend, capabilities=scope).pairs(da, db, scope)
    assert (local >= np.maximum(forward, 0)).all()
    sample = np.arange(3, pairs, 100)                                     # 100 pairs = 1 %
    sub_a, sub_b = _pick(sw, a, sample), _pick(sw, b, sample)
    assert (forward[sample] == orc.nw_pairs(sub_a, sub_b, matrix, open_, extend)).all()
    few = sample[:12]
    assert local[few].tolist() == [orc.nw_score(a[int(i)], b[int(i)], matrix, open_, extend, local=True) for i in few]


def test_config4_full_byte_alphabet(sw, orc, scope):
    """C4's second run: bytes 0-255, all 256 matrix rows in use -> the LDS-gather matrix kernels."""
    pairs = 2_000
    a, b = sw.generate_pairs("bytes4k", pairs, seed=42)
    da, db = a.to_device(scope), b.to_device(scope)
    matrix = sw.substitution_matrix(42, None)
    sample = np.arange(5, pairs, 100)
    for open_, extend in ((-4, -4), (-11, -1)):
        engine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=open_, extend=extend, capabilities=scope)
        forward = engine.pairs(da, db, scope)
        if (matrix == matrix.T).all():
            assert (engine.pairs(db, da, scope) == forward).all()
        assert (forward[sample] == orc.nw_pairs(_pick(sw, a, sample), _pick(sw, b, sample), matrix, open_, extend)).all()


def test_config5_short_words_large(sw, orc, scope):
    a, b = sw.generate_pairs("short_words", 2_000_000, seed=42)
    got = sw.LevenshteinDistances(capabilities=scope).pairs(a, b, scope)
    want = orc.levenshtein_pairs(a, b, algo="hyyro")
    assert (got == want).all()


def test_config5_one_gpu_slice_at_full_size(sw, orc, scope):
    """Config C5 is 100 M word pairs over 8 GPUs: one GPU's slice, 12.5 M pairs, at full size. Too many for the oracle, so:
    three independent routes (k_short_tiled, the general tiled kernel, the planned path with k_direct_short) must agree on
    every pair, the distance must be symmetric and inside its length bounds, and 1 % of the pairs go to the oracle."""
    count = 12_500_000
    a, b = sw.generate_pairs("short_words", count, seed=42)
    pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
    got = sw.LevenshteinDistances(capabilities=scope).pairs(pa, pb, scope)
    for algorithm in ("tiled", "bitparallel"):
        other = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm).pairs(pa, pb, scope)
        assert (other == got).all(), algorithm
    assert (sw.LevenshteinDistances(capabilities=scope).pairs(pb, pa, scope) == got).all()
    la, lb = a.lengths.astype(np.int64), b.lengths.astype(np.int64)
    assert (got >= np.abs(la - lb)).all() and (got <= np.maximum(la, lb)).all()
    sample = np.arange(0, count, 100)
    assert (got[sample] == orc.levenshtein_pairs(_pick(sw, a, sample), _pick(sw, b, sample), algo="hyyro")).all()


def test_fused_planner_gives_up_instead_of_hanging():
    """The one-launch planner waits at a grid-wide barrier; if its grid cannot be resident as a whole (forced here by
    oversubscribing it) it must give up after its time-out, and the call must be redone with the three-pass planner --
    same results, a couple of seconds late, never a hang."""
    import subprocess
    import sys
    code = (
        "import numpy as np, time, oracle, stringwars_amd as sw\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "a, b = sw.generate_pairs('tokens64', 700000, seed=5)\n"
        "engine = sw.LevenshteinDistances(capabilities=scope, algorithm='bitparallel')\n"
        "t = time.time(); got = engine.pairs(a, b, scope); first = time.time() - t\n"
        "t = time.time(); again = engine.pairs(a, b, scope); second = time.time() - t\n"
        "want = oracle.levenshtein_pairs(a, b, algo='hyyro', count=20000)\n"
        "assert (got[:20000] == want).all() and (got == again).all()\n"
        "assert second < 1.0, second\n"
        "print('gave-up ok', round(first, 2), round(second, 3))\n")
    env = dict(os.environ, **TEST_LIBRARY_ENV, STRINGWARS_AMD_FUSED_OVERSUBSCRIBE="1", PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert done.returncode == 0 and "gave-up ok" in done.stdout, (done.stdout[-500:], done.stderr[-2000:])


@pytest.mark.parametrize("shapes", [{"STRINGWARS_AMD_BP_WAVES": "4", "STRINGWARS_AMD_TILED_WAVES": "4"},
                                    {"STRINGWARS_AMD_BP_WAVES": "10", "STRINGWARS_AMD_TILED_WAVES": "8"}])
def test_comparison_knobs_keep_parity(shapes):
    """The environment knobs DESIGN.md quotes A/B numbers from select other launch shapes of the same kernels (four-wave
    workgroups with fixed item lists, no affix cut, k_direct_short for every word-sized batch): each must stay bit-exact."""
    import subprocess
    import sys
    code = (
        "import numpy as np, oracle, stringwars_amd as sw\n"
        "scope = sw.DeviceScope(gpu_device=0)\n"
        "for workload, count in (('tokens64', 120000), ('short_words', 150000), ('utf8_lines', 900)):\n"
        "    a, b = sw.generate_pairs(workload, count, seed=13)\n"
        "    want = oracle.levenshtein_pairs(a, b, algo='hyyro')\n"
        "    pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)\n"
        "    for algorithm in ('auto', 'bitparallel', 'tiled'):\n"
        "        engine = sw.LevenshteinDistances(capabilities=scope, algorithm=algorithm)\n"
        "        assert (engine.pairs(pa, pb, scope) == want).all(), (workload, algorithm)\n"
        "        assert (engine.pairs(a, b, scope, bound=5) == np.minimum(want, 6)).all(), (workload, algorithm)\n"
        "ta, tb = sw.Strs([('\u0416\u4e2d\U0001F600a' * (3 + i % 17))[: 5 + i % 60] for i in range(30000)]), sw.Strs([('\u4e2d\u0416a\U0001F600' * (2 + i % 19))[: 4 + i % 70] for i in range(30000)])\n"
        "assert (sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm='tiled').pairs(ta, tb, scope) == oracle.levenshtein_pairs(ta, tb, utf8=True)).all()\n"
        "ua, ub = sw.generate_pairs('utf8_lines', 700, seed=14)\n"
        "assert (sw.LevenshteinDistancesUTF8(capabilities=scope, algorithm='bitparallel').pairs(ua, ub, scope) == oracle.levenshtein_pairs(ua, ub, utf8=True)).all()\n"
        "rng = np.random.default_rng(5)\n"
        "la = [rng.integers(97, 101, 2200).astype(np.uint8).tobytes() for _ in range(4300)]\n"
        "lb = [x if i % 2 else rng.integers(97, 101, 2100).astype(np.uint8).tobytes() for i, x in enumerate(la)]\n"
        "long_a, long_b = sw.Strs(la), sw.Strs(lb)\n"
        "got = sw.LevenshteinDistances(capabilities=scope).pairs(long_a, long_b, scope)\n"
        "assert (got[1::2] == 0).all() and (got[0::2] >= 100).all() and (got[0::2] <= 2200).all()\n"
        "pick = list(range(0, 4300, 430))\n"
        "assert got[pick].tolist() == oracle.levenshtein_pairs(sw.Strs([la[i] for i in pick]), sw.Strs([lb[i] for i in pick]), algo='hyyro').tolist()\n"
        "pa, pb = sw.generate_pairs('protein4k', 6, seed=5)\n"
        "matrix = sw.substitution_matrix(5)\n"
        "for gaps in ((-4, -4), (-11, -1)):\n"
        "    scope.set_profiling(True)\n"
        "    got = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope).pairs(pa, pb, scope)\n"
        "    assert scope.last_timing()['dominant_name'].startswith('wavefront_class'), scope.last_timing()   # STRINGWARS_AMD_NW=classic\n"
        "    scope.set_profiling(False)\n"
        "    assert (got == oracle.nw_pairs(pa, pb, matrix, *gaps)).all(), gaps\n"
        "print('knobs ok')\n")
    env = dict(os.environ, **TEST_LIBRARY_ENV, **shapes, STRINGWARS_AMD_AFFIX="0", STRINGWARS_AMD_SHORT="direct", STRINGWARS_AMD_NW="classic",
               STRINGWARS_AMD_LONG_TICKET="0", STRINGWARS_AMD_BAND_ITEMS="fixed", STRINGWARS_AMD_BAND_CAP="64",
               PYTHONPATH=child_pythonpath())
    done = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "knobs ok" in done.stdout, (done.stdout[-500:], done.stderr[-2000:])


BENCH_LEGS = ["c1", "c3", "c3_raw", "c3_raw_cold", "c3_raw_forget", "utf8_unbounded_raw", "utf8_unrelated_raw", "c3_k100", "c4_linear", "c4_affine", "c4_bytes",
              "c4_letters52", "c5", "nw_words", "sw_linear", "sw_affine", "cross_lev", "cross_nw"]


def bench_stdout(done):
    """The JSON lines of a bench run's stdout: (every `{"leg": ...}` summary line, the LAST line = the headline)."""
    rows = [json.loads(l) for l in done.stdout.splitlines() if l.startswith("{")]
    return [r for r in rows[:-1] if "leg" in r], rows[-1], len(done.stdout.splitlines()[-1])


def test_bench_line_is_short_and_the_details_carry_every_config(tmp_path):
    """`bench.py` as the driver runs it at N = 1 (scaled down). The LAST stdout line is the headline only -- the contract's keys, the
    full roofline numbers, `cpu_baseline`, one {value, frac, parity} summary per config -- and fits 6 KB (the driver could not parse
    round 5's 24 KB line); one short `{"leg": ...}` line per config precedes it; the full entries (C1, C3 prepared and raw, C4 linear /
    affine / full byte alphabet, C5, Smith-Waterman, the cross-product call: rate, kernel time, roofline object, parity sample) and the
    CPU rows are in --details-out."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    details_path = str(tmp_path / "bench_configs.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--pairs", "50000", "--steps", "5", "--warmup", "1", "--prewarm-seconds", "0.05",
           "--steady-seconds", "0.05", "--cpu-seconds", "0.2", "--leg-pairs", "600", "--details-out", details_path]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert done.returncode == 0, (done.stdout[-1000:], done.stderr[-3000:])
    legs, line, line_bytes = bench_stdout(done)
    assert line_bytes <= 6144 and len(done.stdout) <= 16384, (line_bytes, len(done.stdout))
    assert line["unit"] == "GCUPS" and line["n_gpus"] == 1 and line["steps"] == 5 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["value"] > 0 and line["value_steady"] > 0 and line["value_pipelined"] > 0 and line["parity_vs_oracle"] is True
    assert abs(line["value"] - line["config"]["cells_per_gpu"] * 5 / (line["ms_per_step"] * 5e-3) / 1e9) < 0.02 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "valu" and roof["kernel_ms"] > 0 and roof["kernel"] == "bitparallel_tiled" and roof["peak"] > 70 and roof["unit"] == "Tint32op/s"
    assert {"achieved", "frac", "traffic"} <= set(roof)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 0 and len(line["cpu_baselines"]) == 5
    assert list(line["configs"]) == BENCH_LEGS and [r["leg"] for r in legs] == BENCH_LEGS
    for name, summary in line["configs"].items():
        assert "error" not in summary and summary["value"] > 0 and summary["parity"] is True and summary["kernel_ms"] > 0, (name, summary)
    # the full entries
    details = json.load(open(details_path))
    assert [e["config"] for e in details["configs"]] == BENCH_LEGS and details["headline"]["roofline"]["measured"].startswith("hipEvents")
    for entry in details["configs"]:
        assert "error" not in entry, entry
        assert entry["value"] > 0 and entry["parity_vs_oracle"] is True and entry["roofline"]["kernel_ms"] > 0, entry
        assert entry["pairs"] == (576 if entry["config"].startswith("cross_") else 600), entry      # a 24 x 24 matrix for the cross-product calls
        assert "cells_mismatch" not in entry, entry
    rows = details["cpu_baselines"]
    c1_cpu = [row for row in rows if row["name"] == "c1/cpu::hyyro<1cpu>"]
    assert len(c1_cpu) == 1 and c1_cpu[0]["cores"] == 1 and c1_cpu[0]["value"] > 0           # BASELINE configs[0]: the per-pair CPU row on 10 K words
    many = [row for row in rows if row["cores"] > 1]
    assert len(many) == 1 and many[0]["cores"] == os.cpu_count()                                 # every hardware thread of the box


@pytest.mark.parametrize("config,pairs", [("c2", 60_000), ("c5", 600_000)])
def test_bench_two_ranks_share_the_gpu(config, pairs):
    """`bench.py` as the driver launches it for N > 1 -- one process per rank under torch.distributed -- with both ranks on
    cuda:0 and gloo in RCCL's place (one-GPU boxes): the sharding, the pipelined steps, the gather (weak: `dist.gather`;
    strong: the chunked send/recv of cells-balanced shards) and its checksums run for real; rank 0's shard is compared with
    the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + (os.getpid() % 300) + (0 if config == "c2" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--config", config, "--pairs", str(pairs),
           "--steps", "3", "--warmup", "1", "--backend", "gloo", "--share-gpu", "--details-out", ""]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert done.returncode == 0, (done.stdout[-1000:], done.stderr[-3000:])
    _, line, line_bytes = bench_stdout(done)
    assert line_bytes <= 8192, line_bytes
    assert line["n_gpus"] == 2 and line["gather_ok"] is True and line["parity_vs_oracle"] is True
    assert line["scaling"] == ("weak" if config == "c2" else "strong") and line["value"] > 0


def test_bench_starts_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` with NO torchrun prefix and no WORLD_SIZE (the shape of the driver's N = 1 command): the script
    starts its ranks as a child process before touching the GPU and rank 0's line says `n_gpus: 2`. At N > 1 the headline line (<= 8 KB)
    also summarises BASELINE configs[4] -- C5, strong scaling -- with its checked gather and the in-library sharded call (`single_process`);
    --details-out carries the full entries: C5's shard ranges, the gather's price (u32 on the wire as the north star words it; the u8
    variant beside it), the ranks that took part, the single-process child's own line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    details_path = str(tmp_path / "bench_configs.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--pairs", "60000", "--c5-pairs", "700000", "--steps", "3", "--warmup", "1",
           "--prewarm-seconds", "0.05", "--steady-seconds", "0.05", "--backend", "gloo", "--share-gpu", "--details-out", details_path]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert done.returncode == 0, (done.stdout[-1000:], done.stderr[-3000:])
    legs, line, line_bytes = bench_stdout(done)
    assert line_bytes <= 8192 and len(done.stdout) <= 16384, (line_bytes, len(done.stdout))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["gather_ok"] is True and line["parity_vs_oracle"] is True and line["value"] > 0
    assert line["ranks_seen"] == {"world_size": 2, "backend": "gloo", "rccl_version": None, "distinct_devices": line["ranks_seen"]["distinct_devices"]}
    assert line["gather"]["alone_ms"] > 0 and line["gather"]["bytes_to_root_per_step"] == 4 * 60000
    assert list(line["configs"]) == ["c5_strong"] and [r["leg"] for r in legs] == ["c5_strong"]
    summary = line["configs"]["c5_strong"]
    assert summary["value"] > 0 and summary["gather_ok"] is True and summary["parity"] is True and summary["n_gpus"] == 2 and summary["gather_u8_value"] > 0
    assert line["single_process"]["mode"] == "single-process" and line["single_process"]["n_gpus"] == 2 and line["single_process"]["parity_vs_oracle"] is True
    assert line["single_process"]["value"] > 0 and line["single_process"]["device_count"] == 2
    details = json.load(open(details_path))
    assert [r["rank"] for r in details["ranks_seen"]["ranks"]] == [0, 1]
    (c5,) = details["configs"]
    assert c5["config"] == "c5_strong" and c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["pairs_total"] == 700000, c5
    assert c5["gather_ok"] is True and c5["parity_vs_oracle"] is True and c5["value"] > 0 and c5["ranks_seen"]["world_size"] == 2
    assert c5["shard_ranges"][0][0] == 0 and c5["shard_ranges"][0][1] == c5["shard_ranges"][1][0] and c5["shard_ranges"][1][1] == 700000
    # the entry is the north star's collective -- the u32 distances as they are --, the bytes-on-the-wire variant rides beside it
    assert c5["gather"]["transport"] == "u32" and c5["gather"]["bytes_to_root_per_step"] == 4 * (700000 - c5["shard_ranges"][0][1])
    narrow = c5["gather_u8"]
    assert "error" not in narrow and narrow["gather_ok"] is True and narrow["same_results"] is True and narrow["value"] > 0, narrow
    assert narrow["gather"]["transport"].startswith("u8") and narrow["gather"]["bytes_to_root_per_step"] == 700000 - c5["shard_ranges"][0][1]
    single = details["single_process"]
    assert "error" not in single, single
    assert single["mode"] == "single-process" and single["n_gpus"] == 2 and single["parity_vs_oracle"] is True and single["value"] > 0
    assert single["config"]["device_count"] == 2 and single["config"]["shard_cuts"][-1] == 2 * 60000


def test_bench_failures_at_two_ranks_end_with_a_line():
    """What can go wrong at N > 1 ends in ONE JSON line within a bounded time. (1) A rank that dies while the others are in their first
    collectives: non-zero exit code, a line carrying `error`. (2) The in-library single-process leg (a child of rank 0, bounded by a
    timeout) failing: the parent's line is there all the same, exit code 0, the failure inside `single_process`."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--pairs", "30000", "--steps", "2", "--warmup", "1", "--prewarm-seconds", "0.05",
            "--steady-seconds", "0.05", "--backend", "gloo", "--share-gpu", "--no-configs", "--no-cpu-baseline", "--details-out", ""]
    for at in ("after-init", "measure"):
        started = time.time()
        done = subprocess.run(base + ["--die-rank", "1", "--die-at", at, "--collective-timeout", "40", "--launch-timeout", "200"],
                              capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert done.returncode != 0 and time.time() - started < 400, (at, done.returncode)
        lines = [json.loads(l) for l in done.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and "error" in lines[-1] and lines[-1]["value"] is None, (at, done.stdout[-800:], done.stderr[-800:])   # ONE line: rank 0's or the launcher's
    done = subprocess.run(base + ["--single-process-timeout", "0.01"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert done.returncode == 0, (done.stdout[-1000:], done.stderr[-2000:])
    line = json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["gather_ok"] is True and "error" in line["single_process"], line.get("single_process")


def test_bench_single_process_mode():
    """`bench.py --gpus 3 --single-process --config c5`: one process, one scope over three member devices (all of them device 0 on
    this box), `swh_sharded_prepare_*` + `swh_levenshtein_pairs_sharded` per step with the self-check on every call."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--single-process", "--share-gpu", "--config", "c5", "--pairs", "900000",
           "--steps", "3", "--warmup", "1", "--prewarm-seconds", "0.05"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert done.returncode == 0, (done.stdout[-1000:], done.stderr[-3000:])
    line = json.loads([l for l in done.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["mode"] == "single-process" and line["scaling"] == "strong" and line["parity_vs_oracle"] is True
    assert line["config"]["shard_cuts"][0] == 0 and line["config"]["shard_cuts"][-1] == 900000 and line["value"] > 0


@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("gaps", [(-2, -2), (-5, -1), (-4, -4), (-11, -1)])
@pytest.mark.parametrize("classes", [4, 8, 21, 32])
def test_word_sized_alignment_kernel(sw, orc, scope, local, gaps, classes):
    """alignshort.hip: NW / SW scores of strings of at most 32 bytes on a class table, one pair per lane -- the reference's
    default `words` token mode (bench.rs:271; its rows: perform_linear_benchmarks / perform_affine_benchmarks, bench.rs:641-699,
    :967-1026). Every length 0..32 on both sides (16- and 32-column variants), <= 8 classes (one v_perm per four columns) and up to
    32 (four), asymmetric class costs, linear and affine gaps, global and local, pairwise (prepared tapes, u32 and u64 offsets; raw
    tapes once the scope has seen their lengths) and the cross-product entry point, against the oracle's Gotoh; then a batch with one
    string of 33 bytes: the kernel refuses it, the call is redone on the planned path."""
    rng = np.random.default_rng(classes * 131 + gaps[0] * 7 + local)
    byte_to_class = rng.integers(0, classes, 256).astype(np.uint8)
    costs = np.zeros((32, 32), dtype=np.int8)
    costs[:classes, :classes] = rng.integers(-9, 12, (classes, classes))            # asymmetric on purpose
    full = costs[byte_to_class][:, byte_to_class].astype(np.int8)
    Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
    engine = Engine(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=scope)

    def want_pairs(xs, ys):
        return np.array([orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for x, y in zip(xs, ys)], dtype=np.int64)

    for longest in (16, 32):
        lengths = list(range(0, longest + 1))
        items_a, items_b = random_pairs(rng, 700, lengths, 256, related=0.4)
        items_a = [x[:longest] for x in items_a]
        items_b = [x[:longest] for x in items_b]
        items_a += [b"", b"", bytes(range(longest)), bytes(longest)]
        items_b += [b"", bytes(range(3)), bytes(range(longest)), bytes(range(longest))]
        a, b = sw.Strs(items_a), sw.Strs(items_b)
        want = want_pairs(items_a, items_b)
        for offsets in (np.uint32, np.uint64):
            pa, pb = sw.PreparedTape(scope, a.with_offsets(offsets)), sw.PreparedTape(scope, b.with_offsets(offsets))
            scope.set_profiling(True)
            got = engine.pairs(pa, pb, scope)
            timing = scope.last_timing()
            scope.set_profiling(False)
            assert timing["dominant_name"].startswith("align_short") and timing["dominant_name"].endswith(f"w{longest}"), timing
            assert timing["cells"] == int((a.lengths.astype(np.int64) * b.lengths.astype(np.int64)).sum())
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (longest, offsets, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
            # sub-views of the prepared tapes (items that start in the middle of a wave's 64 pairs)
            assert (engine.pairs(pa[37:500], pb[37:500], scope) == want[37:500]).all()
        # raw tapes: the first call plans (and learns the lengths), the second takes the lane-per-pair kernel; same scores
        fresh = sw.DeviceScope(gpu_device=0)
        first = engine.pairs(a, b, fresh)
        fresh.set_profiling(True)
        second = engine.pairs(a, b, fresh)
        assert fresh.last_timing()["dominant_name"].startswith("align_short"), fresh.last_timing()
        fresh.set_profiling(False)
        assert (first == want).all() and (second == want).all()
        # queries x candidates (compute_into, bench.rs:478-486): more queries than one item holds, candidates beyond one chunk of 64
        q, c = sw.Strs(items_a[:41]), sw.Strs(items_b[100:231])
        pq, pc = sw.PreparedTape(scope, q), sw.PreparedTape(scope, c)
        scope.set_profiling(True)
        matrix = engine(pq, pc, scope)
        assert scope.last_timing()["dominant_name"].startswith("align_short"), scope.last_timing()
        scope.set_profiling(False)
        want_matrix = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in items_b[100:231]] for x in items_a[:41]])
        assert matrix.shape == (41, 131) and (matrix == want_matrix).all()
    # one string too long for the kernel among word-sized ones, on a scope that believes in word-sized strings: refused, redone
    fresh = sw.DeviceScope(gpu_device=0)
    items_a, items_b = random_pairs(rng, 300, list(range(0, 17)), 256)
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    assert (engine.pairs(a, b, fresh) == want_pairs(items_a, items_b)).all()
    items_a[150] = bytes(rng.integers(0, 256, 33, dtype=np.uint8))
    a = sw.Strs(items_a)
    assert (engine.pairs(a, b, fresh) == want_pairs(items_a, items_b)).all()
    items_b[7] = bytes(rng.integers(0, 256, 90, dtype=np.uint8))
    q, c = sw.Strs(items_a[:20]), sw.Strs(items_b[:70])
    want_matrix = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in items_b[:70]] for x in items_a[:20]])
    assert (engine(q, c, fresh) == want_matrix).all()


def test_word_sized_alignment_reference_rows(sw, orc, scope):
    """The reference's own alignment rows on word-sized tokens: `unary_class_costs(2, -1)` (bench.rs:98-108: class = byte % 32, so
    all 32 classes are in play), linear (-2, -2) and affine (-5, -1) gaps (bench.rs:640, :966), NW and SW, a 300 x 300 cross-product of
    the synthetic words -- every score against the oracle."""
    a, b = sw.generate_pairs("words16", 600, seed=5)
    queries, candidates = a.subview(0, 300), b.subview(300, 600)
    byte_to_class, costs = sw.unary_class_costs(2, -1)
    full = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    for Engine, local in ((sw.NeedlemanWunschScores, False), (sw.SmithWatermanScores, True)):
        for gaps in ((-2, -2), (-5, -1)):
            engine = Engine(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=scope)
            first = engine(queries, candidates, scope)           # raw host tapes: plans, learns the lengths
            scope.set_profiling(True)
            again = engine(queries, candidates, scope)
            assert scope.last_timing()["dominant_name"].startswith("align_short"), scope.last_timing()
            scope.set_profiling(False)
            want = np.array([[orc.nw_score(queries[i], candidates[j], full, gaps[0], gaps[1], local=local) for j in range(300)] for i in range(300)])
            assert (first == want).all() and (again == want).all(), (local, gaps)


@pytest.mark.parametrize("local", [False, True])
def test_small_alphabet_cross_product_up_to_128_symbols(sw, orc, scope, local):
    """k_align_cross_wide (alignshort.hip): queries x candidates with linear gaps on strings of 33..128 symbols whose candidates use at
    most eight symbol classes per work item (DNA under the reference's `unary_class_costs`: A, C, G, T are classes 1, 3, 7, 20) -- the
    classes are compacted per item so that one v_perm serves four columns. Every length 0..128 on both sides, the 64- and the
    128-column variant, a random asymmetric class table, global and local; then candidates over 26 letters: the kernel refuses
    them, the call is redone on the planned path and the scope stops trying."""
    rng = np.random.default_rng(77 + local)
    Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
    byte_to_class, costs = sw.unary_class_costs(2, -1)
    random_costs = rng.integers(-9, 12, (32, 32)).astype(np.int8)
    for table, gaps in ((costs, (-2, -2)), (random_costs, (-3, -3))):
        full = np.array([[table[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
        engine = Engine(byte_to_class, table, open=gaps[0], extend=gaps[1], capabilities=scope)
        for longest in (64, 128):
            def dna(n):
                return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(n)))
            queries = [dna(n) for n in list(range(0, longest + 1, 7)) + [longest, longest - 1, 1, 0, 33]]
            candidates = [dna(n) for n in rng.integers(0, longest + 1, 150)] + [dna(longest), b"", dna(1)]
            candidates[5] = queries[3][:longest]                                 # related strings, too
            q, c = sw.PreparedTape(scope, sw.Strs(queries)), sw.PreparedTape(scope, sw.Strs(candidates))
            scope.set_profiling(True)
            got = engine(q, c, scope)
            name = scope.last_timing()["dominant_name"]
            scope.set_profiling(False)
            assert name.startswith("align_wide") and name.endswith(f"w{longest}"), name
            want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in candidates] for x in queries])
            bad = np.argwhere(got != want)
            assert bad.size == 0, (longest, bad[:5], [(len(queries[i]), len(candidates[j]), got[i, j], want[i, j]) for i, j in bad[:5]])
    # text over 26 letters: more than eight classes among a wave's candidates
    fresh = sw.DeviceScope(gpu_device=0)
    engine = Engine(byte_to_class, costs, open=-2, extend=-2, capabilities=fresh)
    full = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    words = [bytes(rng.integers(97, 123, int(n), dtype=np.uint8)) for n in rng.integers(20, 100, 90)]
    q, c = sw.PreparedTape(fresh, sw.Strs(words[:20])), sw.PreparedTape(fresh, sw.Strs(words[20:]))
    want = np.array([[orc.nw_score(x, y, full, -2, -2, local=local) for y in words[20:]] for x in words[:20]])
    for _ in range(2):
        fresh.set_profiling(True)
        got = engine(q, c, fresh)
        name = fresh.last_timing()["dominant_name"]
        fresh.set_profiling(False)
        assert (got == want).all() and not name.startswith("align_wide"), name


def test_engine_clones_go_with_their_engine(sw, orc):
    """A multi-device scope clones an alignment engine per member on first use (its matrix lives on ONE device); freeing the engine
    frees the clones -- a harness that builds an engine per row on a long-lived scope must not pile up matrices on every device."""
    import torch
    scope = sw.DeviceScope(gpu_devices=[0, 0, 0])
    a, b = sw.generate_pairs("words16", 3000, seed=3)
    batch = sw.ShardedPairs(scope, a, b)
    matrix = sw.substitution_matrix(5)
    want = orc.nw_pairs(a, b, matrix, -3, -3)

    def one_engine():
        engine = sw.NeedlemanWunschScores(substitution_matrix=matrix, open=-3, extend=-3, capabilities=scope)
        assert (engine.pairs_sharded(batch, scope) == want).all()
        del engine

    for _ in range(8):
        one_engine()                                   # scratch of the member scopes reaches its size
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(96):
        one_engine()                                   # 96 engines x 3 members x (64 KB matrix + class table) = 19 MB if the clones stayed
    torch.cuda.synchronize()
    assert free_before - torch.cuda.mem_get_info()[0] < 6 << 20


@pytest.mark.parametrize("utf8", [False, True])
def test_banded_windows_of_two_words(sw, orc, scope, utf8):
    """Bounds of 64 .. 127 (`STRINGWARS_ERROR_BOUND` is free-form, README.md:311) on the banded kernel's two-word windows: strings long
    enough for the band to pay (1200 .. 2600 symbols), edit counts below, at and above every bound, length differences up to the bound and
    beyond, both argument orders, bytes and code points -- out = min(d, k + 1) against the oracle; the call must really run on `banded`.
    Shorter strings in the same batch take the bit-parallel kernels and the clamp, and so do bounds beyond 127."""
    rng = np.random.default_rng(91 + utf8)
    alphabet = [chr(c) for c in range(0x61, 0x7B)] + (["é", "я", "語", "😀"] if utf8 else [])
    items_a, items_b = [], []
    for _ in range(260):
        n = int(rng.integers(1200, 2600)) if rng.random() < 0.85 else int(rng.integers(1, 400))
        a = [alphabet[int(i)] for i in rng.integers(0, len(alphabet), n)]
        b = list(a)
        for _ in range(int(rng.choice([0, 3, 40, 63, 64, 65, 90, 100, 127, 128, 129, 160, 191, 192, 200, 255, 256, 300]))):
            op = int(rng.integers(0, 3))
            if op == 0 and b:
                b[int(rng.integers(0, len(b)))] = alphabet[int(rng.integers(0, len(alphabet)))]
            elif op == 1:
                b.insert(int(rng.integers(0, len(b) + 1)), alphabet[int(rng.integers(0, len(alphabet)))])
            elif len(b) > 1:
                del b[int(rng.integers(0, len(b)))]
        if rng.random() < 0.15:  # pure deletions: length difference == distance
            b = a[: max(1, n - int(rng.integers(0, 300)))]
        items_a.append("".join(a))
        items_b.append("".join(b))
    a, b = sw.Strs(items_a), sw.Strs(items_b)
    full = orc.levenshtein_pairs(a, b, utf8=utf8)
    engine = (sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances)(capabilities=scope)
    for k in (64, 65, 95, 96, 97, 100, 126, 127):
        scope.set_profiling(True)
        got = engine.pairs(a, b, scope, bound=k)
        timing = scope.last_timing()
        scope.set_profiling(False)
        want = np.minimum(full, k + 1)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (k, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
        assert timing["dominant_name"] == "banded", (k, timing)
        assert (engine.pairs(b, a, scope, bound=k) == want).all(), k
    for k in (128, 200, 255, 256):                                                      # beyond the band: unbounded kernels + the clamp
        assert (engine.pairs(a, b, scope, bound=k) == np.minimum(full, k + 1)).all(), k


@pytest.mark.parametrize("local", [False, True])
@pytest.mark.parametrize("gaps", [(-2, -2), (-5, -1)])
def test_small_alphabet_cross_product_of_any_length(sw, orc, scope, local, gaps, request):
    """k_align_cross_long (alignshort.hip): queries x candidates of any length (up to 4096 symbols) over a small alphabet, the columns run as
    passes of 128 (local or Gotoh: 64, both: 32) with the boundary column between passes parked in global memory. Candidate lengths around every pass boundary (0, 1,
    63 .. 65, 127 .. 129, 255 .. 257, 300, 384), query lengths odd and even, linear and affine gaps, global and local, the reference's
    unary class costs and a random asymmetric table -- every score against the oracle. The call must run on `align_long`; longer strings,
    or candidates over 26 letters, take the planned path (and score the same)."""
    rng = np.random.default_rng(300 + local + gaps[0])
    Engine = sw.SmithWatermanScores if local else sw.NeedlemanWunschScores
    byte_to_class, costs = sw.unary_class_costs(2, -1)
    random_costs = rng.integers(-9, 12, (32, 32)).astype(np.int8)

    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(n)))

    for table in (costs, random_costs):
        full = np.array([[table[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
        engine = Engine(byte_to_class, table, open=gaps[0], extend=gaps[1], capabilities=scope)
        queries = [dna(n) for n in (0, 1, 2, 33, 100, 129, 300, 383, 384)]
        candidates = [dna(n) for n in [0, 1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 384] + list(rng.integers(0, 385, 70))]
        candidates[20] = queries[6][:280] + dna(15)                              # related strings, too
        q, c = sw.PreparedTape(scope, sw.Strs(queries)), sw.PreparedTape(scope, sw.Strs(candidates))
        scope.set_profiling(True)
        got = engine(q, c, scope)
        name = scope.last_timing()["dominant_name"]
        scope.set_profiling(False)
        assert name.startswith("align_long"), name
        want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in candidates] for x in queries])
        bad = np.argwhere(got != want)
        assert bad.size == 0, (gaps, local, bad[:5], [(len(queries[i]), len(candidates[j]), got[i, j], want[i, j]) for i, j in bad[:5]])
    fresh = sw.DeviceScope(gpu_device=0)
    engine = Engine(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=fresh)
    full = np.array([[costs[i % 32, j % 32] for j in range(256)] for i in range(256)], dtype=np.int8)
    queries, candidates = [dna(200), dna(1400)], [dna(150), dna(40), dna(700)]
    want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in candidates] for x in queries])
    assert (engine(sw.PreparedTape(fresh, sw.Strs(queries)), sw.PreparedTape(fresh, sw.Strs(candidates)), fresh) == want).all()
    # longer ones: the kernel is chosen as far as it measured faster than the column profiles (api.hip: align_long_pays) -- 4096 symbols
    # for linear global alignment, 2048 for Gotoh or local, 384 for both -- and the scores are the oracle's on either side of that
    limit = 384 if (local and gaps[0] != gaps[1]) else (2048 if (local or gaps[0] != gaps[1]) else 4096)
    for top in (2048, 4096):
        queries = [dna(top), dna(top - 1), dna(top // 2 + 3)]
        candidates = [dna(n) for n in (150, 40, 700, top, top // 2 + 1, top - 511, 0)] + [dna(n) for n in rng.integers(1, top + 1, 60)]
        fresh.set_profiling(True)
        got = engine(sw.PreparedTape(fresh, sw.Strs(queries)), sw.PreparedTape(fresh, sw.Strs(candidates)), fresh)
        name = fresh.last_timing()["dominant_name"]
        fresh.set_profiling(False)
        assert name.startswith("align_long") == (top <= limit), (top, limit, name)
        want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in candidates] for x in queries])
        assert (got == want).all(), (top, np.argwhere(got != want)[:5])
    # the boundary columns are budgeted (640 MB: a call that would need more takes the column-profile kernel; STRINGWARS_AMD_ALIGN_BOUNDARY_MB, a
    # hook of the TEST library, gives a launch past the budget fewer waves instead): with 3 MB the 200 x 70 strings below run on one or two
    # workgroups, every wave taking many work items in turn -- the whole test once more in a child process on the test library
    if not run_in_child(request, env={"STRINGWARS_AMD_ALIGN_BOUNDARY_MB": "3"}, test_library=True):
        return
    queries, candidates = [dna(n) for n in rng.integers(300, 385, 200)], [dna(n) for n in rng.integers(1, 385, 70)]
    fresh.set_profiling(True)
    got = engine(sw.PreparedTape(fresh, sw.Strs(queries)), sw.PreparedTape(fresh, sw.Strs(candidates)), fresh)
    name = fresh.last_timing()["dominant_name"]
    fresh.set_profiling(False)
    assert name.startswith("align_long"), name
    want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in candidates] for x in queries])
    assert (got == want).all(), np.argwhere(got != want)[:5]
    text = [bytes(rng.integers(97, 123, int(n), dtype=np.uint8)) for n in rng.integers(150, 420, 80)]
    want = np.array([[orc.nw_score(x, y, full, gaps[0], gaps[1], local=local) for y in text[10:]] for x in text[:10]])
    fresh2 = sw.DeviceScope(gpu_device=0)
    engine2 = Engine(byte_to_class, costs, open=gaps[0], extend=gaps[1], capabilities=fresh2)
    for _ in range(2):
        assert (engine2(sw.PreparedTape(fresh2, sw.Strs(text[:10])), sw.PreparedTape(fresh2, sw.Strs(text[10:])), fresh2) == want).all()


def test_golden_multilingual_words(sw, scope):
    """tests/golden/uwords.npz (made by tests/golden/make_fixtures.py::multilingual_words): 24 x 40 word tokens of four scripts with a few long
    ones among them -- the committed oracle matrices for Levenshtein over bytes and over code points and for NW / SW with the reference's
    unary class costs, against the cross-product routes on prepared tapes, raw host tapes and raw device tapes, each twice (the second
    call acts on what the first one learnt)."""
    z = np.load(os.path.join(GOLDEN, "uwords.npz"))
    q = sw.Strs(data=z["q_data"], offsets=z["q_offsets"])
    c = sw.Strs(data=z["c_data"], offsets=z["c_offsets"])
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    engines = {"lev_bytes": sw.LevenshteinDistances(capabilities=scope), "lev_utf8": sw.LevenshteinDistancesUTF8(capabilities=scope)}
    for tag, gaps in (("linear_m2", (-2, -2)), ("affine_m5_m1", (-5, -1))):
        engines[f"nw_unary_{tag}"] = sw.NeedlemanWunschScores(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=scope)
        engines[f"sw_unary_{tag}"] = sw.SmithWatermanScores(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=scope)
    for name, engine in engines.items():
        want = z[name]
        utf8 = name == "lev_utf8"
        tapes = [(sw.PreparedTape(scope, q, utf8=utf8), sw.PreparedTape(scope, c, utf8=utf8)), (q, c), (q.to_device(scope), c.to_device(scope))]
        for tq, tc in tapes:
            for _ in range(2):
                got = engine(tq, tc, scope)
                assert (got == want).all(), (name, type(tq).__name__, np.argwhere(got != want)[:5])
        # the word-sized part alone (no long tokens): the lane / wave-shared kernels' own range
        short_q, short_c = q.subview(0, 21), c.subview(0, 36)
        for _ in range(2):
            assert (engine(short_q, short_c, scope) == want[:21, :36]).all(), (name, "words only")
        tapes[2][0].free(); tapes[2][1].free()


def test_golden_words_alignment_rows(sw, scope):
    """tests/golden/slices.npz: the reference's alignment rows on word-sized tokens -- `unary_class_costs(2, -1)`, linear -2 / -2 and
    affine -5 / -1 (bench.rs:640, :655, :966), NW and SW -- for the first 256 `words16` pairs (pairwise, prepared and raw tapes) and the
    16 x 16 cross-product of the first 16 strings of either tape: the committed oracle outputs against k_align_short."""
    z = np.load(os.path.join(GOLDEN, "slices.npz"))
    a = sw.Strs(data=z["words16.a_data"], offsets=z["words16.a_offsets"])
    b = sw.Strs(data=z["words16.b_data"], offsets=z["words16.b_offsets"])
    byte_to_class, class_costs = sw.unary_class_costs(2, -1)
    for tag, gaps in (("linear_m2", (-2, -2)), ("affine_m5_m1", (-5, -1))):
        for kind, Engine in (("nw", sw.NeedlemanWunschScores), ("sw", sw.SmithWatermanScores)):
            engine = Engine(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=scope)
            want = z[f"words16.n256.{kind}_unary_{tag}"]
            pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
            scope.set_profiling(True)
            got = engine.pairs(pa, pb, scope)
            assert scope.last_timing()["dominant_name"].startswith("align_short"), scope.last_timing()
            scope.set_profiling(False)
            assert (got == want).all(), (kind, tag)
            assert (engine.pairs(a, b, scope) == want).all() and (engine.pairs(a, b, scope) == want).all(), (kind, tag, "raw tapes, twice")
            cross = engine(sw.PreparedTape(scope, a.subview(0, 16)), sw.PreparedTape(scope, b.subview(0, 16)), scope)
            assert (cross == z[f"words16.cross16.{kind}_unary_{tag}"]).all(), (kind, tag, "cross-product")


def test_code_point_items_on_a_dense_alphabet(sw, orc, request):
    """bp_dense.hpp: a code-point work item of 16 blocks and more gives the distinct symbols of each pair's pattern 8-bit ids (the slot
    of a 251-slot dictionary in LDS that takes a symbol is its id) and then runs on the byte kernel's nibble tables; text symbols the
    pattern does not hold translate to the id with the empty match vector; a pattern with more distinct symbols than the dictionary takes
    (~200: a sketch sends it away, or a lane runs out of probes) goes to the seven group tables as before, and so does every later pass
    of a pattern beyond 2048 symbols (k_bitparallel_long keeps the dictionary from pass to pass). `LevenshteinDistancesUtf8`,
    bench.rs:386-399: lines of 300 ... 6200 code points over alphabets of 2 / 32 / ~190 / 230 / 251 / 256 / 3000 symbols, the edges of
    every UTF-8 length, related and unrelated pairs. (Test library: its kernels count the items and passes of either kind.)"""
    if not run_in_child(request, test_library=True):
        return
    import ctypes as C
    from stringwars_amd import _native as N
    scope = sw.DeviceScope(gpu_device=0)
    rng = np.random.default_rng(77)
    edges = [0, 1, 0x7F, 0x80, 0x7FF, 0x800, 0xD7FF, 0xE000, 0xFFFF, 0x10000, 0x10FFFF]
    pools = {
        "two": [0x430, 0x1F600],
        "few": list(range(0x430, 0x450)),
        "text": list(range(0x20, 0x7F)) + list(range(0x400, 0x460)),
        "edges": edges + list(range(0x3040, 0x3040 + 60)),
        "230": list(range(0x4E00, 0x4E00 + 230)),
        "251": list(range(0x4E00, 0x4E00 + 251)),
        "256": list(range(0x4E00, 0x4E00 + 256)),
        "many": list(range(0x4E00, 0x4E00 + 3000)),
    }
    def draw(pool, n, cover=False):
        picks = rng.choice(pool, n)
        if cover and n >= len(pool):
            picks[rng.permutation(n)[: len(pool)]] = pool      # every symbol of the pool at least once
        return "".join(chr(int(c)) for c in picks)
    def edit(s, pool, edits):
        out = list(s)
        for _ in range(edits):
            at = int(rng.integers(0, len(out) + 1))
            kind = int(rng.integers(0, 3))
            if kind == 0 and out: out.pop(min(at, len(out) - 1))
            elif kind == 1: out.insert(at, chr(int(rng.choice(pool))))
            elif out: out[min(at, len(out) - 1)] = chr(int(rng.choice(pool)))
        return "".join(out)
    items_a, items_b, kinds = [], [], []
    for name, pool in pools.items():
        for n in (481, 512, 513, 700, 1000, 1024, 1500, 2047, 2048, 300, 2100):
            for variant in range(3):
                s = draw(pool, n, cover=True)
                if variant == 0: t = edit(s, pool, int(rng.integers(0, 40)))                     # related, same alphabet
                elif variant == 1: t = draw(pool, int(rng.integers(max(1, n - 200), n + 200)))   # unrelated, same alphabet
                else: t = edit(s, pools["text"], int(rng.integers(20, 200)))                     # symbols the other side does not hold
                items_a.append(s); items_b.append(t); kinds.append(name)
    # patterns of more than 64 blocks (k_bitparallel_long: passes of 64 blocks that share the pair's dictionary): a few symbols all along, a
    # pattern whose SECOND pass brings the symbols that do not fit (the first pass has run dense by then), one whose first pass does
    for n, make in ((2100, lambda: draw(pools["text"], 2100)), (4500, lambda: draw(pools["few"], 4500, cover=True)), (6200, lambda: draw(pools["text"], 6200)),
                    (3300, lambda: draw(pools["few"], 2048) + draw(pools["many"], 1252)), (3300, lambda: draw(pools["many"], 2048) + draw(pools["few"], 1252)),
                    (4200, lambda: draw(pools["few"], 2048) + draw(pools["text"], 2048) + draw(pools["edges"], 104))):
        for variant in range(2):
            s = make()
            t = edit(s, pools["text"], int(rng.integers(0, 60))) if variant == 0 else draw(pools["text"], int(rng.integers(n, n + 300)))
            items_a.append(s); items_b.append(t + draw(pools["few"], 40)); kinds.append(f"long {n}")   # (b is the longer one: a is the pattern)
    order = rng.permutation(len(items_a))
    a, b = sw.Strs([items_a[i] for i in order]), sw.Strs([items_b[i] for i in order])
    want = orc.levenshtein_pairs(a, b, utf8=True)
    counts = (C.c_uint32 * 4)()
    N.lib.swh_test_dense_items.argtypes = [C.POINTER(C.c_uint32)]
    assert N.lib.swh_test_dense_items(counts) == 0
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    for tapes in ((a, b), (sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True))):
        got = engine.pairs(*tapes, scope)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]], [kinds[order[i]] for i in bad[:5]])
        assert N.lib.swh_test_dense_items(counts) == 0
        print("dense / declined items, dense / group-table passes:", list(counts))
        assert counts[0] > 30 and counts[1] > 10, list(counts)          # both kinds of item ran: the dense ones and the ones sent to the group tables
        assert counts[2] >= 16 and counts[3] >= 6, list(counts)         # ... and both kinds of pass of the long kernel
    for bound in (0, 31, 200, 5000):                                    # (bounds beyond the band kernels' clamp what the blocks return)
        assert (engine.pairs(a, b, scope, bound=bound) == np.minimum(want, bound + 1)).all(), bound
    # the same batch with the dense alphabet switched off: the group tables alone give the same distances
    os.environ["STRINGWARS_AMD_BP_DENSE"] = "0"
    try:
        assert N.lib.swh_test_dense_items(counts) == 0                  # (zeroes the counters)
        assert (engine.pairs(a, b, scope) == want).all()
        assert N.lib.swh_test_dense_items(counts) == 0 and counts[0] == 0 and counts[1] == 0 and counts[2] == 0 and counts[3] > 0, list(counts)
    finally:
        del os.environ["STRINGWARS_AMD_BP_DENSE"]


def test_scope_beliefs_can_be_read_and_dropped(sw, orc):
    """`swh_scope_describe` / `swh_scope_forget` (round 6): what a scope remembers between calls is visible and can be dropped -- the
    reference's `compute_into` is a pure function of its arguments (bench.rs:478-486), a caller who wants that, or a timing that
    does not depend on history, forgets first. And the small-alphabet latch holds per engine and tapes, not per scope: DNA scored
    after text on the same scope still takes the compacting kernel."""
    scope = sw.DeviceScope(gpu_device=0)
    fresh = scope.describe()
    assert fresh["lengths_believed"] == "0" and fresh["utf8_tape0"] == "unknown" and fresh["align_wide_off_engine"] == "0"
    # raw UTF-8 tapes: the first call sizes them, the scope then believes the totals (and says so); forget() makes the next call cold again
    ua, ub = sw.generate_pairs("utf8_lines", 300, seed=5)
    want = orc.levenshtein_pairs(ua, ub, utf8=True, bound=32)
    da, db = ua.to_device(scope), ub.to_device(scope)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    assert (engine.pairs(da, db, scope, bound=32) == want).all()
    warm = scope.describe()
    assert warm["utf8_tape0"].startswith("sized") and warm["utf8_tape1"].startswith("sized")
    scope.forget()
    assert scope.describe() == fresh
    assert (engine.pairs(da, db, scope, bound=32) == want).all() and scope.describe()["utf8_tape0"].startswith("sized")
    # the compacting cross-product kernel: text with more than eight classes fails it (the call is redone, the result is right) ...
    rng = np.random.default_rng(8)
    classes, costs = sw.unary_class_costs(2, -1)
    matrix = costs[classes][:, classes].astype(np.int8)
    nw = sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope)
    def tape(alphabet, count, length):
        pool = np.frombuffer(alphabet, np.uint8)
        return sw.Strs([pool[rng.integers(0, len(pool), length)].tobytes() for _ in range(count)])
    text_q, text_c = tape(b"abcdefghijklmnopqrstuvwxyz", 40, 90), tape(b"abcdefghijklmnopqrstuvwxyz", 70, 100)
    dna_q, dna_c = tape(b"ACGT", 40, 90), tape(b"ACGT", 70, 100)
    def check(q, c, pq, pc):
        got = nw(pq, pc, scope)
        for i in (0, 17, 39):
            assert [int(x) for x in got[i]] == [orc.nw_score(q[i], c[j], matrix, -2, -2) for j in range(len(c))]
    ptq, ptc, pdq, pdc = (sw.PreparedTape(scope, t) for t in (text_q, text_c, dna_q, dna_c))
    scope.set_profiling(True)
    check(text_q, text_c, ptq, ptc)
    latched = scope.describe()["align_wide_off_engine"]
    assert latched != "0"
    check(text_q, text_c, ptq, ptc)                      # the same engine on the same tapes: not tried again
    assert not scope.last_timing()["dominant_name"].startswith("align_wide")
    check(dna_q, dna_c, pdq, pdc)                        # ... other tapes get their own try
    assert scope.last_timing()["dominant_name"].startswith("align_wide"), scope.last_timing()
    scope.set_profiling(False)
    scope.forget()
    assert scope.describe()["align_wide_off_engine"] == "0"


@pytest.mark.parametrize("gaps", [(0, 0), (-1, -1), (-4096, -4096), (-3, -1), (-4096, 0), (-60, -60)])
def test_smith_waterman_saturating_cells_at_their_edges(sw, orc, scope, gaps):
    """Round 6's Smith-Waterman cells floor at zero by UNSIGNED SATURATING subtraction (`v_sub_u32 ... clamp`) and, for Gotoh, take their
    plain maxima as `v_max_u16` when no score of the batch can reach 2^16 (largest cost x the shorter side's longest string). The edges:
    free gaps, gaps far larger than any score, and scores on both sides of 2^16 -- identical strings of 2000 symbols under diagonals of 30
    (60 000: the 16-bit maxima) and of 40 (80 000: the 32-bit ones) --, on the column-profile kernel (pairs of more than 384 columns) and on the
    lane kernels (word-sized and 100-symbol strings), pairwise and as a cross-product."""
    rng = np.random.default_rng(77 + abs(gaps[0]) + abs(gaps[1]))
    for diagonal in (30, 40):
        matrix = rng.integers(-12, 6, (256, 256)).astype(np.int8)
        matrix = np.minimum(matrix, matrix.T)
        classes = 12
        fold = (np.arange(256) % classes).astype(np.uint8)                      # 12 symbol classes
        table = matrix[:classes, :classes].copy()
        np.fill_diagonal(table, diagonal)
        full = table[fold][:, fold].astype(np.int8)
        engine = sw.SmithWatermanScores(substitution_matrix=full, open=gaps[0], extend=gaps[1], capabilities=scope)
        items_a, items_b = [], []
        same = rng.integers(0, 256, 2000, dtype=np.uint8).tobytes()
        items_a.append(same); items_b.append(same)                               # score = 2000 x diagonal
        for n in (385, 700, 1300, 2100):
            x = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
            y = bytearray(x)
            for at in rng.integers(0, n, n // 9):
                y[int(at)] = int(rng.integers(0, 256))
            del y[n // 3:n // 3 + 17]
            items_a.append(bytes(x)); items_b.append(bytes(y))
            items_a.append(bytes(y)); items_b.append(bytes(rng.integers(0, 256, n + 5, dtype=np.uint8).tobytes()))
        a, b = sw.Strs(items_a), sw.Strs(items_b)
        got = engine.pairs(a, b, scope)
        want = [orc.nw_score(items_a[i], items_b[i], full, gaps[0], gaps[1], local=True) for i in range(len(items_a))]
        assert [int(x) for x in got] == want, (gaps, diagonal)
        assert int(got[0]) == 2000 * diagonal
        assert (engine.pairs(sw.PreparedTape(scope, a), sw.PreparedTape(scope, b), scope) == got).all()
    if gaps[0] == gaps[1] or gaps == (-3, -1):
        # the lane kernels (alignshort.hip): word-sized pairs and a small-alphabet cross-product of 100-symbol strings
        byte_to_class, class_costs = sw.unary_class_costs(2, -1)
        table = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)
        engine = sw.SmithWatermanScores(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=scope)
        words_a = [rng.integers(97, 101, int(rng.integers(0, 17)), dtype=np.uint8).tobytes() for _ in range(3000)]
        words_b = [w if i % 3 == 0 else rng.integers(97, 101, int(rng.integers(0, 17)), dtype=np.uint8).tobytes() for i, w in enumerate(words_a)]
        wa, wb = sw.Strs(words_a), sw.Strs(words_b)
        got = engine.pairs(sw.PreparedTape(scope, wa), sw.PreparedTape(scope, wb), scope)
        want = [orc.nw_score(words_a[i], words_b[i], table, gaps[0], gaps[1], local=True) for i in range(len(words_a))]
        assert [int(x) for x in got] == want, gaps
        dna = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(60, 129)))].tobytes() for _ in range(90)]
        q, c = sw.Strs(dna[:20]), sw.Strs(dna[20:])
        grid = engine(sw.PreparedTape(scope, q), sw.PreparedTape(scope, c), scope)
        for i in (0, 7, 19):
            assert [int(x) for x in grid[i]] == [orc.nw_score(dna[i], dna[20 + j], table, gaps[0], gaps[1], local=True) for j in range(70)], (gaps, i)
