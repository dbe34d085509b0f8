#!/usr/bin/env python3
"""Randomized parity soak on the GPU: batches with mixed length regimes, alphabets, bounds and engines, every result
compared with the CPU oracle bit for bit. Lives under tests/ (the oracle is test infrastructure) but is not collected
by pytest: minutes, not seconds.

    python tests/soak.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the two-stage (doubling) schedule is taken from 200 000 pair-blocks on; the soak's batches are smaller, so it asks for it always
# (read once per process by the library: set before the import)
os.environ.setdefault("STRINGWARS_AMD_DOUBLING_MIN", "1")
import stringwars_amd as sw  # noqa: E402
import oracle  # noqa: E402

REGIMES = [(0, 8), (0, 40), (20, 140), (100, 700), (500, 2100), (1900, 2300), (2000, 5000), (0, 5000)]
ALPHABETS = {"acgt": [ord(c) for c in "ACGT"], "lower": list(range(97, 123)), "byte": list(range(256)),
             "binary": [48, 49], "ascii": list(range(32, 127))}
SCRIPTS = [0x41, 0x62, 0xE9, 0x416, 0x434, 0x4E2D, 0x6587, 0x1F600, 0x20AC, 0x7F, 0x80, 0x7FF, 0x800, 0xFFFF, 0x10000]
# code points by how many distinct ones a pattern sees (bp_dense.hpp: per-pair dictionaries of 251 slots): a handful; around what a dictionary
# takes -- a sketch or a probe budget decides, pattern by pattern --; far beyond it (the group tables)
CODE_POINT_POOLS = [SCRIPTS, SCRIPTS + list(range(0x430, 0x450)) + list(range(0x20, 0x7F)) + list(range(0x3041, 0x3097)),
                    SCRIPTS + list(range(0x4E00, 0x4E00 + 900)) + list(range(0x1F300, 0x1F340))]


def random_batch(rng, utf8):
    regime = REGIMES[int(rng.integers(0, len(REGIMES)))]
    mixed = rng.random() < 0.3
    budget = 1.5e9
    alphabet = np.array(CODE_POINT_POOLS[int(rng.integers(0, 3))] if utf8 else ALPHABETS[str(rng.choice(list(ALPHABETS)))], np.uint32)
    items_a, items_b, cells = [], [], 0
    while cells < budget and len(items_a) < 20000:
        lo, hi = REGIMES[int(rng.integers(0, len(REGIMES)))] if mixed else regime
        la, lb = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
        a = alphabet[rng.integers(0, len(alphabet), la)]
        if rng.random() < 0.6 and la:
            b = list(a)
            for _ in range(int(rng.integers(0, max(2, la // 8)))):
                op, pos = int(rng.integers(0, 3)), int(rng.integers(0, max(len(b), 1)))
                if op == 0 and b:
                    b[pos] = int(alphabet[rng.integers(0, len(alphabet))])
                elif op == 1:
                    b.insert(pos, int(alphabet[rng.integers(0, len(alphabet))]))
                elif len(b) > 1:
                    del b[pos]
            b = np.array(b, np.uint32)
        else:
            b = alphabet[rng.integers(0, len(alphabet), lb)]
        enc = (lambda x: "".join(map(chr, x)).encode("utf-8")) if utf8 else (lambda x: bytes(x.astype(np.uint8)))
        items_a.append(enc(a))
        items_b.append(enc(b))
        cells += max(len(a), 1) * max(len(b), 1)
    return sw.Strs(items_a), sw.Strs(items_b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true", help="one line per batch before it runs (to find the batch behind a device fault)")
    ap.add_argument("--only", type=int, default=-1, help="draw every batch as usual but run only this one on the device (reproduces one batch of a seed)")
    ap.add_argument("--first", type=int, default=-1, help="with --only N: run batches first .. N on the device (finds the shortest prefix a failure needs)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    scope = sw.DeviceScope(gpu_device=0)
    piped = sw.DeviceScope(gpu_device=0)
    piped.set_async(True)
    piped.set_pipelined(True)
    multi = sw.DeviceScope(gpu_devices=[0, 0, 0])
    t0, rounds, pairs_total = time.time(), 0, 0
    while time.time() - t0 < args.seconds or 0 <= rounds <= args.only:
        kind = str(rng.choice(["lev", "lev", "lev_utf8", "nw", "sw"]))
        if kind in ("lev", "lev_utf8"):
            utf8 = kind == "lev_utf8"
            a, b = random_batch(rng, utf8)
            bound = None if rng.random() < 0.5 else int(rng.integers(0, 141))   # (64 .. 127: the banded kernel's two-word window)
            algorithm = str(rng.choice(["auto", "auto", "bitparallel", "wavefront", "tiled"]))
            if args.verbose:
                print(f"batch {rounds}: {kind} pairs {len(a)} longest {int(max(a.lengths.max(), b.lengths.max()))} bound {bound} algorithm {algorithm}", flush=True)
            if algorithm == "wavefront" and oracle.cells(a, b, utf8=utf8) > 3e8:
                algorithm = "auto"
            if args.only >= 0 and not ((args.first if args.first >= 0 else args.only) <= rounds <= args.only):   # (the draws below are replayed so that the sequence stays the same)
                lo = int(rng.integers(0, len(a)))
                hi = int(rng.integers(lo, len(a) + 1))
                if rounds % 5 == 0:
                    for i in range(min(len(a), 40)): rng.integers(0, 33)
                    for i in range(min(len(b), 90)): rng.integers(0, 33)
                rounds += 1
                if rounds > args.only: break
                continue
            cls = sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances
            engine = cls(capabilities=scope, algorithm=algorithm)
            want = oracle.levenshtein_pairs(a, b, utf8=utf8, algo="wf" if utf8 else "hyyro", bound=bound)
            if args.verbose: print("    raw tapes, twice", flush=True)
            for _ in range(2):   # the second call may take the direct-short path
                got = engine.pairs(a, b, scope, bound=bound)
                bad = np.nonzero(got != want)[0]
                assert bad.size == 0, (kind, algorithm, bound, bad[:5], got[bad[:5]], want[bad[:5]], a.lengths[bad[:5]], b.lengths[bad[:5]])
            if args.verbose: print("    prepared tapes + sub-view", flush=True)
            # prepared tapes (resident, measured, UTF-8 decoded once): whole tapes and a random sub-view, same engine
            pa, pb = sw.PreparedTape(scope, a, utf8=utf8), sw.PreparedTape(scope, b, utf8=utf8)
            got = engine.pairs(pa, pb, scope, bound=bound)
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, ("prepared", kind, algorithm, bound, bad[:5], got[bad[:5]], want[bad[:5]])
            lo = int(rng.integers(0, len(a)))
            hi = int(rng.integers(lo, len(a) + 1))
            assert (engine.pairs(pa[lo:hi], pb[lo:hi], scope, bound=bound) == want[lo:hi]).all(), ("prepared sub-view", kind, algorithm, lo, hi)
            if rounds % 3 == 0 and not utf8:   # the same batch split over a three-member scope on device 0
                if args.verbose: print("    sharded", flush=True)
                sharded_engine = sw.LevenshteinDistances(capabilities=multi)
                batch = sw.ShardedPairs(multi, a, b)
                got = sharded_engine.pairs_sharded(batch, multi, bound=bound)
                assert (got == want).all(), ("sharded", bound)
                batch.free()
            if rounds % 2 == 1:   # raw DEVICE tapes, three calls: the second and third believe what the first one learnt about them
                                  # (byte totals; pure ASCII -> the byte kernels behind k_ascii_check), then the tapes change in place
                da, db = a.to_device(scope), b.to_device(scope)
                for _ in range(3):
                    assert (engine.pairs(da, db, scope, bound=bound) == want).all(), ("device tapes", kind, algorithm, bound)
                if utf8 and len(a) > 1:   # words of either script as a cross-product on device tapes (k_cross_short_cp / its byte twin), twice
                    uq = [bytes(x).decode("utf-8", "ignore")[:int(rng.integers(0, 33))].encode() for x in (a[i] for i in range(min(len(a), 30)))]
                    uc = [bytes(x).decode("utf-8", "ignore")[:int(rng.integers(0, 33))].encode() for x in (b[i] for i in range(min(len(b), 70)))]
                    flat = np.array([[oracle.levenshtein_utf8(x, y) for y in uc] for x in uq])
                    dq, dc = sw.Strs(uq).to_device(scope), sw.Strs(uc).to_device(scope)
                    for _ in range(2):
                        assert (engine(dq, dc, scope) == flat).all(), "cross-product of word-sized code points"
                    dq.free(); dc.free()
                da.free(); db.free()
            if rounds % 5 == 0:   # a word-sized cross-product (k_cross_short) against the same oracle, pair by pair
                words_q = [bytes(x) for x in (a[i][:int(rng.integers(0, 33))] for i in range(min(len(a), 40)))]
                words_c = [bytes(x) for x in (b[i][:int(rng.integers(0, 33))] for i in range(min(len(b), 90)))]
                if not utf8:
                    q, c = sw.Strs(words_q), sw.Strs(words_c)
                    flat = oracle.levenshtein_pairs(sw.Strs([x for x in words_q for _ in words_c]), sw.Strs(words_c * len(words_q)), algo="hyyro")
                    byte_engine = sw.LevenshteinDistances(capabilities=scope)
                    for tapes in ((q, c), (sw.PreparedTape(scope, q), sw.PreparedTape(scope, c))):
                        assert (byte_engine(tapes[0], tapes[1], scope).reshape(-1) == flat).all(), "cross-product of words"
            if rounds % 4 == 0:   # the same batch through the pipelined lanes: device tapes, device outputs, 32-bit offsets
                if args.verbose: print("    pipelined lanes", flush=True)
                import torch
                a32, b32 = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
                da, db = a32.to_device(piped), b32.to_device(piped)
                outs = [torch.zeros(len(a), dtype=torch.int32, device="cuda") for _ in range(4)]
                lane_engine = cls(capabilities=piped, algorithm=algorithm)
                for out in outs:
                    lane_engine.pairs(da, db, piped, bound=bound, out=out)
                piped.synchronize()
                for out in outs:
                    got = out.cpu().numpy().astype(np.uint32)
                    bad = np.nonzero(got != want)[0]
                    assert bad.size == 0, ("pipelined", kind, algorithm, bound, bad[:5], got[bad[:5]], want[bad[:5]])
        else:
            classes = int(rng.choice([2, 4, 8, 9, 21, 24, 25, 32, 40, 53, 100, 127, 128, 256]))   # (33 .. 128 classes: the wide class table of the profile kernel)
            alphabet = np.arange(classes if classes < 256 else 256, dtype=np.uint32) + (65 if classes <= 32 else 0)
            matrix = rng.integers(-6, 7, (256, 256)).astype(np.int8)
            if rng.random() < 0.7:
                matrix = np.minimum(matrix, matrix.T)   # symmetric (the kernels may then put the shorter string on the columns)
            if classes < 256:   # bytes outside the alphabet share one class
                other = np.setdiff1d(np.arange(256), alphabet.astype(np.int64))
                matrix[other, :] = matrix[other[0], :][None, :]
                matrix[:, other] = matrix[:, other[0]][:, None]
                matrix[np.ix_(other, other)] = matrix[other[0], other[0]]
            gaps = [(-4, -4), (-11, -1), (-2, -2), (-5, -1), (-1, -1)][int(rng.integers(0, 5))]
            items_a, items_b, cells = [], [], 0
            lo, hi = [(0, 16), (0, 32), (0, 40), (20, 300), (200, 1500), (1000, 5000)][int(rng.integers(0, 6))]   # (<= 32: the lane-per-pair kernel)
            while cells < 1.5e8 and len(items_a) < 4000:
                la, lb = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
                items_a.append(bytes(alphabet[rng.integers(0, len(alphabet), la)].astype(np.uint8)))
                items_b.append(bytes(alphabet[rng.integers(0, len(alphabet), lb)].astype(np.uint8)))
                cells += max(la, 1) * max(lb, 1)
            a, b = sw.Strs(items_a), sw.Strs(items_b)
            if args.verbose:
                print(f"batch {rounds}: {kind} classes {classes} gaps {gaps} pairs {len(a)} lengths {lo}..{hi} symmetric {bool((matrix == matrix.T).all())}", flush=True)
            if args.only >= 0 and not ((args.first if args.first >= 0 else args.only) <= rounds <= args.only):
                rounds += 1
                if rounds > args.only: break
                continue
            cls = sw.NeedlemanWunschScores if kind == "nw" else sw.SmithWatermanScores
            engine = cls(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=scope)
            got = engine.pairs(a, b, scope)
            want = np.array([oracle.nw_score(x, y, matrix, gaps[0], gaps[1], local=(kind == "sw")) for x, y in zip(items_a, items_b)])
            bad = np.nonzero(got != want)[0]
            assert bad.size == 0, (kind, classes, gaps, bad[:5], got[bad[:5]], want[bad[:5]])
            # again on the same scope (word-sized batches now take the plan-free lane-per-pair kernel on the lengths the first call saw), on
            # prepared tapes, and a cross-product of strings cut to <= 384 bytes (small alphabets: the class-compacting kernels)
            assert (engine.pairs(a, b, scope) == want).all(), ("second call", kind, classes, gaps, lo, hi)
            pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
            assert (engine.pairs(pa, pb, scope) == want).all(), ("prepared", kind, classes, gaps, lo, hi)
            if rounds % 2 == 0:   # word tokens with a few long ones among them: the lane kernel + a redo that plans only the pairs it left
                wq = [x[:int(rng.integers(0, 20))] for x in items_a[:40]] + [x[:int(n)] for x, n in zip(items_a[40:43], (70, 150, 65))]
                wc = [x[:int(rng.integers(0, 24))] for x in items_b[:120]] + [x[:int(n)] for x, n in zip(items_b[120:123], (66, 400, 64))]
                wflat = np.array([[oracle.nw_score(x, y, matrix, gaps[0], gaps[1], local=(kind == "sw")) for y in wc] for x in wq])
                pwq, pwc = sw.PreparedTape(scope, sw.Strs(wq)), sw.PreparedTape(scope, sw.Strs(wc))
                for _ in range(2):
                    assert (engine(pwq, pwc, scope) == wflat).all(), ("words with a few long tokens", kind, classes, gaps)
            cut = int(rng.choice([16, 32, 64, 128, 200, 384, 1000, 2048, 4096]))   # (beyond 128: columns in passes, k_align_cross_long)
            few = cut > 384   # (the oracle's share: 7 x 70 strings of up to 4 K symbols are ~10^9 cells)
            qs, cs = [x[:cut] for x in items_a[:7 if few else 23]], [x[:cut] for x in items_b[:70 if few else 150]]
            flat = np.array([[oracle.nw_score(x, y, matrix, gaps[0], gaps[1], local=(kind == "sw")) for y in cs] for x in qs])
            assert (engine(sw.PreparedTape(scope, sw.Strs(qs)), sw.PreparedTape(scope, sw.Strs(cs)), scope) == flat).all(), ("cross-product", kind, classes, gaps, cut)
            if rounds % 3 == 0:   # the same batch over the three-member scope (per-member engine clones), and a small cross-product
                sharded_engine = cls(substitution_matrix=matrix, open=gaps[0], extend=gaps[1], capabilities=multi)
                batch = sw.ShardedPairs(multi, a, b)
                assert (sharded_engine.pairs_sharded(batch, multi) == want).all(), ("sharded", kind, classes, gaps)
                batch.free()
                q, c = sw.Strs(items_a[:7]), sw.Strs(items_b[:5])
                product = sw.ShardedCross(multi, q, c)
                flat = np.array([[oracle.nw_score(x, y, matrix, gaps[0], gaps[1], local=(kind == "sw")) for y in items_b[:5]] for x in items_a[:7]])
                assert (sharded_engine.cross_sharded(product, multi) == flat).all(), ("sharded cross-product", kind, classes, gaps)
                product.free()
        rounds += 1
        pairs_total += len(a)
        if 0 <= args.only < rounds:
            break
    print(f"soak ok: {rounds} batches, {pairs_total} pairs, {time.time() - t0:.0f} s, seed {args.seed}")


if __name__ == "__main__":
    main()
