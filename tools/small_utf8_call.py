"""Measuring tool: what a SMALL synchronous call costs on raw UTF-8 tapes against byte tapes (10 K word pairs, pairwise; and a
256 x 256 cross-product) -- the fixed cost of the UTF-8 staging launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.generate_pairs("words16", 10_000, seed=42)
da, db = a.to_device(scope), b.to_device(scope)
def rate(fn, seconds=0.5):
    for _ in range(50): fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
    return (time.perf_counter() - t0) / n * 1e6
bytes_engine = sw.LevenshteinDistances(capabilities=scope)
chars_engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
print("pairs 10 K words: bytes %.1f us, utf8 raw %.1f us" % (rate(lambda: bytes_engine.pairs(da, db, scope)), rate(lambda: chars_engine.pairs(da, db, scope))))
qa, qb = a.subview(0, 256).to_device(scope), b.subview(0, 256).to_device(scope)
print("cross 256 x 256 words: bytes %.1f us, utf8 raw %.1f us" % (rate(lambda: bytes_engine(qa, qb, scope)), rate(lambda: chars_engine(qa, qb, scope))))
pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)
print("pairs 10 K words, utf8 prepared: %.1f us" % rate(lambda: chars_engine.pairs(pa, pb, scope)))
if os.environ.get("STRINGWARS_AMD_STAMPS"):
    scope.set_profiling(True)
    chars_engine.pairs(da, db, scope); scope.last_timing()
