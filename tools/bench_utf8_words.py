"""Measuring tool: unbounded Levenshtein over code points on token-sized strings (20-120 symbols, 11 scripts): the
code-point flavour of the tiled kernel (k_bitparallel_tiled<u32>), which no BASELINE config reaches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw, oracle
rng = np.random.default_rng(5)
scripts = [0x41, 0x62, 0xE9, 0x416, 0x434, 0x4E2D, 0x6587, 0x1F600, 0x20AC, 0x7FF, 0x800]
if "--low" in sys.argv:     # Latin and Cyrillic only: every symbol below U+0800 (the tables' first four 3-bit groups)
    scripts = [0x41, 0x62, 0x65, 0x74, 0xE9, 0xFC, 0x416, 0x434, 0x43E, 0x442, 0x451]
if "--bmp" in sys.argv:     # no astral symbols: six groups
    scripts = [0x41, 0x62, 0xE9, 0x416, 0x434, 0x4E2D, 0x6587, 0x20AC, 0x7FF, 0x800, 0x3042]
def word(n): return "".join(chr(scripts[i]) for i in rng.integers(0, len(scripts), n))
A, B = [], []
for _ in range(60000):
    a = word(int(rng.integers(20, 120)))
    if rng.random() < 0.5:
        b = list(a)
        for _ in range(int(rng.integers(0, 8))):
            pos = int(rng.integers(0, len(b))); b[pos] = chr(scripts[int(rng.integers(0, len(scripts)))])
        b = "".join(b)
    else:
        b = word(int(rng.integers(20, 120)))
    A.append(a); B.append(b)
A = A * 5; B = B * 5
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.Strs(A), sw.Strs(B)
pa, pb = sw.PreparedTape(scope, a, utf8=True), sw.PreparedTape(scope, b, utf8=True)
eng = sw.LevenshteinDistancesUTF8(capabilities=scope)
got = eng.pairs(pa, pb, scope)
want = oracle.levenshtein_pairs(sw.Strs(A[:20000]), sw.Strs(B[:20000]), utf8=True)
assert (got[:20000] == want).all()
scope.set_profiling(True)
import time
until = time.perf_counter() + 0.3   # clocks
while time.perf_counter() < until:
    eng.pairs(pa, pb, scope)
best = 1e9
for _ in range(7):
    eng.pairs(pa, pb, scope); t = scope.last_timing(); best = min(best, t["compute_ms"])
print("scripts", "low" if "--low" in sys.argv else ("bmp" if "--bmp" in sys.argv else "all"), "pairs", len(A), "kernel ms", round(best, 4), t["dominant_name"], "TCUPS", round(t["cells"] / best / 1e9, 1))
