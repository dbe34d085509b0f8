#!/usr/bin/env python3
"""C5 probe: parity of k_short_tiled against the oracle at sizes that give a workgroup one, two and many chunks, then the
time of a synchronous call on 20 M pairs (run under `timeout`: a variant of the kernel that hangs must not cost the box)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw, oracle
scope = sw.DeviceScope(gpu_device=0)
engine = sw.LevenshteinDistances(capabilities=scope)
for pairs in (150_000, 1_100_000, 3_000_000):
    a, b = sw.generate_pairs("short_words", pairs, seed=7)
    got = engine.pairs(sw.PreparedTape(scope, a), sw.PreparedTape(scope, b), scope)
    want = oracle.levenshtein_pairs(a, b, algo="hyyro", count=min(pairs, 300_000))
    print(pairs, "ok" if (got[:len(want)] == want).all() else "MISMATCH", flush=True)
a, b = sw.generate_pairs("short_words", 20_000_000, seed=42)
a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
out_ptr = None
import ctypes as C
from stringwars_amd import _native as N
ptr, err = C.c_void_p(), C.c_char_p()
N.check(N.lib.swh_device_alloc(scope.handle, 4 * 20_000_000 + 64, C.byref(ptr), C.byref(err)), err)
call = lambda: engine.pairs(pa, pb, scope, out=int(ptr.value))
until = time.perf_counter() + 0.5
while time.perf_counter() < until:
    call()
scope.set_profiling(True)
best = 1e9
for _ in range(20):
    call(); best = min(best, scope.last_timing()["compute_ms"])
scope.set_profiling(False)
t0 = time.perf_counter()
for _ in range(50):
    call()
print("20M pairs: kernel_ms best", round(best, 4), "call ms", round((time.perf_counter() - t0) / 50 * 1e3, 4), flush=True)
