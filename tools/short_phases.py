"""Diagnostic: wall-clock time per step of k_short_tiled for the first chunks of every 16th workgroup
(library built with EXTRA=-DSWH_SHORT_PROFILE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
from stringwars_amd import _native as N
scope = sw.DeviceScope(gpu_device=0)
workload, pairs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("short_words", 20_000_000)
a, b = sw.generate_pairs(workload, pairs, seed=42)
pa, pb = sw.PreparedTape(scope, a.with_offsets(np.uint32)), sw.PreparedTape(scope, b.with_offsets(np.uint32))
engine = sw.LevenshteinDistances(capabilities=scope)
out = C.c_void_p(); err = C.c_char_p()
N.lib.swh_device_alloc(scope.handle, pairs * 4, C.byref(out), C.byref(err))
spans = np.zeros((64, 8, 8), np.uint64)
N.lib.swh_debug_short_spans.argtypes = [C.c_void_p]
scope.set_profiling(True)
for _ in range(3):
    engine.pairs(pa, pb, scope, out=int(out.value))
    t = scope.last_timing()
    N.lib.swh_debug_short_spans(spans.ctypes.data)
    s = spans.astype(np.int64)[:, 1:7, :]          # chunks 1..6 (steady state)
    # stamps: 7 top, 0 after the copy, 1 after barrier 1, 2 after B, 3 after C+D, 4 after E, 5 after barrier 5, 6 after F
    order = [7, 0, 1, 2, 3, 4, 5, 6]
    names = ["A copy", "barrier 1", "B affixes+count", "C+D scan, scatter", "E items (+ requests)", "barrier 5", "F results"]
    d = np.stack([s[:, :, order[i + 1]] - s[:, :, order[i]] for i in range(7)], -1) / 100.0   # us
    whole = (s[:, :, 6] - s[:, :, 7]) / 100.0
    print(f"kernel {t['compute_ms']*1e3:.1f} us | chunk {np.median(whole):.2f} us (p10 {np.percentile(whole,10):.2f}, p90 {np.percentile(whole,90):.2f}) | " +
          " | ".join(f"{n} {np.median(d[..., i]):.2f}" for i, n in enumerate(names)))
