"""Diagnostic: wave cycles per phase of k_short_tiled (library built with EXTRA=-DSWH_SHORT_PROFILE)."""
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
buf = (C.c_ulonglong * 10)()
N.lib.swh_debug_short_phases.argtypes = [C.c_void_p]
engine.pairs(pa, pb, scope, out=int(out.value))
N.lib.swh_debug_short_phases(buf)
scope.set_profiling(True)
names = ["A requests+copy", "A barrier", "B affixes+count", "C+D scan+scatter", "E items", "E wait", "F results"]
for _ in range(3):
    engine.pairs(pa, pb, scope, out=int(out.value))
    t = scope.last_timing()
    N.lib.swh_debug_short_phases(buf)
    vals = [int(x) for x in buf[:9]]
    waves, items = vals[7], vals[8]
    total = sum(vals[:7])
    print(f"kernel {t['compute_ms']*1e3:.1f} us | waves {waves} items {items} | cycles/wave {total/waves:.0f} | cycles/item {vals[4]/max(items,1):.0f} | " +
          " | ".join(f"{n} {v/total:.3f}" for n, v in zip(names, vals[:7])))
