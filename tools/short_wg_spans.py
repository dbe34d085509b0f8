"""Diagnostic: when the workgroups of k_short_tiled start and finish (library built with EXTRA=-DSWH_SHORT_WG_SPANS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
from stringwars_amd import _native as N
scope = sw.DeviceScope(gpu_device=0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
a, b = sw.generate_pairs("short_words", pairs, seed=42)
pa, pb = sw.PreparedTape(scope, a.with_offsets(np.uint32)), sw.PreparedTape(scope, b.with_offsets(np.uint32))
engine = sw.LevenshteinDistances(capabilities=scope)
out = C.c_void_p(); err = C.c_char_p()
N.lib.swh_device_alloc(scope.handle, pairs * 4, C.byref(out), C.byref(err))
buf = np.zeros((4096, 2), np.uint64)
N.lib.swh_debug_short_wg.argtypes = [C.c_void_p]
for _ in range(3):
    for _ in range(5):
        engine.pairs(pa, pb, scope, out=int(out.value))
    N.lib.swh_debug_short_wg(buf.ctypes.data)
    used = buf[buf[:, 1] > 0].astype(np.int64)
    t0 = used[:, 0].min()
    us = (used - t0) / 100.0
    print(f"workgroups {len(used)} | start p50 {np.median(us[:,0]):.1f} max {us[:,0].max():.1f} us | end p10 {np.percentile(us[:,1],10):.1f} p50 {np.median(us[:,1]):.1f} p90 {np.percentile(us[:,1],90):.1f} max {us[:,1].max():.1f}")
    print("  finish-time histogram (10 us bins):", np.histogram(us[:, 1], bins=np.arange(0, us[:,1].max() + 10, 10))[0].tolist())
