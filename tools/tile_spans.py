"""Diagnostic: when the workgroups of k_bitparallel_tiled start and finish (library built with EXTRA=-DSWH_TILE_PROFILE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
from stringwars_amd import _native as N
scope = sw.DeviceScope(gpu_device=0)
workload, pairs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("tokens64", 1_000_000)
a, b = sw.generate_pairs(workload, pairs, seed=42)
pa, pb = sw.PreparedTape(scope, a.with_offsets(np.uint32)), sw.PreparedTape(scope, b.with_offsets(np.uint32))
engine = sw.LevenshteinDistances(capabilities=scope, algorithm="tiled")
out = C.c_void_p(); err = C.c_char_p()
N.lib.swh_device_alloc(scope.handle, pairs * 4, C.byref(out), C.byref(err))
buf = np.zeros((2048, 4), np.uint64)
N.lib.swh_debug_tile_spans.argtypes = [C.c_void_p]
for _ in range(3):
    engine.pairs(pa, pb, scope, out=int(out.value))
    N.lib.swh_debug_tile_spans(buf.ctypes.data)
    used = buf[buf[:, 3] > 0].astype(np.int64)
    t0 = used[:, 0].min()
    us = (used - t0) / 100.0
    print(f"workgroups {len(used)} | start: p50 {np.median(us[:,0]):.1f} max {us[:,0].max():.1f} us | planned at: p50 {np.median(us[:,1]):.1f} max {us[:,1].max():.1f} | items done: p10 {np.percentile(us[:,2],10):.1f} p50 {np.median(us[:,2]):.1f} p90 {np.percentile(us[:,2],90):.1f} max {us[:,2].max():.1f} | end max {us[:,3].max():.1f}")
    print("  finish-time histogram (10 us bins):", np.histogram(us[:, 2], bins=np.arange(0, us[:,2].max() + 10, 10))[0].tolist())
