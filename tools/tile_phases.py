"""Diagnostic: wave cycles per phase of k_bitparallel_tiled (library built with EXTRA=-DSWH_TILE_PROFILE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw
from stringwars_amd import _native as N
scope = sw.DeviceScope(gpu_device=0)
workload, pairs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("tokens64", 1_000_000)
a, b = sw.generate_pairs(workload, pairs, seed=42)
pa, pb = sw.PreparedTape(scope, a), sw.PreparedTape(scope, b)
engine = sw.LevenshteinDistances(capabilities=scope, algorithm="tiled")
out = C.c_void_p(); err = C.c_char_p()
N.lib.swh_device_alloc(scope.handle, pairs * 4, C.byref(out), C.byref(err))
buf = (C.c_ulonglong * 8)()
N.lib.swh_debug_tile_phases.argtypes = [C.c_void_p]
engine.pairs(pa, pb, scope, out=int(out.value))
N.lib.swh_debug_tile_phases(buf)
scope.set_profiling(True)
for _ in range(3):
    engine.pairs(pa, pb, scope, out=int(out.value))
    t = scope.last_timing()
    N.lib.swh_debug_tile_phases(buf)
    plan, items, wait, waves, nitems = [int(x) for x in buf[:5]]
    total = plan + items + wait
    print(f"kernel {t['compute_ms']*1e3:.1f} us | waves {waves} items {nitems} ({nitems/waves:.2f}/wave) | per wave cycles: plan {plan/waves:.0f} items {items/waves:.0f} wait {wait/waves:.0f} "
          f"| shares plan {plan/total:.3f} items {items/total:.3f} wait {wait/total:.3f} | cycles/item {items/max(nitems,1):.0f}")
