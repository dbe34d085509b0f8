"""Diagnostic: per-phase wave cycles of k_bitparallel on a workload (library built with EXTRA=-DSWH_BP_PROFILE)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringwars_amd as sw  # noqa: E402
from stringwars_amd import _native as N  # noqa: E402

workload, pairs = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("tokens64", 1_000_000)
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.generate_pairs(workload, pairs, seed=42)
da, db = a.to_device(scope), b.to_device(scope)
out_ptr, err = C.c_void_p(), C.c_char_p()
N.check(N.lib.swh_device_alloc(scope.handle, pairs * 4 + 16, C.byref(out_ptr), C.byref(err)), err)
engine = sw.LevenshteinDistances(capabilities=scope)
fn = N.lib.swh_debug_bp_phases
fn.argtypes = [C.c_void_p]
fn.restype = None
buf = (C.c_ulonglong * 10)()
for _ in range(3):
    engine.pairs(da, db, scope, out=int(out_ptr.value))
fn(buf)
engine.pairs(da, db, scope, out=int(out_ptr.value))
fn(buf)
waves, items = buf[5], max(buf[6], 1)
names = ["locate+extents", "strings+build", "steps", "result+clear"]
print(f"waves {waves} items {items} lifetime/wave {buf[4] / waves:.0f} | " + " | ".join(f"{names[k]} {buf[k] / items:.0f}/item" for k in range(4)))
