"""Diagnostic: per-phase wave-cycle totals of k_bitparallel_mw<W> on workload C2.

Needs a library built with `make -C stringwars_amd/csrc EXTRA=-DSWH_MW_PROFILE` (never the shipped build).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringwars_amd as sw  # noqa: E402
from stringwars_amd import _native as N  # noqa: E402

pairs = 1_000_000
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.generate_pairs("tokens64", pairs, seed=42)
da, db = a.to_device(scope), b.to_device(scope)
out_ptr, err = C.c_void_p(), C.c_char_p()
N.check(N.lib.swh_device_alloc(scope.handle, pairs * 4 + 16, C.byref(out_ptr), C.byref(err)), err)
engine = sw.LevenshteinDistances(capabilities=scope)
for _ in range(3):
    engine.pairs(da, db, scope, out=int(out_ptr.value))
buf = (C.c_ulonglong * 40)()
fn = N.lib.swh_debug_mw_phases
fn.argtypes = [C.c_void_p]
fn.restype = None
fn(buf)
engine.pairs(da, db, scope, out=int(out_ptr.value))
fn(buf)
names = ["prologue", "strings", "build", "steps", "store+clear", "lifetime"]
for w in range(1, 5):
    row = [buf[w * 8 + k] for k in range(8)]
    waves, items = row[6], row[7]
    if not waves:
        continue
    print(f"mw<{w}>: waves {waves} items {items} | " + " | ".join(
        f"{names[k]} {row[k] / waves:.0f}/wave" + (f" ({row[k] / max(items, 1):.0f}/item)" if 1 <= k <= 4 else "")
        for k in range(6)))
