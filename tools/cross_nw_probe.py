#!/usr/bin/env python3
"""Which kernel a 2048 x 2048 cross-product of ACGT-100 strings takes for NW linear on a fresh scope, and its time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw
scope = sw.DeviceScope(gpu_device=0)
rng = np.random.default_rng(42)
side, length = 2048, 100
offsets = np.arange(2 * side + 1, dtype=np.uint64) * length
tape = sw.Strs(data=np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2 * side * length)], offsets=offsets)
q, c = tape.subview(0, side).to_device(scope), tape.subview(side, 2 * side).to_device(scope)
classes, costs = sw.unary_class_costs(2, -1)
import ctypes as C
from stringwars_amd import _native as N
ptr, err = C.c_void_p(), C.c_char_p()
N.check(N.lib.swh_device_alloc(scope.handle, side * side * 8, C.byref(ptr), C.byref(err)), err)
for name, engine in (("nw_linear", sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope)),
                     ("sw_linear", sw.SmithWatermanScores(classes, costs, open=-2, extend=-2, capabilities=scope))):
    call = lambda: engine(q, c, scope, out=int(ptr.value))
    until = time.perf_counter() + 0.3
    while time.perf_counter() < until:
        call()
    scope.set_profiling(True)
    call()
    timing = scope.last_timing()
    scope.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    print(name, timing["dominant_name"], "kernel_ms", round(timing["compute_ms"], 3), "call_ms", round((time.perf_counter() - t0) * 100, 3), scope.describe(), flush=True)
