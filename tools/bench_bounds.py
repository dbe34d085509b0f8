#!/usr/bin/env python3
"""Measuring tool: config C3's lines (100 K pairs of ~1 KB UTF-8, tapes prepared) at bounds beyond one 64-bit window -- the banded
kernel's windows of two to four words against the unbounded kernels + the clamp (`STRINGWARS_AMD_BAND_MAX=63`), per synchronous call.
    python tools/bench_bounds.py [--bytes]        # --bytes: the same tapes as byte strings"""
import os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BOUNDS = (32, 63, 64, 100, 127, 128, 255)
if os.environ.get("BENCH_BOUNDS_CHILD") != "1":
    for knob in ("255", "63"):
        # (A / B switches are test hooks since round 6: the TEST library reads them, the shipped one does not)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, BENCH_BOUNDS_CHILD="1", STRINGWARS_AMD_BAND_MAX=knob, STRINGWARS_AMD_LIBRARY=os.path.join(root, "stringwars_amd", "libstringwars_amd_test.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=True)
    raise SystemExit(0)
import numpy as np
import stringwars_amd as sw
scope = sw.DeviceScope(gpu_device=0)
as_bytes = "--bytes" in sys.argv
a, b = sw.generate_pairs("utf8_lines", 100_000, seed=42)
pa, pb = sw.PreparedTape(scope, a, utf8=not as_bytes), sw.PreparedTape(scope, b, utf8=not as_bytes)
engine = (sw.LevenshteinDistances if as_bytes else sw.LevenshteinDistancesUTF8)(capabilities=scope)
import torch
out = torch.zeros(100_000, dtype=torch.int32, device="cuda")
for k in BOUNDS + (None,):
    call = engine.bind_pairs(pa, pb, scope, out, bound=k)
    call()
    until = time.perf_counter() + 0.3
    while time.perf_counter() < until:
        call()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
    scope.set_profiling(True); call(); timing = scope.last_timing(); scope.set_profiling(False)
    print(json.dumps({"band_max": os.environ["STRINGWARS_AMD_BAND_MAX"], "symbols": "bytes" if as_bytes else "code points", "bound": k,
                      "tcups": round(timing["cells"] / best / 1e12, 2), "call_ms": round(best * 1e3, 3), "kernel": timing["dominant_name"],
                      "kernel_ms": round(timing["compute_ms"], 3)}), flush=True)
