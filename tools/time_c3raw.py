"""Measuring tool: synchronous C3 calls on raw UTF-8 device tapes, host-timed (no event stamps). usage: time_c3raw.py [pairs] [bound]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
bound = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.generate_pairs("utf8_lines", pairs, seed=42)
da, db = a.to_device(scope), b.to_device(scope)
engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    engine.pairs(da, db, scope, bound=bound)
times = []
for _ in range(300):
    t0 = time.perf_counter(); engine.pairs(da, db, scope, bound=bound); times.append(time.perf_counter() - t0)
times.sort()
print(f"pairs {pairs} bound {bound}: median {times[len(times)//2]*1e3:.4f} ms  best {times[0]*1e3:.4f} ms  p90 {times[int(len(times)*0.9)]*1e3:.4f} ms")
