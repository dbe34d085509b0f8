"""Measuring tool: synchronous raw-UTF-8 calls of 4 ... 63 MB of tapes (k = 32): where one staging launch pair for both tapes
stops paying against a launch pair and a stream per tape (STRINGWARS_AMD_UTF8_MERGED_MB)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("STRINGWARS_AMD_LIBRARY", os.path.join(os.getcwd(), "stringwars_amd", "libstringwars_amd_test.so"))   # (the switch is a test hook since round 6)
import stringwars_amd as sw
scope = sw.DeviceScope(gpu_device=0)
for pairs in (2000, 8000, 16000, 32000):
    a, b = sw.generate_pairs("utf8_lines", pairs, seed=42)
    da, db = a.to_device(scope), b.to_device(scope)
    engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
    for _ in range(20): engine.pairs(da, db, scope, bound=32)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        engine.pairs(da, db, scope, bound=32); n += 1
    print(pairs, "pairs", round((a.data.nbytes + b.data.nbytes) / 2**20, 1), "MB:", round((time.perf_counter() - t0) / n * 1e6, 1), "us per call", flush=True)
