"""Measuring tool: the staging kernels of ONE ~100 MB UTF-8 tape alone (a call of a tape against itself stages it once)."""
import os, sys
os.environ["STRINGWARS_AMD_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw
scope = sw.DeviceScope(gpu_device=0)
a, _ = sw.generate_pairs("utf8_lines", 100_000, seed=42)
da = a.to_device(scope)
engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
for _ in range(10):
    engine.pairs(da, da, scope, bound=32)
scope.set_profiling(True)
for _ in range(2):
    print("--- call", file=sys.stderr)
    engine.pairs(da, da, scope, bound=32)
    scope.last_timing()
