"""Measuring tool: the kernels of one C3 call on raw UTF-8 device tapes, with their event times (STRINGWARS_AMD_STAMPS=1)."""
import os, sys
os.environ["STRINGWARS_AMD_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stringwars_amd as sw
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
scope = sw.DeviceScope(gpu_device=0)
a, b = sw.generate_pairs("utf8_lines", pairs, seed=42)
da, db = a.to_device(scope), b.to_device(scope)
engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
for _ in range(20):
    engine.pairs(da, db, scope, bound=32)
scope.set_profiling(True)
for _ in range(3):
    print("--- call", file=sys.stderr)
    engine.pairs(da, db, scope, bound=32)
    scope.last_timing()
