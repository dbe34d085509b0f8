import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def lap(msg):
    global t0
    print(f"{msg}: {time.time() - t0:.2f}s", flush=True); t0 = time.time()
import numpy as np
import stringwars_amd as sw
lap("import")
a, b = sw.generate_pairs("tokens64", 60_000, seed=21)
lap("generate")
for devices in ([0], [0, 0]):
    scope = sw.DeviceScope(gpu_devices=devices); lap(f"scope {devices}")
    engine = sw.LevenshteinDistances(capabilities=scope); lap("engine")
    batch = sw.ShardedPairs(scope, a, b); lap("sharded prepare")
    r = engine.pairs_sharded(batch, scope); lap("pairs_sharded 1")
    r = engine.pairs_sharded(batch, scope); lap("pairs_sharded 2")
    batch.free(); lap("batch free")
    del engine; scope.close(); lap("scope free")
