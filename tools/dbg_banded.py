import sys; sys.path.insert(0, '/root/repo')
import numpy as np, stringwars_amd as sw, oracle
scope = sw.DeviceScope(gpu_device=0)
eng = sw.LevenshteinDistances(capabilities=scope)
engu = sw.LevenshteinDistancesUTF8(capabilities=scope)
rng = np.random.default_rng(0)
for n in (30, 64, 65, 100, 200, 400):
    a = bytes(rng.integers(97, 123, n, dtype=np.uint8))
    b = bytearray(a); b[n//2] = 65
    for k in (35, 36, 40, 63):
        r = eng.pairs([a]*3, [bytes(b)]*3, scope, bound=k)
        ru = engu.pairs([a]*3, [bytes(b)]*3, scope, bound=k)
        print(n, k, r.tolist(), ru.tolist(), oracle.levenshtein(a, bytes(b)))
