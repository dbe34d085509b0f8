import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw, oracle
scope = sw.DeviceScope(gpu_device=0)
for utf8 in (False, True):
    for algo in ("bitparallel", "tiled"):
        cls = sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances
        eng = cls(capabilities=scope, algorithm=algo)
        a = sw.Strs(["kitten", "intention", "flaw", "abcdefghijklmnopqrstuvwxyzabcdefghijklmnop"])
        b = sw.Strs(["sitting", "execution", "lawn", "abcdefghijklmnopqrstuvwxyzabcdefghijklmnopq"])
        print(utf8, algo, eng.pairs(a, b, scope).tolist(), oracle.levenshtein_pairs(a, b, utf8=utf8).tolist())
for wl, n in (("tokens64", 3000), ("words16", 3000), ("utf8_lines", 100)):
    a, b = sw.generate_pairs(wl, n, seed=1)
    for utf8 in (False, True):
        cls = sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances
        got = cls(capabilities=scope, algorithm="tiled").pairs(a, b, scope)
        want = oracle.levenshtein_pairs(a, b, utf8=utf8)
        bad = np.nonzero(got != want)[0]
        print(wl, utf8, "bad", bad.size, bad[:10], got[bad[:5]], want[bad[:5]])
