import sys, os
sys.path.insert(0, os.getcwd())
import stringwars_amd as sw, oracle
scope = sw.DeviceScope(gpu_device=0)
a = sw.Strs(["héllo wörld", "abc", "中文"]); b = sw.Strs(["hello world", "abd", "中"])
eng = sw.LevenshteinDistancesUTF8(capabilities=scope)
print("call 1", flush=True)
print(eng.pairs(a, b, scope), flush=True)
print("call 2", flush=True)
print(eng.pairs(a, b, scope), flush=True)
big_a, big_b = sw.generate_pairs("utf8_lines", 300, seed=3)
print("call 3", flush=True)
got = eng.pairs(big_a, big_b, scope)
print((got == oracle.levenshtein_pairs(big_a, big_b, utf8=True)).all(), flush=True)
