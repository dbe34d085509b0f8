"""Measuring tool: the stamped launches of one UTF-8 cross-product call (STRINGWARS_AMD_STAMPS=1). usage: stamps_cross_utf8.py [kind]"""
import os, sys
os.environ["STRINGWARS_AMD_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_cross import tokens
kind = sys.argv[1] if len(sys.argv) > 1 else "words"
scope = sw.DeviceScope(gpu_device=0)
rng = np.random.default_rng(42)
side = 2048
tape = tokens(kind, 2 * side, rng)
q, c = tape.subview(0, side).to_device(scope), tape.subview(side, 2 * side).to_device(scope)
import torch
out = torch.zeros((side, side), dtype=torch.int64, device="cuda")
engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
for _ in range(10):
    engine(q, c, scope, out=out)
scope.set_profiling(True)
for _ in range(2):
    print("--- call", file=sys.stderr)
    engine(q, c, scope, out=out)
    print(scope.last_timing(), file=sys.stderr)
