"""Measuring tool: the stamped launches of one NW cross-product call on a bench_cross kind (STRINGWARS_AMD_STAMPS=1). usage: stamps_cross_nw.py [kind]"""
import os, sys
os.environ["STRINGWARS_AMD_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import stringwars_amd as sw
from bench_cross import tokens
kind = sys.argv[1] if len(sys.argv) > 1 else "twords"
scope = sw.DeviceScope(gpu_device=0)
rng = np.random.default_rng(42)
side = 2048
tape = tokens(kind, 2 * side, rng)
q, c = tape.subview(0, side).to_device(scope), tape.subview(side, 2 * side).to_device(scope)
import torch
out = torch.zeros((side, side), dtype=torch.int64, device="cuda")
classes, costs = sw.unary_class_costs(2, -1)
engine = sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope)
for _ in range(5):
    engine(q, c, scope, out=out)
scope.set_profiling(True)
print("--- call", file=sys.stderr)
engine(q, c, scope, out=out)
print(scope.last_timing(), file=sys.stderr)
