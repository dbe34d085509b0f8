#!/usr/bin/env python3
"""Probe: does C3's raw call overlap with itself? Two scopes (two streams) on one device, each scoring HALF of the 100 K line pairs on raw
device tapes in a loop from its own host thread, against one scope scoring all of them -- aggregate cells per second either way.
If the halves' staging and band kernels shared the device well, slicing a raw call inside the library would pay."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw
import torch

pairs = 100_000
a, b = sw.generate_pairs("utf8_lines", pairs, seed=42)
leads = lambda t: np.concatenate([[0], np.cumsum((t.data & 0xC0) != 0x80)])
def cps(t):
    l = leads(t); return l[t.offsets[1:].astype(np.int64)] - l[t.offsets[:-1].astype(np.int64)]
cells_all = cps(a).astype(np.int64) * cps(b)
def run(parts, seconds=1.5):
    scopes = [sw.DeviceScope(gpu_device=0) for _ in parts]
    work = []
    for scope, (lo, hi) in zip(scopes, parts):
        ta, tb = a.subview(lo, hi).to_device(scope), b.subview(lo, hi).to_device(scope)
        out = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        engine = sw.LevenshteinDistancesUTF8(capabilities=scope)
        work.append((scope, engine, ta, tb, out, int(cells_all[lo:hi].sum())))
    for scope, engine, ta, tb, out, _ in work:
        for _ in range(20): engine.pairs(ta, tb, scope, bound=32, out=out)
    done = [0] * len(work)
    stop = time.perf_counter() + seconds
    def loop(i):
        scope, engine, ta, tb, out, _ = work[i]
        while time.perf_counter() < stop:
            engine.pairs(ta, tb, scope, bound=32, out=out); done[i] += 1
    threads = [threading.Thread(target=loop, args=(i,)) for i in range(len(work))]
    t0 = time.perf_counter()
    for t in threads: t.start()
    for t in threads: t.join()
    dt = time.perf_counter() - t0
    total = sum(done[i] * work[i][5] for i in range(len(work)))
    return total / dt / 1e12, [d / dt for d in done]

for parts in ([(0, pairs)], [(0, pairs // 2), (pairs // 2, pairs)], [(0, pairs // 4), (pairs // 4, pairs // 2), (pairs // 2, 3 * pairs // 4), (3 * pairs // 4, pairs)],
              [(0, pairs), (0, pairs)]):
    tcups, rates = run(parts)
    print(len(parts), "scope(s)", [hi - lo for lo, hi in parts], "aggregate TCUPS", round(tcups, 1), "calls/s per scope", [round(r) for r in rates], flush=True)
