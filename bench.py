#!/usr/bin/env python3
"""bench.py -- GCUPS of batched Levenshtein on MI355X (BASELINE.json metric): rank 0's stdout ENDS with one short JSON line (the headline,
held to 6 KB at N = 1 and 8 KB beyond: `fit_line`), preceded by one short `{"leg": ...}` line per other config; the full per-config entries,
CPU rows and rank identities go to --details-out (bench_configs.json) and to stderr.

A "step" is ONE SYNCHRONOUS CALL of the hot path over one batch held resident in HBM -- results visible on return,
the reference's own metric (`compute_into` inside `measure_throughput`, bench.rs:478-486, utils.rs:721-799) -- with tapes
prepared once outside the timed region (as the reference builds its tape views once, bench.rs:292-306), plus, for
N > 1, the RCCL gather of the u32 distances to rank 0 that the north-star names: enqueued behind the call, it travels
while the next step's call computes (two result buffers) and is waited for before its buffer is reused and by the
barrier + synchronize that closes the timed region.

  --config c2 (default)  BASELINE configs[1]: 1,000,000 printable-ASCII token pairs per GPU, lengths U[32,96],
                         unbounded. Weak scaling: rank r scores pairs [r*P, (r+1)*P) of the seeded stream.
  --config c5            BASELINE configs[4]: 100,000,000 short-word pairs (<= 16 B) in total, strong scaling: the
                         ranks split the stream into cells-balanced contiguous shards, score them in pieces and
                         send each piece to rank 0 (ncclSend / ncclRecv group) while the next one is scored.

CUPS accounting is the reference's (similarities/bench.rs:413-414): cells = sum len(a_i)*len(b_i), whatever the
algorithm skips. Fields of the line:

  value              rate of the K timed synchronous steps (barrier + synchronize on both sides, max over ranks)
  value_steady       the same calls over at least `--steady-seconds` (default 1 s) of wall time
  value_pipelined    K steps enqueued asynchronously on two internal lanes (not the reference's metric; what a caller
                     that does not need each result before the next call gets)
  roofline           dominant kernel of the synchronous call: executed lane-ops (PMC constants, profiles/r6) over the kernel
                     time measured live with hipEvents inside the library, on the kernel's own stream
  configs            (N = 1) every other BASELINE config at full size, same measurement per entry: C1, C3 prepared / raw,
                     C4 linear / affine / full byte alphabet, C5 at 20 M pairs per GPU, Smith-Waterman on C4's sequences, the
                     2048 x 2048 cross-product call -- in the line: {name: value, ms, kernel, kernel_ms, frac, parity}; in
                     --details-out: each with its roofline object, traffic and parity sample
  cpu_baseline(s)    the oracle's CPU rows on this box's host cores, bounded samples (rank 0, N = 1)
  details            where the full entries were written

Order of a run: data, tapes, `--prewarm-seconds` of untimed steps (an idle MI355X needs a few hundred milliseconds of work
to reach its clocks), W warm-up steps, the K timed steps, a profiled repeat, the steady-state loop, the pipelined steps,
the checks, the other configs, the CPU rows.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N              # the same N ranks: without WORLD_SIZE in the environment the script starts
                                          # torch.distributed.run itself, as a child, before anything touches a GPU
    python bench.py --gpus N --single-process   # one process, one scope over N devices: swh_scope_init_gpus +
                                          # swh_levenshtein_pairs_sharded, the RCCL gather INSIDE the library

At N > 1 the line's `value` is still config C2 (weak: N x 1 M pairs); `configs` then holds BASELINE configs[4] -- C5, the
100 M short-word pairs split over the ranks (strong scaling), with `gather_ok`, the gather's cost per step and the ranks
and devices that took part -- and `single_process` the in-library sharded call over the same N devices, run by rank 0 as
a child process once the ranks are done. A world size that differs from --gpus is an error, never a smaller measurement.
    python bench.py --only-config c4_linear --calls 3     # exactly 3 engine calls of one config (PMC passes)
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_sources import KERNELS as KERNEL_SOURCES, source_digest  # noqa: E402

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 32-bit
# integer lane-ops/s (the FP32 vector peak 157.3 TFLOP/s counts an FMA as two); HBM3E 8 TB/s.
PEAK_VALU_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0
# SURVEY.md 8d's scalar-DP models (32-bit integer ops per cell): Levenshtein 5, NW linear 6 (+1 LDS read), Gotoh 11 (+1)
NOMINAL_OPS_PER_CELL = {"lev": 5, "lev_utf8": 5, "nw_linear": 6, "nw_affine": 11}
# Executed VALU instructions and HBM traffic per call come from rocprofv3 PMC passes over `bench.py --only-config ...`
# (tools/refresh_profiles.sh -> tools/pmc_constants.py); they cannot be read from inside the process.
PMC_CONSTANTS = os.path.join(ROOT, "profiles", "r6", "pmc_constants.json")
PMC_CONSTANTS_NAME = os.path.relpath(PMC_CONSTANTS, ROOT)

CONFIGS = {
    "c2": dict(workload="tokens64", pairs=1_000_000, scaling="weak",
               text="C2 tokens64: {pairs} ASCII token pairs per GPU, lengths U[32,96], unbounded Levenshtein"),
    "c5": dict(workload="short_words", pairs=100_000_000, scaling="strong",
               text="C5 short_words: {pairs} word pairs (<= 16 B, mean ~6) in total, split over the GPUs, unbounded Levenshtein"),
}

# The other BASELINE configs at full size, measured after the headline legs at N = 1 (`configs` in the line) and one at a
# time under the profiler (`--only-config`). kind: lev (bytes) / lev_utf8 (code points) / nw; `check` = pairs compared
# with the oracle; README rows mirrored: /root/reference/similarities/README.md:33-40, :67-70, :107-110.
LEGS = {
    "c1": dict(workload="words16", pairs=10_000, kind="lev", prepared=True, check=10_000,
               text="C1: 10 K ASCII word pairs <= 16 B, unbounded Levenshtein (BASELINE configs[0] shape, on the GPU)"),
    "c2": dict(workload="tokens64", pairs=1_000_000, kind="lev", prepared=True, check=20_000,
               text="C2: 1 M ASCII token pairs ~64 B, unbounded Levenshtein (the headline config)"),
    "c3": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=32, prepared=True, variant="k32", check=5_000,
               text="C3: 100 K UTF-8 line pairs ~1 KB, bounded Levenshtein k = 32 over code points, tapes prepared (decoded once)"),
    "c3_raw": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=32, prepared=False, variant="k32", check=5_000,
                   text="C3 on raw device tapes: UTF-8 validated and decoded inside every call"),
    # the reference's literal UTF-8 calls (similarities/bench.rs:538-546, :625-629: raw tapes handed over on every call, no bound)
    "utf8_unbounded_raw": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", prepared=False, variant="unbounded", check=5_000,
                               text="C3's lines, UNBOUNDED Levenshtein over code points on raw device tapes: LevenshteinDistancesUtf8's literal call "
                                    "(bench.rs:538-546) -- validated and decoded inside every call"),
    "c3_raw_cold": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=32, prepared=False, cold=3, variant="k32", check=5_000,
                        text="C3 on raw device tapes the scope has not seen in its previous call (three copies of the tapes in turn: the library's "
                             "beliefs about a tape's byte total and its ASCII-ness never apply)"),
    "c3_raw_forget": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=32, prepared=False, forget=True, variant="k32", check=5_000,
                          text="C3 on raw device tapes with swh_scope_forget() before every call: a history-free call, the reference's compute_into "
                               "semantics (bench.rs:478-486) -- what the scope's beliefs are worth is the distance to c3_raw"),
    "utf8_unrelated_raw": dict(workload="script_lines", pairs=50_000, kind="lev_utf8", prepared=False, variant="unbounded", check=2_500,
                               text="UNRELATED article lines of 700-1300 code points, one script each (Latin / Cyrillic / Greek / Arabic / Devanagari), unbounded "
                                    "Levenshtein over code points on raw device tapes: what the reference's cross-product of XLSum lines pairs up for "
                                    "LevenshteinDistancesUtf8 (bench.rs:386-399, similarities/README.md:18, :39-40) -- nothing for a band to settle; the "
                                    "block kernel on per-pair dense alphabets (bp_dense.hpp)"),
    "c4_linear": dict(workload="protein4k", pairs=10_000, kind="nw", gaps=(-4, -4), prepared=True, variant="linear", check=128,
                      text="C4: NW, 256x256 i8 matrix (20 amino acids + other), 10 K pairs ~4 KB, linear gaps -4"),
    "c4_affine": dict(workload="protein4k", pairs=10_000, kind="nw", gaps=(-11, -1), prepared=True, variant="affine", check=128,
                      text="C4 with affine gaps (-11, -1)"),
    "c4_bytes": dict(workload="bytes4k", pairs=10_000, kind="nw", gaps=(-4, -4), prepared=True, variant="linear", check=128,
                     text="C4 over the full byte alphabet (all 256 classes of the matrix in use), 10 K pairs ~4 KB, linear gaps -4 (BASELINE configs[3] as worded)"),
    "c4_letters52": dict(workload="bytes4k", pairs=10_000, kind="nw", gaps=(-4, -4), prepared=True, variant="letters52", check=128, letters=52,
                         text="C4 over a 52-letter alphabet (a-z, A-Z: what a rust-bio style scoring closure over mixed-case text distinguishes, "
                              "bench.rs:746-752; 53 symbol classes), 10 K pairs ~4 KB, linear gaps -4: the column-profile kernel on the wide class table"),
    # SmithWatermanScores (bench.rs:882-963) on C4's sequences, and the reference's own call shape -- compute_into(queries, candidates, &mut matrix),
    # bench.rs:478-486, :599-603 -- at the side its H100 tables use (similarities/README.md:22: 16384 per core -> side 2048 on 256 CUs)
    "sw_linear": dict(workload="protein4k", pairs=10_000, kind="nw", local=True, gaps=(-4, -4), prepared=True, variant="sw_linear", check=32,
                      text="Smith-Waterman local score on C4's sequences: 256x256 i8 matrix, 10 K pairs ~4 KB, linear gaps -4"),
    "sw_affine": dict(workload="protein4k", pairs=10_000, kind="nw", local=True, gaps=(-11, -1), prepared=True, variant="sw_affine", check=32,
                      text="Smith-Waterman local score on C4's sequences with affine gaps (-11, -1)"),
    "cross_lev": dict(workload="acgt100", side=2048, pairs=2048 * 2048, kind="lev", cross=True, prepared=True, variant="cross", check=4096,
                      text="cross-product: 2048 queries x 2048 candidates of 100 ACGT bytes (the reference's compute_into shape, bench.rs:478-486), "
                           "unit-cost Levenshtein, u64 matrix on the device"),
    "cross_nw": dict(workload="acgt100", side=2048, pairs=2048 * 2048, kind="nw", cross=True, gaps=(-2, -2), unary=(2, -1), prepared=True,
                     variant="cross_linear", check=4096,
                     text="cross-product: 2048 x 2048 ACGT-100 strings, NeedlemanWunschScores with unary_class_costs(2, -1), linear gaps -2 (bench.rs:658-662, :814-821)"),
    "c5": dict(workload="short_words", pairs=20_000_000, kind="lev", prepared=True, check=200_000,
               text="C5: one GPU's share of the 100 M short-word pairs (20 M pairs <= 16 B, mean ~6), unbounded Levenshtein"),
    # beyond BASELINE's five: what round 4 added kernels for
    "c3_k100": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=100, prepared=True, variant="k100", check=5_000,
                    text="C3's lines at k = 100 (STRINGWARS_ERROR_BOUND is free-form, README.md:311): the banded kernel's two-word window"),
    "nw_words": dict(workload="words16", pairs=4_000_000, kind="nw", gaps=(-2, -2), unary=(2, -1), prepared=True, variant="unary_linear", check=20_000,
                     text="NW on word-sized strings (the reference's default `words` token mode, bench.rs:271): 4 M pairs <= 16 B, "
                          "unary_class_costs(2, -1) as a 32-class table, linear gaps -2 -- one pair per lane (alignshort.hip)"),
}
DEFAULT_LEGS = ["c1", "c3", "c3_raw", "c3_raw_cold", "c3_raw_forget", "utf8_unbounded_raw", "utf8_unrelated_raw", "c3_k100", "c4_linear", "c4_affine", "c4_bytes", "c4_letters52", "c5", "nw_words",
                "sw_linear", "sw_affine", "cross_lev", "cross_nw"]


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--prewarm-seconds", type=float, default=0.5, help="untimed steps before the W warm-up steps, by the clock: brings an idle device to its clocks (0: none)")
    p.add_argument("--steady-seconds", type=float, default=1.0, help="length of the steady-state loop of synchronous calls behind `value_steady` (0: none)")
    p.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    p.add_argument("--workload", default=None, help="override the config's synthetic workload")
    p.add_argument("--pairs", type=int, default=None, help="pairs per GPU (weak configs) / in total (strong configs)")
    p.add_argument("--chunks", type=int, default=0,
                   help="pieces a shard is scored and gathered in (strong configs); 0 = 4 when there is a gather to overlap, 1 on one GPU")
    p.add_argument("--algorithm", default="auto", choices=["auto", "wavefront", "bitparallel", "tiled"])
    p.add_argument("--seed", type=int, default=int(os.environ.get("STRINGWARS_SEED", "42")))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each CPU baseline row (the Gotoh row takes half)")
    p.add_argument("--no-configs", action="store_true", help="leave the other BASELINE configs out of the line")
    p.add_argument("--no-pipelined", action="store_true", help="leave the pipelined steps out (the profiler then sees synchronous launches only)")
    p.add_argument("--legs", default=",".join(DEFAULT_LEGS), help="which configs go into `configs`")
    p.add_argument("--only-config", default=None, choices=sorted(LEGS), help="run ONE config's leg and print its entry (PMC / rocprof passes)")
    p.add_argument("--calls", type=int, default=0, help="with --only-config: make exactly this many engine calls (no clock-driven loops)")
    p.add_argument("--leg-pairs", type=int, default=0, help="with --only-config / --legs: scale every leg to this many pairs (testing)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --share-gpu exercises the multi-rank control flow on a one-GPU box (testing only)")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (testing only; never a result)")
    p.add_argument("--single-process", action="store_true",
                   help="one process, one scope over --gpus devices: swh_scope_init_gpus + swh_levenshtein_pairs_sharded (the RCCL gather inside the library)")
    p.add_argument("--no-single-process", action="store_true", help="at N > 1: leave the single-process leg (a child of rank 0) out of the line")
    p.add_argument("--single-process-timeout", type=float, default=300.0)
    p.add_argument("--gather-transport", default="u32", choices=["auto", "u32", "u8"],
                   help="strong configs: what travels to rank 0 -- the u32 distances as they are (what BASELINE's north star names: the default), "
                        "or bytes when no distance can exceed 255 (u8 / auto). At N > 1 the C5 entry is measured with u32 and carries the u8 variant beside it (`gather_u8`)")
    p.add_argument("--collective-timeout", type=float, default=600.0, help="seconds a collective may take before the process group gives up (a dead rank ends the run, with an error line, within this)")
    p.add_argument("--launch-timeout", type=float, default=7200.0, help="with --gpus N and no WORLD_SIZE: seconds the ranks this script starts may take in all")
    p.add_argument("--die-rank", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: this rank exits with code 3 ...
    p.add_argument("--die-at", default="start", choices=["start", "after-init", "measure"], help=argparse.SUPPRESS)   # ... at this point
    p.add_argument("--c5-pairs", type=int, default=0, help="at N > 1: total pairs of the C5 strong-scaling entry (default 100 M; testing)")
    p.add_argument("--details-out", default=os.path.join(ROOT, "bench_configs.json"),
                   help="where rank 0 writes the FULL per-config entries, CPU rows and rank list (the stdout line carries their summary only)")
    return p.parse_args()


def load_pmc_constants(path=PMC_CONSTANTS):
    try:
        with open(path) as handle:
            return json.load(handle)
    except (OSError, ValueError):
        return {"kernels": {}}


def kernel_family(stamp):
    """The library stamps launches with the instantiation's name (`wavefront_class_g64_w80`, `nwprofile_w16`); PMC constants and
    source digests are kept per kernel family."""
    for family in ("wavefront", "nwprofile", "align_short", "align_wide", "align_long"):
        if stamp.startswith(family):
            return family
    return stamp


# kernel families whose launches one engine call can hold together (api.hip: the two-stage schedule; the planned alignment path)
CO_RUNNING = ({"banded", "bitparallel", "bitparallel_u32", "bitparallel_long", "bitparallel_long_u32"}, {"nwprofile", "wavefront"})


def roofline_of(kernel, kernel_ms, cells, algorithmic_bytes, workload, pairs, constants, extra=None, variant="", model="lev"):
    """Roofline object of one call's dominant kernel family. `achieved` is EXECUTED work: SQ_INSTS_VALU (wave instructions
    of every dispatch of the family in one call, PMC) x 64 lanes / the time those kernels cover, measured live with hipEvents
    on their own streams; `frac` = achieved / the integer VALU peak. SURVEY 8d's nominal ops-per-cell model is carried
    separately (a bit-parallel kernel executes ~1.2 lane-ops per cell, so that figure exceeds the peak by construction and is
    not a roofline fraction). `pmc_stale` is true when the kernel's sources changed since the counters were taken."""
    seconds = kernel_ms * 1e-3
    stamp = kernel
    kernel = kernel_family(stamp)
    key = f"{kernel}|{workload}" + (f"|{variant}" if variant else "")
    entry = constants.get("kernels", {}).get(key)
    per_call = entry.get("pairs_per_call", entry.get("pairs_per_launch")) if entry else None
    scale = pairs / per_call if entry else 1.0   # the counters were taken on calls of `pairs_per_call` pairs: work is linear in pairs
    hbm_gbs = algorithmic_bytes / seconds / 1e9 if seconds > 0 else 0.0
    nominal = NOMINAL_OPS_PER_CELL.get(model, 5)
    roof = {
        "bound": "valu", "kernel": kernel, "longest_launch": stamp, "kernel_ms": round(kernel_ms, 4), "unit": "Tint32op/s", "peak": round(PEAK_VALU_TOPS, 1),
        "achieved": None, "frac": None, "traffic": None, "cells_per_launch": cells,
        "nominal_ops_per_cell_equiv": round(nominal * cells / seconds / 1e12, 3) if seconds > 0 else None,
        "hbm": {"achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(hbm_gbs / PEAK_HBM_GBS, 4),
                "algorithmic_bytes": algorithmic_bytes},
    }
    if entry and seconds > 0:
        # a call may run DP kernels of several families (the two stages of a doubling call: the band, then bit-parallel blocks; an alignment
        # call's profile kernel and wavefront classes): `kernel_ms` is the time they cover together, so their instructions are summed
        # (only families that can run in ONE call beside the dominant one -- the two stages of a doubling call, an alignment call's profile
        # kernel and wavefront classes --, never whatever else happens to share the workload's name: advisor, round 5)
        suffix = key[len(kernel):]
        beside = next((group for group in CO_RUNNING if kernel in group), {kernel})
        others = {k: v for k, v in constants.get("kernels", {}).items()
                  if k != key and k.endswith(suffix) and k[:len(k) - len(suffix)] in beside
                  and v.get("pairs_per_call", v.get("pairs_per_launch")) == per_call}
        lane_ops = (entry["valu_insts"] + sum(v["valu_insts"] for v in others.values())) * scale * 64
        if others:
            roof["families_summed"] = sorted([key] + list(others))
        roof["achieved"] = round(lane_ops / seconds / 1e12, 3)
        roof["frac"] = round(lane_ops / seconds / 1e12 / PEAK_VALU_TOPS, 4)
        roof["valu_wave_insts_per_launch"] = round(lane_ops / 64, 1)
        roof["lane_ops_per_cell"] = round(lane_ops / max(cells, 1), 3)
        if entry.get("fetch_kb") is not None and entry.get("write_kb") is not None:
            # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of the bytes
            # of wide reads (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE as is. Infinity-Cache hits are counted.
            # Calibrated on this box (profiles/r2/fetch_calibration.txt): x2 holds for coalesced streams and for windows in
            # adjacent sectors, a lone 64-byte sector is fetched and tallied as 64 B (x1) -- kernels whose work items read
            # scattered 16-byte windows lie between `vs_algorithmic_lower` (x1) and `vs_algorithmic` (x2).
            roof["traffic"] = int((entry["fetch_kb"] * 1024 * 2 + entry["write_kb"] * 1024) * scale)
            lower = (entry["fetch_kb"] * 1024 + entry["write_kb"] * 1024) * scale
            roof["traffic_detail"] = {"fetch_kb_raw": round(entry["fetch_kb"] * scale, 1), "write_kb_raw": round(entry["write_kb"] * scale, 1),
                                      "fetch_correction": 2, "vs_algorithmic": round(roof["traffic"] / max(algorithmic_bytes, 1), 2),
                                      "vs_algorithmic_lower": round(lower / max(algorithmic_bytes, 1), 2)}
        roof["pmc_source"] = f"{entry.get('source')}; calls of {per_call} pairs"
        if kernel in KERNEL_SOURCES and entry.get("source_digest") is not None:
            roof["pmc_stale"] = entry["source_digest"] != source_digest(kernel)
        else:
            roof["pmc_stale"] = None   # an entry without a digest cannot be tied to the tree
    else:
        roof["note"] = f"no PMC constants for {key} in {PMC_CONSTANTS_NAME}"
    if extra:
        roof.update(extra)
    return roof


def cpu_rows(a, b, budget_s=6.0):
    """The CPU rows BASELINE.md section 3 lists, on the host cores of this box, same inputs, bounded samples:
    hyyro<1cpu> (the row that corresponds to rapidfuzz::levenshtein<Bytes,1cpu>, bench.rs:407), hyyro<Ncpu>,
    wagner_fischer<1cpu> (~ bio::levenshtein, bench.rs:443-459)."""
    import oracle  # checker + reported baseline only; never on the timed GPU path
    rows = []
    pairs = len(a.offsets) - 1

    def timed(name, cores, fn, sample_pairs, what):
        cells = int((a.lengths[:sample_pairs] * b.lengths[:sample_pairs]).sum())
        repeats, start = 0, time.perf_counter()
        while True:
            fn(sample_pairs)
            repeats += 1
            spent = time.perf_counter() - start
            if spent >= budget_s:
                break
        rows.append({"name": name, "value": round(cells * repeats / spent / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
                     "sample": f"first {sample_pairs} pairs of the same workload x {repeats} repeats, {what}"})

    one = min(pairs, 1_000_000)
    timed("cpu::hyyro<1cpu>", 1, lambda n: oracle.levenshtein_pairs(a, b, algo="hyyro", count=n), one,
          "oracle Hyyro/Myers 64-bit bit-parallel (the algorithm family of rapidfuzz), one pair per call")
    # "the host cores" of the north-star: every hardware thread the box has (not a cap), the physical core count stated
    cores = os.cpu_count() or 1
    physical = physical_cores()
    if cores > 1:
        pool = ThreadPoolExecutor(cores)

        def sharded(n):
            bounds = [n * t // cores for t in range(cores + 1)]
            jobs = [pool.submit(oracle.levenshtein_pairs, a, b, False, "hyyro", None, bounds[t], bounds[t + 1] - bounds[t]) for t in range(cores)]
            for job in jobs:
                job.result()
        timed(f"cpu::hyyro<{cores}cpu>", cores, sharded, one,
              f"the same over {cores} host threads = every hardware thread of the box ({physical} physical cores), contiguous slices")
        pool.shutdown()
    timed("cpu::wagner_fischer<1cpu>", 1, lambda n: oracle.levenshtein_pairs(a, b, algo="wf", count=n), min(pairs, 100_000),
          "oracle two-row Wagner-Fischer (the algorithm of bio::levenshtein)")
    # BASELINE configs[0] as it is worded: the per-pair CPU call on 10 K ASCII word pairs <= 16 B, one thread, in the harness
    # loop (rapidfuzz::levenshtein::distance per pair inside measure_throughput, bench.rs:404-423, utils.rs:721-799): the
    # oracle's bit-parallel routine called pair by pair over the `words16` tapes -- the plumbing row, no GPU
    import stringwars_amd as sw
    wa, wb = sw.generate_pairs("words16", 10_000, seed=42)
    word_cells = int((wa.lengths.astype(np.int64) * wb.lengths.astype(np.int64)).sum())
    repeats, start = 0, time.perf_counter()
    while True:
        oracle.levenshtein_pairs(wa, wb, algo="hyyro")
        repeats += 1
        spent = time.perf_counter() - start
        if spent >= budget_s / 3:
            break
    rows.append({"name": "c1/cpu::hyyro<1cpu>", "value": round(word_cells * repeats / spent / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
                 "config": "C1: 10 K ASCII word pairs <= 16 B (words16), one thread, one pair per call",
                 "pairs_per_second": round(10_000 * repeats / spent, 1),
                 "sample": f"all 10000 pairs x {repeats} repeats, oracle Hyyro/Myers bit-parallel called per pair (the shape of the rapidfuzz row, bench.rs:404-423)"})
    # cpu::gotoh<1cpu> (~ bio::pairwise::Aligner::global, bench.rs:746-765) has no pairs of this workload to run on: it is
    # timed on config C4's shape (4 KB amino-acid sequences, 256x256 i8 matrix, affine gaps), cells of that sample
    pa, pb = sw.generate_pairs("protein4k", 4, seed=42)
    matrix = sw.substitution_matrix(42)
    gotoh_cells = int((pa.lengths * pb.lengths).sum())
    repeats, start = 0, time.perf_counter()
    while True:
        oracle.nw_pairs(pa, pb, matrix, -11, -1)
        repeats += 1
        spent = time.perf_counter() - start
        if spent >= budget_s / 2:
            break
    rows.append({"name": "cpu::gotoh<1cpu>", "value": round(gotoh_cells * repeats / spent / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
                 "sample": f"4 pairs of config C4 (protein4k, affine -11/-1) x {repeats} repeats, oracle Gotoh score-only DP "
                           "(bio::pairwise also keeps traceback matrices)"})
    return rows


def physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo; falls back to the logical count."""
    seen, socket = set(), 0
    try:
        for row in open("/proc/cpuinfo"):
            if row.startswith("physical id"):
                socket = int(row.split(":")[1])
            elif row.startswith("core id"):
                seen.add((socket, int(row.split(":")[1])))
    except (OSError, ValueError):
        pass
    return len(seen) or (os.cpu_count() or 1)


def codepoint_lengths(strs):
    """Code points per string of a UTF-8 tape (the reference counts cells in `chars().count()`, bench.rs:230-247)."""
    leads = ((strs.data & 0xC0) != 0x80).astype(np.int64)
    prefix = np.concatenate(([0], np.cumsum(leads)))
    return prefix[strs.offsets[1:].astype(np.int64)] - prefix[strs.offsets[:-1].astype(np.int64)]


def run_leg(name, sw, scope, torch, device, seed, constants, calls=0, pairs_override=0, check=True, algorithm="auto"):
    """One BASELINE config, measured like the headline: tapes resident in HBM, synchronous calls timed by the host, the DP
    kernels' time from the library's hipEvents, executed lane-ops from the PMC constants, a sample against the oracle.
    With `calls` > 0 exactly that many engine calls are made (the profiler's passes need a known count)."""
    leg = LEGS[name]
    pairs = pairs_override or leg["pairs"]
    started = time.perf_counter()
    cross = bool(leg.get("cross"))
    if cross:
        # queries [0, side) x candidates [side, 2*side) of one token tape (bench.rs:113-148): fixed-length ACGT strings, the
        # reference's DNA datasets in synthetic form (similarities/README.md:30-40)
        side = max(2, int(round(pairs ** 0.5))) if pairs_override else leg["side"]
        pairs = side * side
        length = int(leg["workload"][4:])
        rng = np.random.default_rng(seed)
        offsets = (np.arange(2 * side + 1, dtype=np.uint64) * length)
        tape = sw.Strs(data=np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2 * side * length)], offsets=offsets)
        a, b = sw.Strs(data=tape.data[:side * length], offsets=offsets[:side + 1].copy()), sw.Strs(data=tape.data[side * length:], offsets=offsets[:side + 1].copy())
    else:
        a, b = sw.generate_pairs(leg["workload"], pairs, seed=seed)
    letters = None
    if leg.get("letters"):   # the byte workload folded onto an alphabet of that many letters (a-z, A-Z, 0-9 ...): same lengths, same edits
        letters = (bytes(range(97, 123)) + bytes(range(65, 91)) + bytes(range(48, 58)))[:leg["letters"]]
        fold = np.frombuffer(letters, dtype=np.uint8)
        a.data[:] = fold[a.data % len(letters)]
        b.data[:] = fold[b.data % len(letters)]
    generate_s = time.perf_counter() - started
    if int(a.offsets[-1]) < 2 ** 32 and int(b.offsets[-1]) < 2 ** 32:
        a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    as_torch = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else (x.view(np.int32) if x.dtype == np.uint32 else x)).to(device)
    tensors = [as_torch(x) for x in (a.data, a.offsets, b.data, b.offsets)]
    da = sw.DeviceTape(tensors[0].data_ptr(), tensors[1].data_ptr(), a.count, a.offsets.dtype, keepalive=tensors[:2])
    db = sw.DeviceTape(tensors[2].data_ptr(), tensors[3].data_ptr(), b.count, b.offsets.dtype, keepalive=tensors[2:])
    kind, utf8 = leg["kind"], leg["kind"] == "lev_utf8"
    out = torch.zeros(pairs + 4, dtype=torch.int64 if cross else torch.int32, device=device)
    nw_engine = sw.SmithWatermanScores if leg.get("local") else sw.NeedlemanWunschScores
    if kind == "nw" and leg.get("unary"):
        byte_to_class, class_costs = sw.unary_class_costs(*leg["unary"])       # bench.rs:98-108: class = byte % 32
        matrix = class_costs[byte_to_class][:, byte_to_class].astype(np.int8)  # the same scoring as a 256 x 256 table (what the oracle takes)
        engine = nw_engine(byte_to_class, class_costs, open=leg["gaps"][0], extend=leg["gaps"][1], capabilities=scope)
        model = "nw_linear" if leg["gaps"][0] == leg["gaps"][1] else "nw_affine"
    elif kind == "nw":
        alphabet = letters if letters else (None if leg["workload"] == "bytes4k" else sw.synth.AMINO_ACIDS)
        matrix = sw.substitution_matrix(seed, alphabet)
        engine = nw_engine(substitution_matrix=matrix, open=leg["gaps"][0], extend=leg["gaps"][1], capabilities=scope)
        model = "nw_linear" if leg["gaps"][0] == leg["gaps"][1] else "nw_affine"
    else:
        engine = (sw.LevenshteinDistancesUTF8 if utf8 else sw.LevenshteinDistances)(capabilities=scope, algorithm=algorithm)
        model = kind
    if cross:
        pa, pb = sw.PreparedTape(scope, da, utf8=utf8), sw.PreparedTape(scope, db, utf8=utf8)
        call = lambda: engine(pa, pb, scope, out=out.data_ptr())     # compute_into(queries, candidates, &mut matrix): 64-bit entries, row-major
    elif leg["prepared"]:
        pa, pb = sw.PreparedTape(scope, da, utf8=utf8), sw.PreparedTape(scope, db, utf8=utf8)
        call = engine.bind_pairs(pa, pb, scope, out, bound=leg.get("bound")) if kind != "nw" else engine.bind_pairs(pa, pb, scope, out)
    elif kind == "nw":
        call = lambda: engine.pairs(da, db, scope, out=out)
    elif leg.get("cold"):
        # copies of the tapes taken in turn: no call meets the tapes of the call before it
        copies = [(da, db)]
        for _ in range(leg["cold"] - 1):
            more = [t.clone() for t in tensors]
            copies.append((sw.DeviceTape(more[0].data_ptr(), more[1].data_ptr(), a.count, a.offsets.dtype, keepalive=more[:2]),
                           sw.DeviceTape(more[2].data_ptr(), more[3].data_ptr(), b.count, b.offsets.dtype, keepalive=more[2:])))
        turn = [0]

        def call():
            ta, tb = copies[turn[0] % len(copies)]
            turn[0] += 1
            return engine.pairs(ta, tb, scope, bound=leg.get("bound"), out=out)
    elif leg.get("forget"):
        def call():
            scope.forget()
            return engine.pairs(da, db, scope, bound=leg.get("bound"), out=out)
    else:
        call = lambda: engine.pairs(da, db, scope, bound=leg.get("bound"), out=out)
    made = 0
    if calls > 0:                              # the profiler's passes: a fixed number of calls, the last one stamped
        for _ in range(calls - 1):
            call()
        scope.set_profiling(True)
        call()
        timing = scope.last_timing()
        scope.set_profiling(False)
        made, elapsed, timed_calls, best_ms = calls, None, 0, None
    else:
        call()
        warm_until = time.perf_counter() + 0.25   # by the clock, not by a count: the device must be at its clocks
        while time.perf_counter() < warm_until:
            call()
        torch.cuda.synchronize()
        walls, start = [], time.perf_counter()
        while True:
            t0 = time.perf_counter()
            call()
            now = time.perf_counter()
            walls.append(now - t0)
            if (now - start >= 0.35 and len(walls) >= 5) or now - start >= 3.0:
                break
        elapsed, timed_calls, best_ms = now - start, len(walls), min(walls) * 1e3
        scope.set_profiling(True)
        timings = []
        for _ in range(3):
            call()
            timings.append(scope.last_timing())
        scope.set_profiling(False)
        timing = min(timings, key=lambda t: t["compute_ms"])
    if utf8:
        lengths_a, lengths_b = codepoint_lengths(a), codepoint_lengths(b)
    else:
        lengths_a, lengths_b = a.lengths.astype(np.int64), b.lengths.astype(np.int64)
    cells = int(lengths_a.sum()) * int(lengths_b.sum()) if cross else int((lengths_a * lengths_b).sum())
    algorithmic_bytes = int(a.data.nbytes + b.data.nbytes + 2 * len(lengths_a) * a.offsets.dtype.itemsize + (8 if cross else 4) * pairs)
    entry = {"config": name, "workload": leg["text"], "pairs": pairs, "cells": cells, "dtype": "i32" if kind == "nw" else "u32",
             "tapes": "prepared" if leg["prepared"] else "raw device tapes", "offsets": str(a.offsets.dtype)}
    if timed_calls:
        entry.update({"value": round(cells * timed_calls / elapsed / 1e9, 2), "unit": "GCUPS",
                      "value_is": f"{timed_calls} synchronous calls over {elapsed:.2f} s timed by the host, after 0.25 s of untimed calls",
                      "ms_per_call": round(elapsed / timed_calls * 1e3, 4), "best_call_ms": round(best_ms, 4),
                      "gcups_kernels": round(cells / max(timing["compute_ms"], 1e-9) / 1e6, 2)})
    else:
        entry["calls_made"] = made
    entry["kernels_per_call"] = timing["kernels"]
    entry["all_kernels_ms"] = round(timing["total_ms"], 4)
    entry["roofline"] = roofline_of(timing["dominant_name"], timing["compute_ms"], cells, algorithmic_bytes, leg["workload"], pairs, constants,
                                    variant=leg.get("variant", ""), model=model,
                                    extra={"measured": "hipEvents inside the library: time covered by the DP kernels of one synchronous call"})
    if timing["cells"] != cells:
        entry["cells_mismatch"] = {"library": timing["cells"], "host": cells}
    if check and leg.get("check"):
        import oracle  # the checker; the timed calls above never touch it
        count = min(pairs, leg["check"])
        got = out[:pairs].cpu().numpy()
        if cross:
            # the first and the last rows of the matrix (and as many in between as `check` allows), entry by entry
            rows = sorted(set(np.linspace(0, len(lengths_a) - 1, max(2, count // len(lengths_b))).astype(int).tolist()))
            grid = got.reshape(len(lengths_a), len(lengths_b))
            ok = True
            for r in rows:
                if kind == "nw":
                    want = [oracle.nw_score(a[r], b[c], matrix, leg["gaps"][0], leg["gaps"][1], local=bool(leg.get("local"))) for c in range(len(lengths_b))]
                else:
                    want = [oracle.levenshtein(a[r], b[c], algo="hyyro") for c in range(len(lengths_b))]
                ok = ok and bool((grid[r].astype(np.int64) == np.array(want, dtype=np.int64)).all())
            count = len(rows) * len(lengths_b)
        elif kind == "nw" and leg.get("local"):
            want = np.array([oracle.nw_score(a[i], b[i], matrix, leg["gaps"][0], leg["gaps"][1], local=True) for i in range(count)], dtype=np.int64)
            ok = bool((got[:count].astype(np.int64) == want).all())
        elif kind == "nw":
            want = oracle.nw_pairs(a, b, matrix, leg["gaps"][0], leg["gaps"][1], count=count)
            ok = bool((got[:count].astype(np.int64) == want.astype(np.int64)).all())
        else:
            want = oracle.levenshtein_pairs(a, b, utf8=utf8, algo="hyyro" if not utf8 else "wf", bound=leg.get("bound"), count=count)
            ok = bool((got[:count].astype(np.uint32) == want.astype(np.uint32)).all())
        entry["parity_vs_oracle"] = ok
        entry["parity_sample"] = (f"{count} entries (whole rows) of the call's matrix against oracle/" if cross else
                                  f"first {count} pairs of the call's results against oracle/")
    entry["generate_s"] = round(generate_s, 2)
    del call, engine
    return entry


# ---- what goes where ---------------------------------------------------------------------------------------------------------------
# stdout ends with ONE short headline line (the reference's reporter prints one short line per variant, utils.rs:652-692): the
# driver parses that line, so it is held to LINE_BUDGET bytes. Before it, one short summary line per config (`{"leg": ...}`); the
# full entries -- prose, traffic detail, PMC provenance, CPU samples, every rank's identity -- go to --details-out (JSON) and, one
# line each, to stderr.
LINE_BUDGET = {1: 6144}
LINE_BUDGET_MANY = 8192


def line_budget(world):
    return LINE_BUDGET.get(world, LINE_BUDGET_MANY)


ROOFLINE_KEYS = ("bound", "kernel", "kernel_ms", "unit", "peak", "achieved", "frac", "traffic", "lane_ops_per_cell", "nominal_ops_per_cell_equiv",
                 "hbm", "pmc_stale", "launches_timed", "all_kernels_ms", "kernel_ms_scaled_to_step", "families_summed")


def compact_roofline(roof):
    """The roofline object of the headline line: every number, none of the prose (that stays in --details-out)."""
    if not roof:
        return roof
    out = {k: roof[k] for k in ROOFLINE_KEYS if k in roof}
    if roof.get("traffic_detail"):
        out["traffic_vs_algorithmic"] = roof["traffic_detail"]["vs_algorithmic"]
    return out


def compact_leg(entry):
    """One config's summary for the headline's `configs` map and its own short stdout line."""
    if "error" in entry:
        return {"error": str(entry["error"])[:160]}
    roof = entry.get("roofline") or {}
    out = {"value": entry.get("value"), "ms": entry.get("ms_per_call", entry.get("ms_per_step")), "pairs": entry.get("pairs", entry.get("pairs_total")),
           "kernel": roof.get("kernel", entry.get("dominant_kernel")), "kernel_ms": roof.get("kernel_ms", entry.get("kernel_ms_rank0_per_step")),
           "frac": roof.get("frac"), "parity": entry.get("parity_vs_oracle")}
    if "gather_ok" in entry:
        out["gather_ok"] = entry["gather_ok"]
        out["n_gpus"] = entry.get("n_gpus")
        if entry.get("gather"):
            out["gather_exposed_ms"] = entry["gather"].get("exposed_ms_per_step")
        if isinstance(entry.get("gather_u8"), dict):
            out["gather_u8_value"] = entry["gather_u8"].get("value", entry["gather_u8"].get("error"))
    return out


def compact_single_process(child):
    if child is None:
        return None
    if "error" in child:
        return {"error": str(child["error"])[:200]}
    return {"mode": child.get("mode"), "value": child.get("value"), "ms_per_step": child.get("ms_per_step"), "n_gpus": child.get("n_gpus"),
            "device_count": (child.get("config") or {}).get("device_count"), "parity_vs_oracle": child.get("parity_vs_oracle"), "gather_ok": child.get("gather_ok")}


def write_details(path, details):
    """The full entries: a JSON file beside the script (or wherever --details-out points) and one line each on stderr."""
    for key, value in details.items():
        rows = value if isinstance(value, list) else [value]
        for row in rows:
            print(json.dumps({"detail": key, **row} if isinstance(row, dict) else {"detail": key, "value": row}), file=sys.stderr, flush=True)
    if not path:
        return None
    try:
        with open(path, "w") as handle:
            json.dump(details, handle, indent=1)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    except OSError as error:
        return f"not written ({error})"


def fit_line(line, world):
    """Serialises the headline line and holds it to its byte budget: what does not fit is dropped from the least important end
    (the per-config map loses its kernel names and times first), never the contract's keys."""
    budget = line_budget(world)
    text = json.dumps(line)
    for shed in ("kernel", "kernel_ms", "ms", "pairs"):
        if len(text) <= budget:
            break
        for summary in (line.get("configs") or {}).values():
            summary.pop(shed, None)
        text = json.dumps(line)
    if len(text) > budget:
        for key in ("cpu_baselines", "gather", "single_process", "configs"):
            if len(text) <= budget:
                break
            line[key] = "see " + str(line.get("details"))
            text = json.dumps(line)
    return text


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def last_json_line(text):
    for row in reversed(text.splitlines()):
        if row.startswith("{"):
            try:
                return json.loads(row)
            except ValueError:
                continue
    return None


def error_line(args, message, **more):
    """The one JSON line of a run that could not measure: same keys a reader of the line looks at first, `value` null, `error` set."""
    line = {"metric": "GCUPS (DP cell updates/s) batched Levenshtein", "value": None, "unit": "GCUPS", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "error": message}
    line.update(more)
    return json.dumps(line)


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE around: start the N ranks ourselves. This process has not imported
    torch, loaded the library or made any HIP call -- the ranks are a CHILD process (torch.distributed.run), never an exec --
    and it leaves with the child's exit code; rank 0's JSON line goes straight to our stdout. A run that fails -- a rank that
    dies, RCCL that does not come up, ranks that hang past --launch-timeout -- still ends with ONE JSON line (`error`) and a non-zero
    exit code, within a bounded time: torch.distributed.run tears the other ranks down when one exits, the process group's
    collectives give up after --collective-timeout, and whatever is left is killed here."""
    import signal
    import subprocess
    import threading
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    seen = {"line": False, "tail": []}

    def pump():
        for row in child.stdout:
            sys.stdout.write(row)
            sys.stdout.flush()
            if row.startswith('{"metric"'):
                seen["line"] = True
            seen["tail"] = (seen["tail"] + [row.rstrip()])[-5:]
    reader = threading.Thread(target=pump, daemon=True)
    reader.start()
    timed_out = False
    try:
        code = child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(child.pid, signal.SIGKILL)          # the launcher and every rank it started (its own session: nothing else)
        except ProcessLookupError:
            pass
        code = child.wait()
    reader.join(timeout=10)
    if timed_out or (code != 0 and not seen["line"]):
        why = f"the ranks did not finish within --launch-timeout {args.launch_timeout} s" if timed_out else f"torch.distributed.run exited with code {code} before rank 0 printed its line"
        print(error_line(args, why, stdout_tail=seen["tail"]), flush=True)
    raise SystemExit(code if code != 0 else (1 if timed_out else 0))


def visible_devices(torch):
    return int(torch.cuda.device_count())     # counting devices does not initialise the GPU on this image


def run_single_process(args):
    """One process, ONE scope over N devices (`swh_scope_init_gpus`, include/stringwars_amd.h): the batch is cut into
    cells-balanced shards, shard r prepared on device r (`swh_sharded_prepare_*`), and every step is one
    `swh_levenshtein_pairs_sharded` call -- all shards scored side by side, pieces sent to the first device with
    ncclSend / ncclRecv INSIDE the library (csrc/sharded.hip), self-checked on the scope's first call. The distances land in
    memory of the first device. C2 weak (N x 1 M pairs) by default, `--config c5` for the 100 M strong config."""
    import torch

    import stringwars_amd as sw

    cfg = CONFIGS[args.config]
    workload = args.workload or cfg["workload"]
    strong = cfg["scaling"] == "strong"
    n = args.gpus
    have = visible_devices(torch)
    if not args.share_gpu and have < n:
        raise SystemExit(f"--gpus {n} --single-process but only {have} device(s) are visible")
    devices = [0] * n if args.share_gpu else list(range(n))
    per_gpu = args.pairs or cfg["pairs"]
    total_pairs = per_gpu if strong else per_gpu * n
    started = time.perf_counter()
    a, b = sw.generate_pairs(workload, total_pairs, seed=args.seed)
    cells = int((a.lengths.astype(np.int64) * b.lengths.astype(np.int64)).sum())
    os.environ.setdefault("STRINGWARS_AMD_SHARD_CHECK", "1")   # every call checks its gather (sum + keyed xor per shard)
    scope = sw.DeviceScope(gpu_devices=devices)
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=args.algorithm)
    batch = sw.ShardedPairs(scope, a, b)
    setup_s = time.perf_counter() - started
    torch.cuda.set_device(devices[0])
    out = torch.zeros(total_pairs + 4, dtype=torch.int32, device=torch.device("cuda", devices[0]))

    def step():
        engine.pairs_sharded(batch, scope, out=out)       # synchronous: results visible on return

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    step()
    if args.prewarm_seconds > 0:
        until = time.perf_counter() + args.prewarm_seconds
        while time.perf_counter() < until:
            step()
    for _ in range(args.warmup):
        step()
    sync_all()
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - start
    shard = scope.shard_timing()
    got = out[:total_pairs].cpu().numpy().astype(np.uint32)
    parity = None
    if not args.no_cpu_baseline:
        import oracle  # the checker; never on the timed path
        cuts = batch.cuts
        ok = True
        for r in range(n):                                  # the head of EVERY shard: each device's results reached their place
            count = min(cuts[r + 1] - cuts[r], 4000)
            want = oracle.levenshtein_pairs(a, b, algo="hyyro", first=cuts[r], count=count)
            ok = ok and bool((want == got[cuts[r]:cuts[r] + count]).all())
        parity = ok
    line = {
        "metric": "GCUPS (DP cell updates/s) batched Levenshtein", "value": round(cells * args.steps / elapsed / 1e9, 2), "unit": "GCUPS",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "mode": "single-process",
        "config": {"workload": cfg["text"].format(pairs=total_pairs if strong else per_gpu) + ", shards prepared and resident in HBM",
                   "value_is": "rate of K synchronous swh_levenshtein_pairs_sharded calls on ONE scope over all devices (one process); "
                               "the gather to the first device is part of every call",
                   "devices": devices, "device_count": scope.device_count, "shard_cuts": batch.cuts, "pairs_total": total_pairs, "cells_total": cells,
                   "collective": "ncclSend/ncclRecv groups inside the library (RCCL bound with dlopen), four pieces per shard" if len(set(devices)) > 1
                                 else "members share a device: copies in RCCL's place (testing only)",
                   "self_check": "STRINGWARS_AMD_SHARD_CHECK=" + os.environ["STRINGWARS_AMD_SHARD_CHECK"] + ": a checksum mismatch after the gather fails the call",
                   "setup_s": round(setup_s, 2), "seed": args.seed},
        "shard_timing_last_call": shard, "parity_vs_oracle": parity, "gather_ok": True,   # a failed self-check raises
    }
    print(json.dumps(line), flush=True)


def measure(cfg_name, args, env, steps, warmup, prewarm_seconds, steady_seconds, with_pipelined, pairs_arg=None, gather_cost=False, gather_transport=None):
    """The measurement of one bench config on this rank's GPU (+ the collective at N > 1): returns what the line needs.
    Every rank makes the same calls in the same order (a step contains the collective)."""
    torch, dist, sw, sharding = env["torch"], env["dist"], env["sw"], env["sharding"]
    world, rank, device, comm_device, scope = env["world"], env["rank"], env["device"], env["comm_device"], env["scope"]
    cfg = CONFIGS[cfg_name]
    workload = (args.workload if cfg_name == args.config else None) or cfg["workload"]
    strong = cfg["scaling"] == "strong"
    chunks = args.chunks
    gather_transport = gather_transport or args.gather_transport
    # ---- this rank's shard of the seeded stream -------------------------------------------------------------------
    if strong:
        total_pairs = pairs_arg or cfg["pairs"]
        lo, hi = sharding.shard_range(total_pairs, rank, world)
        if world > 1:
            # cells-balanced cuts without any rank holding all the lengths: per-block cell sums of the count-balanced
            # slices, all-reduced, then cut on their prefix (sharding.shard_ranges_by_block_cells)
            block = 1 << 20
            a, b = sw.generate_pairs(workload, hi - lo, seed=args.seed, first=lo)
            pair_cells = (a.lengths * b.lengths).astype(np.int64)
            sums = np.zeros((total_pairs + block - 1) // block, dtype=np.int64)
            for k in range(lo // block, (hi - 1) // block + 1 if hi > lo else 0):
                sums[k] = pair_cells[max(k * block, lo) - lo:min((k + 1) * block, hi) - lo].sum()
            reduced = torch.from_numpy(sums).to(comm_device)
            dist.all_reduce(reduced)
            ranges = sharding.shard_ranges_by_block_cells(reduced.cpu().numpy(), block, total_pairs, world)
            if ranges[rank] != (lo, hi):
                lo, hi = ranges[rank]
                a, b = sw.generate_pairs(workload, hi - lo, seed=args.seed, first=lo)
        else:
            ranges = [(0, total_pairs)]
            a, b = sw.generate_pairs(workload, total_pairs, seed=args.seed)
        pairs = hi - lo
    else:
        pairs = pairs_arg or cfg["pairs"]
        total_pairs = pairs * world
        ranges = [(r * pairs, (r + 1) * pairs) for r in range(world)]
        a, b = sw.generate_pairs(workload, pairs, seed=args.seed, first=sharding.weak_shard_first(rank, pairs))
    cells = int((a.lengths * b.lengths).sum())
    # per-shard u32 offsets (half the offset traffic of u64) whenever the shard's bytes fit them
    if int(a.offsets[-1]) < 2 ** 32 and int(b.offsets[-1]) < 2 ** 32:
        a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    offsets_dtype = a.offsets.dtype
    as_torch = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else (x.view(np.int32) if x.dtype == np.uint32 else x)).to(device)
    tensors = [as_torch(x) for x in (a.data, a.offsets, b.data, b.offsets)]
    da = sw.DeviceTape(tensors[0].data_ptr(), tensors[1].data_ptr(), a.count, offsets_dtype, keepalive=tensors[:2])
    db = sw.DeviceTape(tensors[2].data_ptr(), tensors[3].data_ptr(), b.count, offsets_dtype, keepalive=tensors[2:])

    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=args.algorithm)
    # prepared once, outside every timed region: resident, measured (the engine then needs no planning pre-pass)
    pa, pb = sw.PreparedTape(scope, da), sw.PreparedTape(scope, db)

    # ---- the step ------------------------------------------------------------------------------------------------------
    # Two result buffers: consecutive steps alternate between them (the pipelined steps need that; the synchronous ones do
    # it too so that both modes run the same calls). In synchronous mode every call returns with its results visible
    # and the step also waits for its gather; in pipelined mode calls are enqueued and only `fence()` waits.
    outs = [torch.zeros(max(pairs, 1), dtype=torch.int32, device=device) for _ in range(2)]
    counter = [0]
    pipelined = [False]
    collective = [True]            # switched off for the compute-only steps that price the gather
    if strong:
        if chunks <= 0:
            chunks = 4 if world > 1 else 1
        pieces = sharding.chunk_ranges(pairs, chunks)
        gathers = [None, None]
        # The root's vector: one per slot, reused from step to step. Transport: u8 when no distance of any shard can exceed
        # 255 (the longest strings are that short: every rank knows its own from the prepared tapes, one all-reduce agrees),
        # a quarter of the bytes over the root's inbound xGMI links; --gather-transport u32 sends the distances as they are.
        fulls = [torch.zeros(total_pairs, dtype=torch.int32, device=comm_device) if rank == 0 and world > 1 else None for _ in range(2)]
        transport = None
        if world > 1:
            longest = torch.tensor([int(max(a.lengths.max(initial=0), b.lengths.max(initial=0)))], dtype=torch.int64, device=comm_device)
            dist.all_reduce(longest, op=dist.ReduceOp.MAX)
            if gather_transport == "u8" and int(longest.item()) > 255:
                raise RuntimeError("gather transport u8 needs every string <= 255 bytes")
            if gather_transport != "u32" and int(longest.item()) <= 255:
                transport = torch.uint8
        # one pre-bound call per (buffer, piece): sub-views of the prepared tapes, the piece's slice of the result buffer
        piece_calls = [[engine.bind_pairs(pa[p_lo:p_hi], pb[p_lo:p_hi], scope, outs[slot][p_lo:p_hi]) if p_hi > p_lo else None
                        for p_lo, p_hi in pieces] for slot in range(2)]

        def step():
            slot = counter[0] & 1
            counter[0] += 1
            if gathers[slot] is not None:
                gathers[slot].wait()            # this buffer's previous gather has left it
            gather = sharding.ChunkedGather(ranges, chunks, torch.int32, device, full=fulls[slot], transport=transport) if world > 1 and collective[0] else None
            for j, (p_lo, p_hi) in enumerate(pieces):
                if p_hi > p_lo:
                    piece_calls[slot][j]()
                if gather is not None:
                    if pipelined[0]:
                        scope.join()            # the send is ordered on torch's stream: make it wait for this piece
                    gather.send_chunk(outs[slot], j)
            gathers[slot] = gather              # (waited for when this buffer comes round again, and by the final fence)
    else:
        pieces = [(0, pairs)]
        transport = None
        gathered = [[torch.zeros(pairs, dtype=torch.int32, device=comm_device) for _ in range(world)] for _ in range(2)] \
            if rank == 0 and world > 1 else [None, None]
        works = [None, None]
        gathers = works
        calls = [engine.bind_pairs(pa, pb, scope, outs[slot]) for slot in range(2)]

        def step():
            slot = counter[0] & 1
            counter[0] += 1
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
            calls[slot]()               # engine.pairs(pa, pb, scope, out=outs[slot]) with its arguments bound once
            if world > 1 and collective[0]:
                if pipelined[0]:
                    scope.join()        # the gather is ordered on torch's stream: make that stream wait for this call
                if args.backend == "nccl":
                    works[slot] = dist.gather(outs[slot], gathered[slot], dst=0, async_op=True)   # overlaps the next step's call
                else:
                    dist.gather(outs[slot].cpu(), gathered[slot], dst=0)

    def fence():
        for slot in range(2):
            if gathers[slot] is not None:
                gathers[slot].wait()
                if not strong:
                    gathers[slot] = None
        scope.synchronize()         # both pipeline lanes
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(count):
        fence()
        start = time.perf_counter()
        for _ in range(count):
            step()
        fence()
        return time.perf_counter() - start

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # An idle MI355X takes a few hundred milliseconds of work to reach its clocks (profiles/r2: the same 20 steps measure
    # 8 % apart right after start-up and after a second of calls). The device is brought there first -- every rank runs the same
    # number of untimed steps, the collective included --, then come the W warm-up steps and the K timed ones.
    if prewarm_seconds > 0:
        fence()
        start = time.perf_counter()
        for _ in range(8):                      # what a step costs here
            step()
        fence()
        per_step = (time.perf_counter() - start) / 8
        extra = int(min(max(prewarm_seconds / max(per_step, 1e-6) - 8, 0), 100000))
        if world > 1:                           # the same count on every rank: a step contains the collective
            agreed = torch.tensor([extra], dtype=torch.int64, device=comm_device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
            extra = int(agreed.item())
        for i in range(extra):
            step()
            if i % 64 == 63:
                fence()
    # ---- `value`: K synchronous steps -----------------------------------------------------------------------------------
    for _ in range(warmup):
        step()
    elapsed = max_over_ranks(timed_region(steps))
    if world > 1:
        c = torch.tensor([cells], dtype=torch.int64, device=comm_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = int(c.item())
    else:
        total_cells = cells
    last_slot = (counter[0] - 1) & 1
    sync_result = outs[last_slot][:pairs].cpu().numpy().astype(np.uint32)   # produced by the timed steps

    # ---- every rank's slice of the gathered vector, against a checksum the rank computed locally ---------------------
    gather_ok = None
    if world > 1:
        mine = torch.tensor([int(sync_result.astype(np.int64).sum()), int(np.bitwise_xor.reduce(sync_result)) if pairs else 0],
                            dtype=torch.int64, device=comm_device)
        sums = [torch.zeros(2, dtype=torch.int64, device=comm_device) for _ in range(world)]
        dist.all_gather(sums, mine)
        if rank == 0:
            if strong:
                full = gathers[last_slot].full.cpu().numpy().astype(np.uint32)
                slices = [full[l:h] for l, h in ranges]
            else:
                slices = [g.cpu().numpy().astype(np.uint32) for g in gathered[last_slot]]
            gather_ok = all(int(s.astype(np.int64).sum()) == int(sums[r][0]) and
                            (int(np.bitwise_xor.reduce(s)) if s.size else 0) == int(sums[r][1]) for r, s in enumerate(slices))
            gather_ok = bool(gather_ok and (slices[0] == sync_result).all())

    # ---- what the gather costs a step: the same K steps with the collective switched off -------------------------------
    gather = None
    if world > 1 and gather_cost:
        collective[0] = False
        fence()
        compute_only = max_over_ranks(timed_region(steps))
        collective[0] = True
        # and one step's gather by itself: results already there, every piece sent, waited for (enqueue -> complete on this rank)
        fence()
        start = time.perf_counter()
        slot = counter[0] & 1
        if strong:
            alone = sharding.ChunkedGather(ranges, chunks, torch.int32, device, full=fulls[slot], transport=transport)
            for j in range(len(pieces)):
                alone.send_chunk(outs[slot], j)
            alone.wait()
        elif args.backend == "nccl":
            dist.gather(outs[slot], gathered[slot], dst=0)
        else:
            dist.gather(outs[slot].cpu(), gathered[slot], dst=0)
        torch.cuda.synchronize()
        alone_s = max_over_ranks(time.perf_counter() - start)
        gather = {"exposed_ms_per_step": round((elapsed - compute_only) / steps * 1e3, 4),
                  "exposed_is": "ms per step of the K timed steps minus the same K steps without the collective (max over ranks both)",
                  "alone_ms": round(alone_s * 1e3, 4), "alone_is": "one step's gather by itself, enqueue to completion, max over ranks",
                  "transport": "u8 over the links, widened to u32 on the root" if transport is not None else "u32",
                  "bytes_to_root_per_step": int((1 if transport is not None else 4) * (total_pairs - (ranges[0][1] - ranges[0][0]))),
                  "compute_only_ms_per_step": round(compute_only / steps * 1e3, 4)}

    # ---- the same steps with the library's hipEvent pairs on (kernel durations on the kernels' own streams) ----------
    # They cost a few microseconds per call, so `value` above comes from the run without them; same calls, same conditions
    # -- what `rocprofv3 --kernel-trace --stats` of `bench.py --no-pipelined --no-configs --no-cpu-baseline` averages over.
    scope.set_profiling(True)
    profiled_elapsed = max_over_ranks(timed_region(steps))
    totals = scope.timing_totals()
    sync_timing = scope.last_timing()
    scope.set_profiling(False)

    # ---- steady state: the same synchronous steps for at least --steady-seconds ---------------------------------------
    steady = None
    if steady_seconds > 0:
        steps_for = max(int(steady_seconds / max(elapsed / steps, 1e-6) * 1.05) + 1, steps)
        if world > 1:
            agreed = torch.tensor([steps_for], dtype=torch.int64, device=comm_device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
            steps_for = int(agreed.item())
        steady_elapsed = max_over_ranks(timed_region(steps_for))
        steady = {"steps": steps_for, "seconds": round(steady_elapsed, 4), "value": round(total_cells * steps_for / steady_elapsed / 1e9, 2)}

    # ---- pipelined: K steps enqueued asynchronously on two internal lanes (host work of step i+1 overlaps step i) ----
    pipelined_rate, pipelined_result, pipelined_ms = None, None, None
    if with_pipelined:
        scope.set_async(True)
        scope.set_pipelined(True)
        pipelined[0] = True
        for _ in range(max(warmup, 4)):
            step()
        pipelined_elapsed = max_over_ranks(timed_region(steps))
        pipelined_ms = pipelined_elapsed / steps * 1e3
        pipelined_rate = round(total_cells * steps / pipelined_elapsed / 1e9, 2)
        pipelined_result = outs[(counter[0] - 1) & 1][:pairs].cpu().numpy().astype(np.uint32)
        pipelined[0] = False
        scope.set_pipelined(False)
        scope.set_async(False)
    fence()
    return dict(cfg=cfg, workload=workload, strong=strong, a=a, b=b, pairs=pairs, total_pairs=total_pairs, cells=cells, total_cells=total_cells,
                offsets_dtype=offsets_dtype, n_pieces=len(pieces), elapsed=elapsed, steps=steps, sync_result=sync_result, gather_ok=gather_ok,
                gather=gather, totals=totals, sync_timing=sync_timing, profiled_elapsed=profiled_elapsed, steady=steady, pipelined_rate=pipelined_rate,
                pipelined_result=pipelined_result, pipelined_ms=pipelined_ms, ranges=ranges)


def collective_text(strong, world):
    if world == 1:
        return "none"
    return ("ncclSend/ncclRecv group per piece to rank 0 (variable-size gather of u32 distances)" if strong
            else "RCCL gather of u32 distances to rank 0")


def main():
    args = parse_args()
    try:
        return run(args)
    except SystemExit:
        raise
    except BaseException as error:   # a rank that cannot go on says so in a JSON line (rank 0's is THE line; the launcher adds one if none came)
        rank = int(os.environ.get("RANK", "0"))
        # ONE line on stdout: rank 0's (the launcher adds one if rank 0 never spoke); the other ranks say theirs on stderr
        print(error_line(args, f"rank {rank}: {type(error).__name__}: {error}", rank=rank), file=sys.stdout if rank == 0 else sys.stderr, flush=True)
        import traceback
        traceback.print_exc()
        os._exit(1)        # not sys.exit: a process group whose peer is gone can hang in its destructors


def run(args):
    world_env = os.environ.get("WORLD_SIZE")
    if args.single_process:
        if world_env is not None and int(world_env) > 1:
            raise SystemExit(f"--single-process is ONE process over {args.gpus} devices, but WORLD_SIZE={world_env}")
        return run_single_process(args)
    if args.gpus > 1 and world_env is None:
        return self_launch(args)               # before torch, the library or any HIP call
    world = int(world_env or "1")
    if world != args.gpus:                     # never a silent measurement of fewer GPUs than asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure a different number of GPUs")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.die_rank == rank and args.die_at == "start":
        os._exit(3)                            # test hook: a rank that never comes up
    if world > 1:
        # torch.distributed.run ends the other ranks with SIGTERM when one of them exits: say why before going
        import signal

        def terminated(signum, frame):
            print(error_line(args, f"rank {rank}: terminated by signal {signum} (another rank failed -- its own message is on stderr --, or the launcher gave up)",
                             rank=rank), file=sys.stdout if rank == 0 else sys.stderr, flush=True)
            os._exit(1)
        signal.signal(signal.SIGTERM, terminated)

    import datetime

    import torch
    import torch.distributed as dist

    import stringwars_amd as sw
    from stringwars_amd import sharding

    if not args.share_gpu and visible_devices(torch) < (local_rank + 1 if world > 1 else 1):
        raise SystemExit(f"rank {rank}: device {local_rank} is not there ({visible_devices(torch)} visible)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # a stream of our own: on the legacy default stream the library's side stream (wavefront class kernels run on two streams)
    # would serialise with it instead of overlapping
    torch.cuda.set_stream(torch.cuda.Stream(device))

    if args.only_config:
        scope = sw.DeviceScope(gpu_device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
        entry = run_leg(args.only_config, sw, scope, torch, device, args.seed, load_pmc_constants(), calls=args.calls,
                        pairs_override=args.leg_pairs, check=not args.no_cpu_baseline, algorithm=args.algorithm)
        print(json.dumps(entry), flush=True)
        return

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        limit = datetime.timedelta(seconds=args.collective_timeout)   # a collective whose peer is gone raises instead of waiting for ever
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=limit)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo", timeout=limit)
        if args.die_rank == rank and args.die_at == "after-init":
            os._exit(3)                        # test hook: a rank that dies once the group is up
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    comm_device = device if args.backend == "nccl" else torch.device("cpu")
    scope = sw.DeviceScope(gpu_device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    env = dict(torch=torch, dist=dist, sw=sw, sharding=sharding, world=world, rank=rank, device=device, comm_device=comm_device, scope=scope)

    # ---- who is here: ranks, devices, the collective library ------------------------------------------------------------
    ranks_seen = None
    if world > 1:
        props = torch.cuda.get_device_properties(device)
        me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "device": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": int(getattr(props, "pci_bus_id", -1))}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        version = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        ranks_seen = {"world_size": dist.get_world_size(), "backend": args.backend, "rccl_version": version,
                      "distinct_devices": len({(e["uuid"], e["pci_bus_id"]) for e in everyone}), "ranks": everyone}

    if args.die_rank == rank and args.die_at == "measure":
        os._exit(3)                            # test hook: a rank that dies while the others are in their first collective
    head = measure(args.config, args, env, args.steps, args.warmup, args.prewarm_seconds, args.steady_seconds, not args.no_pipelined,
                   pairs_arg=args.pairs, gather_cost=world > 1)

    # ---- N > 1: BASELINE configs[4] beside the headline -- the 100 M short-word pairs, strong scaling -------------------
    c5 = None
    if world > 1 and args.config == "c2" and not args.no_configs:
        c5_pairs = args.c5_pairs or CONFIGS["c5"]["pairs"]
        # the entry is what the north star names -- the u32 distances gathered as they are --; the same steps with bytes on the wire
        # (no distance of word-sized pairs exceeds 255; widened on the root) ride beside it as `gather_u8`
        c5_u8 = None
        try:
            c5 = measure("c5", args, env, args.steps, min(args.warmup, 2), min(args.prewarm_seconds, 0.25), 0.0, False, pairs_arg=c5_pairs, gather_cost=True,
                         gather_transport="u32")
        except Exception as error:   # must not cost the line its headline; every rank raises or none (same calls everywhere)
            c5 = {"error": f"{type(error).__name__}: {error}"}
        if "error" not in c5:
            try:
                torch.cuda.empty_cache()
                c5_u8 = measure("c5", args, env, args.steps, min(args.warmup, 2), min(args.prewarm_seconds, 0.25), 0.0, False, pairs_arg=c5_pairs,
                                gather_cost=True, gather_transport="u8")
            except Exception as error:
                c5_u8 = {"error": f"{type(error).__name__}: {error}"}

    line = None
    if rank == 0:
        constants = load_pmc_constants()
        cfg, strong, pairs, cells = head["cfg"], head["strong"], head["pairs"], head["cells"]
        n_pieces = head["n_pieces"]
        totals, sync_timing = head["totals"], head["sync_timing"]
        calls_timed = max(totals["calls"], 1)
        kernel_ms = totals["compute_ms"] / calls_timed
        # The event pairs slow a step down (a few microseconds per call, 8 % of a step of C5 at 100 M pairs): a kernel cannot outlast the
        # step that contains it, so where the profiled repeat's kernel time exceeds the timed steps' ms_per_step it is brought to their scale
        profiled_ms_per_step = head["profiled_elapsed"] / args.steps * 1e3
        unprofiled_ms_per_step = head["elapsed"] / args.steps * 1e3
        # (only then: where the profiled repeat's kernel time fits inside the timed steps it is reported as measured -- the figure the
        # committed `rocprofv3 --kernel-trace --stats` average must agree with)
        # (advisor, round 5: the headline kernel time is the MEASURED one; the figure brought to the timed steps' scale rides beside it)
        scaled_ms = None
        if kernel_ms > unprofiled_ms_per_step and profiled_ms_per_step > 0:
            scaled_ms = kernel_ms * min(1.0, unprofiled_ms_per_step / profiled_ms_per_step)
        extra = {"all_kernels_ms": round(totals["total_ms"] / calls_timed, 4), "launches_timed": totals["calls"],
                 "profiled_repeat_ms_per_step": round(profiled_ms_per_step, 4),
                 "measured": "hipEvents inside the library over a repeat of the K timed synchronous steps (average per call), on the kernel's own stream"}
        if scaled_ms is not None:
            extra["kernel_ms_scaled_to_step"] = round(scaled_ms, 4)
        roofline = roofline_of(sync_timing["dominant_name"], kernel_ms, int(cells / n_pieces), int(sync_timing["bytes"]), head["workload"],
                               pairs // n_pieces, constants, extra=extra)
        roofline["unit_note"] = "bound = integer VALU issue (not MFMA: min-plus has no dense contraction; not HBM: see `hbm`)"
        parity = None
        cpu_baseline, cpu_baselines = None, None
        if not args.no_cpu_baseline:
            import oracle  # checker + reported baseline only; never on the timed GPU path
            check = min(pairs, 20_000)   # (rank 0's shard; the other ranks' slices are covered by `gather_ok`)
            want = oracle.levenshtein_pairs(head["a"], head["b"], algo="hyyro", count=check)
            parity = bool((want == head["sync_result"][:check]).all() and
                          (head["pipelined_result"] is None or (head["pipelined_result"] == head["sync_result"]).all()))
        leg_entries = None
        if world == 1 and not args.no_configs:
            leg_entries = []
            for leg_name in [n for n in args.legs.split(",") if n]:
                try:
                    leg_entries.append(run_leg(leg_name, sw, scope, torch, device, args.seed, constants, pairs_override=args.leg_pairs,
                                               check=not args.no_cpu_baseline))
                except Exception as error:   # one config failing must not cost the line its headline
                    leg_entries.append({"config": leg_name, "error": f"{type(error).__name__}: {error}"})
                torch.cuda.empty_cache()
        elif c5 is not None:
            if "error" in c5:
                leg_entries = [{"config": "c5_strong", **c5}]
            else:
                c5_parity = None
                if not args.no_cpu_baseline:
                    check = min(c5["pairs"], 200_000)
                    want = oracle.levenshtein_pairs(c5["a"], c5["b"], algo="hyyro", count=check)
                    c5_parity = bool((want == c5["sync_result"][:check]).all())
                c5_calls = max(c5["totals"]["calls"], 1)
                leg_entries = [{
                    "config": "c5_strong", "workload": CONFIGS["c5"]["text"].format(pairs=c5["total_pairs"]) + ", tapes prepared and resident in HBM",
                    "scaling": "strong", "n_gpus": world, "pairs_total": c5["total_pairs"], "pairs_rank0": c5["pairs"], "cells_total": c5["total_cells"],
                    "shard_ranges": [list(r) for r in c5["ranges"]], "pieces_per_step": c5["n_pieces"],
                    "value": round(c5["total_cells"] * c5["steps"] / c5["elapsed"] / 1e9, 2), "unit": "GCUPS", "steps": c5["steps"],
                    "ms_per_step": round(c5["elapsed"] / c5["steps"] * 1e3, 4),
                    "value_is": "K synchronous steps, barrier + synchronize on both sides, max over ranks: every rank scores its cells-balanced "
                                "shard in pieces and sends each piece to its place in rank 0's vector while the next is scored",
                    "collective": collective_text(True, world), "gather_ok": c5["gather_ok"], "gather": c5["gather"], "ranks_seen": ranks_seen,
                    "kernel_ms_rank0_per_step": round(c5["totals"]["compute_ms"] / c5_calls * c5["n_pieces"], 4),
                    "dominant_kernel": kernel_family(c5["sync_timing"]["dominant_name"]), "parity_vs_oracle": c5_parity,
                    "parity_sample": "first 200000 pairs of rank 0's shard against oracle/; every other rank's slice by checksum (gather_ok)"}]
                if c5_u8 is not None:
                    leg_entries[0]["gather_u8"] = c5_u8 if "error" in c5_u8 else {
                        "value": round(c5_u8["total_cells"] * c5_u8["steps"] / c5_u8["elapsed"] / 1e9, 2), "unit": "GCUPS",
                        "ms_per_step": round(c5_u8["elapsed"] / c5_u8["steps"] * 1e3, 4), "gather_ok": c5_u8["gather_ok"], "gather": c5_u8["gather"],
                        "same_results": bool((c5_u8["sync_result"] == c5["sync_result"]).all()),
                        "is": "the same steps with the distances travelling as bytes (a quarter of the u32 bytes over rank 0's inbound links), widened on the root"}
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed at N = 1 only
            cpu_baselines = cpu_rows(head["a"], head["b"], budget_s=args.cpu_seconds)
            cpu_baseline = {k: v for k, v in cpu_baselines[0].items() if k != "name"}
        elapsed, steady = head["elapsed"], head["steady"]
        ms_per_step = elapsed / args.steps * 1e3
        # the full entries (--details-out, stderr) ...
        details = {
            "headline": {"config": args.config, "workload": cfg["text"].format(pairs=head["total_pairs"] if strong else pairs) + ", tapes prepared and resident in HBM",
                         "value_is": "rate of the K timed steps; a step is one synchronous call (results visible on return; the reference's "
                                     "compute_into metric, utils.rs:721-799); a call returns when every result has been written through and "
                                     "acknowledged (the kernel's summary in host-mapped memory), "
                                     + ("as built" if os.environ.get("STRINGWARS_AMD_EARLY_RETURN", "1") != "0" else "switched off: it waits for the stream")
                                     + " (DESIGN.md 3)" + (" followed by the gather of the distances to rank 0, which overlaps the next step's call" if world > 1 else ""),
                         "value_steady_is": f"the same steps over >= {args.steady_seconds} s: {steady}" if steady else None,
                         "value_pipelined_is": "K steps enqueued asynchronously on two internal lanes" + (f", {round(head['pipelined_ms'], 4)} ms per step" if head["pipelined_ms"] else ""),
                         "device_prewarm_s": args.prewarm_seconds, "roofline": roofline, "gather": head["gather"]},
            "configs": leg_entries or [], "cpu_baselines": cpu_baselines or [], "ranks_seen": ranks_seen,
        }
        # ... and the headline line: the contract's keys, every number of the roofline object, one summary per config
        line = {
            "metric": "GCUPS (DP cell updates/s) batched Levenshtein", "value": round(head["total_cells"] * args.steps / elapsed / 1e9, 2),
            "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "value_steady": steady["value"] if steady else None, "value_pipelined": head["pipelined_rate"],
            "config": {"workload": cfg["text"].format(pairs=head["total_pairs"] if strong else pairs) + ", tapes prepared and resident in HBM",
                       "step": "one synchronous call (results visible on return)" + (" + gather of the u32 distances to rank 0" if world > 1 else ""),
                       "pairs_per_gpu": pairs, "pairs_total": head["total_pairs"], "cells_per_gpu": cells, "algorithm": args.algorithm,
                       "offsets": str(head["offsets_dtype"]), "pieces_per_step": n_pieces, "collective": collective_text(strong, world), "seed": args.seed},
            "roofline": compact_roofline(roofline),
            "cpu_baseline": {k: (v[:200] if isinstance(v, str) else v) for k, v in cpu_baseline.items()} if cpu_baseline else None,
            "cpu_baselines": {row["name"]: {"value": row["value"], "cores": row["cores"]} for row in cpu_baselines} if cpu_baselines else None,
            "configs": {entry["config"]: compact_leg(entry) for entry in leg_entries} if leg_entries is not None else None,
            "parity_vs_oracle": parity, "gather_ok": head["gather_ok"],
            "gather": {k: head["gather"][k] for k in ("exposed_ms_per_step", "alone_ms", "transport", "bytes_to_root_per_step", "compute_only_ms_per_step")} if head["gather"] else None,
            "ranks_seen": {k: ranks_seen[k] for k in ("world_size", "backend", "rccl_version", "distinct_devices")} if ranks_seen else None,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None and world > 1 and not args.no_single_process:
        # The in-library multi-device path (one scope over all N devices, ncclSend / ncclRecv inside csrc/sharded.hip) on the same
        # devices, once the ranks are done with them: a CHILD process of rank 0 (never an exec), bounded by a timeout so that a
        # hang there cannot cost the line. The other ranks have nothing left to do and exit.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--config", args.config, "--seed", str(args.seed), "--prewarm-seconds", str(min(args.prewarm_seconds, 0.25))]
        if args.pairs:
            cmd += ["--pairs", str(args.pairs)]
        if args.share_gpu:
            cmd.append("--share-gpu")
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        child_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                        "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            done = subprocess.run(cmd, env=child_env, capture_output=True, text=True, timeout=args.single_process_timeout)
            child = last_json_line(done.stdout)
            if done.returncode != 0 or child is None:
                child = {"error": f"exit code {done.returncode}", "stderr_tail": done.stderr[-600:]}
        except subprocess.TimeoutExpired:
            child = {"error": f"no line within {args.single_process_timeout} s"}
        details["single_process"] = child
        line["single_process"] = compact_single_process(child)
    if line is not None:
        for entry in details["configs"]:                     # one short line per config, before the headline line
            print(json.dumps({"leg": entry["config"], **compact_leg(entry)}), flush=True)
        line["details"] = write_details(args.details_out, details)
        print(fit_line(line, world), flush=True)


if __name__ == "__main__":
    main()
