"""Deterministic synthetic workloads for the BASELINE.json configs.

Thin wrapper over `swh_synth_generate` (stringwars_amd/csrc/synth.cpp): pair i depends only on
(workload, seed, i), so any rank can generate its own shard of the same global stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from .engines import Strs

WORKLOADS = {
    "words16": 1,      # C1
    "tokens64": 2,     # C2 (headline)
    "utf8_lines": 3,   # C3
    "protein4k": 4,    # C4
    "short_words": 5,  # C5
    "script_lines": 6, # unrelated article lines, one script each (the UTF-8 engine's cross-product regime)
    "bytes4k": 40,     # C4, full byte alphabet
}
AMINO_ACIDS = b"ACDEFGHIKLMNPQRSTVWY"


def generate_pairs(workload, count: int, seed: int = 42, first: int = 0, threads: int = 0) -> Tuple[Strs, Strs]:
    """Returns (a, b) host tapes with u64 offsets. STRINGWARS_SEED defaults to 42 (bench.py:848)."""
    wid = WORKLOADS[workload] if isinstance(workload, str) else int(workload)
    out = N.Synth()
    err = C.c_char_p()
    N.check(N.lib.swh_synth_generate(wid, seed, first, count, threads, C.byref(out), C.byref(err)), err)
    try:
        def grab(data_ptr, offs_ptr):
            offsets = np.ctypeslib.as_array(C.cast(offs_ptr, C.POINTER(C.c_uint64)), shape=(count + 1,)).copy()
            total = int(offsets[-1])
            data = (np.ctypeslib.as_array(C.cast(data_ptr, C.POINTER(C.c_uint8)), shape=(total,)).copy()
                    if total else np.zeros(0, np.uint8))
            return Strs(data=data, offsets=offsets)
        return grab(out.data_a, out.offsets_a), grab(out.data_b, out.offsets_b)
    finally:
        N.lib.swh_synth_free(C.byref(out))


def substitution_matrix(seed: int = 42, alphabet: Optional[bytes] = AMINO_ACIDS) -> np.ndarray:
    matrix = np.zeros((256, 256), dtype=np.int8)
    N.lib.swh_synth_matrix(seed, alphabet, matrix.ctypes.data)
    return matrix


def unary_class_costs(match: int, mismatch: int):
    """`unary_class_costs(match, mismatch)` (bench.rs:95-108, bench.py:308-318)."""
    byte_to_class = np.zeros(256, dtype=np.uint8)
    costs = np.zeros((32, 32), dtype=np.int8)
    N.lib.swh_unary_class_costs(match, mismatch, byte_to_class.ctypes.data, costs.ctypes.data)
    return byte_to_class, costs
