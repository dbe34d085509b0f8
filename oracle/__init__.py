"""ctypes loader of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py. Nothing under stringwars_amd/ imports this package. PARITY UNPINNED by the reference
(it ships no tests and no arithmetic for this path): see the header of oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB


build()
# ORACLE_LIBRARY: another build of the same source, e.g. `make -C oracle asan` -> liboracle_asan.so under LD_PRELOAD=libasan
# (tests/test_sanitizers.py)
lib = C.CDLL(os.environ.get("ORACLE_LIBRARY") or _LIB)
_u8p, _vp = C.c_void_p, C.c_void_p
lib.orc_utf8_decode.restype = C.c_long
lib.orc_utf8_decode.argtypes = [_u8p, C.c_size_t, _vp]
for _n in ("orc_lev_bytes", "orc_hyyro_bytes", "orc_lev_u32"):
    getattr(lib, _n).restype = C.c_uint32
    getattr(lib, _n).argtypes = [_vp, C.c_size_t, _vp, C.c_size_t]
lib.orc_lev_costs_bytes.restype = C.c_int64
lib.orc_lev_costs_bytes.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
for _n in ("orc_nw_score", "orc_sw_score"):
    getattr(lib, _n).restype = C.c_int64
    getattr(lib, _n).argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int]
lib.orc_lev_pairs.restype = C.c_long
lib.orc_lev_pairs.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_uint32, _vp]
lib.orc_nw_pairs.restype = C.c_long
lib.orc_nw_pairs.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, C.c_size_t, _vp, C.c_int, C.c_int, _vp]
lib.orc_lev_costs_pairs.restype = C.c_long
lib.orc_lev_costs_pairs.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
lib.orc_align_score_general.restype = C.c_int64
lib.orc_align_score_general.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int, C.c_int]
lib.orc_lev_antidiagonal.restype = C.c_uint32
lib.orc_lev_antidiagonal.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t]
for _n in ("orc_selfcheck_levenshtein", "orc_selfcheck_alignment"):
    getattr(lib, _n).restype = C.c_long
    getattr(lib, _n).argtypes = [C.c_uint64, C.c_size_t, C.c_uint32, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_uint64)]
lib.orc_cells.restype = C.c_uint64
lib.orc_cells.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, C.c_int]

UNBOUNDED = 0xFFFFFFFF


def _buf(x):
    if isinstance(x, str):
        x = x.encode("utf-8")
    return np.frombuffer(bytes(x), dtype=np.uint8) if not isinstance(x, np.ndarray) else x


def levenshtein(a, b, algo: str = "wf") -> int:
    a, b = _buf(a), _buf(b)
    fn = lib.orc_hyyro_bytes if algo == "hyyro" else lib.orc_lev_bytes
    return int(fn(a.ctypes.data, a.size, b.ctypes.data, b.size))


def utf8_decode(s) -> np.ndarray:
    s = _buf(s)
    out = np.zeros(max(s.size, 1), dtype=np.uint32)
    n = lib.orc_utf8_decode(s.ctypes.data, s.size, out.ctypes.data)
    if n < 0:
        raise ValueError("invalid UTF-8")
    return out[:n]


def levenshtein_utf8(a, b) -> int:
    ca, cb = utf8_decode(a), utf8_decode(b)
    return int(lib.orc_lev_u32(ca.ctypes.data, ca.size, cb.ctypes.data, cb.size))


def levenshtein_costs(a, b, match, mismatch, open, extend) -> int:
    a, b = _buf(a), _buf(b)
    return int(lib.orc_lev_costs_bytes(a.ctypes.data, a.size, b.ctypes.data, b.size, match, mismatch, open, extend))


def nw_score(a, b, matrix: np.ndarray, open: int, extend: int, local: bool = False) -> int:
    a, b = _buf(a), _buf(b)
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    fn = lib.orc_sw_score if local else lib.orc_nw_score
    return int(fn(a.ctypes.data, a.size, b.ctypes.data, b.size, m.ctypes.data, open, extend))


def align_score_general(a, b, matrix: np.ndarray, open: int, extend: int, local: bool = False) -> int:
    """The second, independent alignment scorer (Waterman-Smith-Beyer general-gap table, cubic)."""
    a, b = _buf(a), _buf(b)
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    return int(lib.orc_align_score_general(a.ctypes.data, a.size, b.ctypes.data, b.size, m.ctypes.data, open, extend, int(local)))


def levenshtein_antidiagonal(a, b) -> int:
    a, b = _buf(a), _buf(b)
    return int(lib.orc_lev_antidiagonal(a.ctypes.data, a.size, b.ctypes.data, b.size))


def selfcheck(kind: str, seed: int, cases: int, alphabet: int, max_len: int):
    """Runs the C cross-check loop; returns (disagreements, index of the first one or -1, DP cells covered)."""
    fn = lib.orc_selfcheck_levenshtein if kind == "levenshtein" else lib.orc_selfcheck_alignment
    first, cells = C.c_long(-1), C.c_uint64(0)
    bad = fn(seed, cases, alphabet, max_len, C.byref(first), C.byref(cells))
    return int(bad), int(first.value), int(cells.value)


def _width(offsets: np.ndarray) -> int:
    return 8 if offsets.dtype == np.uint64 else 4


def levenshtein_pairs(a, b, utf8=False, algo="wf", bound=None, first=0, count=None) -> np.ndarray:
    """a, b: objects with .data (uint8) and .offsets (uint32/uint64) numpy arrays (stringwars_amd.Strs)."""
    n = len(a.offsets) - 1
    count = n - first if count is None else count
    out = np.zeros(n, dtype=np.uint32)
    rc = lib.orc_lev_pairs(a.data.ctypes.data, a.offsets.ctypes.data, b.data.ctypes.data, b.offsets.ctypes.data,
                           _width(a.offsets), first, count, int(utf8), 1 if algo == "hyyro" else 0,
                           UNBOUNDED if bound is None else bound, out.ctypes.data)
    if rc != 0:
        raise ValueError(f"invalid UTF-8 in pair {-rc - 1}")
    return out[first:first + count]


def nw_pairs(a, b, matrix, open, extend, first=0, count=None) -> np.ndarray:
    n = len(a.offsets) - 1
    count = n - first if count is None else count
    out = np.zeros(n, dtype=np.int64)
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    lib.orc_nw_pairs(a.data.ctypes.data, a.offsets.ctypes.data, b.data.ctypes.data, b.offsets.ctypes.data,
                     _width(a.offsets), first, count, m.ctypes.data, open, extend, out.ctypes.data)
    return out[first:first + count]


def levenshtein_costs_pairs(a, b, match, mismatch, open, extend) -> np.ndarray:
    n = len(a.offsets) - 1
    out = np.zeros(n, dtype=np.int64)
    lib.orc_lev_costs_pairs(a.data.ctypes.data, a.offsets.ctypes.data, b.data.ctypes.data, b.offsets.ctypes.data,
                            _width(a.offsets), 0, n, match, mismatch, open, extend, out.ctypes.data)
    return out


def cells(a, b, utf8=False) -> int:
    n = len(a.offsets) - 1
    return int(lib.orc_cells(a.data.ctypes.data, a.offsets.ctypes.data, b.data.ctypes.data, b.offsets.ctypes.data,
                             _width(a.offsets), n, int(utf8)))
