"""ctypes binding of ``libstringwars_amd.so`` (the C ABI of ``include/stringwars_amd.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C stringwars_amd/csrc``.
There is no Python or CPU fallback: if the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STRINGWARS_AMD_LIBRARY: another build of the same C ABI -- the tests that need a test hook point it at libstringwars_amd_test.so
# (built with -DSWH_TEST_HOOKS, `make -C stringwars_amd/csrc test-lib`) in a child process; nothing else sets it.
LIBRARY_PATH = os.environ.get("STRINGWARS_AMD_LIBRARY") or os.path.join(_HERE, "libstringwars_amd.so")
TEST_LIBRARY_PATH = os.path.join(_HERE, "libstringwars_amd_test.so")

if not os.path.exists(LIBRARY_PATH):
    raise ImportError(
        f"{LIBRARY_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C stringwars_amd/csrc` (hipcc, --offload-arch=gfx950). There is no fallback path."
    )



def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so; if this library
    pulled in /opt/rocm's copy first, a later `import torch` would find a foreign runtime already bound to
    the SONAME and report "No HIP GPUs are available". So when torch is installed, map its copy first and
    let our DT_NEEDED resolve to it. Without torch (C++ harness, plain ctypes users) the system runtime is used."""
    if os.environ.get("STRINGWARS_AMD_SYSTEM_HIP") == "1":
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    candidate = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(candidate):
        try:
            C.CDLL(candidate, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_preload_torch_hip_runtime()
lib = C.CDLL(LIBRARY_PATH)

SUCCESS = 0
STATUS_NAMES = {
    0: "success", 1: "bad_alloc", 2: "invalid_argument", 3: "invalid_utf8", 4: "unsupported_length",
    5: "no_device", 6: "device_error", 7: "not_implemented", 8: "rccl_error",
}
UNBOUNDED = 0xFFFFFFFF
ALGORITHM_AUTO, ALGORITHM_WAVEFRONT, ALGORITHM_BITPARALLEL, ALGORITHM_TILED = 0, 1, 2, 3


class TapeU32(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("count", C.c_size_t)]


class TapeU64(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("count", C.c_size_t)]


class PreparedInfo(C.Structure):
    """``swh_prepared_info_t``"""
    _fields_ = [("count", C.c_size_t), ("bytes", C.c_uint64), ("symbols", C.c_uint64), ("longest", C.c_uint32),
                ("utf8", C.c_int), ("ascii", C.c_int)]


class ShardTiming(C.Structure):
    """``swh_shard_timing_t``"""
    _fields_ = [("compute_ms", C.c_double), ("gather_ms", C.c_double), ("cells", C.c_uint64), ("pairs", C.c_uint64)]


class PreparedView(C.Structure):
    """``swh_prepared_view_t``: strings [first, first + count) of a prepared tape."""
    _fields_ = [("tape", C.c_void_p), ("first", C.c_size_t), ("count", C.c_size_t)]


class Timing(C.Structure):
    _fields_ = [
        ("total_ms", C.c_double), ("dominant_ms", C.c_double), ("compute_ms", C.c_double),
        ("dominant_name", C.c_char * 64),
        ("cells", C.c_uint64), ("bytes", C.c_uint64), ("kernels", C.c_uint32),
    ]


class TimingTotals(C.Structure):
    """``swh_timing_totals_t``: sums over the profiled calls (synchronous, asynchronous and pipelined)."""
    _fields_ = [("total_ms", C.c_double), ("dominant_ms", C.c_double), ("compute_ms", C.c_double), ("calls", C.c_uint64)]


class Synth(C.Structure):
    _fields_ = [
        ("data_a", C.c_void_p), ("offsets_a", C.c_void_p), ("data_b", C.c_void_p), ("offsets_b", C.c_void_p),
        ("count", C.c_size_t),
    ]


_ERR = C.POINTER(C.c_char_p)
_P = C.c_void_p

# name -> (restype, argtypes); exactly the declarations of include/stringwars_amd.h + _harness.h
SIGNATURES = {
    "swh_scope_init_gpu": (C.c_int, [C.c_int, C.POINTER(_P), _ERR]),
    "swh_scope_init_gpu_stream": (C.c_int, [C.c_int, _P, C.POINTER(_P), _ERR]),
    "swh_scope_init_cpu": (C.c_int, [C.c_size_t, C.POINTER(_P), _ERR]),
    "swh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "swh_scope_init_gpus": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P), _ERR]),
    "swh_scope_device_count": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "swh_scope_free": (C.c_int, [_P]),
    "swh_scope_compute_units": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "swh_scope_set_async": (C.c_int, [_P, C.c_int]),
    "swh_scope_synchronize": (C.c_int, [_P, _ERR]),
    "swh_scope_set_pipelined": (C.c_int, [_P, C.c_int, _ERR]),
    "swh_scope_join": (C.c_int, [_P, _ERR]),
    "swh_scope_forget": (C.c_int, [_P]),
    "swh_scope_describe": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "swh_scope_set_profiling": (C.c_int, [_P, C.c_int]),
    "swh_scope_last_timing": (C.c_int, [_P, C.POINTER(Timing)]),
    "swh_scope_timing_totals": (C.c_int, [_P, C.POINTER(TimingTotals)]),
    "swh_unified_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), _ERR]),
    "swh_unified_free": (C.c_int, [_P, _P]),
    "swh_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), _ERR]),
    "swh_device_free": (C.c_int, [_P, _P]),
    "swh_copy_to_device": (C.c_int, [_P, _P, _P, C.c_size_t, _ERR]),
    "swh_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_init": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P), _ERR]),
    "swh_levenshtein_free": (C.c_int, [_P]),
    "swh_levenshtein_set_algorithm": (C.c_int, [_P, C.c_int]),
    "swh_levenshtein_pairs_u32tape": (C.c_int, [_P, _P, C.POINTER(TapeU32), C.POINTER(TapeU32), C.c_uint32, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_pairs_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_uint32, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_utf8_pairs_u32tape": (C.c_int, [_P, _P, C.POINTER(TapeU32), C.POINTER(TapeU32), C.c_uint32, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_utf8_pairs_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_uint32, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_cross_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_levenshtein_utf8_cross_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_sharded_prepare_u32tape": (C.c_int, [_P, C.POINTER(TapeU32), C.POINTER(TapeU32), C.c_int, C.POINTER(_P), _ERR]),
    "swh_sharded_prepare_u64tape": (C.c_int, [_P, C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_int, C.POINTER(_P), _ERR]),
    "swh_sharded_free": (C.c_int, [_P]),
    "swh_sharded_cuts": (C.c_int, [_P, C.POINTER(C.c_size_t), C.c_size_t]),
    "swh_levenshtein_pairs_sharded": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _ERR]),
    "swh_sharded_cross_prepare_u64tape": (C.c_int, [_P, C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_int, C.POINTER(_P), _ERR]),
    "swh_sharded_cross_free": (C.c_int, [_P]),
    "swh_levenshtein_cross_sharded": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _ERR]),
    "swh_nw_cross_sharded": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _ERR]),
    "swh_sw_cross_sharded": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _ERR]),
    "swh_nw_pairs_sharded": (C.c_int, [_P, _P, _P, _P, _ERR]),
    "swh_sw_pairs_sharded": (C.c_int, [_P, _P, _P, _P, _ERR]),
    "swh_levenshtein_pairs_sharded_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_uint32, _P, _ERR]),
    "swh_scope_shard_timing": (C.c_int, [_P, C.POINTER(ShardTiming)]),
    "swh_nw_init": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P), _ERR]),
    "swh_nw_init_classes": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(_P), _ERR]),
    "swh_nw_free": (C.c_int, [_P]),
    "swh_nw_pairs_u32tape": (C.c_int, [_P, _P, C.POINTER(TapeU32), C.POINTER(TapeU32), _P, C.c_size_t, _ERR]),
    "swh_nw_pairs_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_nw_cross_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_sw_init": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P), _ERR]),
    "swh_sw_init_classes": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(_P), _ERR]),
    "swh_sw_free": (C.c_int, [_P]),
    "swh_sw_pairs_u32tape": (C.c_int, [_P, _P, C.POINTER(TapeU32), C.POINTER(TapeU32), _P, C.c_size_t, _ERR]),
    "swh_sw_pairs_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_sw_cross_u64tape": (C.c_int, [_P, _P, C.POINTER(TapeU64), C.POINTER(TapeU64), _P, C.c_size_t, _ERR]),
    "swh_tape_prepare_u32": (C.c_int, [_P, C.POINTER(TapeU32), C.c_int, C.POINTER(_P), _ERR]),
    "swh_tape_prepare_u64": (C.c_int, [_P, C.POINTER(TapeU64), C.c_int, C.POINTER(_P), _ERR]),
    "swh_prepared_info": (C.c_int, [_P, C.POINTER(PreparedInfo)]),
    "swh_prepared_free": (C.c_int, [_P]),
    "swh_levenshtein_pairs_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), C.c_uint32, _P, C.c_size_t, _ERR]),
    "swh_levenshtein_cross_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), _P, C.c_size_t, _ERR]),
    "swh_nw_pairs_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), _P, C.c_size_t, _ERR]),
    "swh_nw_cross_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), _P, C.c_size_t, _ERR]),
    "swh_sw_pairs_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), _P, C.c_size_t, _ERR]),
    "swh_sw_cross_prepared": (C.c_int, [_P, _P, C.POINTER(PreparedView), C.POINTER(PreparedView), _P, C.c_size_t, _ERR]),
    "swh_version": (C.c_char_p, []),
    "swh_capabilities": (C.c_char_p, []),
    # harness header
    "swh_synth_generate": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(Synth), _ERR]),
    "swh_synth_free": (None, [C.POINTER(Synth)]),
    "swh_synth_matrix": (None, [C.c_uint64, C.c_char_p, _P]),
    "swh_unary_class_costs": (None, [C.c_int8, C.c_int8, _P, _P]),
    "swh_shard_cuts_u32tape": (None, [C.POINTER(TapeU32), C.POINTER(TapeU32), C.c_size_t, C.POINTER(C.c_size_t)]),
    "swh_shard_cuts_u64tape": (None, [C.POINTER(TapeU64), C.POINTER(TapeU64), C.c_size_t, C.POINTER(C.c_size_t)]),
    "swh_crossproduct_side": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "swh_auto_batch_size": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "swh_format_si_rate": (C.c_size_t, [C.c_double, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "swh_format_seconds": (C.c_size_t, [C.c_double, C.c_char_p, C.c_size_t]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class StringWarsError(RuntimeError):
    """Raised for every non-success status; ``.status`` holds the ``swh_status_t`` name."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = STATUS_NAMES.get(status, str(status))


def check(status: int, err: C.c_char_p) -> None:
    if status != SUCCESS:
        message = err.value.decode("utf-8", "replace") if err.value else ""
        raise StringWarsError(status, message)
