"""Host-side mirror of the engine interface the reference benchmarks drive.

Names, argument meaning and error behaviour follow the ``stringzillas`` Python surface used by
``similarities/bench.py`` (``szs.DeviceScope(gpu_device=0)`` :360, ``sz.Strs(list)`` :399,
``engine_class(capabilities=scope)`` :403, ``engine(queries, candidates, scope, out=matrix)``
:421-422, ``NeedlemanWunschScores(byte_to_class, costs, open=, extend=, capabilities=)``
:466-472) and the Rust ``szs`` types of ``similarities/bench.rs:79-82``. On top of the reference's
cross-product call every engine has the pairwise batch the north-star asks for
(``engine.pairs(a, b, scope, bound=..., out=...)``), the shape of the only pairwise batched call in
the reference, ``cudf ... str.edit_distance`` (``bench.py:596-604``).

Everything here is plumbing over the C ABI (``include/stringwars_amd.h``); all arithmetic happens
in the HIP kernels. Inputs may live on the host (numpy) or on the device (``DeviceTape``, torch
CUDA tensors); device-resident inputs are used in place.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence, Union

import numpy as np

from . import _native as N

__all__ = [
    "DeviceScope", "Strs", "DeviceTape", "PreparedTape", "ShardedPairs", "shard_cuts", "LevenshteinDistances", "LevenshteinDistancesUTF8",
    "NeedlemanWunschScores", "SmithWatermanScores", "edit_distance", "StringWarsError", "UNBOUNDED",
]

StringWarsError = N.StringWarsError
UNBOUNDED = N.UNBOUNDED


def _pointer(obj) -> int:
    """Raw address of a numpy array, torch tensor, ctypes pointer or int."""
    if obj is None:
        return 0
    if isinstance(obj, int):
        return obj
    if isinstance(obj, np.ndarray):
        return obj.ctypes.data
    if hasattr(obj, "data_ptr"):  # torch.Tensor
        return int(obj.data_ptr())
    if hasattr(obj, "value"):
        return int(obj.value or 0)
    raise TypeError(f"cannot take the address of {type(obj)!r}")


class DeviceScope:
    """``szs.DeviceScope(gpu_device=0)`` / ``DeviceScope::gpu_device(0)`` (bench.rs:379).

    ``cpu_cores=`` scopes are refused: this backend has no CPU path and says so loudly
    (the reference prints ``SKIPPED (<reason>)`` for a variant whose scope cannot be built).
    ``stream`` may be a raw ``hipStream_t`` address (e.g. ``torch.cuda.current_stream().cuda_stream``).
    """

    def __init__(self, gpu_device: Optional[int] = None, cpu_cores: Optional[int] = None, stream: Optional[int] = None,
                 gpu_devices: Optional[Sequence[int]] = None):
        handle = C.c_void_p()
        err = C.c_char_p()
        if gpu_devices is not None:
            # several GPUs of one node behind one scope (`swh_scope_init_gpus`): batches are split over them by the
            # `*_sharded` calls, distances gathered with RCCL inside the library
            devices = (C.c_int * len(gpu_devices))(*[int(d) for d in gpu_devices])
            status = N.lib.swh_scope_init_gpus(devices, len(gpu_devices), C.byref(handle), C.byref(err))
            gpu_device = gpu_devices[0] if len(gpu_devices) else 0
        elif cpu_cores is not None and gpu_device is None:
            status = N.lib.swh_scope_init_cpu(int(cpu_cores), C.byref(handle), C.byref(err))
        elif stream is not None:
            status = N.lib.swh_scope_init_gpu_stream(int(gpu_device or 0), C.c_void_p(int(stream)), C.byref(handle), C.byref(err))
        else:
            status = N.lib.swh_scope_init_gpu(int(gpu_device or 0), C.byref(handle), C.byref(err))
        N.check(status, err)
        self._handle = handle
        self.gpu_device = int(gpu_device or 0)

    @property
    def handle(self) -> C.c_void_p:
        return self._handle

    @property
    def compute_units(self) -> int:
        value = C.c_size_t()
        N.lib.swh_scope_compute_units(self._handle, C.byref(value))
        return int(value.value)

    @property
    def device_count(self) -> int:
        value = C.c_size_t()
        N.lib.swh_scope_device_count(self._handle, C.byref(value))
        return int(value.value)

    def shard_timing(self) -> dict:
        timing = N.ShardTiming()
        N.lib.swh_scope_shard_timing(self._handle, C.byref(timing))
        return {"compute_ms": timing.compute_ms, "gather_ms": timing.gather_ms, "cells": int(timing.cells), "pairs": int(timing.pairs)}

    def set_async(self, enabled: bool) -> None:
        N.lib.swh_scope_set_async(self._handle, int(bool(enabled)))

    def set_pipelined(self, enabled: bool) -> None:
        """Alternate calls between two internal lanes so the next call's planning overlaps this call's DP kernel;
        consumers ordered on the scope's stream call ``join()`` first, everyone else ``synchronize()``."""
        err = C.c_char_p()
        N.check(N.lib.swh_scope_set_pipelined(self._handle, int(bool(enabled)), C.byref(err)), err)

    def join(self) -> None:
        err = C.c_char_p()
        N.check(N.lib.swh_scope_join(self._handle, C.byref(err)), err)

    def forget(self) -> None:
        """Drops everything the scope believes about earlier calls (``swh_scope_forget``): the next call is routed as on a new scope."""
        N.lib.swh_scope_forget(self._handle)

    def describe(self) -> dict:
        """The scope's beliefs (``swh_scope_describe``) as a dict of strings."""
        text = C.create_string_buffer(512)
        N.lib.swh_scope_describe(self._handle, text, len(text))
        return dict(item.split("=", 1) for item in text.value.decode().split())

    def set_profiling(self, enabled: bool) -> None:
        N.lib.swh_scope_set_profiling(self._handle, int(bool(enabled)))

    def synchronize(self) -> None:
        err = C.c_char_p()
        N.check(N.lib.swh_scope_synchronize(self._handle, C.byref(err)), err)

    def last_timing(self) -> dict:
        timing = N.Timing()
        N.lib.swh_scope_last_timing(self._handle, C.byref(timing))
        return {
            "total_ms": timing.total_ms, "dominant_ms": timing.dominant_ms, "compute_ms": timing.compute_ms,
            "dominant_name": timing.dominant_name.decode(), "cells": int(timing.cells),
            "bytes": int(timing.bytes), "kernels": int(timing.kernels),
        }


    def timing_totals(self) -> dict:
        """Sums of ``last_timing``'s durations over every call since profiling was switched on, including
        asynchronous / pipelined calls (``swh_scope_timing_totals``)."""
        totals = N.TimingTotals()
        N.lib.swh_scope_timing_totals(self._handle, C.byref(totals))
        return {"total_ms": totals.total_ms, "dominant_ms": totals.dominant_ms, "compute_ms": totals.compute_ms,
                "calls": int(totals.calls)}

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle:
            N.lib.swh_scope_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Strs:
    """Arrow-style string tape on the host: ``sz.Strs(list)`` (bench.py:399), ``BytesTape<u64>``
    (bench.rs:292). ``offsets`` has ``count + 1`` entries of dtype uint64 or uint32."""

    def __init__(self, items: Union[Iterable[Union[bytes, str]], None] = None, *, data: Optional[np.ndarray] = None,
                 offsets: Optional[np.ndarray] = None):
        if items is not None:
            encoded = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in items]
            lengths = np.fromiter((len(s) for s in encoded), dtype=np.uint64, count=len(encoded))
            offsets = np.zeros(len(encoded) + 1, dtype=np.uint64)
            np.cumsum(lengths, out=offsets[1:])
            data = np.frombuffer(b"".join(encoded), dtype=np.uint8).copy() if encoded else np.zeros(0, np.uint8)
        if data is None or offsets is None:
            raise ValueError("Strs needs either items or data+offsets")
        if offsets.dtype not in (np.uint32, np.uint64):
            raise TypeError("offsets must be uint32 or uint64")
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets)
        self.count = len(self.offsets) - 1

    def __len__(self) -> int:
        return self.count

    def __getitem__(self, index):
        if isinstance(index, slice):
            start, stop, step = index.indices(self.count)
            if step != 1:
                raise ValueError("only contiguous sub-views")
            return self.subview(start, stop)
        lo, hi = int(self.offsets[index]), int(self.offsets[index + 1])
        return self.data[lo:hi].tobytes()

    def subview(self, start: int, stop: int) -> "Strs":
        """Zero-copy sub-range, ``BytesTapeView::subview(lo, hi)`` (bench.rs:134-139)."""
        return Strs(data=self.data, offsets=self.offsets[start:stop + 1])

    def with_offsets(self, dtype) -> "Strs":
        return Strs(data=self.data, offsets=self.offsets.astype(dtype))

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets.astype(np.int64))

    def to_device(self, scope: DeviceScope) -> "DeviceTape":
        return DeviceTape.upload(scope, self)


class DeviceTape:
    """A tape whose data and offsets are device pointers (hipMalloc or torch CUDA tensors)."""

    def __init__(self, data_ptr: int, offsets_ptr: int, count: int, offsets_dtype, keepalive=None, scope=None, owned=False):
        self.data_ptr, self.offsets_ptr, self.count = int(data_ptr), int(offsets_ptr), int(count)
        self.offsets_dtype = np.dtype(offsets_dtype)
        self._keepalive, self._scope, self._owned = keepalive, scope, owned

    @classmethod
    def upload(cls, scope: DeviceScope, strs: Strs) -> "DeviceTape":
        err = C.c_char_p()
        pointers = []
        for array in (strs.data, strs.offsets):
            pointer = C.c_void_p()
            N.check(N.lib.swh_device_alloc(scope.handle, array.nbytes + 16, C.byref(pointer), C.byref(err)), err)
            if array.nbytes:
                N.check(N.lib.swh_copy_to_device(scope.handle, pointer, array.ctypes.data, array.nbytes, C.byref(err)), err)
            pointers.append(pointer)
        # a sub-view's offsets index into the full data buffer, which was uploaded whole
        return cls(pointers[0].value, pointers[1].value, strs.count, strs.offsets.dtype, scope=scope, owned=True)

    @classmethod
    def from_torch(cls, data, offsets) -> "DeviceTape":
        import torch
        dtype = {torch.int32: np.uint32, torch.int64: np.uint64, torch.uint8: None}.get(offsets.dtype)
        if hasattr(torch, "uint32"):
            dtype = {torch.uint32: np.uint32, torch.uint64: np.uint64}.get(offsets.dtype, dtype)
        if dtype is None:
            raise TypeError("offsets tensor must be (u)int32 or (u)int64")
        return cls(data.data_ptr(), offsets.data_ptr(), offsets.numel() - 1, dtype, keepalive=(data, offsets))

    def __len__(self) -> int:
        return self.count

    def free(self) -> None:
        if self._owned and self._scope is not None and self._scope.handle:
            N.lib.swh_device_free(self._scope.handle, C.c_void_p(self.data_ptr))
            N.lib.swh_device_free(self._scope.handle, C.c_void_p(self.offsets_ptr))
        self._owned = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PreparedTape:
    """A tape made ready once (``swh_tape_prepare_*``): resident on the scope's device, measured, and -- with
    ``utf8=True`` -- validated and decoded to code points. The counterpart of the ``BytesTapeView`` / ``CharsTapeView`` the
    reference builds once outside its timed closures (bench.rs:292-306; ``try_into`` is where invalid UTF-8 surfaces, and
    so it does here: ``StringWarsError('invalid_utf8')`` from the constructor). Slicing gives zero-copy sub-views
    (``subview(lo, hi)``, bench.rs:134-139). Engines take prepared tapes wherever they take tapes; both sides of a call
    must then be prepared, and in the same mode."""

    def __init__(self, scope: DeviceScope, tape, utf8: bool = False, _parent: Optional["PreparedTape"] = None,
                 first: int = 0, count: Optional[int] = None):
        if _parent is not None:
            self._handle, self._root, self.utf8 = _parent._handle, _parent._root, _parent.utf8
            self.first, self.count = first, count
            return
        tape = _as_tape(tape)
        struct, is64, keep = _c_tape(tape)
        handle, err = C.c_void_p(), C.c_char_p()
        fn = N.lib.swh_tape_prepare_u64 if is64 else N.lib.swh_tape_prepare_u32
        N.check(fn(scope.handle, C.byref(struct), int(bool(utf8)), C.byref(handle), C.byref(err)), err)
        self._handle, self._root, self.utf8 = handle, self, bool(utf8)
        self._keepalive = keep if isinstance(keep, DeviceTape) else None   # device tapes are used in place
        self.first, self.count = 0, len(tape)

    def __len__(self) -> int:
        return self.count

    @property
    def info(self) -> dict:
        info = N.PreparedInfo()
        N.lib.swh_prepared_info(self._handle, C.byref(info))
        return {"count": int(info.count), "bytes": int(info.bytes), "symbols": int(info.symbols), "longest": int(info.longest),
                "utf8": bool(info.utf8), "ascii": bool(info.ascii)}

    def subview(self, start: int, stop: int) -> "PreparedTape":
        if not 0 <= start <= stop <= self.count:
            raise IndexError("sub-view outside the tape")
        return PreparedTape(None, None, _parent=self, first=self.first + start, count=stop - start)

    def __getitem__(self, index):
        if not isinstance(index, slice):
            raise TypeError("prepared tapes are sliced, not indexed")
        start, stop, step = index.indices(self.count)
        if step != 1:
            raise ValueError("only contiguous sub-views")
        return self.subview(start, max(start, stop))

    def view(self) -> "N.PreparedView":
        return N.PreparedView(self._handle, self.first, self.count)

    def free(self) -> None:
        if self._root is self and getattr(self, "_handle", None) and getattr(N, "lib", None) is not None:
            N.lib.swh_prepared_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def shard_cuts(a: Strs, b: Strs, shards: int) -> list:
    """Cells-balanced contiguous cuts of a pairwise batch (`swh_shard_cuts_*`): shard r = pairs [cuts[r], cuts[r+1])."""
    ta, a64, keep_a = _c_tape(a)
    tb, b64, keep_b = _c_tape(b, want64=a64 or None)
    if a64 != b64:
        ta, a64, keep_a = _c_tape(a, want64=True)
        tb, b64, keep_b = _c_tape(b, want64=True)
    cuts = (C.c_size_t * (shards + 1))()
    (N.lib.swh_shard_cuts_u64tape if a64 else N.lib.swh_shard_cuts_u32tape)(C.byref(ta), C.byref(tb), shards, cuts)
    return [int(c) for c in cuts]


class ShardedPairs:
    """A pairwise batch made resident on every device of a multi-GPU scope (`swh_sharded_prepare_*`): contiguous
    cells-balanced shards, shard r uploaded to and prepared on device r. The steady state of the `<Ngpu>` rows:
    ``engine.pairs_sharded(batch, scope)`` scores all shards and gathers the distances with RCCL."""

    def __init__(self, scope: DeviceScope, a: Strs, b: Strs, utf8: bool = False):
        if not isinstance(a, Strs) or not isinstance(b, Strs):
            raise TypeError("sharding reads host tapes (Strs)")
        ta, a64, keep_a = _c_tape(a)
        tb, b64, keep_b = _c_tape(b, want64=a64 or None)
        if a64 != b64:
            ta, a64, keep_a = _c_tape(a, want64=True)
            tb, b64, keep_b = _c_tape(b, want64=True)
        handle, err = C.c_void_p(), C.c_char_p()
        fn = N.lib.swh_sharded_prepare_u64tape if a64 else N.lib.swh_sharded_prepare_u32tape
        N.check(fn(scope.handle, C.byref(ta), C.byref(tb), int(bool(utf8)), C.byref(handle), C.byref(err)), err)
        self._handle, self.count, self.utf8, self._scope = handle, len(a), bool(utf8), scope

    @property
    def cuts(self) -> list:
        n = self._scope.device_count + 1
        cuts = (C.c_size_t * n)()
        N.lib.swh_sharded_cuts(self._handle, cuts, n)
        return [int(c) for c in cuts]

    def free(self) -> None:
        if getattr(self, "_handle", None) and getattr(N, "lib", None) is not None:
            N.lib.swh_sharded_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ShardedCross:
    """A dense queries x candidates product made resident on every device of a multi-GPU scope
    (`swh_sharded_cross_prepare_u64tape`): row blocks of equal query symbols, block r and all candidates prepared on device
    r. ``engine.cross_sharded(product, scope, out=matrix)`` is the `<Ngpu>` twin of the reference's `compute_into`."""

    def __init__(self, scope: DeviceScope, queries: Strs, candidates: Strs, utf8: bool = False):
        if not isinstance(queries, Strs) or not isinstance(candidates, Strs):
            raise TypeError("sharding reads host tapes (Strs)")
        tq, _, keep_q = _c_tape(queries, want64=True)
        tc, _, keep_c = _c_tape(candidates, want64=True)
        handle, err = C.c_void_p(), C.c_char_p()
        N.check(N.lib.swh_sharded_cross_prepare_u64tape(scope.handle, C.byref(tq), C.byref(tc), int(bool(utf8)), C.byref(handle), C.byref(err)), err)
        self._handle, self.shape, self.utf8, self._scope = handle, (len(queries), len(candidates)), bool(utf8), scope

    def free(self) -> None:
        if getattr(self, "_handle", None) and getattr(N, "lib", None) is not None:
            N.lib.swh_sharded_cross_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


TapeLike = Union[Strs, DeviceTape, PreparedTape, Sequence[Union[bytes, str]]]


def _as_tape(obj: TapeLike):
    if isinstance(obj, (Strs, DeviceTape, PreparedTape)):
        return obj
    return Strs(obj)


def _c_tape(tape, want64: Optional[bool] = None):
    """ctypes tape struct + its width; keeps arrays alive through the returned tuple."""
    if isinstance(tape, Strs):
        if want64 is True and tape.offsets.dtype != np.uint64:
            tape = tape.with_offsets(np.uint64)
        is64 = tape.offsets.dtype == np.uint64
        struct = (N.TapeU64 if is64 else N.TapeU32)(tape.data.ctypes.data, tape.offsets.ctypes.data, tape.count)
        return struct, is64, tape
    is64 = tape.offsets_dtype == np.dtype(np.uint64)
    if want64 is True and not is64:
        raise TypeError("this call needs u64 offsets on the device tape")
    struct = (N.TapeU64 if is64 else N.TapeU32)(tape.data_ptr, tape.offsets_ptr, tape.count)
    return struct, is64, tape


class _Engine:
    _utf8 = False
    _abi_prefix = "swh_levenshtein"

    def __init__(self):
        self._handle = None

    def _prepared(self, suffix, a, b, scope, out, out_dtype, extra=(), cross=False):
        """The ``*_prepared`` twin of a call: both sides are PreparedTape views."""
        if not isinstance(a, PreparedTape) or not (b is None or isinstance(b, PreparedTape)):
            raise TypeError("both tapes of a call must be prepared, or neither")
        if self._utf8 != a.utf8:
            raise ValueError("a %s engine needs tapes prepared with utf8=%s" % (type(self).__name__, self._utf8))
        va, vb = a.view(), (b.view() if b is not None else None)
        if out is None:
            shape = (len(a), len(b if b is not None else a)) if cross else (len(a),)
            out = np.zeros(shape, dtype=out_dtype)
        stride = 0
        if isinstance(out, np.ndarray):
            stride = out.strides[0] if (cross or out.size > 1) else 0
        fn = getattr(N.lib, self._abi_prefix + suffix)
        err = C.c_char_p()
        status = fn(self._handle, scope.handle, C.byref(va), C.byref(vb) if vb is not None else None, *extra,
                    C.c_void_p(_pointer(out)), stride, C.byref(err))
        N.check(status, err)
        return out

    def cross_sharded(self, product: "ShardedCross", scope: DeviceScope, out=None):
        """The dense matrix of a sharded product: every device of the scope fills its rows and copies them into `out` (host
        memory or memory of the first device; 64-bit entries)."""
        if self._utf8 != product.utf8:
            raise ValueError("engine and sharded product disagree on UTF-8")
        if out is None:
            out = np.zeros(product.shape, dtype=np.uint64 if self._abi_prefix == "swh_levenshtein" else np.int64)
        if isinstance(out, np.ndarray) and (out.dtype.itemsize != 8 or out.shape != product.shape):
            raise ValueError("out must be a (len(queries), len(candidates)) matrix of 64-bit integers")
        row_stride = out.strides[0] if isinstance(out, np.ndarray) else product.shape[1] * 8
        err = C.c_char_p()
        status = getattr(N.lib, self._abi_prefix + "_cross_sharded")(self._handle, scope.handle, product._handle, C.c_void_p(_pointer(out)), row_stride, C.byref(err))
        N.check(status, err)
        return out

    def bind_pairs(self, a: "PreparedTape", b: "PreparedTape", scope: DeviceScope, out, bound: Optional[int] = None):
        """A pre-bound pairwise call on prepared views: every ctypes argument is built once, the returned callable
        only crosses the FFI (what a compiled harness pays per call; `pairs()` re-derives views, pointers and strides
        in Python each time, ~10 us). `out` must stay alive and in place."""
        if not isinstance(a, PreparedTape) or not isinstance(b, PreparedTape) or len(a) != len(b):
            raise TypeError("bind_pairs takes two prepared views of equal length")
        if self._utf8 != a.utf8:
            raise ValueError("engine and tapes disagree on UTF-8")
        va, vb, err = a.view(), b.view(), C.c_char_p()
        fn = getattr(N.lib, self._abi_prefix + "_pairs_prepared")
        extra = (C.c_uint32(N.UNBOUNDED if bound is None else int(bound)),) if self._abi_prefix == "swh_levenshtein" else ()
        args = (self._handle, scope.handle, C.byref(va), C.byref(vb), *extra, C.c_void_p(_pointer(out)),
                out.strides[0] if isinstance(out, np.ndarray) and out.size > 1 else 0, C.byref(err))
        keep = (va, vb, err, a, b, out)

        def call(_fn=fn, _args=args, _keep=keep):
            status = _fn(*_args)
            if status != N.SUCCESS:
                N.check(status, _keep[2])
        return call

    def _pairs(self, fn32, fn64, a, b, scope, out, out_dtype, extra=()):
        a, b = _as_tape(a), _as_tape(b)
        if len(a) != len(b):
            raise ValueError("pairwise scoring needs two collections of equal length")
        if isinstance(a, PreparedTape) or isinstance(b, PreparedTape):
            return self._prepared("_pairs_prepared", a, b, scope, out, out_dtype, extra)
        ta, a64, keep_a = _c_tape(a)
        tb, b64, keep_b = _c_tape(b, want64=a64 or None)
        if a64 != b64:
            ta, a64, keep_a = _c_tape(a, want64=True)
            tb, b64, keep_b = _c_tape(b, want64=True)
        if out is None:
            out = np.empty(len(a), dtype=out_dtype)
        stride = out.strides[0] if isinstance(out, np.ndarray) and out.ndim == 1 and out.size > 1 else 0
        err = C.c_char_p()
        fn = fn64 if a64 else fn32
        status = fn(self._handle, scope.handle, C.byref(ta), C.byref(tb), *extra, C.c_void_p(_pointer(out)), stride, C.byref(err))
        N.check(status, err)
        del keep_a, keep_b
        return out

    def _cross(self, fn, queries, candidates, scope, out, out_dtype):
        queries = _as_tape(queries)
        candidates = queries if candidates is None else _as_tape(candidates)
        if isinstance(queries, PreparedTape) or isinstance(candidates, PreparedTape):
            if isinstance(out, np.ndarray) and (out.dtype.itemsize != 8 or out.shape != (len(queries), len(candidates))):
                raise ValueError("out must be a (len(queries), len(candidates)) matrix of 64-bit integers")
            return self._prepared("_cross_prepared", queries, candidates, scope, out, out_dtype, cross=True)
        tq, _, keep_q = _c_tape(queries, want64=True)
        tc, _, keep_c = _c_tape(candidates, want64=True)
        if out is None:
            out = np.zeros((len(queries), len(candidates)), dtype=out_dtype)
        row_stride = out.strides[0] if isinstance(out, np.ndarray) else len(candidates) * 8
        if isinstance(out, np.ndarray) and (out.dtype.itemsize != 8 or out.shape != (len(queries), len(candidates))):
            raise ValueError("out must be a (len(queries), len(candidates)) matrix of 64-bit integers")
        err = C.c_char_p()
        status = fn(self._handle, scope.handle, C.byref(tq), C.byref(tc), C.c_void_p(_pointer(out)), row_stride, C.byref(err))
        N.check(status, err)
        del keep_q, keep_c
        return out


class LevenshteinDistances(_Engine):
    """``szs.LevenshteinDistances`` / ``LevenshteinDistances::new(&scope, 0, 1, 1, 1)`` (bench.rs:382).

    ``engine(queries, candidates, scope, out=matrix)`` is the reference's dense cross-product
    (bench.py:421-422, bench.rs:478-486); ``engine.pairs(a, b, scope, bound=k)`` scores
    ``a[i]`` against ``b[i]`` and returns ``min(d, k+1)``.
    """

    def __init__(self, match: int = 0, mismatch: int = 1, open: int = 1, extend: int = 1, *,
                 capabilities: Optional[DeviceScope] = None, algorithm: str = "auto"):
        super().__init__()
        if capabilities is None:
            raise ValueError("capabilities=DeviceScope(gpu_device=...) is required: there is no default CPU scope")
        handle, err = C.c_void_p(), C.c_char_p()
        N.check(N.lib.swh_levenshtein_init(capabilities.handle, match, mismatch, open, extend, C.byref(handle), C.byref(err)), err)
        self._handle = handle
        self.set_algorithm(algorithm)

    def set_algorithm(self, algorithm: str) -> None:
        code = {"auto": N.ALGORITHM_AUTO, "wavefront": N.ALGORITHM_WAVEFRONT, "bitparallel": N.ALGORITHM_BITPARALLEL,
                "tiled": N.ALGORITHM_TILED}[algorithm]
        N.lib.swh_levenshtein_set_algorithm(self._handle, code)

    def __call__(self, queries: TapeLike, candidates: Optional[TapeLike] = None, scope: Optional[DeviceScope] = None, out=None):
        if scope is None:
            raise ValueError("a DeviceScope is required")
        fn = N.lib.swh_levenshtein_utf8_cross_u64tape if self._utf8 else N.lib.swh_levenshtein_cross_u64tape
        return self._cross(fn, queries, candidates, scope, out, np.uint64)

    def pairs(self, a: TapeLike, b: TapeLike, scope: DeviceScope, bound: Optional[int] = None, out=None):
        bound_value = N.UNBOUNDED if bound is None else int(bound)
        if self._utf8:
            fns = (N.lib.swh_levenshtein_utf8_pairs_u32tape, N.lib.swh_levenshtein_utf8_pairs_u64tape)
        else:
            fns = (N.lib.swh_levenshtein_pairs_u32tape, N.lib.swh_levenshtein_pairs_u64tape)
        return self._pairs(fns[0], fns[1], a, b, scope, out, np.uint32, extra=(C.c_uint32(bound_value),))

    def pairs_sharded(self, batch: "ShardedPairs", scope: DeviceScope, bound: Optional[int] = None, out=None):
        """One batch over every GPU of a multi-device scope; the distances come back gathered, in pair order."""
        if self._utf8 != batch.utf8:
            raise ValueError("engine and sharded batch disagree on UTF-8")
        if out is None:
            out = np.zeros(batch.count, dtype=np.uint32)
        err = C.c_char_p()
        status = N.lib.swh_levenshtein_pairs_sharded(self._handle, scope.handle, batch._handle,
                                                     N.UNBOUNDED if bound is None else int(bound), C.c_void_p(_pointer(out)), C.byref(err))
        N.check(status, err)
        return out

    def __del__(self):
        if getattr(self, "_handle", None) and getattr(N, "lib", None) is not None:   # module globals go first at exit
            N.lib.swh_levenshtein_free(self._handle)
            self._handle = None


class LevenshteinDistancesUTF8(LevenshteinDistances):
    """``szs.LevenshteinDistancesUTF8`` / ``LevenshteinDistancesUtf8`` (bench.rs:386-399): symbols are
    Unicode scalar values; invalid UTF-8 raises ``StringWarsError('invalid_utf8')``."""

    _utf8 = True


class NeedlemanWunschScores(_Engine):
    """``szs.NeedlemanWunschScores(byte_to_class, class_costs, open=, extend=, capabilities=)``
    (bench.py:466-472, bench.rs:658-662). ``substitution_matrix=`` takes a full 256x256 int8 table
    instead (config C4). gap(k) = open + (k-1)*extend."""

    _prefix = "swh_nw"
    _abi_prefix = "swh_nw"

    def __init__(self, byte_to_class: Optional[np.ndarray] = None, class_costs: Optional[np.ndarray] = None, *,
                 open: int = -2, extend: int = -2, capabilities: Optional[DeviceScope] = None,
                 substitution_matrix: Optional[np.ndarray] = None):
        super().__init__()
        if capabilities is None:
            raise ValueError("capabilities=DeviceScope(gpu_device=...) is required")
        handle, err = C.c_void_p(), C.c_char_p()
        if substitution_matrix is not None:
            matrix = np.ascontiguousarray(substitution_matrix, dtype=np.int8)
            if matrix.shape != (256, 256):
                raise ValueError("substitution_matrix must be 256x256 int8")
            status = getattr(N.lib, self._prefix + "_init")(capabilities.handle, matrix.ctypes.data, open, extend, C.byref(handle), C.byref(err))
        else:
            classes = np.ascontiguousarray(byte_to_class, dtype=np.uint8)
            costs = np.ascontiguousarray(class_costs, dtype=np.int8)
            if classes.shape != (256,) or costs.shape != (32, 32):
                raise ValueError("byte_to_class must have 256 entries and class_costs must be 32x32")
            status = getattr(N.lib, self._prefix + "_init_classes")(capabilities.handle, classes.ctypes.data, costs.ctypes.data,
                                                                   open, extend, C.byref(handle), C.byref(err))
        N.check(status, err)
        self._handle = handle

    def __call__(self, queries: TapeLike, candidates: Optional[TapeLike] = None, scope: Optional[DeviceScope] = None, out=None):
        if scope is None:
            raise ValueError("a DeviceScope is required")
        return self._cross(getattr(N.lib, self._prefix + "_cross_u64tape"), queries, candidates, scope, out, np.int64)

    def pairs(self, a: TapeLike, b: TapeLike, scope: DeviceScope, out=None):
        return self._pairs(getattr(N.lib, self._prefix + "_pairs_u32tape"), getattr(N.lib, self._prefix + "_pairs_u64tape"),
                           a, b, scope, out, np.int32)

    def pairs_sharded(self, batch: "ShardedPairs", scope: DeviceScope, out=None):
        """One batch over every GPU of a multi-device scope (the matrix is cloned to every device on first use); the scores
        come back gathered, in pair order."""
        if batch.utf8:
            raise ValueError("alignment engines score bytes: the sharded batch was prepared as UTF-8")
        if out is None:
            out = np.zeros(batch.count, dtype=np.int32)
        err = C.c_char_p()
        status = getattr(N.lib, self._prefix + "_pairs_sharded")(self._handle, scope.handle, batch._handle, C.c_void_p(_pointer(out)), C.byref(err))
        N.check(status, err)
        return out

    def __del__(self):
        if getattr(self, "_handle", None) and getattr(N, "lib", None) is not None:   # module globals go first at exit
            getattr(N.lib, self._prefix + "_free")(self._handle)
            self._handle = None


class SmithWatermanScores(NeedlemanWunschScores):
    """``szs.SmithWatermanScores`` (bench.py:789, bench.rs:882-963): local alignment score, same arguments."""

    _prefix = "swh_sw"
    _abi_prefix = "swh_sw"


def edit_distance(column_a: TapeLike, column_b: TapeLike, scope: DeviceScope, utf8: bool = True,
                  bound: Optional[int] = None) -> np.ndarray:
    """Element-wise edit distance of two equal-length string columns, the call shape of
    ``cudf.Series.str.edit_distance(other)`` (similarities/bench.py:602)."""
    engine = (LevenshteinDistancesUTF8 if utf8 else LevenshteinDistances)(capabilities=scope)
    return engine.pairs(column_a, column_b, scope, bound=bound)
