"""Python counterpart of the reference's shared harness for the similarity benches (``utils.py`` +
the measuring helpers of ``similarities/bench.py``), restated for this backend.

Behaviours reproduced (reference file:line):
  * reporter line ``f"{name:<42} {' | '.join(columns)}"``, rate/seconds formatting, p50/p99 by index
    ``round(q * (n - 1))``                                            -- utils.py:251-336
  * env helpers, ``should_run`` (compiled regex, ``search``)           -- utils.py:18-63, :497-502
  * batch sizing ``auto_batch_size`` / ``crossproduct_side``           -- utils.py:202-209, bench.py:152-161
  * tokenisation: ``lines`` = split on LF, ``words`` = ``str.split()`` (ALL whitespace, unlike the Rust
    loader), ``file``; STRINGWARS_UNIQUE keeps first appearances        -- utils.py:410-451
  * seeded shuffle ``random.seed(STRINGWARS_SEED or 42); random.shuffle(tokens)`` -- bench.py:847-867
  * time-budgeted loop: uncounted warm-up, at least one measured call  -- bench.py:198-236
"""
from __future__ import annotations

import os
import random
import re
import time
from typing import Callable, List, Optional, Sequence

REPORT_NAME_WIDTH = 42


def get_env(name: str) -> Optional[str]:
    value = os.environ.get(name)
    return value if value not in (None, "") else None


def get_env_parsed(name: str, default, parser=int):
    value = get_env(name)
    if value is None:
        return default
    try:
        return parser(value)
    except ValueError:
        return default


def get_env_bool(name: str) -> bool:
    value = get_env(name)
    return value is not None and value.lower() in ("1", "true", "yes", "on")


# ---- reporter ---------------------------------------------------------------------------------------------
# One formatter for every column of the report line. The LINE is the contract (utils.py:291-336 prints
# "<name padded to 42> <col> | <col> ..." with two decimals, decimal k/M/G prefixes and ns/µs/ms/s latencies);
# how it is produced is this module's own: a ladder of (threshold, suffix) steps walked from the top.
_DECIMAL_STEPS = ((1e9, "G"), (1e6, "M"), (1e3, "k"))          # the reference stops at G: 1500.00 GCUPS, never 1.50 T
_TIME_STEPS = ((1.0, 1.0, "s"), (1e-3, 1e3, "ms"), (1e-6, 1e6, "µs"))
# report kind -> (unit, "space between prefix and unit")
_RATE_UNITS = {"cups": ("CUPS", False), "hashes": ("hashes/s", True), "bits": ("bits/s", True), "comparisons": ("cmp/s", True),
               "bytes": ("B/s", False)}


def _scaled(amount: float, unit: str, spaced: bool = False) -> str:
    for threshold, prefix in _DECIMAL_STEPS:
        if amount >= threshold:
            return f"{amount / threshold:.2f} {prefix}{' ' if spaced else ''}{unit}"
    return f"{amount:.2f} {unit}"


def format_seconds(seconds: float) -> str:
    for threshold, factor, suffix in _TIME_STEPS:
        if seconds >= threshold:
            return f"{seconds * factor:.2f} {suffix}"
    return f"{seconds * 1e9:.2f} ns"


def _quantile(ordered: Sequence[float], q: float) -> float:
    return ordered[min(round(q * (len(ordered) - 1)), len(ordered) - 1)]


def stats_line(name: str, report: str, elapsed_seconds: float, elements: int, total_bytes: int,
               latencies_seconds: Optional[Sequence[float]] = None) -> str:
    """The report line for one variant: primary rate, byte rate (when bytes were counted), p50 / p99."""
    if report not in _RATE_UNITS:
        raise ValueError(f"Unknown report unit: {report!r}")
    seconds = max(elapsed_seconds, 1e-12)
    columns: List[str] = []
    if report != "bytes":
        columns.append(_scaled(elements / seconds, *_RATE_UNITS[report]))
    if report == "bytes" or total_bytes > 0:
        columns.append(_scaled(total_bytes / seconds, *_RATE_UNITS["bytes"]))
    if latencies_seconds:
        ordered = sorted(latencies_seconds)
        columns.append(f"p50 {format_seconds(_quantile(ordered, 0.5))} p99 {format_seconds(_quantile(ordered, 0.99))}")
    return f"{name:<{REPORT_NAME_WIDTH}} {' | '.join(columns)}"


def report_stats(name, report, elapsed_seconds, elements, total_bytes, latencies_seconds=None) -> None:
    print(stats_line(name, report, elapsed_seconds, elements, total_bytes, latencies_seconds), flush=True)


def should_run(name: str, pattern: Optional[re.Pattern]) -> bool:
    return True if pattern is None else bool(pattern.search(name))


def auto_batch_size(cores: int, base: Optional[int] = None, default_base: int = 128) -> int:
    per_core = base if base is not None else get_env_parsed("STRINGWARS_BATCH_PER_CORE", default_base)
    return max(1, max(1, per_core) * max(1, cores))


def crossproduct_side(budget: int, num_tokens: int) -> int:
    target = max(1, round(budget ** 0.5))
    return max(1, min(target, num_tokens // 2))


def tokenize(haystack, tokens_mode: Optional[str] = None, unique: Optional[bool] = None):
    mode = tokens_mode or os.environ.get("STRINGWARS_TOKENS", "lines")
    if mode == "lines":
        tokens = haystack.split(b"\n" if isinstance(haystack, bytes) else "\n")
    elif mode == "words":
        tokens = haystack.split()
    elif mode == "file":
        tokens = [haystack]
    else:
        raise ValueError(f"Unknown tokens mode: {mode}. Use 'lines', 'words', or 'file'.")
    if unique is None:
        unique = get_env_bool("STRINGWARS_UNIQUE")
    if mode != "file" and unique:
        tokens = list(dict.fromkeys(tokens))
    return tokens


_SIZE_UNITS = {"": 1, "b": 1, "kb": 1 << 10, "mb": 1 << 20, "gb": 1 << 30}


def size_in_bytes(text: str) -> int:
    """`--dataset-limit` values: a number with an optional b / kb / mb / gb suffix, powers of 1024, any case
    (utils.py:340-367: '128mb', '1gb', '500kb', '1.5 mb')."""
    spelled = (text or "").strip().lower()
    digits = spelled.rstrip("kmgb").strip()
    unit = spelled[len(spelled.rstrip("kmgb")):]
    if not digits or unit not in _SIZE_UNITS or not re.fullmatch(r"\d+(\.\d+)?", digits):
        raise ValueError(f"Invalid size format: {text}. Use formats like '128mb', '1gb', '500kb'")
    return int(float(digits) * _SIZE_UNITS[unit])


def load_tokens(path: Optional[str] = None, tokens_mode: str = "words", shuffle: bool = True, size_limit: Optional[str] = None) -> List[str]:
    """Dataset -> shuffled token list exactly as similarities/bench.py:857-867 prepares it; `size_limit` reads at most that
    much of the file (`--dataset-limit`, utils.py:489-494; counted in characters of the decoded text, as `f.read(n)` does)."""
    path = path or get_env("STRINGWARS_DATASET")
    if path is None:
        raise ValueError("No dataset path provided and STRINGWARS_DATASET not set")
    with open(path, encoding="utf-8", errors="ignore") as handle:
        text = handle.read(size_in_bytes(size_limit)) if size_limit else handle.read()
    tokens = tokenize(text, os.environ.get("STRINGWARS_TOKENS", tokens_mode))
    max_tokens = get_env_parsed("STRINGWARS_MAX_TOKENS", None)
    if max_tokens is not None and max_tokens > 0:
        tokens = tokens[:max_tokens]
    if shuffle:
        random.seed(get_env_parsed("STRINGWARS_SEED", 42))
        random.shuffle(tokens)
    return tokens


def measure(name: str, compute: Callable[[], None], cells_per_call: int, bytes_per_call: int, warmup_seconds: float,
            time_limit_seconds: float, report: str = "cups") -> Optional[dict]:
    """Uncounted warm-up, then cycle `compute` until the deadline (at least one measured call, also with a zero
    budget) and print the canonical line. Per-call latencies feed the p50/p99 column."""
    if warmup_seconds > 0:
        deadline = time.perf_counter_ns() + int(warmup_seconds * 1e9)
        while time.perf_counter_ns() < deadline:
            compute()
    deadline = time.perf_counter_ns() + int(time_limit_seconds * 1e9)
    start = last = time.perf_counter_ns()
    iterations, latencies = 0, []
    while True:
        compute()
        iterations += 1
        now = time.perf_counter_ns()
        latencies.append((now - last) / 1e9)
        last = now
        if now >= deadline:
            break
    elapsed = (time.perf_counter_ns() - start) / 1e9
    report_stats(name, report, elapsed, cells_per_call * iterations, bytes_per_call * iterations, latencies)
    return {"name": name, "elapsed": elapsed, "iterations": iterations, "cells": cells_per_call * iterations}
