"""stringwars_amd -- MI355X-native batched Levenshtein / bounded Levenshtein / Needleman-Wunsch.

A drop-in for the similarity hot path of ashvardanian/StringWars (`similarities/bench.rs`,
`similarities/bench.py`): hand-written HIP kernels for gfx950 behind a C ABI
(`include/stringwars_amd.h`), with this package as the thin host-side mirror of the
`stringzillas` engine interface the reference drives. No CPU fallback exists.
"""
from ._native import LIBRARY_PATH, StringWarsError, lib as _lib  # noqa: F401  (import fails loudly if unbuilt)
from .engines import (  # noqa: F401
    UNBOUNDED, DeviceScope, DeviceTape, PreparedTape, ShardedPairs, ShardedCross, shard_cuts, LevenshteinDistances, LevenshteinDistancesUTF8, NeedlemanWunschScores, SmithWatermanScores, Strs,
    edit_distance,
)
from .synth import WORKLOADS, generate_pairs, substitution_matrix, unary_class_costs  # noqa: F401

__version__ = _lib.swh_version().decode()


def capabilities() -> str:
    """`log_stringzilla_metadata` counterpart (utils.rs:78-92)."""
    return _lib.swh_capabilities().decode()
