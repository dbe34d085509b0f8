"""Which source files a kernel's executed-instruction counts depend on, and a digest of them.

`tools/pmc_constants.py` stamps every entry of `profiles/<round>/pmc_constants.json` with the digest of the sources the
profiled library was built from; `bench.py` recomputes it and marks the roofline `pmc_stale` when they differ -- the
per-launch instruction and traffic figures cannot be read from inside the process, so a committed constant must not
outlive the kernel it was measured on unnoticed.
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stringwars_amd", "csrc")

# stamp name used by the library's timing (swh_timing_t::dominant_name) -> (substring of the kernel symbol, sources)
KERNELS = {
    "bitparallel": ("swh::k_bitparallel<unsigned char,", ("bitparallel.hip", "bp_item.hpp", "bp_window.hpp", "common.hpp")),
    "bitparallel_u32": ("swh::k_bitparallel<unsigned int,", ("bitparallel.hip", "bp_item.hpp", "bp_dense.hpp", "bp_window.hpp", "common.hpp")),
    "bitparallel_tiled": ("swh::k_bitparallel_tiled<unsigned char,", ("tiled.hip", "bp_item.hpp", "bp_window.hpp", "common.hpp")),
    "bitparallel_tiled_u32": ("swh::k_bitparallel_tiled<unsigned int,", ("tiled.hip", "bp_item.hpp", "bp_window.hpp", "common.hpp")),
    "bitparallel_long": ("swh::k_bitparallel_long<unsigned char", ("bitparallel.hip", "bp_item.hpp", "bp_window.hpp", "common.hpp")),
    "direct_short": ("swh::k_direct_short<", ("prepass.hip", "bp_window.hpp", "common.hpp")),
    "utf8_strings": ("swh::k_utf8_strings<", ("prepass.hip", "common.hpp")),
    "short_tiled": ("swh::k_short_tiled<", ("short.hip", "common.hpp")),
    "banded": ("swh::k_banded<", ("banded.hip", "common.hpp")),
    "wavefront": ("swh::k_wavefront<", ("wavefront.hip", "common.hpp")),
    "nwprofile": ("k_nwprofile<", ("nwprofile.hip", "common.hpp")),
    "cross_short": ("swh::k_cross_short<", ("cross.hip", "bp_window.hpp", "common.hpp")),
    "cross_short_u32": ("swh::k_cross_short_cp<", ("cross.hip", "bp_window.hpp", "common.hpp")),
    "align_short": ("swh::k_align_short<", ("alignshort.hip", "bp_window.hpp", "common.hpp")),
    "align_wide": ("swh::k_align_cross_wide<", ("alignshort.hip", "bp_window.hpp", "common.hpp")),
    "align_long": ("swh::k_align_cross_long<", ("alignshort.hip", "bp_window.hpp", "common.hpp")),
}


def source_digest(stamp: str) -> str:
    """sha256 (first 16 hex digits) over the sources of one kernel family, in the listed order; a missing file hashes as empty."""
    digest = hashlib.sha256()
    for name in KERNELS[stamp][1]:
        path = os.path.join(CSRC, name)
        digest.update(name.encode() + b"\0")
        if os.path.exists(path):
            with open(path, "rb") as handle:
                digest.update(handle.read())
    return digest.hexdigest()[:16]
