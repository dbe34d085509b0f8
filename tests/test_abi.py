"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol the
headers declare, and refuses to compute without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swh_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["stringwars_amd.h", "stringwars_amd_harness.h"])
def test_every_declared_symbol_is_exported(sw, header):
    lib = ctypes.CDLL(sw.LIBRARY_PATH)
    names = declared_symbols(header)
    assert len(names) >= 8
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/{header} but not exported: {missing}"


def test_binding_table_matches_header(sw):
    from stringwars_amd import _native
    declared = set(declared_symbols("stringwars_amd.h")) | set(declared_symbols("stringwars_amd_harness.h"))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)


# C parameter type -> the Rust spelling the crate must use for it (after the crate's own aliases are expanded:
# `Handle` = `*mut c_void`, `Err` = `*mut *const c_char`). Opaque handles are pointers to incomplete structs.
_HANDLES = ("swh_scope_t", "swh_prepared_t", "swh_levenshtein_t", "swh_nw_t", "swh_sw_t", "swh_sharded_t", "swh_sharded_cross_t")
_C_TO_RUST = {
    "int": "c_int", "swh_algorithm_t": "c_int", "swh_status_t": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64",
    "int *": "*mut c_int", "const int *": "*const c_int", "size_t *": "*mut usize", "ptrdiff_t *": "*mut isize",
    "uint32_t *": "*mut u32", "int32_t *": "*mut i32", "const int8_t *": "*const i8", "const uint8_t *": "*const u8",
    "void *": "*mut c_void", "const void *": "*const c_void", "void **": "*mut *mut c_void", "const char **": "*mut *const c_char",
    "const char *": "*const c_char", "char *": "*mut c_char",
    "const swh_tape_u32_t *": "*const TapeU32", "const swh_tape_u64_t *": "*const TapeU64", "const swh_prepared_view_t *": "*const PreparedView",
    "swh_prepared_info_t *": "*mut PreparedInfo", "swh_timing_t *": "*mut Timing", "swh_timing_totals_t *": "*mut TimingTotals",
    "swh_shard_timing_t *": "*mut ShardTiming",
}
for _h in _HANDLES:
    _C_TO_RUST[_h] = "*mut c_void"
    _C_TO_RUST[_h + " *"] = "*mut *mut c_void"


def _c_parameter_types(declaration):
    types = []
    for parameter in declaration.split(","):
        parameter = " ".join(parameter.replace("*", " * ").split())
        if not parameter or parameter == "void":
            continue
        words = parameter.split()
        if words[-1] != "*":                      # drop the parameter's name
            words = words[:-1]
        text = " ".join(words).replace(" * *", " **").replace(" *", " *")
        types.append(re.sub(r"\s*\*\s*\*", " **", text).replace("* *", "**"))
    return types


def _rust_parameter_types(declaration):
    types = []
    for parameter in declaration.split(","):
        if not parameter.strip():
            continue
        spelled = " ".join(parameter.split(":", 1)[1].split())
        spelled = re.sub(r"\bHandle\b", "*mut c_void", spelled)
        types.append(re.sub(r"\bErr\b", "*mut *const c_char", spelled))
    return types


def rust_abi_mismatches(header_text, rust_text, names):
    """Every (symbol, position, C type, Rust type found, Rust type wanted) where the crate's extern block and the header differ
    in parameter count, parameter type or return type."""
    wrong = []
    for name in names:
        c_decl = re.search(r"([A-Za-z_][A-Za-z_0-9 ]*?[ \*]+)%s\s*\(([^;]*?)\)\s*;" % name, header_text, flags=re.S)
        r_decl = re.search(r"\bfn %s\s*\(([^;]*?)\)\s*(?:->\s*([^;]*?))?\s*;" % name, rust_text, flags=re.S)
        if not c_decl or not r_decl:
            wrong.append((name, "declaration", None, None, None))
            continue
        c_types, r_types = _c_parameter_types(c_decl.group(2)), _rust_parameter_types(r_decl.group(1))
        if len(c_types) != len(r_types):
            wrong.append((name, "count", len(c_types), len(r_types), None))
            continue
        for position, (c_type, r_type) in enumerate(zip(c_types, r_types)):
            want = _C_TO_RUST.get(c_type)
            if want is None or want != r_type:
                wrong.append((name, position, c_type, r_type, want))
        c_return = " ".join(c_decl.group(1).replace("*", " * ").split()).replace(" *", " *")
        c_return = re.sub(r"^(extern|SWH_API|SWH_EXPORT)\s+", "", c_return)
        want = _C_TO_RUST.get(c_return)
        found = " ".join((r_decl.group(2) or "()").split())
        if want != found:
            wrong.append((name, "return", c_return, found, want))
    return wrong


def test_rust_crate_declares_the_c_abi():
    """stringwars_amd/rust (north-star: "exposed from a new Rust crate through a thin extern "C" FFI") cannot be compiled
    here, so its extern block is held to the header: every symbol of include/stringwars_amd.h is declared in src/lib.rs with
    the same number of parameters, every parameter and the return value of the corresponding TYPE (C -> Rust map above: a
    `uint32_t` must be `u32`, a `size_t` `usize`, `const T *` `*const T`, ...), and the patch for the reference's bench.rs /
    Cargo.toml is present with the Levenshtein rows and the NW / SW rows of both gap groups."""
    crate = os.path.join(ROOT, "stringwars_amd", "rust")
    for name in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), "bench.rs.patch"):
        assert os.path.exists(os.path.join(crate, name)), name
    rust = open(os.path.join(crate, "src", "lib.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "stringwars_amd.h")).read(), flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)
    names = declared_symbols("stringwars_amd.h")
    assert not rust_abi_mismatches(header, rust, names), rust_abi_mismatches(header, rust, names)
    # the check has teeth: a u32 <-> usize swap, a *const <-> *mut swap and a dropped parameter are each caught
    seeded = rust.replace("fn swh_levenshtein_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, bound: u32,",
                          "fn swh_levenshtein_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, bound: usize,")
    assert seeded != rust and [w[:2] for w in rust_abi_mismatches(header, seeded, names)] == [("swh_levenshtein_pairs_u64tape", 4)]
    seeded = rust.replace("fn swh_nw_init(scope: Handle, substitution_256x256: *const i8,", "fn swh_nw_init(scope: Handle, substitution_256x256: *mut i8,")
    assert seeded != rust and [w[:2] for w in rust_abi_mismatches(header, seeded, names)] == [("swh_nw_init", 1)]
    seeded = rust.replace("fn swh_sharded_cuts(sharded: Handle, cuts: *mut usize, capacity: usize)", "fn swh_sharded_cuts(sharded: Handle, cuts: *mut usize)")
    assert seeded != rust and [w[:2] for w in rust_abi_mismatches(header, seeded, names)] == [("swh_sharded_cuts", "count")]
    patch = open(os.path.join(crate, "bench.rs.patch")).read()
    assert "similarities/bench.rs" in patch and "Cargo.toml" in patch and "stringwars_amd::levenshtein_pairs<" in patch
    # the alignment rows of `perform_linear_benchmarks` / `perform_affine_benchmarks` (bench.rs:641-699, :967-1026), added where both
    # groups meet (`align_score_benchmarks`), with the reference's own scoring inputs (`unary_class_costs(2, -1)`, bench.rs:655)
    for row in ('{group_name}/stringwars_amd::NeedlemanWunschScores<1gpu>', '{group_name}/stringwars_amd::SmithWatermanScores<1gpu>'):
        assert row in patch, row
    assert "unary_class_costs(2, -1)" in patch and "align_score_benchmarks_rocm" in patch
    assert "verify-rapidfuzz" in open(os.path.join(crate, "Cargo.toml")).read() and "assert_eq!" in rust


def test_no_oracle_in_product():
    """The product path must never touch oracle/ (tier rule 3)."""
    for base, _, files in os.walk(os.path.join(ROOT, "stringwars_amd")):
        for name in files:
            if name.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(base, name), errors="replace").read()
                assert "import oracle" not in text and "liboracle" not in text and "oracle/" not in text, name


def test_version_and_capabilities(sw):
    assert sw.__version__.count(".") == 2
    caps = sw.capabilities().split(",")
    assert "gfx950" in caps and "wavefront" in caps and "bitparallel" in caps


def test_cpu_scope_is_refused(sw):
    with pytest.raises(sw.StringWarsError) as info:
        sw.DeviceScope(cpu_cores=1)
    assert info.value.status == "not_implemented"


def test_helpers_match_reference_formulas(sw):
    from stringwars_amd import _native as N
    # crossproduct_side (bench.rs:113-117): round(sqrt(budget)) clamped to tape_len / 2, at least 1
    assert N.lib.swh_crossproduct_side(256, 10_000) == 16
    assert N.lib.swh_crossproduct_side(256 * 132, 1_000_000) == 184   # H100: 132 SMs (similarities/README.md:23)
    assert N.lib.swh_crossproduct_side(256 * 256, 1_000_000) == 256   # MI355X: 256 CUs
    assert N.lib.swh_crossproduct_side(10_000, 10) == 5
    assert N.lib.swh_crossproduct_side(0, 0) == 1
    buf = ctypes.create_string_buffer(64)
    for rate, expect in [(1.2e12, "1200.00 GCUPS"), (3.3e9, "3.30 GCUPS"), (1.5e6, "1.50 MCUPS"), (999.0, "999.00 CUPS")]:
        N.lib.swh_format_si_rate(rate, b"CUPS", 0, buf, 64)
        assert buf.value.decode() == expect  # utils.rs:487-500: no "T" prefix
    N.lib.swh_format_si_rate(2.5e6, b"hashes/s", 1, buf, 64)
    assert buf.value.decode() == "2.50 M hashes/s"
    for seconds, expect in [(2.5e-7, "250.00 ns"), (2.5e-5, "25.00 µs"), (2.5e-2, "25.00 ms"), (2.5, "2.50 s")]:
        N.lib.swh_format_seconds(seconds, buf, 64)
        assert buf.value.decode("utf-8") == expect
    classes, costs = sw.unary_class_costs(2, -1)
    assert classes[33] == 1 and costs[3, 3] == 2 and costs[3, 4] == -1


def test_gpu_entry_points_fail_loudly_without_device(sw):
    """On this CPU-only container there is no device: scope creation must raise, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    with pytest.raises(sw.StringWarsError) as info:
        sw.DeviceScope(gpu_device=0)
    assert info.value.status == "no_device"
