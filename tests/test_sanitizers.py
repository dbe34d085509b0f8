"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side C / C++ of this repo (CPU only --
GPU sanitizers are not available on the pool): the oracle (`make -C oracle asan`) under its known-answer and cross-check
suites, and the product's host code that needs no device -- the synthetic generators / harness helpers of synth.cpp, the
harness of include/stringwars_amd.hpp, and the C++ counterpart of bench.rs with every GPU row SKIPPED."""
import json
import os
import subprocess
import sys

import pytest

from conftest import child_pythonpath

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sanitizer_runtime():
    paths = []
    for name in ("libasan.so", "libubsan.so"):
        found = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(found) or not os.path.exists(found):
            pytest.skip(f"{name} not installed")
        paths.append(os.path.realpath(found))
    return ":".join(paths)


def clean(result):
    text = result.stdout + result.stderr
    assert result.returncode == 0, text[-3000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text and "LeakSanitizer" not in text, text[-3000:]
    return text


def test_oracle_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "-s"], check=True, capture_output=True, timeout=300)
    library = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    code = r'''
import json, numpy as np, oracle
assert oracle.lib._name.endswith("liboracle_asan.so")
kat = json.load(open("tests/golden/kat.json"))
for a, b, points, bytes_ in kat["levenshtein"]:
    assert oracle.levenshtein_utf8(a, b) == points and oracle.levenshtein(a, b) == bytes_ == oracle.levenshtein(a, b, algo="hyyro")
    assert oracle.levenshtein_antidiagonal(a, b) == bytes_
for name in ("nw_unary_2_m1", "sw_unary_2_m1"):
    m = np.full((256, 256), -1, np.int8); np.fill_diagonal(m, 2)
    for a, b, linear, affine in kat[name]["cases"]:
        local = name.startswith("sw")
        assert oracle.nw_score(a, b, m, -2, -2, local=local) == linear and oracle.nw_score(a, b, m, -5, -1, local=local) == affine
        assert oracle.align_score_general(a, b, m, -5, -1, local=local) == affine
for alphabet in (2, 4, 26, 256):
    assert oracle.selfcheck("levenshtein", 7, 3000, alphabet, 300)[0] == 0
    assert oracle.selfcheck("alignment", 11, 400, alphabet, 40)[0] == 0
import stringwars_amd as sw   # host-side generators only (no device is touched)
for workload, count in (("words16", 300), ("tokens64", 300), ("utf8_lines", 20), ("short_words", 500)):
    a, b = sw.generate_pairs(workload, count, seed=3)
    utf8 = workload == "utf8_lines"
    wf = oracle.levenshtein_pairs(a, b, utf8=utf8)
    assert (wf == oracle.levenshtein_pairs(a, b, utf8=utf8, algo="wf", bound=None)).all()
    if not utf8:
        assert (wf == oracle.levenshtein_pairs(a, b, algo="hyyro")).all()
    assert (oracle.levenshtein_pairs(a, b, utf8=utf8, bound=3) == np.minimum(wf, 4)).all()
    assert oracle.cells(a, b, utf8=utf8) > 0
pa, pb = sw.generate_pairs("protein4k", 1, seed=3)
assert oracle.nw_pairs(pa, pb, sw.substitution_matrix(3), -11, -1).shape == (1,)
for bad in (b"\xc0\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xe4\xb8", b"\x80", b"\xff"):
    try:
        oracle.utf8_decode(bad)
    except ValueError:
        continue
    raise SystemExit("invalid UTF-8 accepted: %r" % bad)
print("oracle clean")
'''
    env = dict(os.environ, LD_PRELOAD=sanitizer_runtime(), ORACLE_LIBRARY=library, PYTHONPATH=child_pythonpath(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    result = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert "oracle clean" in clean(result)


HOST_DRIVER = r'''
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include "stringwars_amd.hpp"
using namespace swa;
using namespace swa::harness;
int main(int argc, char **argv) {
    // synthetic generators: every workload, several shards of the same stream, odd thread counts
    const int workloads[] = {1, 2, 3, 4, 5, 40};
    uint64_t digest = 0;
    for (int w : workloads) {
        const size_t count = (w == 4 || w == 40) ? 3 : 257;
        for (int threads : {1, 3}) {
            swh_synth_t whole{}, part{};
            const char *err = nullptr;
            if (swh_synth_generate(w, 42, 0, count, threads, &whole, &err) != swh_success_k) return 2;
            if (swh_synth_generate(w, 42, count / 2, count - count / 2, threads, &part, &err) != swh_success_k) return 3;
            // pair i depends only on (workload, seed, i): the second half generated on its own is the second half
            const uint64_t from = whole.offsets_a[count / 2];
            if (std::memcmp(whole.data_a + from, part.data_a, part.offsets_a[part.count]) != 0) return 4;
            for (size_t i = 0; i < whole.offsets_b[count]; ++i) digest = digest * 1099511628211ull + whole.data_b[i];
            swh_synth_free(&whole); swh_synth_free(&part);
        }
    }
    swh_synth_t none{};
    if (swh_synth_generate(99, 42, 0, 4, 1, &none, nullptr) != swh_invalid_argument_k) return 5;
    if (swh_synth_generate(1, 42, 0, 0, 0, &none, nullptr) != swh_success_k) return 6;
    swh_synth_free(&none);
    int8_t matrix[65536];
    swh_synth_matrix(42, "ACDEFGHIKLMNPQRSTVWY", matrix);
    swh_synth_matrix(42, nullptr, matrix);
    uint8_t classes[256]; int8_t costs[1024];
    swh_unary_class_costs(2, -1, classes, costs);
    if (swh_crossproduct_side(65536, 1000000) != 256 || swh_crossproduct_side(1, 3) != 1 || swh_auto_batch_size(256, 256) != 65536) return 7;
    // the harness: dataset loader + tokenizer + reporter + measurement loop on host work
    BytesTape tape = load_dataset_with_default_mode("words");
    if (tape.size() < 2) return 8;
    volatile uint64_t sink = 0;
    BenchStats stats = measure_throughput("host/loop", ReportAs::Cups, BenchBudget{0.0, 0.02}, [&] {
        for (size_t i = 0; i < tape.size(); ++i) sink = sink + tape.view().length(i);
        return WorkUnits{tape.size(), tape.size()};
    });
    if (!stats.calls) return 9;
    std::printf("host clean %" PRIu64 "\n", digest);
    return 0;
}
'''


def test_host_side_of_the_product_under_asan_and_ubsan(tmp_path):
    """synth.cpp (generators, harness helpers) and include/stringwars_amd.hpp (loader, tokenizer, reporter, measurement loop)
    compiled with -fsanitize=address,undefined; then the C++ counterpart of bench.rs itself, sanitized, against the real
    library with no device: every GPU row must be SKIPPED, nothing may leak out of bounds on the way."""
    library_dir = os.path.join(ROOT, "stringwars_amd")
    if not os.path.exists(os.path.join(library_dir, "libstringwars_amd.so")):
        pytest.skip("library not built")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include")]
    driver = tmp_path / "driver.cpp"
    driver.write_text(HOST_DRIVER)
    binary = tmp_path / "host_asan"
    subprocess.run(["g++", *flags, str(driver), os.path.join(library_dir, "csrc", "synth.cpp"), "-o", str(binary), "-lpthread"],
                   check=True, capture_output=True, text=True, timeout=600)
    env = dict(os.environ, STRINGWARS_DATASET=os.path.join(ROOT, "README.md"), ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    assert "host clean" in clean(subprocess.run([str(binary)], env=env, capture_output=True, text=True, timeout=300))
    bench = tmp_path / "bench_asan"
    subprocess.run(["g++", *flags, os.path.join(library_dir, "csrc", "bench_similarities.cpp"), "-o", str(bench), "-L", library_dir,
                    "-lstringwars_amd", f"-Wl,-rpath,{library_dir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        env["HIP_VISIBLE_DEVICES"] = "-1"    # the sanitized run covers the no-device paths; GPU rows are test_cpp_harness.py's
    env.update(STRINGWARS_TIME="0", STRINGWARS_WARMUP="0", ASAN_OPTIONS="detect_leaks=0")   # (the HIP runtime keeps its own allocations)
    text = clean(subprocess.run([str(bench)], env=env, capture_output=True, text=True, timeout=300))
    assert text.count("SKIPPED (") >= 9 and "# affine" in text
