"""The C++ counterpart of similarities/bench.rs (stringwars_amd/bench_similarities): output format of the
harness restated in include/stringwars_amd.hpp. CPU: every GPU row is SKIPPED loudly; GPU: rows are measured."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "stringwars_amd", "bench_similarities")
ROWS = [
    "uniform/stringwars_amd::LevenshteinDistances<1gpu>", "uniform/stringwars_amd::LevenshteinDistancesUtf8<1gpu>",
    "uniform/stringwars_amd::levenshtein_pairs<1gpu>", "linear/stringwars_amd::NeedlemanWunschScores<1gpu>",
    "affine/stringwars_amd::NeedlemanWunschScores<1gpu>", "linear/stringwars_amd::SmithWatermanScores<1gpu>",
    "affine/stringwars_amd::SmithWatermanScores<1gpu>",
    "uniform/stringwars_amd::levenshtein_pairs<k=32,1gpu>", "uniform/stringwars_amd::levenshtein_pairs_utf8<k=32,1gpu>",
]


def run(extra_env):
    env = dict(os.environ, STRINGWARS_DATASET=os.path.join(ROOT, "README.md"), STRINGWARS_WARMUP="0", **extra_env)
    return subprocess.run([BINARY], env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not os.path.exists(BINARY), reason="bench_similarities not built")
def test_rows_skip_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    result = run({"STRINGWARS_TIME": "0"})
    assert result.returncode == 0
    for header in ("# uniform", "# linear", "# affine"):  # bench.rs:331/:343/:354
        assert header in result.stdout
    for row in ROWS:
        assert f"{row}: SKIPPED (no HIP device visible" in result.stdout
    assert "Dataset:" in result.stderr and "Distribution:" in result.stderr  # utils.rs:402-430


@pytest.mark.skipif(not os.path.exists(BINARY), reason="bench_similarities not built")
def test_filter_env_is_honoured():
    result = run({"STRINGWARS_TIME": "0", "STRINGWARS_FILTER": "NeedlemanWunsch"})
    assert result.returncode == 0 and "STRINGWARS_FILTER active" in result.stderr or "SKIPPED" in result.stdout


def test_report_line_with_and_without_hardware_counters(tmp_path):
    """utils.rs:652-692: `cyc/B` and `IPC` sit between the byte rate and the latencies when the perf_event counters were
    readable and are left out -- nothing else moves -- when they were not. A measured loop of host work prints whichever
    this machine allows."""
    source = tmp_path / "line.cpp"
    source.write_text(r'''
#include "stringwars_amd.hpp"
using namespace swa::harness;
int main() {
    BenchStats stats;
    stats.elapsed_seconds = 2.0; stats.calls = 4; stats.elements = 3000000000ull; stats.bytes = 500000;
    std::printf("%s\n", stats.line("plain", ReportAs::Cups).c_str());
    stats.has_cycles = true; stats.cycles = 1250000;
    std::printf("%s\n", stats.line("cycles-only", ReportAs::Cups).c_str());
    stats.has_instructions = true; stats.instructions = 2500000;
    std::printf("%s\n", stats.line("both", ReportAs::Bytes).c_str());
    volatile uint64_t sink = 0;
    BenchStats looped = measure_throughput("host/loop", ReportAs::Bytes, BenchBudget{0.0, 0.05}, [&] {
        for (int i = 0; i < 20000; ++i) sink = sink + (uint64_t)i * 3u;
        return WorkUnits{20000, 20000};
    });
    return looped.calls > 0 && (!looped.has_cycles || looped.cycles > 0) ? 0 : 1;
}
''')
    binary = tmp_path / "line"
    library_dir = os.path.join(ROOT, "stringwars_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(source), "-o", str(binary), "-L", library_dir,
                    "-lstringwars_amd", f"-Wl,-rpath,{library_dir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"], check=True, capture_output=True, text=True, timeout=300)
    result = subprocess.run([str(binary)], capture_output=True, text=True, timeout=60)
    assert result.returncode == 0, result.stderr
    lines = result.stdout.splitlines()
    assert lines[0] == f"{'plain':<42} 1.50 GCUPS | 250.00 kB/s"
    assert lines[1] == f"{'cycles-only':<42} 1.50 GCUPS | 250.00 kB/s | 2.50 cyc/B"
    assert lines[2] == f"{'both':<42} 250.00 kB/s | 2.50 cyc/B | IPC 2.00"
    assert re.match(r"^host/loop\s+\d+\.\d\d [kMG]?B/s( \| \d+\.\d\d cyc/B)?( \| IPC \d+\.\d\d)? \| p50 .* p99 .*$", lines[3]), lines[3]


@pytest.mark.gpu
def test_rows_are_measured_on_gpu():
    result = run({"STRINGWARS_TIME": "0.3"})
    assert result.returncode == 0, result.stderr
    line = re.compile(r"^(\S+)\s+\d+\.\d\d [kMG]?CUPS \| \d+\.\d\d [kMG]?B/s( \| \d+\.\d\d cyc/B)?( \| IPC \d+\.\d\d)? \| p50 \d+\.\d\d (ns|µs|ms|s) p99 \d+\.\d\d (ns|µs|ms|s)$")
    measured = {m.group(1) for m in map(line.match, result.stdout.splitlines()) if m}
    assert measured == set(ROWS) | {"uniform/stringwars_amd::levenshtein_pairs<prepared,1gpu>"}, result.stdout
    filtered = run({"STRINGWARS_TIME": "0.1", "STRINGWARS_FILTER": "uniform/.*pairs<1gpu"})
    names = {m.group(1) for m in map(line.match, filtered.stdout.splitlines()) if m}
    assert names == {"uniform/stringwars_amd::levenshtein_pairs<1gpu>"}
    # the `<Ngpu>` row: three member scopes on device 0 (one-GPU box), results identical to the single-GPU row
    multi = run({"STRINGWARS_TIME": "0.1", "STRINGWARS_AMD_GPUS": "0,0,0", "STRINGWARS_FILTER": "uniform/.*pairs"})
    names = {m.group(1) for m in map(line.match, multi.stdout.splitlines()) if m}
    assert multi.returncode == 0 and "uniform/stringwars_amd::levenshtein_pairs<3gpu>" in names, (multi.stdout, multi.stderr[-800:])
    assert "Skipping: linear/stringwars_amd::NeedlemanWunschScores<1gpu>" in filtered.stderr
    # the bounded rows take k from STRINGWARS_ERROR_BOUND (reference README.md:311); k = 2 is exceeded by README words and the
    # binary itself checks bounded == min(unbounded, k + 1) on every pair (exit code 2 otherwise)
    bounded = run({"STRINGWARS_TIME": "0.1", "STRINGWARS_ERROR_BOUND": "2", "STRINGWARS_FILTER": "k="})
    names = {m.group(1) for m in map(line.match, bounded.stdout.splitlines()) if m}
    assert bounded.returncode == 0 and names == {"uniform/stringwars_amd::levenshtein_pairs<k=2,1gpu>", "uniform/stringwars_amd::levenshtein_pairs_utf8<k=2,1gpu>"}, (bounded.stdout, bounded.stderr[-800:])
    assert re.search(r"exceeded=[1-9]\d* of", bounded.stderr)
