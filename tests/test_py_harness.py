"""Python harness counterpart (stringwars_amd/harness.py, bench_similarities.py): line layout, formulas and the
dataset preparation of the reference's Python side (utils.py / similarities/bench.py)."""
import os
import random
import re
import subprocess
import sys

import pytest

from conftest import child_pythonpath

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reporter_layout():
    from stringwars_amd import harness as H
    line = H.stats_line("uniform/x<1gpu>", "cups", 2.0, 3_000_000_000_000, 4_000_000, [1e-3, 2e-3, 3e-3])
    assert line == f"{'uniform/x<1gpu>':<42} 1500.00 GCUPS | 2.00 MB/s | p50 2.00 ms p99 3.00 ms"  # no "T" prefix
    assert H.stats_line("n", "hashes", 1.0, 2_500_000, 0) == f"{'n':<42} 2.50 M hashes/s"
    assert H.stats_line("n", "bytes", 1.0, 0, 999) == f"{'n':<42} 999.00 B/s"
    assert [H.format_seconds(s) for s in (2.5e-7, 2.5e-5, 2.5e-2, 2.5)] == ["250.00 ns", "25.00 µs", "25.00 ms", "2.50 s"]
    with pytest.raises(ValueError):
        H.stats_line("n", "furlongs", 1.0, 1, 1)


def test_batch_formulas_match_the_c_helpers(sw):
    from stringwars_amd import _native as N, harness as H
    for budget, tokens in ((256, 10_000), (256 * 132, 1_000_000), (256 * 256, 1_000_000), (10_000, 10), (1, 3)):
        assert H.crossproduct_side(budget, tokens) == N.lib.swh_crossproduct_side(budget, tokens)
    assert H.auto_batch_size(256, base=None, default_base=256) == 65536
    assert H.auto_batch_size(1, base=7) == 7


def test_tokenisation_and_shuffle(tmp_path, monkeypatch):
    from stringwars_amd import harness as H
    text = "alpha beta\tgamma\n\ndelta  beta\n"
    assert H.tokenize(text, "words") == ["alpha", "beta", "gamma", "delta", "beta"]      # str.split(): all whitespace
    assert H.tokenize(text, "lines") == ["alpha beta\tgamma", "", "delta  beta", ""]      # LF only, empties kept
    assert H.tokenize(text, "words", unique=True) == ["alpha", "beta", "gamma", "delta"]
    assert H.tokenize(text, "file") == [text]
    path = tmp_path / "data.txt"
    path.write_text(text)
    monkeypatch.delenv("STRINGWARS_SEED", raising=False)
    monkeypatch.delenv("STRINGWARS_MAX_TOKENS", raising=False)
    expected = ["alpha", "beta", "gamma", "delta", "beta"]
    random.seed(42)
    random.shuffle(expected)                                                               # bench.py:847-867
    assert H.load_tokens(str(path), "words") == expected
    pattern = re.compile("uniform/.*UTF8")
    assert H.should_run("uniform/stringwars_amd.LevenshteinDistancesUTF8<1gpu>", pattern)
    assert not H.should_run("linear/stringwars_amd.NeedlemanWunschScores<1gpu>", pattern)


def test_dataset_limit_sizes_and_truncation(tmp_path):
    """`--dataset-limit` (utils.py:340-367, :489-494): 1024-based b / kb / mb / gb suffixes; the loader reads at most that much."""
    from stringwars_amd import harness as H
    for text, want in (("128mb", 128 << 20), ("1gb", 1 << 30), ("500kb", 500 << 10), ("10", 10), ("1.5 MB", int(1.5 * (1 << 20))), ("7b", 7)):
        assert H.size_in_bytes(text) == want, text
    for bad in ("", "mb", "12tb", "-3kb", "1e3"):
        with pytest.raises(ValueError):
            H.size_in_bytes(bad)
    path = tmp_path / "words.txt"
    path.write_text(" ".join(f"w{i:04d}" for i in range(1000)))
    assert len(H.load_tokens(str(path), shuffle=False)) == 1000
    assert H.load_tokens(str(path), shuffle=False, size_limit="60b") == [f"w{i:04d}" for i in range(10)]


def run_script(extra):
    env = dict(os.environ, PYTHONPATH=child_pythonpath())
    return subprocess.run([sys.executable, "-m", "stringwars_amd.bench_similarities", "--dataset", os.path.join(ROOT, "README.md"),
                           *extra], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)


def test_script_skips_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    result = run_script(["--time-limit", "0"])
    assert result.returncode == 0, result.stderr
    assert result.stdout.count("SKIPPED (no_device") == 9 and "# uniform" in result.stdout and "# affine" in result.stdout


@pytest.mark.gpu
def test_script_measures_rows_on_gpu():
    result = run_script(["--time-limit", "0.2"])
    assert result.returncode == 0, result.stderr
    rows = [l for l in result.stdout.splitlines() if "CUPS" in l]
    assert len(rows) == 9, result.stdout
    assert sum("<k=32,1gpu>" in row for row in rows) == 2                   # the bounded rows (bytes, code points)
    for row in rows:
        assert re.match(r"^\S+<(k=32,)?1gpu>\s+\d+\.\d\d [kMG]?CUPS \| \d+\.\d\d [kMG]?B/s \| p50 .* p99 .*$", row), row
    env_bound = subprocess.run([sys.executable, "-m", "stringwars_amd.bench_similarities", "--dataset", os.path.join(ROOT, "README.md"),
                                "--time-limit", "0.05", "-k", "k="], env=dict(os.environ, PYTHONPATH=child_pythonpath(), STRINGWARS_ERROR_BOUND="2"),
                               capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert env_bound.returncode == 0 and len([l for l in env_bound.stdout.splitlines() if "<k=2,1gpu>" in l and "CUPS" in l]) == 2, env_bound.stdout + env_bound.stderr
    assert re.search(r"exceeded=[1-9]\d* of", env_bound.stderr), env_bound.stderr   # k = 2 is exceeded by README words
    only = run_script(["--time-limit", "0.05", "-k", "SmithWaterman"])
    assert len([l for l in only.stdout.splitlines() if "CUPS" in l]) == 2
