"""CPU checks of the measurement plumbing: the roofline arithmetic of bench.py and the PMC folding of
tools/pmc_constants.py (the numbers themselves come from the GPU box; the formulas are checked here)."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def test_roofline_fraction_is_executed_work_over_peak():
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module")
    constants = {"kernels": {"bitparallel_tiled|tokens64": {
        "pairs_per_launch": 1_000_000, "valu_insts": 86.0e6, "fetch_kb": 190_000.0, "write_kb": 4_000.0, "source": "unit test"}}}
    roof = bench.roofline_of("bitparallel_tiled", 0.180, 4_273_938_131, 140_000_000, "tokens64", 1_000_000, constants)
    lane_ops = 86.0e6 * 64
    assert abs(roof["achieved"] - lane_ops / 0.180e-3 / 1e12) < 1e-2
    assert abs(roof["frac"] - lane_ops / 0.180e-3 / 1e12 / 78.6432) < 1e-3 and roof["frac"] < 1.0
    assert roof["traffic"] == int(190_000 * 1024 * 2 + 4_000 * 1024)          # FETCH x 2 (gfx950 correction) + WRITE
    assert roof["nominal_ops_per_cell_equiv"] > roof["peak"]                     # the 5-ops-per-cell model is not a fraction
    # a launch of another size of the same workload is priced per pair
    half = bench.roofline_of("bitparallel_tiled", 0.090, 2_136_969_065, 70_000_000, "tokens64", 500_000, constants)
    assert abs(half["frac"] - roof["frac"]) < 1e-3 and half["traffic"] * 2 == roof["traffic"]
    missing = bench.roofline_of("banded", 0.3, 1, 1, "utf8_lines", 10, constants)
    assert missing["frac"] is None and missing["traffic"] is None and "no PMC constants" in missing["note"]


def test_pmc_constants_folds_counter_passes(tmp_path):
    run = tmp_path / "pmc"
    run.mkdir()
    rows = []
    for dispatch in (1, 2, 3):
        rows.append({"Dispatch_Id": dispatch, "Kernel_Name": "void swh::k_bitparallel_tiled<unsigned char, 8>(swh::TiledArgs)",
                     "Counter_Name": "SQ_INSTS_VALU", "Counter_Value": 80e6 + dispatch * 1e6})
        rows.append({"Dispatch_Id": dispatch, "Kernel_Name": "void swh::k_tape_longest<unsigned int>(...)",
                     "Counter_Name": "SQ_INSTS_VALU", "Counter_Value": 5.0})
    with open(run / "a_counter_collection.csv", "w", newline="") as handle:
        writer = csv.DictWriter(handle, fieldnames=list(rows[0]))
        writer.writeheader()
        writer.writerows(rows)
    with open(run / "b_counter_collection.csv", "w", newline="") as handle:
        writer = csv.DictWriter(handle, fieldnames=list(rows[0]))
        writer.writeheader()
        for dispatch in (7, 8):   # FETCH_SIZE arrives as several rows per dispatch (one per XCD): they add up
            for part in range(8):
                writer.writerow({"Dispatch_Id": dispatch, "Kernel_Name": rows[0]["Kernel_Name"], "Counter_Name": "FETCH_SIZE", "Counter_Value": 1000.0})
    out = tmp_path / "constants.json"
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_constants.py"), str(run), "--workload", "tokens64",
                           "--pairs", "1000000", "--out", str(out)], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    entry = json.load(open(out))["kernels"]["bitparallel_tiled|tokens64"]
    assert entry["dispatches"] == 3 and abs(entry["valu_insts"] - 82e6) < 1
    assert abs(entry["valu_insts_per_pair"] - 82.0) < 1e-6 and entry["fetch_kb"] == 8000.0 and entry["write_kb"] is None
