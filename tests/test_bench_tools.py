"""CPU checks of the measurement plumbing: the roofline arithmetic of bench.py and the PMC folding of
tools/pmc_constants.py (the numbers themselves come from the GPU box; the formulas are checked here)."""
import csv
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def test_roofline_fraction_is_executed_work_over_peak():
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module")
    constants = {"kernels": {"bitparallel_tiled|tokens64": {
        "pairs_per_call": 1_000_000, "valu_insts": 86.0e6, "fetch_kb": 190_000.0, "write_kb": 4_000.0, "source": "unit test",
        "source_digest": bench.source_digest("bitparallel_tiled")},
        "wavefront|protein4k|affine": {"pairs_per_call": 10, "valu_insts": 1e6, "fetch_kb": None, "write_kb": None, "source_digest": "0" * 16}}}
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
    # an entry is tied to the sources it was measured on: same digest -> fresh, another digest -> stale; variants are keyed
    assert roof["pmc_stale"] is False
    stale = bench.roofline_of("wavefront", 1.0, 100, 100, "protein4k", 10, constants, variant="affine", model="nw_affine")
    assert stale["pmc_stale"] is True and stale["frac"] is not None and stale["traffic"] is None
    assert bench.roofline_of("wavefront", 1.0, 100, 100, "protein4k", 10, constants, variant="linear")["frac"] is None


def test_source_digest_follows_the_kernel_sources(tmp_path, monkeypatch):
    ks = load(os.path.join(ROOT, "tools", "kernel_sources.py"), "kernel_sources_module")
    before = ks.source_digest("short_tiled")
    assert before == ks.source_digest("short_tiled") and len(before) == 16
    for name in ("short.hip", "common.hpp"):
        (tmp_path / name).write_bytes(open(os.path.join(ks.CSRC, name), "rb").read())
    monkeypatch.setattr(ks, "CSRC", str(tmp_path))
    assert ks.source_digest("short_tiled") == before
    with open(tmp_path / "short.hip", "ab") as handle:
        handle.write(b"\n// one more line\n")
    assert ks.source_digest("short_tiled") != before
    for stamp, (needle, sources) in ks.KERNELS.items():
        assert "k_" in needle and "<" in needle and sources   # a substring of the kernel symbol as rocprofv3 prints it


def test_bench_legs_cover_every_baseline_config():
    """`configs` of the bench line: C1, C3 (prepared and raw), C4 linear / affine / full byte alphabet, C5 -- BASELINE.json's
    configs beside the headline C2, each at its full size -- and the two shapes round 4 added kernels for (C3's lines at k = 100,
    NW on word-sized strings)."""
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module_legs")
    assert bench.DEFAULT_LEGS == ["c1", "c3", "c3_raw", "c3_raw_cold", "c3_raw_forget", "utf8_unbounded_raw", "utf8_unrelated_raw", "c3_k100", "c4_linear", "c4_affine", "c4_bytes", "c4_letters52", "c5", "nw_words",
                                  "sw_linear", "sw_affine", "cross_lev", "cross_nw"]
    # Smith-Waterman on C4's sequences (bench.rs:882-963) and the reference's own call shape, compute_into(queries, candidates, &mut matrix) (bench.rs:478-486)
    assert bench.LEGS["sw_linear"]["local"] and bench.LEGS["sw_affine"]["gaps"] == (-11, -1) and bench.LEGS["cross_lev"]["side"] == bench.LEGS["cross_nw"]["side"] == 2048
    # the reference's literal UTF-8 calls (bench.rs:538-546): raw tapes, no bound -- and raw tapes the scope's beliefs do not cover
    assert not bench.LEGS["utf8_unbounded_raw"]["prepared"] and "bound" not in bench.LEGS["utf8_unbounded_raw"] and bench.LEGS["c3_raw_cold"]["cold"] >= 2
    assert all(bench.LEGS[name]["check"] >= 100 for name in ("c4_linear", "c4_affine", "c4_bytes", "c4_letters52"))
    assert bench.LEGS["c4_letters52"]["letters"] == 52
    sizes = {name: (leg["workload"], leg["pairs"]) for name, leg in bench.LEGS.items()}
    assert sizes["c1"] == ("words16", 10_000) and sizes["c2"] == ("tokens64", 1_000_000) and sizes["c3"] == ("utf8_lines", 100_000)
    assert sizes["c4_linear"] == sizes["c4_affine"] == ("protein4k", 10_000) and sizes["c4_bytes"] == sizes["c4_letters52"] == ("bytes4k", 10_000)   # configs[3] as worded: 10 K sequences
    assert sizes["c5"] == ("short_words", 20_000_000) and bench.LEGS["c3"]["bound"] == 32 and not bench.LEGS["c3_raw"]["prepared"]
    assert sizes["c3_k100"] == ("utf8_lines", 100_000) and bench.LEGS["c3_k100"]["bound"] == 100 and sizes["nw_words"] == ("words16", 4_000_000)
    assert bench.kernel_family("align_short_affine_w16") == "align_short" and bench.kernel_family("align_wide_local_w128") == "align_wide"


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
                           "--pairs", "1000000", "--calls", "3", "--out", str(out)], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    entry = json.load(open(out))["kernels"]["bitparallel_tiled|tokens64"]
    assert entry["calls"] == 3 and entry["dispatches_per_call"] == 1.0 and abs(entry["valu_insts"] - 82e6) < 1
    # FETCH_SIZE: 2 dispatches x 8 rows x 1000 KB over the 3 calls of the command (every pass runs the same command)
    assert abs(entry["valu_insts_per_pair"] - 82.0) < 1e-6 and abs(entry["fetch_kb"] - 16000.0 / 3) < 0.1 and entry["write_kb"] is None
    assert len(entry["source_digest"]) == 16
    # a second variant of the same kernel and workload lands beside it
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_constants.py"), str(run), "--workload", "tokens64", "--pairs", "1000000",
                    "--calls", "1", "--variant", "k8", "--out", str(out)], check=True, capture_output=True)
    book = json.load(open(out))["kernels"]
    assert set(book) == {"bitparallel_tiled|tokens64", "bitparallel_tiled|tokens64|k8"} and abs(book["bitparallel_tiled|tokens64|k8"]["valu_insts"] - 246e6) < 1


def test_bench_refuses_a_world_size_other_than_gpus():
    """`--gpus N` is a promise: a process group of another size is an error before anything touches a GPU, never a quiet
    measurement of fewer devices (the driver computes scaling efficiency from `n_gpus`)."""
    bench = os.path.join(ROOT, "bench.py")
    for gpus, world in (("2", "3"), ("8", "1"), ("1", "2")):
        env = dict(os.environ, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        done = subprocess.run([sys.executable, bench, "--gpus", gpus, "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
        assert done.returncode != 0 and f"--gpus {gpus} but WORLD_SIZE={world}" in done.stderr, (gpus, world, done.stderr[-400:])
        assert not [row for row in done.stdout.splitlines() if row.startswith("{")]          # and no line


def test_bench_starts_its_own_ranks_when_there_is_no_world(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: the N ranks are started as a CHILD (`python -m torch.distributed.run
    --nproc-per-node N ... bench.py <same arguments>`) by a process that has imported neither torch nor the library, and the
    child's exit code is handed on."""
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module_launch")
    seen = {}

    class Child:
        pid = 0
        stdout = iter(['{"metric": "GCUPS", "value": 1.0}\n'])

        def wait(self, timeout=None):
            return 7

    def fake_popen(cmd, env=None, **kwargs):
        seen["cmd"], seen["env"] = cmd, env
        return Child()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("main() must leave with the child's exit code")
    except SystemExit as leave:
        assert leave.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    source = open(os.path.join(ROOT, "bench.py")).read()
    # nothing at module level pulls torch or the library in: both are imported inside the functions that need a GPU
    assert not [row for row in source.splitlines() if row.startswith(("import torch", "import stringwars_amd", "from stringwars_amd"))]


def test_bench_run_that_fails_still_ends_with_one_error_line(monkeypatch, capsys):
    """A run whose ranks die before rank 0 has printed its line, or hang past --launch-timeout, ends with a non-zero exit code AND one JSON
    line carrying `error` (the driver reads the last JSON line of stdout): the launcher prints it when no rank did."""
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module_fail")
    killed = []

    def child_of(code, rows, hang=False):
        class Child:
            pid = 424242
            stdout = iter(rows)

            def __init__(self):
                self.waits = 0

            def wait(self, timeout=None):
                self.waits += 1
                if hang and self.waits == 1:
                    raise subprocess.TimeoutExpired("torchrun", timeout)
                return code
        return Child()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "killpg", lambda pid, sig: killed.append(pid))
    for code, rows, hang, want_code, want_error in ((1, ["Traceback ...\n", "RuntimeError: ncclCommInitRank failed\n"], False, 1, "exited with code 1"),
                                                     (-9, [], True, -9, "did not finish within --launch-timeout"),
                                                     (1, ['{"metric": "m", "value": null, "error": "rank 0: boom"}\n'], False, 1, None)):
        monkeypatch.setattr(subprocess, "Popen", lambda cmd, **kw: child_of(code, rows, hang))
        monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--launch-timeout", "5"])
        try:
            bench.main()
            raise AssertionError("a failed run must not return")
        except SystemExit as leave:
            assert leave.code == want_code
        lines = [json.loads(row) for row in capsys.readouterr().out.splitlines() if row.startswith("{")]
        assert len(lines) == 1 and lines[0]["value"] is None
        if want_error:
            assert want_error in lines[0]["error"] and lines[0]["steps"] == 2 and lines[0]["n_gpus"] == 8
        else:
            assert lines[0]["error"] == "rank 0: boom"          # the rank's own line is THE line: the launcher adds none
    assert killed == [424242]


def test_bench_rank_that_dies_ends_the_run_with_an_error_line():
    """`python bench.py --gpus 2` whose rank 1 exits before the process group is up (a GPU that did not come up, an RCCL that failed to
    initialise): torch.distributed.run tears rank 0 down, the run ends with a non-zero exit code and a JSON line carrying `error`, within a
    bounded time -- no hang, no half line. (Runs without a GPU: rank 0 then fails on its own, which is an error line all the same.)"""
    bench = os.path.join(ROOT, "bench.py")
    started = time.time()
    done = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "1", "--warmup", "0", "--pairs", "2000",
                           "--no-cpu-baseline", "--no-configs", "--die-rank", "1", "--die-at", "start", "--collective-timeout", "30", "--launch-timeout", "150"],
                          capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert done.returncode != 0 and time.time() - started < 300
    lines = [json.loads(row) for row in done.stdout.splitlines() if row.startswith("{")]
    assert lines and "error" in lines[-1] and lines[-1]["value"] is None and lines[-1]["n_gpus"] == 2, (done.stdout[-600:], done.stderr[-600:])


def test_trace_tools_read_rocprofv3_csvs(tmp_path):
    """tools/kernel_gaps.py and tools/api_timeline.py on a hand-made pair of rocprofv3 traces: the kernels of the last call in start
    order with their lengths, and the host API calls of that call on the same time line."""
    kernels = tmp_path / "run" / "x_kernel_trace.csv"
    kernels.parent.mkdir()
    rows = [("swh::k_utf8_tile_decode(a)", 1000, 251000, 1), ("void swh::k_banded<unsigned int, 36, 32>(b)", 300000, 610000, 1),
            ("swh::k_utf8_tile_decode(a)", 1000000, 1250000, 1), ("other_kernel(c)", 1260000, 1270000, 1),
            ("void swh::k_banded<unsigned int, 36, 32>(b)", 1300000, 1612000, 1)]
    with open(kernels, "w", newline="") as handle:
        writer = csv.writer(handle)
        writer.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id"])
        writer.writerows(rows)
    api = tmp_path / "run" / "x_hip_api_trace.csv"
    with open(api, "w", newline="") as handle:
        writer = csv.writer(handle)
        writer.writerow(["Function", "Start_Timestamp", "End_Timestamp"])
        writer.writerows([("hipLaunchKernel", 990000, 995000), ("hipStreamSynchronize", 1305000, 1620000), ("hipLaunchKernel", 100, 200)])
    gaps = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_gaps.py"), str(kernels), "1"], capture_output=True, text=True, timeout=60)
    assert gaps.returncode == 0, gaps.stderr
    lines = gaps.stdout.strip().splitlines()
    assert len(lines) == 3 and "k_banded" in lines[-1] and "312.0" in lines[-1] and "other_kernel" not in gaps.stdout   # the last 3 x 1 swh:: kernels
    line = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "api_timeline.py"), str(tmp_path / "run"), "k_banded"], capture_output=True, text=True, timeout=60)
    assert line.returncode == 0, line.stderr
    out = line.stdout.strip().splitlines()
    assert len(out) == 4, out
    assert "A hipLaunchKernel" in out[0] and "K swh::k_utf8_tile_decode" in out[1] and "K swh::k_banded" in out[2] and "A hipStreamSynchronize" in out[3]


def test_headline_line_fits_its_budget():
    """The driver parses the LAST stdout line: it is the headline only, held to 6 KB at N = 1 and 8 KB at N = 8 whatever the
    per-config entries hold (round 5's 24 KB line could not be parsed). `fit_line` sheds the least important fields first and
    never the contract's keys; the full entries go to --details-out."""
    bench = load(os.path.join(ROOT, "bench.py"), "bench_module_line")
    roof = {"bound": "valu", "kernel": "bitparallel_tiled", "kernel_ms": 0.1345, "unit": "Tint32op/s", "peak": 78.6, "achieved": 33.9, "frac": 0.4321,
            "traffic": 600744140, "traffic_detail": {"vs_algorithmic": 4.29}, "pmc_source": "x" * 400, "measured": "y" * 400,
            "hbm": {"achieved": 1041.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.13, "algorithmic_bytes": 140028467}}
    compact = bench.compact_roofline(roof)
    assert compact["frac"] == 0.4321 and compact["traffic"] == 600744140 and compact["traffic_vs_algorithmic"] == 4.29 and "pmc_source" not in compact
    legs = [{"config": name, "value": 1234.5, "ms_per_call": 1.5, "pairs": 10_000, "roofline": dict(roof), "parity_vs_oracle": True, "workload": "w" * 300}
            for name in bench.DEFAULT_LEGS] + [{"config": "broken", "error": "RuntimeError: " + "z" * 900}]
    for world in (1, 8):
        line = {"metric": "GCUPS", "value": 31000.0, "unit": "GCUPS", "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.137, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "C2 " + "c" * 150},
                "roofline": compact, "cpu_baseline": {"value": 13.9, "unit": "GCUPS", "cores": 1, "kind": "port", "sample": "s" * 200},
                "cpu_baselines": {f"cpu::row{i}": {"value": 1.0, "cores": 1} for i in range(5)},
                "configs": {entry["config"]: bench.compact_leg(entry) for entry in legs}, "parity_vs_oracle": True, "gather_ok": None,
                "ranks_seen": {"world_size": world, "backend": "nccl", "rccl_version": "2.26.6", "distinct_devices": world}, "details": "bench_configs.json"}
        text = bench.fit_line(line, world)
        assert len(text) <= bench.line_budget(world) and "\n" not in text
        back = json.loads(text)
        assert back["roofline"]["frac"] == 0.4321 and back["cpu_baseline"]["cores"] == 1 and back["configs"]["c5"]["parity"] is True
        assert len(back["configs"]["broken"]["error"]) <= 160
    # a pathological line (a config map that cannot fit) still fits: the map is replaced by a pointer to the details file
    line["configs"] = {f"leg{i}": {"value": 1.0, "frac": 0.5, "parity": True, "note": "n" * 200} for i in range(60)}
    text = bench.fit_line(line, 1)
    assert len(text) <= bench.line_budget(1) and json.loads(text)["value"] == 31000.0


def test_no_kernel_lost_an_occupancy_class():
    """Every kernel of the built objects keeps at least the waves per SIMD its registers allowed when profiles/r6/kernel_resources.json
    was written (tools/kernel_occupancy.py --write; no GPU needed). Round 6 lost 1.6 x on a cross-product to one register -- 256 -> 257
    VGPRs, two waves per SIMD -> one -- and only a benchmark table noticed; a deliberate change regenerates the table."""
    tool = load(os.path.join(ROOT, "tools", "kernel_occupancy.py"), "kernel_occupancy_module")
    assert tool.waves_per_simd(256) == 2 and tool.waves_per_simd(257) == 1 and tool.waves_per_simd(128) == 4 and tool.waves_per_simd(40) == 8
    committed = json.load(open(tool.TABLE))["kernels"]
    built = tool.build_table()
    assert len(built) >= 300 and set(committed) <= set(built), sorted(set(committed) - set(built))[:5]
    lost = {k: (committed[k]["waves_per_simd"], built[k]["waves_per_simd"], built[k]["vgpr_count"]) for k in committed
            if built[k]["waves_per_simd"] < committed[k]["waves_per_simd"]}
    assert not lost, lost
    # the hot kernels, by name: the classes DESIGN.md quotes
    by = lambda needle: [e for k, e in built.items() if needle in k]
    assert all(e["waves_per_simd"] >= 4 for e in by("k_short_tiled<")) and all(e["waves_per_simd"] >= 2 for e in by("k_align_cross_wide<128"))
