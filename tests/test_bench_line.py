"""The line bench.py prints for the driver: ONE line, compact, parseable, carrying the contract's keys, `roofline` and
`cpu_baseline`.  Round 4's line (profiles/r04/bench_line.json, 35.7 KB) was longer than the driver's 2 000-character tail and was
recorded as `parsed: null`; bench.compact_line() is held to that tail here, on the recorded full object of that very run."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _recorded():
    return json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_line.json")))


def test_compact_line_fits_the_drivers_tail_and_round_trips():
    b = _bench()
    full = _recorded()
    line = json.dumps(b.compact_line(full), separators=(",", ":"))
    assert "\n" not in line
    assert len(line) < 2000, len(line)
    back = json.loads(line)
    for key in CONTRACT:
        assert key in back, key
    assert back["config"]["workload"].startswith("uniform 1000000x100000")
    assert "model" not in back["config"]
    rf = back["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-3
    assert rf["dominant_kernel"]["kernel"].startswith("half_sweep_")
    cb = back["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(back["value"] / full["value"] - 1) < 1e-4 and abs(back["ms_per_step"] / full["ms_per_step"] - 1) < 1e-3
    # every line of the metric is there, each with its two roofline fractions; the CPU legs beside the ones that have one
    byc = back["by_config"]
    for name in ("pg_maxupd10_f32_defaults", "pg_maxupd1_f32", "pg_maxupd10_f32_finite", "cg_f32", "tncg_f32", "cg_f64", "tncg_f64_c5"):
        assert {"ms", "frac", "valu"} <= set(byc[name]), name
    for name in ("pg_maxupd10_f32_defaults", "pg_maxupd10_f32_finite", "cg_f64", "tncg_f64_c5"):
        assert byc[name]["cpu_s"] > 0
    assert set(back["results_alive"]) == {"A_nonzero_frac", "B_nonzero_frac"}


def test_compact_line_of_a_minimal_object():
    """--no-cpu / --no-extra / N > 1 runs have no cpu_baseline, no extra and no by_config: still one short valid line"""
    b = _bench()
    full = {"metric": "nonzeros/sec per full A+B sweep", "value": 1.0e9, "unit": "nnz/s", "n_gpus": 8, "steps": 10, "warmup": 3,
            "ms_per_step": 1.5, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "uniform ...", "baseline_config": "C4: x", "sharding": "one fixed matrix: " + "r" * 2000},
            "roofline": {"bound": "hbm", "achieved": 100.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.0125, "traffic": None,
                         "traffic_source": None, "valu": {"frac": 0.1}, "dominant_kernel": None}}
    line = json.dumps(b.compact_line(full), separators=(",", ":"))
    assert len(line) < 2000
    back = json.loads(line)
    assert back["n_gpus"] == 8 and back["roofline"]["frac"] == 0.0125 and "cpu_baseline" not in back


def test_main_prints_the_compact_line_last():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tail = src[src.rindex("write_full(final_line)"):]
    assert "compact_line(final_line)" in tail and tail.count("print(") == 1


def test_compact_line_of_a_multi_gpu_object_carries_what_the_communicator_said():
    b = _bench()
    full = dict(_recorded())
    full["n_gpus"] = 8
    full["comm"] = {"backend": "nccl", "comm_ranks": 8, "rccl_ranks": 8, "rccl_version": "2.22.3",
                    "per_rank_nnz": {"A_half_csr": [12493725] * 8, "B_half_csc": [12493725] * 8},
                    "exchange": {"rows_ms_per_sweep": 0.4321987, "colsum_partials_ms_per_sweep": 0.0123456, "exchanges_per_sweep": 7}}
    line = json.dumps(b.compact_line(full), separators=(",", ":"))
    assert len(line) < 2000, len(line)
    back = json.loads(line)
    assert back["rccl_ranks"] == 8 and back["comm"]["ranks"] == 8 and back["comm"]["backend"] == "nccl"
    assert len(back["comm"]["nnz_per_rank"]) == 8 and back["comm"]["exchange_ms"] == 0.4322


def test_bench_gpus_n_starts_its_own_ranks_and_returns_their_exit_code():
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset launches the ranks itself (no GPU here: every rank stops with the "needs an
    MI355X" message, the launcher fails, and the parent hands that exit code on instead of a JSON line)"""
    import subprocess
    import sys
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                         capture_output=True, text=True, timeout=600, env=env)
    out = res.stdout + res.stderr
    import torch
    if torch.cuda.is_available():
        return   # (on a GPU box the real run is tests/test_gpu_dist2.py's)
    assert res.returncode != 0
    assert "needs an MI355X" in out and "launch with torch.distributed.run" not in out
