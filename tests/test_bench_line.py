"""The bench line the driver parses: ONE stdout line that fits the driver's 8 KB stdout tail whole.

CPU only: the line is built from a canned full result (round 3's 23 KB record, committed under profiles/) through the same
`emit_line` bench.py's main() calls.
"""
import io
import json
import math
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

CANNED = ROOT / "profiles" / "r03_bench_default.json"


@pytest.fixture()
def full():
    return json.loads(CANNED.read_text())


def _emit(full, tmp_path):
    buf = io.StringIO()
    line = bench.emit_line(full, buf, tmp_path / "bench_detail.json")
    assert buf.getvalue() == line + "\n"
    assert "\n" not in line
    return line


def test_line_fits_the_drivers_tail_and_parses(full, tmp_path):
    assert len(json.dumps(full)) > 20000  # the record that broke round 3
    line = _emit(full, tmp_path)
    assert len(line) < 8000
    for tok in ("NaN", "Infinity"):
        assert tok not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["dtype"] == "u16" and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    wl = d["config"]["workload"]
    assert "BoxBlur" in wl and "13" in wl and "3840x2160" in wl and "YUV420P16" in wl
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert math.isclose(rf["frac"], rf["achieved"] / rf["peak"], rel_tol=1e-4)
    assert {"kernel", "avg_launch_us", "traffic", "frac_median"} <= set(rf)
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] in ("port", "reference")
    # config: scalars and small all-scalar records only (no prose blocks, no candidate lists)
    for k, v in d["config"].items():
        assert not isinstance(v, list), k
        if isinstance(v, dict):
            assert len(json.dumps(v)) <= 300, k
    assert "note" not in d["config"].get("placement", {}) and "rank0_pcie_path" not in d["config"].get("pcie_fed", {})
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 3185049600
    assert isinstance(d["config"]["pcie_fed_fps"], float)
    # every leg: value / unit / frac / limit only
    assert set(d["others"]) == set(full["others"])
    for name, leg in d["others"].items():
        assert set(leg) <= {"value", "unit", "frac", "whole_call_frac", "limit", "cpu", "error", "threads", "host_link_GBps"}, name
        assert "value" in leg or "error" in leg, name
    # the CPU side of every BASELINE config travels on the line: value, cores and the ratio
    for name in ("bilateral_1080p", "bilateral_4k", "ssimulacra2_4k", "eedi3_1080p", "boxblur_1080p", "boxblur_1080p_5pass"):
        if "cpu_baseline" in full["others"].get(name, {}):
            cpu = d["others"][name]["cpu"]
            assert {"value", "cores", "x"} <= set(cpu) <= {"value", "cores", "x", "single", "scaling"} and cpu["cores"] >= 1, name
            assert math.isclose(cpu["x"], d["others"][name]["value"] / cpu["value"], rel_tol=1e-4), name
    assert {"bilateral_1080p", "bilateral_4k", "ssimulacra2_4k", "eedi3_1080p"} <= {n for n, leg in d["others"].items() if "cpu" in leg}
    assert d["others"]["bilateral_1080p"]["limit"]["bound"] in ("valu", "lds")
    assert d["detail"] == "bench_detail.json"


def test_sidecar_keeps_everything(full, tmp_path):
    _emit(full, tmp_path)
    side = json.loads((tmp_path / "bench_detail.json").read_text())
    assert set(side) == set(full)
    assert side["config"]["placement"]["note"] == full["config"]["placement"]["note"]
    assert side["others"]["eedi3_1080p"]["workload"] == full["others"]["eedi3_1080p"]["workload"]
    assert math.isclose(side["value"], full["value"], rel_tol=1e-8)


def test_values_survive_rounding(full, tmp_path):
    d = json.loads(_emit(full, tmp_path))
    assert math.isclose(d["value"], full["value"], rel_tol=1e-5)
    assert math.isclose(d["ms_per_step"], full["ms_per_step"], rel_tol=1e-5)
    assert math.isclose(d["roofline"]["frac"], full["roofline"]["frac"], rel_tol=1e-5)
    for name, leg in full["others"].items():
        assert math.isclose(d["others"][name]["value"], leg["value"], rel_tol=1e-5), name
    # integers stay integers
    assert d["steps"] == full["steps"] and isinstance(d["steps"], int) and isinstance(d["n_gpus"], int)


def test_line_stays_bounded_when_the_record_grows(full, tmp_path):
    """twice as many legs, each failing with a long message, non-finite numbers, numpy scalars: still one parseable line < 8000 B"""
    import numpy as np

    for i in range(40):
        full["others"][f"extra_leg_number_{i}_with_a_long_name"] = {"error": "x" * 5000} if i % 2 else {
            "value": np.float64(1234.56789), "unit": "frames/s", "roofline": {"frac": float("nan")}, "limit": {"bound": "valu", "frac": np.float32(0.5)}}
    full["config"]["huge"] = {"note": "y" * 3000}
    full["config"]["nan_scalar"] = float("inf")
    full["roofline"]["launch_us"] = {"each": list(range(5000))}
    line = _emit(full, tmp_path)
    assert len(line) < 8000
    d = json.loads(line)
    assert "huge" not in d["config"] and d["config"]["nan_scalar"] is None
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["cores"] > 0


def test_other_workload_lines(tmp_path):
    """the --workload bilateral / ssimulacra2 / pipeline records go through the same emit"""
    rec = {"metric": "pairs/sec: vszip.SSIMULACRA2 3840x2160 RGBS", "value": 5500.0, "unit": "pairs/s", "n_gpus": 1, "steps": 50, "warmup": 5,
           "ms_per_step": 2.9, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "vszip.SSIMULACRA2 ref vs dist, 3840x2160 RGBS linear", "pairs_per_step_per_gpu": 16},
           "roofline": {"bound": "hbm", "achieved": 1100.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1375, "traffic": None, "kernel": "whole pipeline"}}
    d = json.loads(_emit(rec, tmp_path))
    assert d["config"]["pairs_per_step_per_gpu"] == 16 and d["roofline"]["traffic"] is None and "others" not in d


def test_unwritable_sidecar_does_not_cost_the_line(full, tmp_path):
    buf = io.StringIO()
    line = bench.emit_line(full, buf, tmp_path / "no_such_dir" / "bench_detail.json")
    d = json.loads(line)
    assert "detail" not in d and d["value"] > 0


def test_effective_cpus_takes_the_smallest_of_nominal_affinity_and_quota():
    """the GPU boxes show 256 logical CPUs under a 16-CPU cgroup quota"""
    e = bench.effective_cpus(quota=16.0, affinity=256, nominal=256)
    assert e == {"nominal": 256, "affinity": 256, "quota": 16.0, "effective": 16}
    assert bench.effective_cpus(quota=None, affinity=8, nominal=64)["effective"] == 8
    assert bench.effective_cpus(quota=2.5, affinity=8, nominal=8)["effective"] == 3
    assert bench.effective_cpus(quota=0.5, affinity=8, nominal=8)["effective"] == 1
    here = bench.effective_cpus()
    assert 1 <= here["effective"] <= here["nominal"]


def test_cpu_record_carries_threads_single_thread_rate_and_scaling(monkeypatch):
    """a CPU leg's record: threads, value, single_thread_value, scaling = value / (threads x single); cores = what ran"""
    import time

    monkeypatch.setenv("VSZIP_BENCH_CPU_THREADS", "2")
    assert bench.cpu_threads() == 2

    def unit(_):
        time.sleep(0.01)  # releases the GIL like the oracle's C calls do

    rec = bench._timed_pool(unit, 2, 0.2, "sleep unit", 0.01, single_budget_s=0.1)
    assert {"value", "unit", "cores", "kind", "sample", "threads", "single_thread_value", "scaling", "cores_nominal", "cores_effective"} <= set(rec)
    assert rec["cores"] == rec["threads"] == 2
    assert math.isclose(rec["scaling"], rec["value"] / (2 * rec["single_thread_value"]), rel_tol=1e-9)
    assert 0.7 <= rec["scaling"] <= 1.3 and "scaling_note" not in rec
    leg = bench.compact_leg({"value": 1000.0, "unit": "frames/s", "cpu_baseline": rec})
    assert set(leg["cpu"]) == {"value", "cores", "x", "single", "scaling"}
    assert math.isclose(leg["cpu"]["x"], 1000.0 / rec["value"], rel_tol=1e-9)
