"""bench.py's output contract, checked without a GPU (the round-3 line grew to 24.5 KB, the driver keeps an 8 KB
tail, BENCH_r03.parsed was null).  The LAST stdout line is one compact JSON object < 4 KB; everything else goes to
bench_detail.json.  Also: the per-kernel roofline rows carry each launch's own bytes."""
import copy
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def recorded():
    """the full result dict of a real run (round 3's 24.5 KB line, kept under profiles/)"""
    with open(os.path.join(ROOT, "profiles", "r03_bench_v5.json")) as f:
        return json.load(f)


def test_compact_line_of_a_recorded_run_fits_and_round_trips(bench, recorded):
    assert len(json.dumps(recorded)) > 20000                     # the input really is the oversized result
    text = bench.compact_line(copy.deepcopy(recorded))
    assert "\n" not in text and len(text) < 4096, len(text)
    line = json.loads(text)
    for k in REQUIRED:
        assert k in line, k
    assert line["value"] == pytest.approx(recorded["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(recorded["ms_per_step"], rel=1e-5)
    assert line["dtype"] == "f32" and line["unit"] == "images/s" and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert rf["kernel"] == recorded["roofline"]["kernel"] and "traffic" in rf
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    other = line["images_per_s_other_protocols"]
    assert other["one_image_at_a_time"] == pytest.approx(recorded["one_image_at_a_time"]["value"], rel=1e-5)
    assert all(not isinstance(v, (dict, list)) for v in other.values())      # one scalar per alternative measurement


def test_compact_line_of_an_eight_rank_run_summarises_the_ranks(bench, recorded):
    out = copy.deepcopy(recorded)
    out["n_gpus"] = 8
    out["ranks"] = [{"rank": r, "device": r, "host": "node-with-a-rather-long-hostname-%d" % r, "ms_per_step": 4.4 + 0.01 * r,
                     "cpu_affinity": 32, "device_mem_gb": 5.0 + 0.1 * r} for r in range(8)]
    out["gather_transport"], out["rccl_version"], out["dist_backend"] = "rccl", 22304, "nccl"
    text = bench.compact_line(out)
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    assert "ranks" not in line
    assert line["ranks_ms_per_step"] == {"min": 4.4, "max": 4.47, "mean": pytest.approx(4.435), "n": 8}
    assert line["ranks_device_mem_gb"]["max"] == pytest.approx(5.7)
    assert line["gather_transport"] == "rccl" and "8 ranks" in line["config"]["parallelism"]


def test_compact_line_sheds_optional_blocks_rather_than_overflow(bench, recorded):
    out = copy.deepcopy(recorded)
    out["kernel_ms_per_image"] = {("some_kernel_with_a_long_name_%03d" % i): 0.001 * i for i in range(400)}
    for i in range(200):
        out["alt_math_%03d" % i] = {"math": "mode_%03d" % i, "value": float(i), "roofline": {"frac": 0.1},
                                    "max_rel_diff_vs_fp32": {"a": 1e-4}}
    text = bench.compact_line(out)
    assert len(text) < 4096
    line = json.loads(text)
    for k in REQUIRED + ("roofline", "cpu_baseline"):
        assert k in line, k


def test_emit_prints_the_compact_line_last_and_writes_the_detail_file(bench, recorded, tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out = copy.deepcopy(recorded)
    out["config"]["config_name"] = "vgg16"
    bench.emit(out)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    last = json.loads(lines[-1])
    assert len(lines[-1]) < 4096 and last["detail"] == "bench_detail.json"
    with open(tmp_path / "bench_detail.json") as f:
        detail = json.load(f)
    assert detail["roofline_by_kernel"] == recorded["roofline_by_kernel"]       # nothing is lost, it just is not on the last line


def test_roofline_rows_carry_each_launchs_own_bytes(bench):
    """Two layers with the same algorithmic flop and different bytes (conv1_2 and conv4_2 are both 44.24 GFLOP: 307 and 48 MB):
    separate rows, and the scope's aggregate row sums the launches' own bytes (round 3 charged the first member's bytes to all)."""
    fl = 44.2368e9
    records = []
    for _ in range(3):                                            # three event images
        records += [("conv3x3_wino_mfma", 0.30, fl, 307.3e6), ("conv3x3_wino_mfma", 0.20, fl, 47.8e6),
                    ("conv3x3_wino_mfma", 0.21, fl, 47.8e6), ("fc_mfma", 0.45, 61.66e9, 446e6)]
    rows = bench.roofline_by_kernel(records, 3)
    conv = [r for r in rows if r["scope"] == "conv3x3_wino_mfma" and "all launches" not in r["what"]]
    assert sorted(round(r["algorithmic_mb_per_launch"], 1) for r in conv) == [47.8, 307.3]
    by_mb = {round(r["algorithmic_mb_per_launch"], 1): r for r in conv}
    assert by_mb[307.3]["launches_per_image"] == 1 and by_mb[47.8]["launches_per_image"] == 2
    assert by_mb[47.8]["avg_launch_ms"] == pytest.approx(0.205)
    agg = [r for r in rows if r["scope"] == "conv3x3_wino_mfma" and "all launches" in r["what"]]
    assert len(agg) == 1
    assert agg[0]["algorithmic_mb_per_image"] == pytest.approx(307.3 + 2 * 47.8)
    assert agg[0]["launches_per_image"] == 3
