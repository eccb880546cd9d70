"""CPU-side checks of bench.py's contract pieces that do not need a GPU."""
import importlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constants_match_baseline_and_survey():
    bench = importlib.import_module("bench")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert (bench.H, bench.W, bench.VN, bench.HN, bench.BATCH) == (480, 640, 9, 1024, 32)  # BASELINE.json configs[2]
    assert bench.BYTES_PER_VOTING == 24_576_072  # SURVEY.md section 8(d), int64 mask
    assert base["published"] == {}  # => vs_baseline must stay null
    assert "votings/s" in base["metric"]


def test_usable_cores_and_traffic_helpers():
    bench = importlib.import_module("bench")
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    t = bench.measured_traffic("score_kernel")
    assert t is None or t > 0
    assert bench.measured_traffic("no_such_kernel") is None


def test_cpu_baseline_leg_runs_on_a_tiny_budget():
    bench = importlib.import_module("bench")
    from pvnet_amd import synth
    mask, planar, _ = synth.make_batch(bench.BATCH, radius=12, h=bench.H, w=bench.W, noise=True)
    r = bench.cpu_baseline([(None, None, mask, planar)], 0.5)
    assert r["kind"] == "port" and r["unit"] == "votings/s" and r["value"] > 0 and r["cores"] >= 1
    assert "sample" in r and r["value_1_core"] > 0
