"""CPU-side checks of bench.py's contract pieces that do not need a GPU."""
import importlib
import json
import os
import subprocess
import sys

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
    t, tag, name = bench.measured_traffic("score_exact_kernel")
    assert t is None or (t > 0 and name.startswith("score_exact_kernel<"))
    assert bench.measured_traffic("no_such_kernel")[0] is None


def test_cpu_baseline_leg_runs_on_a_tiny_budget():
    bench = importlib.import_module("bench")
    from pvnet_amd import synth
    mask, planar, _ = synth.make_batch(bench.BATCH, radius=12, h=bench.H, w=bench.W, noise=True)
    r = bench.cpu_baseline([(None, None, mask, planar)], 0.5)
    assert r["kind"] == "port" and r["unit"] == "votings/s" and r["value"] > 0 and r["cores"] >= 1
    assert "sample" in r and r["value_1_core"] > 0


def _run_bench(*args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # exactly ONE JSON line on stdout, whatever the launcher prints elsewhere
    return json.loads(lines[0])


def test_gpus_n_launches_itself_world_2_gloo_stub():
    """`python bench.py --gpus 2` (the driver's command form, no torchrun around it) must start its own two ranks, run
    the process-group / all-gather / max-over-ranks path and print one line with n_gpus 2.  CPU: gloo + the stub voter."""
    r = _run_bench("--gpus", "2", "--steps", "20", "--warmup", "2", "--stub")
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 2 and r["stub"] is True
    assert r["gather_ok"] is True  # every rank's block arrived in rank order
    assert len(r["per_rank_votings_per_s"]) == 2 and all(x > 0 for x in r["per_rank_votings_per_s"])
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["config"]["global_batch"] == 64
    # whole-job value = all ranks' units / the slowest rank's time
    assert abs(r["value"] - 2 * min(r["per_rank_votings_per_s"])) / r["value"] < 1e-6


def test_plain_and_single_rank_launcher_runs_agree_in_shape():
    a = _run_bench("--gpus", "1", "--steps", "10", "--warmup", "1", "--stub")
    assert a["n_gpus"] == 1 and len(a["per_rank_votings_per_s"]) == 1
    assert abs(a["value"] - a["per_rank_votings_per_s"][0]) / a["value"] < 1e-6


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.pop("MASTER_ADDR", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_roofline_helpers_read_the_committed_profiles():
    bench = importlib.import_module("bench")
    u, tag = bench.measured_mfma_busy()
    assert u is None or 0.0 < u < 1.0
    for k in bench.PATH_KERNELS:
        t, tag, name = bench.measured_traffic(k)
        # one instantiation, never a mix of them (a path kernel may have variants: the mask kernel's 16-byte-load form)
        assert t is None or (t >= 0 and name.split("<")[0] in bench.KERNEL_VARIANTS.get(k, (k,)))
    # a profile is only quoted when it was taken from this build's kernel sources
    tag, kernels, fresh = bench.committed_profile("_traffic.json")
    if tag is not None and not fresh:
        assert all(bench.measured_traffic(k)[0] is None for k in bench.PATH_KERNELS)
    assert len(bench.source_hash()) == 16


def test_gpu_sampler_is_harmless_without_a_gpu():
    bench = importlib.import_module("bench")
    import time
    with bench.GpuSampler(0, period=0.001) as s:
        time.sleep(0.01)
    r = s.summary()
    assert set(r) == {"sclk_mhz", "power_w", "source"}
