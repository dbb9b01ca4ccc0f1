"""bench.py's one-line JSON contract (driver-facing): keys, types, roofline / cpu_baseline objects, the multi-rank switch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_without_a_launcher_starts_n_ranks(n):
    """`python bench.py --gpus N` with WORLD_SIZE unset must run N ranks (fresh child processes over
    torch.distributed.run), never one rank under an n_gpus: 1 line.  --dry-run takes the whole launch / rendezvous /
    barrier / max-over-ranks / one-JSON-line path with gloo and no GPU work."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--dry-run", "--backend", "gloo"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["global_batch"] == 8 * n and d["dry_run"] is True and d["steps"] == 3
    # under a process group a rank replays LINEAR graphs, three in flight (no branch streams whose hardware-queue placement
    # could differ between ranks: DESIGN.md section 6)
    assert d["config"]["launch"] == "3 batches in flight, linear graphs"


def test_world_size_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_host_cores_respects_affinity():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    n = bench.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


@pytest.mark.gpu
def test_bench_line_schema(dev):
    env = dict(os.environ, MSPI_BENCH_CPU_BUDGET_S="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--stream-layouts", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "clips_per_sec" and d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"] and d["value"] > 50
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["avg_launch_us"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "clips/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["value_b1"] > 0 and cb["batch"] == 8
    # SURVEY 8d metric (2): ms/clip including the post-process kernels, from the product's own launch path
    assert d["ms_per_clip_with_postproc"] >= 0.9 * d["ms_per_clip"] and d["saliency_map_ms_per_clip"] > 0
    assert "GraphPipeline" in d["config"]["launch"]
