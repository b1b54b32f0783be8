"""The bench line's contract (the driver parses ONE JSON line from `python bench.py ...`): run the script as the driver
does - a child process, a few steps - and check the fields the contract names, including the reference-equivalent pass."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 6 * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 6e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 38 * 512 * 384 * d["config"]["batch_per_gpu"]
    # the same workload with the reference's per-sample background preparation, in the same line
    e = d["reference_equivalent"]
    assert e["background_prep"] == 1 and e["unit"] == "samples/s" and 0 < e["value"] < d["value"]
    # the context proves how it was set up: chains, hardware queues, shard indices of rank 0
    assert d["config"]["chains"] >= 3 and d["config"]["shards"]["first_index_of_steps_0_and_1_by_rank"] == [[0, d["config"]["batch_per_gpu"]]]
