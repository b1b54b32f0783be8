"""The bench line's contract (the driver parses ONE JSON line from `python bench.py ...`): run the script as the driver
does - a child process, a few steps - and check the fields the contract names; the headline is the reference-equivalent
workload (background_prep = 1), the centre-crop form rides along.  The same through the script's own launcher path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", *extra],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_contract(d):
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
    # the context proves how it was set up: chains, hardware queues, shard indices of rank 0
    assert d["config"]["chains"] >= 3 and d["config"]["shards"]["first_index_of_steps_0_and_1_by_rank"] == [[0, d["config"]["batch_per_gpu"]]]


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    d = bench()
    check_contract(d)
    # the headline does the reference's per-sample background preparation; the lighter centre-crop form is the secondary
    assert d["config"]["background_prep"] == 1 and "getRandomizedCrop" in d["config"]["workload"]
    assert d["config"]["launched_by"] == "direct"
    e = d["centre_crop_backgrounds"]
    assert e["background_prep"] == 0 and e["unit"] == "samples/s" and e["value"] > d["value"] > 0
    assert "reference_equivalent" not in d
    # the step's other heavy kernel is reported beside the compose kernel's roofline: its launch in the pipeline and alone
    k = d["background_prep_kernel"]
    assert "neither hbm nor mfma" in k["bound"] and k["peak"] == 256 * 4 * 2.4 / 2  # (a wave64 vector instruction issues over 2 cycles of a 32-lane SIMD: MI355X_MICROARCH.md)
    assert k["kernel_ms"] > 0 and k["kernel_ms_alone"] > 0 and d["kernel_ms_alone"]["background_prep"] == k["kernel_ms_alone"]
    if k["valu_instructions_per_launch"]:   # (a PMC pass of this configuration is committed)
        assert abs(k["frac_alone"] - k["valu_instructions_per_launch"] / (k["kernel_ms_alone"] * 1e-3) / 1e9 / k["peak"]) < 1e-9


@pytest.mark.gpu
def test_bench_line_through_the_launcher_path_has_the_same_contract():
    """`--launcher`: the rank is a child of the script's own launcher (what `--gpus N > 1` does by itself); the relayed
    line obeys the same contract as the direct form."""
    d = bench("--launcher", "--no-secondary")
    check_contract(d)
    assert d["config"]["launched_by"] == "bench.py launcher" and d["config"]["background_prep"] == 1
    assert d["centre_crop_backgrounds"] is None


@pytest.mark.gpu
def test_bench_centre_crop_headline_says_so():
    d = bench("--background-prep", "0", "--no-secondary")
    check_contract(d)
    assert d["config"]["background_prep"] == 0 and "CENTRE-CROP" in d["config"]["workload"]
    assert "background_prep_kernel" not in d
