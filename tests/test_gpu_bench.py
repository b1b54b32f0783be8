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
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "per_launch", "algorithmic_bytes_per_step", "ms_per_step",
                "traffic_whole_step", "dominant_kernel_by_gpu_time"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # every fraction of the block can be recomputed from the line alone: the whole step's first ...
    assert r["algorithmic_bytes_per_step"] == 38 * 512 * 384 * d["config"]["batch_per_gpu"] and r["ms_per_step"] == d["ms_per_step"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-9 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["frac"] == r["whole_step_frac"]
    assert abs(r["achieved"] - d["value"] / d["n_gpus"] * 38 * 512 * 384 / 1e9) < 1e-9 * r["achieved"]
    # ... then one launch of the compose kernel: live HIP events in the pipeline (an overlapped span) and with the device to itself
    q = r["per_launch"]
    assert q["kernel"] == r["kernel"] and q["algorithmic_bytes_per_launch"] == r["algorithmic_bytes_per_step"]
    assert abs(q["achieved"] - q["algorithmic_bytes_per_launch"] / (q["kernel_ms"] * 1e-3) / 1e9) < 1e-9 * q["achieved"]
    assert abs(q["frac"] - q["achieved"] / r["peak"]) < 1e-12
    assert abs(q["frac_alone"] - q["algorithmic_bytes_per_launch"] / (q["kernel_ms_alone"] * 1e-3) / 1e9 / r["peak"]) < 1e-12
    assert abs(q["launches_in_flight"] - q["kernel_ms"] / d["ms_per_step"]) < 1e-9
    if r["traffic_whole_step"]:  # (PMC passes of this configuration are committed: the step's bytes at the L2s' memory side, by kernel)
        t = r["traffic_whole_step"]
        assert t["hbm_bytes_per_step"] == sum(t["by_kernel"].values()) and t["by_kernel"][r["kernel"]] == r["traffic"]
    if r["dominant_kernel_by_gpu_time"]:
        k = r["dominant_kernel_by_gpu_time"]
        assert k["kernel"] in k["all"] and abs(sum(v["share"] for v in k["all"].values()) - 1.0) < 1e-6
    # every rank's own rate beside the aggregate
    assert len(d["samples_per_s_by_rank"]) == d["n_gpus"] and abs(sum(d["samples_per_s_by_rank"]) - d["value"]) < 1e-6 * d["value"]
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
    check_native_startup(d)


def check_native_startup(d):
    """The rank took bench.py's N > 1 start-up on a one-rank RCCL communicator (Comm.from_store, bcast_setup, params_of,
    pool_from_setup, agree, nccl_count) and ran on the context a receiving rank builds from the broadcast."""
    c = d["config"]
    assert c["rccl_ranks"] == 1, c
    assert c["startup"].startswith("ofdg_comm_bcast_setup: one ncclBroadcast") and "built from the broadcast" in c["startup"], c["startup"]
    assert "gloo" in c["plumbing"]
    assert c["shards"]["first_index_of_steps_0_and_1_by_rank"] == [[0, c["batch_per_gpu"]]]


@pytest.mark.gpu
@pytest.mark.parametrize("config", [4, 5])
def test_bench_native_startup_with_one_rank(config):
    """`--native-startup`: bench.py's own multi-rank branch at world = 1, for the two configurations BASELINE quotes on 8 GPUs, at
    their per-GPU batch (config 4: 8 samples of 1024 x 768; config 5: 32 samples on the 10 000 x 1 MP pool) - the 8-GPU run
    differs from this by the world size only (data_generation_layer.hpp:54 has no counterpart: the reference does not shard)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                          "--no-secondary", "--native-startup", "--config", str(config)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    check_native_startup(d)
    assert d["config"]["baseline_config"] == config and d["config"]["batch_per_gpu"] == (8 if config == 4 else 32)
    assert d["config"]["launched_by"] == "direct" and d["value"] > 0 and len(d["samples_per_s_by_rank"]) == 1


@pytest.mark.gpu
def test_bench_centre_crop_headline_says_so():
    d = bench("--background-prep", "0", "--no-secondary")
    check_contract(d)
    assert d["config"]["background_prep"] == 0 and "CENTRE-CROP" in d["config"]["workload"]
    assert "background_prep_kernel" not in d
