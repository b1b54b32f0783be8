#!/usr/bin/env python3
"""Benchmark of the hot path: training samples/s (image0 + image1 + flow) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): FlyingChairs mode 5, 512x384, batch 32 per GPU,
16 objects, affine-only motion, AA on, synthetic 1000 x 1024x768 texture pool.
A "step" is one pass of the whole hot path over one batch of 32 NEW samples:

  --sampler counter (default): motion/shape sampling + realize (cs_sample_realize kernel,
      Philox counter streams) -> geom -> raster -> compose, everything on the device; the
      only input resident in HBM is the texture pool.  Every step renders samples never
      rendered before (global indices step*B*world + rank*B + [0, B)).
  --sampler resident: the reference's 45 mt19937 streams are sampled on the host before the
      timed region; NSLOT realised batches are resident in HBM and rotated (their
      backgrounds together exceed the 256 MiB Infinity Cache); a step is geom -> raster ->
      compose.

The calls are made the way a prefetch ring makes them: call k renders into output buffer set k mod 4 on the
context's next internal stream (ofdg_stream), so the kernels of neighbouring steps overlap on the device; the
timed region ends with a device-wide synchronisation.

Samples shard across ranks with no data-path collective ("weak" scaling): rank r renders
block r of every B*world consecutive samples of the stream.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, MODE, BATCH, NOBJ = 512, 384, 5, 32, 16
POOL_N, POOL_W, POOL_H, POOL_SEED = 1000, 1024, 768, 2024
NSLOT = 12
NBUF = 4    # output buffer sets the bench cycles (one per call in flight)
SEED = 20261003
ALG_BYTES_PER_SAMPLE = 38 * W * H       # 32 B/px written (8 fp32 planes) + 6 B/px background read
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(ofdg, gen, budget_s=15.0):
    """The oracle (CPU restatement of the reference path, per-object full-frame work like
    the reference) timed on this host's cores with the reference's threading: one sample
    worker per core (first_level_threads = cores, second_level_threads = 1,
    DataGenerator.cpp:1023-1027).  A bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as oracle
    cores = os.cpu_count() or 1
    n = max(cores, 8)  # one task per worker thread and round
    tasks, bps, n_bps = ofdg.HostSampler(MODE, W, H, NOBJ).next(n, cap=n * 64)
    sub = np.stack([gen.pool_download(i) for i in range(4)])  # tex_id % 4: same work, small host pool
    prm = oracle.default_params(W, H, MODE, 1, 1, NOBJ)
    t0 = time.perf_counter()
    oracle.render(prm, tasks, n, bps, n_bps, sub, n_threads=cores)
    first = time.perf_counter() - t0
    done = n
    reps = int(max(0, min((budget_s - first) / max(first, 1e-6), 64)))
    for _ in range(reps):
        oracle.render(prm, tasks, n, bps, n_bps, sub, n_threads=cores)
        done += n
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d samples of this workload (mode %d, %dx%d, %d objects; 4-texture host pool subset) in %.1f s; "
                      "oracle/ restatement with the reference's per-object full-frame work, %d worker threads" %
                      (done, MODE, W, H, NOBJ, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sampler", choices=("counter", "resident"), default="counter")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")

    header = torch.tensor([MODE, W, H, NOBJ, POOL_N, POOL_W, POOL_H, POOL_SEED, SEED], dtype=torch.int64, device="cuda")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
        # the one collective of this path: rank 0's seed + stream/pool description, over RCCL
        if rank != 0:
            header.zero_()
        dist.broadcast(header, src=0)
    mode, w, h, nobj, pool_n, pool_w, pool_h, pool_seed, seed = [int(v) for v in header.tolist()]

    counter = args.sampler == "counter"
    prm = ofdg.default_params(width=w, height=h, mode=mode, num_objects=nobj, batch_size=BATCH,
                              rank=rank, world_size=world, device=local_rank, sampler=1 if counter else 0, seed=seed)
    gen = ofdg.Generator(prm)
    gen.pool_synthetic(pool_n, pool_w, pool_h, pool_seed)
    stream = torch.cuda.current_stream().cuda_stream
    # a prefetch ring of NBUF output buffer sets (data_param.prefetch): every call renders into the next set on the
    # context's next internal stream (ofdg_stream), so the calls in flight overlap and never share an output
    outs = [ofdg.alloc_outputs(BATCH, h, w) for _ in range(NBUF)]

    host_sampler_rate = None
    if counter:
        def step(i):
            gen.forward(*outs[i % NBUF], gen.next_stream())  # samples (step*world + rank)*B + [0, B) on the device, then renders
    else:
        # every rank walks the same reference stream and keeps its own block of each B*world tasks
        sampler = ofdg.HostSampler(mode, w, h, nobj)
        t_s = time.perf_counter()
        for slot in range(NSLOT):
            tasks, bps, n_bps = sampler.next(BATCH * world, cap=BATCH * world * 64)
            mine = (ofdg.Task * BATCH)(*[tasks[rank * BATCH + i] for i in range(BATCH)])
            gen.upload_slot(slot, mine, BATCH, bps, n_bps, stream)
        host_sampler_rate = NSLOT * BATCH * world / (time.perf_counter() - t_s)

        def step(i):
            gen.render_slot(i % NSLOT, *outs[i % NBUF], gen.next_stream())
    gen.synchronize(stream)

    for i in range(args.warmup):
        step(i)
    gen.synchronize(stream)
    gen.set_profiling(1)  # HIP events around the compose kernel, on the launch stream

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    gen.synchronize(stream)  # raises if a kernel flagged a capacity error
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    compose_ms = gen.kernel_ms("compose")
    if rank == 0:
        # second short pass with all three kernels timed (not part of `value`)
        gen.set_profiling(2)
        for i in range(min(args.steps, 48)):
            step(i)
        gen.synchronize(stream)
        parts = {k: gen.kernel_ms(k) for k in ("geom", "raster", "compose")}
        gen.set_profiling(0)
        samples = args.steps * BATCH * world
        value = samples / dt
        achieved = BATCH * ALG_BYTES_PER_SAMPLE / (compose_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per compose launch from the committed PMC passes (profiles/traffic.json)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("compose_kernel_hbm_bytes_per_launch")
        out = {
            "metric": "training samples/sec (img0+img1+flow, 512x384)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 blends / fp64 affines -> f32 planes", "data": "synthetic",
            "config": {"workload": "FlyingChairs mode 5, 512x384, batch=32 per GPU, 16 objects, affine-only motion, "
                                   "AA on, synthetic 1000x(1024x768) texture pool (BASELINE configs[1])",
                       "batch_per_gpu": BATCH,
                       "sampler": ("counter (Philox, on the device, inside the timed region; every step renders new samples)"
                                   if counter else
                                   "ref (host mt19937 streams) outside the timed region; %d resident batches rotated" % NSLOT)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "compose_pow2_kernel", "kernel_ms": compose_ms,
                         "algorithmic_bytes_per_launch": BATCH * ALG_BYTES_PER_SAMPLE,
                         # the pipeline runs three in-order chains: compose launches of neighbouring steps overlap each
                         # other (and the preparation kernels) on the device, so a launch's duration is longer than the
                         # step; launches in flight on average = kernel_ms / ms_per_step
                         "launches_in_flight": compose_ms / (dt / args.steps * 1e3),
                         "note": "per-launch duration of overlapping launches; whole pipeline: hbm_gbs_whole_step"},
            "kernel_ms": parts,
            "hbm_gbs_whole_step": value / world * ALG_BYTES_PER_SAMPLE / 1e9,
        }
        if host_sampler_rate is not None:
            out["host_ref_sampler_samples_per_s"] = host_sampler_rate
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ofdg, gen)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
