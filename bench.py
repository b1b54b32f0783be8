#!/usr/bin/env python3
"""Benchmark of the hot path: training samples/s (image0 + image1 + flow) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}]
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Default workload = BASELINE.json configs[1] (--config 2): FlyingChairs mode 5, 512x384, batch 32 per GPU,
16 objects, affine-only motion, AA on, synthetic 1000 x 1024x768 texture pool.  The other BASELINE
configurations print the same JSON line (see CONFIGS below).  A "step" is one pass of the whole hot path
over one batch of NEW samples:

  --sampler counter (default): motion/shape sampling + realize (cs_sample_realize kernel, Philox counter
      streams) -> geom -> raster -> compose, everything on the device; the only input resident in HBM is the
      texture pool (mode 9: + the warp crops).  Every step renders samples never rendered before (global
      indices step*B*world + rank*B + [0, B)).
  --sampler resident: the reference's 45 mt19937 streams are sampled on the host before the timed region;
      NSLOT realised batches are resident in HBM and rotated; a step is geom -> raster -> compose.

The calls are made the way a prefetch ring makes them: call k renders into output buffer set k mod NBUF on the
context's next internal stream (ofdg_stream); NBUF = 2 x the number of internal streams, so a buffer set is
always written by the same in-order stream and the calls in flight never share an output.  The timed region
ends with a device-wide synchronisation.

Samples shard across ranks with no data-path collective ("weak" scaling): rank r renders block r of every
B*world consecutive samples of the stream.  Start-up for N > 1: ONE native RCCL broadcast of rank 0's setup
header + texture index table (ofdg_comm_bcast_setup, csrc/comm.cpp); the JSON line carries the proof (`rccl_ranks` = what
ncclCommCount says, `shards` = every rank's first global index of steps 0 and 1).  If the native start-up fails the run
exits non-zero on every rank (--allow-fallback: the same header over torch.distributed instead, and the line says so).

`reference_equivalent` in the same line: the same workload with background_prep = 1 - Texture::getRandomizedCrop(2W, 2H,
rot, zoom, shift) on every sample's background, which the reference runs per sample (DataGenerator.cpp:1186-1192) and
the headline skips - on the GPU and in the CPU-baseline leg.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# launch configuration of the product (README): eight HIP hardware queues, so that the context's four in-order chains
# each own one (must be in the environment before the HIP runtime starts, i.e. before torch touches the GPU)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# BASELINE.json configs.  pool = (n, w, h); batch = samples per GPU and step.
CONFIGS = {
    1: dict(mode=7, W=512, H=384, batch=1, nobj=1, pool=(1000, 1024, 768), sampler="ref",
            name="FlyingChairs default mode (7), 512x384, batch=1, 1 object, fixed seeds 0..44 (BASELINE configs[0])"),
    2: dict(mode=5, W=512, H=384, batch=32, nobj=16, pool=(1000, 1024, 768), sampler="counter",
            name="FlyingChairs mode 5, 512x384, batch=32 per GPU, 16 objects, affine-only motion, AA on, "
                 "synthetic 1000x(1024x768) texture pool (BASELINE configs[1])"),
    3: dict(mode=9, W=512, H=384, batch=32, nobj=16, pool=(1000, 1024, 768), sampler="counter",
            name="ThinPlate/deformation mode 9, 512x384, batch=32 per GPU, 16 objects, warp fields generated on the "
                 "device (BASELINE configs[2])"),
    4: dict(mode=7, W=1024, H=768, batch=8, nobj=32, pool=(1000, 2048, 1536), sampler="counter",
            name="FlyingChairs mode 7, 1024x768, batch=8 per GPU (64 over 8 GPUs), 32 objects, mixed motion, "
                 "synthetic 1000x(2048x1536) texture pool (BASELINE configs[3])"),
    5: dict(mode=7, W=512, H=384, batch=32, nobj=0, pool=(10000, 1024, 1024), sampler="counter",
            name="large-pool stress: 10000 x 1 MP textures (42 GB BGRX) resident in HBM, mode 7, 512x384, batch=32 per "
                 "GPU (256 over 8 GPUs) (BASELINE configs[4])"),
}
POOL_SEED = 2024
NSLOT = 12
SEED = 20261003
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(ofdg, gen, cfg, budget_s=24.0, host_pool=128, background_prep=0):
    """The oracle (CPU restatement of the reference path) timed on this host's cores, on a bounded sample of the
    same workload: 1 thread and all cores (the reference's threading: one sample worker per core,
    first_level_threads = cores, second_level_threads = 1, DataGenerator.cpp:1023-1027), in the reference's work
    pattern ("faithful": 4 rasterisations + full-frame warps and blits per shape, DataGenerator.cpp:337-349,
    465-479) and in a "lean" form (one rasterisation per frame, work inside the outlines' boxes; same bytes)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as oracle
    W, H, mode, nobj = cfg["W"], cfg["H"], cfg["mode"], cfg["nobj"]
    cores = os.cpu_count() or 1
    n_pool = min(host_pool, cfg["pool"][0])
    sub = np.stack([gen.pool_download(i) for i in range(n_pool)])  # tex_id % n_pool
    prm = oracle.default_params(W, H, mode, 1, 1, nobj)
    prm.background_prep = background_prep
    crops = None
    if mode == 9:
        crops = np.stack([gen.warp_download(i) for i in range(min(gen.warp_count(), 8))])
    sampler = ofdg.HostSampler(mode, W, H, nobj)

    def run(n_threads, lean, budget):
        n = max(n_threads, 2)
        tasks, bps, n_bps = sampler.next(n, cap=n * 320)
        done, t0 = 0, time.perf_counter()
        while True:
            if lean:
                with oracle.lean():
                    oracle.render(prm, tasks, n, bps, n_bps, sub, warp_crops=crops, n_threads=n_threads)
            else:
                oracle.render(prm, tasks, n, bps, n_bps, sub, warp_crops=crops, n_threads=n_threads)
            done += n
            dt = time.perf_counter() - t0
            if dt >= budget or dt * (done + n) / done > 1.5 * budget:
                return done / dt, done, dt

    all_threads = 1 if mode == 9 else cores  # (the oracle serves warp crops in order: mode 9 is sequential)
    shares = {"faithful_all": 0.35, "lean_all": 0.25, "faithful_1": 0.25, "lean_1": 0.15}
    res = {}
    for key, share in shares.items():
        lean = key.startswith("lean")
        threads = all_threads if key.endswith("_all") else 1
        rate, done, dt = run(threads, lean, budget_s * share)
        res[key] = {"samples_per_s": rate, "threads": threads, "samples": done, "seconds": dt}
    total = sum(v["seconds"] for v in res.values())
    return {"value": res["faithful_all"]["samples_per_s"], "unit": "samples/s", "cores": res["faithful_all"]["threads"], "kind": "port",
            "threads_all": res["faithful_all"]["samples_per_s"], "threads_1": res["faithful_1"]["samples_per_s"],
            "lean": {"threads_all": res["lean_all"]["samples_per_s"], "threads_1": res["lean_1"]["samples_per_s"]},
            "host_logical_cpus": cores, "detail": res, "background_prep": background_prep,
            "sample": "oracle/ restatement on this workload (mode %d, %dx%d, %s objects), host pool = the first %d of the %d "
                      "textures; 'value' = the reference's work pattern (4 rasterisations + full-frame warps / blits per shape) on "
                      "%d worker threads; lean = one rasterisation per frame, work inside the outlines' boxes; %.0f s of CPU time in all"
                      % (mode, W, H, nobj or "16-23", n_pool, cfg["pool"][0], res["faithful_all"]["threads"], total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pool", type=int, default=128, help="textures of the pool the CPU baseline works on")
    ap.add_argument("--sampler", choices=("counter", "resident"), default=None)
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the native RCCL start-up fails, broadcast the header over torch.distributed instead of exiting")
    ap.add_argument("--no-reference-equivalent", action="store_true", help="skip the background_prep = 1 pass")
    ap.add_argument("--background-prep", action="store_true",
                    help="apply Texture::getRandomizedCrop(2W, 2H, rot, zoom, shift) to every background (DataGenerator.cpp:1186-1192)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.sampler:
        cfg["sampler"] = "counter" if args.sampler == "counter" else "resident"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)  # one process per GPU: the device is bound before any HIP work
    ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")

    counter = cfg["sampler"] == "counter"
    W, H, BATCH = cfg["W"], cfg["H"], cfg["batch"]
    prm = ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=BATCH,
                              rank=rank, world_size=world, device=local_rank, sampler=1 if counter else 0, seed=SEED,
                              background_prep=1 if args.background_prep else 0)
    startup = "single process"
    gen = None
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")  # (barriers, the gathers of the proof fields and the max over ranks of the timing)
        # the one collective of this path, native: rank 0's seed / stream / pool header + texture index table in ONE
        # ncclBroadcast on the library's own RCCL communicator (the unique id travels through the launcher's store).
        # Success or failure is decided by all ranks together: a root that cannot set itself up broadcasts a failure
        # status (bcast_abort) instead of leaving the others in the collective.
        err = None
        try:
            comm = ofdg.Comm.from_store(dist.distributed_c10d._get_default_store(), rank, world, local_rank)
            rccl_ranks = comm.nccl_count()
            if rank == 0:
                try:
                    gen = ofdg.Generator(prm)
                    gen.pool_synthetic(*cfg["pool"], POOL_SEED)
                except Exception:
                    comm.bcast_abort()
                    raise
            setup, table = comm.bcast_setup(gen)
            if rank != 0:
                prm = comm.params_of(setup)
                gen = ofdg.Generator(prm)
                gen.pool_from_setup(setup, table)
            comm.close()
            startup = "ofdg_comm_bcast_setup: one ncclBroadcast of the setup header + %d-entry texture index table" % setup.n_table
        except Exception as e:
            err = e
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # (every rank gets here: nobody waits in a broadcast the root never entered)
        if int(ok.item()) != 1:
            if not args.allow_fallback:
                sys.stderr.write("rank %d: native multi-GPU start-up failed%s\n" % (rank, ": %s" % err if err else " on another rank"))
                dist.destroy_process_group()
                raise SystemExit(3)
            header = torch.tensor([cfg["mode"], W, H, cfg["nobj"], *cfg["pool"], POOL_SEED, SEED], dtype=torch.int64, device="cuda")
            if rank != 0:
                header.zero_()
            dist.broadcast(header, src=0)
            mode, W, H, nobj, pn, pw, ph, pseed, seed = [int(v) for v in header.tolist()]
            prm = ofdg.default_params(width=W, height=H, mode=mode, num_objects=nobj, batch_size=BATCH, rank=rank, world_size=world,
                                      device=local_rank, sampler=1 if counter else 0, seed=seed,
                                      background_prep=1 if args.background_prep else 0)
            gen = ofdg.Generator(prm)
            gen.pool_synthetic(pn, pw, ph, pseed)
            rccl_ranks = None
            startup = "FALLBACK: torch.distributed broadcast (--allow-fallback; the native start-up failed: %s)" % str(err)[:200]
    else:
        gen = ofdg.Generator(prm)
        gen.pool_synthetic(*cfg["pool"], POOL_SEED)
    if cfg["mode"] == 9:
        gen.warp_generate(2, SEED)  # seeded displacer lists: every rank generates the same fields
    stream = torch.cuda.current_stream().cuda_stream
    NBUF = 2 * gen.num_chains()
    outs = [ofdg.alloc_outputs(BATCH, H, W) for _ in range(NBUF)]

    host_sampler_rate = None
    if counter:
        def step(i):
            gen.forward(*outs[i % NBUF], gen.next_stream())  # samples (step*world + rank)*B + [0, B) on the device, then renders
    elif cfg["sampler"] == "ref":
        def step(i):  # config 1: the reference-stream sampler on the host inside the step, like load_batch
            gen.forward(*outs[i % NBUF], gen.next_stream())
    else:
        # every rank walks the same reference stream and keeps its own block of each B*world tasks
        sampler = ofdg.HostSampler(cfg["mode"], W, H, cfg["nobj"])
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

    # every rank's first global sample index of steps 0 and 1 (the sharding rule the library applies, gathered)
    mine = torch.tensor([ofdg.shard_first_index(k, BATCH, world, rank) for k in (0, 1)], dtype=torch.int64, device="cuda")
    shards = [mine.clone() for _ in range(world)]
    if world > 1:
        import torch.distributed as dist
        dist.all_gather(shards, mine)
    shards = [[int(v) for v in t.tolist()] for t in shards]

    compose_ms = gen.kernel_ms("compose")
    parts = alone = cpu_base = None
    if rank == 0:
        # second short pass with all three kernels timed (not part of `value`)
        gen.set_profiling(2)
        for i in range(min(args.steps, 48)):
            step(i)
        gen.synchronize(stream)
        parts = {k: gen.kernel_ms(k) for k in ("geom", "raster", "compose")}
        # third short pass, one batch at a time (device idle between the steps): the kernels' durations with nothing
        # else in flight - what one launch of the compose kernel takes when it has the GPU to itself
        gen.set_profiling(2)
        for i in range(min(args.steps, 32)):
            step(i)
            gen.synchronize(stream)
        alone = {k: gen.kernel_ms(k) for k in ("geom", "raster", "compose")}
        gen.set_profiling(0)
        if world == 1 and not args.no_cpu_baseline:
            cpu_base = cpu_baseline(ofdg, gen, cfg, host_pool=args.cpu_pool)
    ctx_info, n_chains = gen.info(), gen.num_chains()
    # The headline context is closed before the reference-equivalent one is made: two contexts would own ten streams on the
    # process's eight hardware queues, and chains that share a queue run one after the other.
    gen.synchronize(stream)
    gen.close()
    del gen

    # the reference-equivalent pass: the same workload with Texture::getRandomizedCrop on every background
    ref_eq = None
    if not args.no_reference_equivalent and not args.background_prep and counter:
        prm2 = ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=BATCH, rank=rank,
                                   world_size=world, device=local_rank, sampler=1, seed=SEED, background_prep=1)
        gen2 = ofdg.Generator(prm2)
        gen2.pool_synthetic(*cfg["pool"], POOL_SEED)
        if cfg["mode"] == 9:
            gen2.warp_generate(2, SEED)
        steps2 = min(args.steps, 1000)
        for i in range(args.warmup):
            gen2.forward(*outs[i % NBUF], gen2.next_stream())
        gen2.synchronize(stream)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps2):
            gen2.forward(*outs[i % NBUF], gen2.next_stream())
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        barrier()
        gen2.synchronize(stream)
        t2 = torch.tensor([dt2], dtype=torch.float64, device="cuda")
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dt2 = float(t2.item())
        v2 = steps2 * BATCH * world / dt2
        ref_eq = {"value": v2, "unit": "samples/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3,
                  "whole_step_frac": v2 / world * 38 * W * H / 1e9 / HBM_PEAK_GBS, "background_prep": 1,
                  "note": "background_prep = 1: getRandomizedCrop(2W, 2H, rot, zoom, shift) per sample (DataGenerator.cpp:1186-1192), "
                          "the CImg chain stage by stage on the device; same algorithmic bytes as the headline (the prepared "
                          "textures are extra traffic)"}
        if world == 1 and not args.no_cpu_baseline:
            ref_eq["cpu_baseline"] = cpu_baseline(ofdg, gen2, cfg, budget_s=12.0, host_pool=args.cpu_pool, background_prep=1)
        gen2.close()
        del gen2

    if rank == 0:
        alg_bytes_per_sample = 38 * W * H  # 32 B/px written (8 fp32 planes) + 6 B/px background read
        samples = args.steps * BATCH * world
        value = samples / dt
        achieved = BATCH * alg_bytes_per_sample / (compose_ms * 1e-3) / 1e9
        kernel = ("compose_deform" if cfg["mode"] == 9 else "compose_rigid") + ("_pow2_kernel" if W & (W - 1) == 0 else "_kernel")
        # HBM bytes per launch of that kernel from the PMC passes committed for THIS configuration (tools/profile_round.sh:
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script); null when no pass of this config is committed
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get("config%d" % args.config)
            if ent and ent.get("kernel") == kernel and not args.background_prep:
                traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
        out = {
            "metric": "training samples/sec (img0+img1+flow, %dx%d)" % (W, H),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 blends / fp64 affines -> f32 planes", "data": "synthetic",
            "config": {"workload": cfg["name"], "baseline_config": args.config, "batch_per_gpu": BATCH,
                       "background_prep": bool(args.background_prep), "startup": startup, "rccl_ranks": rccl_ranks,
                       "shards": {"first_index_of_steps_0_and_1_by_rank": shards}, "context": ctx_info, "output_buffer_sets": NBUF,
                       "chains": n_chains, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "sampler": ("counter (Philox, on the device, inside the timed region; every step renders new samples)"
                                   if counter else "ref (host mt19937 streams) inside the timed region" if cfg["sampler"] == "ref" else
                                   "ref (host mt19937 streams) outside the timed region; %d resident batches rotated" % NSLOT)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_ms": compose_ms,
                         # the same launch with the device to itself (serialised pass after the timed region)
                         "kernel_ms_alone": alone["compose"],
                         "frac_alone": BATCH * alg_bytes_per_sample / (alone["compose"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": BATCH * alg_bytes_per_sample,
                         # the pipeline runs independent in-order chains: compose launches of neighbouring steps overlap each
                         # other (and the preparation kernels) on the device, so a launch's duration can exceed the step;
                         # launches in flight on average = kernel_ms / ms_per_step
                         "launches_in_flight": compose_ms / (dt / args.steps * 1e3),
                         "whole_step_frac": value / world * alg_bytes_per_sample / 1e9 / HBM_PEAK_GBS,
                         # the two halves of the algorithmic bytes by themselves (32 B/px written, 6 B/px read)
                         "whole_step_write_frac": value / world * 32 * W * H / 1e9 / HBM_PEAK_GBS,
                         "whole_step_read_frac": value / world * 6 * W * H / 1e9 / HBM_PEAK_GBS,
                         "note": "achieved = algorithmic bytes of one launch / its live HIP-event duration (launches overlap); "
                                 "whole_step_frac = algorithmic bytes per second of the whole pipeline / peak"},
            "kernel_ms": parts, "kernel_ms_alone": alone,
            "hbm_gbs_whole_step": value / world * alg_bytes_per_sample / 1e9,
        }
        out["reference_equivalent"] = ref_eq
        if host_sampler_rate is not None:
            out["host_ref_sampler_samples_per_s"] = host_sampler_rate
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
