#!/usr/bin/env python3
"""Benchmark of the hot path: training samples/s (image0 + image1 + flow) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}]
    N > 1, started by hand: the script launches its N ranks itself (one child process per GPU with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its environment; the parent never touches the GPU, relays rank 0's JSON
    line and exits non-zero if any rank does).  Started by `python -m torch.distributed.run --nproc-per-node N ...`
    (WORLD_SIZE already set) it is one of the ranks.

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
context's next internal stream (OFDG_STREAM_OWN); NBUF = 2 x the number of internal streams, so a buffer set is
always written by the same in-order stream and the calls in flight never share an output.  The timed region
ends with a device-wide synchronisation.

Samples shard across ranks with no data-path collective ("weak" scaling): rank r renders block r of every
B*world consecutive samples of the stream.  Start-up for N > 1: ONE native RCCL broadcast of rank 0's setup
header + texture index table (ofdg_comm_bcast_setup, csrc/comm.cpp); the JSON line carries the proof (`rccl_ranks` = what
ncclCommCount says, `shards` = every rank's first global index of steps 0 and 1).  If the native start-up fails the run
exits non-zero on every rank (--allow-fallback: the same header over torch.distributed instead, and the line says so).

The headline `value` (and the `cpu_baseline` next to it) is the REFERENCE-EQUIVALENT workload: background_prep = 1 -
Texture::getRandomizedCrop(2W, 2H, rot, zoom, shift) on every sample's background, which the reference runs per sample
(DataGenerator.cpp:1186-1192) - the CImg chain stage by stage on the device.  `centre_crop_backgrounds` in the same line is
the lighter form (background_prep = 0: every background is the centre crop of its pool image; the headline of rounds 1-3),
measured after the headline context is closed.  --background-prep 0 makes that form the headline instead (and says so in
`config`).
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
EXIT_PORT_TAKEN = 98                    # a rank's exit code: the rendezvous port was taken between the launcher's probe and rank 0's bind
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2     # G wave64 vector instructions/s: 256 CUs x 4 SIMDs (32 lanes wide) at 2.4 GHz, 2 cycles per wave64 instruction (MI355X_MICROARCH.md; ONE wave sustains one per 4 cycles)


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


def parse_args(argv=None):
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
    ap.add_argument("--background-prep", type=int, choices=(0, 1, 2), default=1,
                    help="the headline's background mode: 1 (default) = Texture::getRandomizedCrop(2W, 2H, rot, zoom, shift) on every "
                         "background like the reference (DataGenerator.cpp:1186-1192); 0 = centre crops; 2 = one resampling")
    ap.add_argument("--no-secondary", action="store_true", help="skip the pass with the other background mode (centre_crop_backgrounds)")
    ap.add_argument("--chains", type=int, default=0, help="ofdg_params.chains (0: the context's own choice; 1: every kernel of a step runs alone, one step after the other)")
    ap.add_argument("--launcher", action="store_true", help="start the rank(s) as child processes even for --gpus 1 (what --gpus N > 1 does by itself)")
    ap.add_argument("--native-startup", action="store_true",
                    help="take the N > 1 start-up (RCCL communicator, ONE ncclBroadcast of the setup, receivers' context + pool from it, "
                         "agreement) also with one rank - implied when the rank is a child of this script's launcher; the rank then runs "
                         "on the context a RECEIVER builds from the broadcast")
    ap.add_argument("--launch-only", action="store_true",
                    help="launcher check: every rank prints its rank environment as one JSON line and exits (no torch, no GPU)")
    return ap.parse_args(argv)


# ---- the launcher: `python3 bench.py --gpus N` starts its own ranks ----------------------------------------------------
def launch(args, argv):
    """Parent of an N-rank run.  It must not touch the GPU (no torch.cuda, no libofdg, no HIP call): the children are
    started before anything in this process could initialise it, one per GPU, with the rendezvous in their environment;
    rank 0's stdout (the ONE JSON line) is relayed, everybody's stderr passes through, and the exit code is non-zero if
    any rank's is.  A rank that dies takes the others down (they would wait for it in a collective)."""
    import signal
    import socket
    import subprocess
    import threading
    n = args.gpus
    child_argv = [a for a in argv if a != "--launcher"]
    for attempt in range(3):
        # The rendezvous port: bound and released here, bound again by rank 0's store.  Somebody else can take it in
        # between; rank 0 then exits with EXIT_PORT_TAKEN and the whole start is repeated on another port.
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        procs, outs, readers = [], [[] for _ in range(n)], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OFDG_BENCH_LAUNCHED="1", OFDG_BENCH_ATTEMPT=str(attempt))
            env.setdefault("GLOO_SOCKET_IFNAME", "lo")       # the timing barrier runs over gloo on the loopback interface
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this driver)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 or args.launch_only else subprocess.DEVNULL, text=True))
            if procs[r].stdout:  # drained while the child runs: a rank that prints more than a pipe holds must not block on it
                t = threading.Thread(target=lambda f=procs[r].stdout, o=outs[r]: o.append(f.read()), daemon=True)
                t.start()
                readers.append(t)
        rc = 0
        alive = set(range(n))
        deadline = None
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py launcher: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                    deadline = time.time() + 20.0  # (they normally fail by themselves: the start-up is decided collectively)
            if rc != 0 and alive and time.time() > deadline:
                for r in alive:
                    procs[r].send_signal(signal.SIGTERM)  # exactly the PIDs started above
                time.sleep(2.0)
                for r in alive:
                    if procs[r].poll() is None:
                        procs[r].kill()
                deadline = time.time() + 60.0
            time.sleep(0.02)
        for t in readers:
            t.join()
        if rc != EXIT_PORT_TAKEN:
            break
        sys.stderr.write("bench.py launcher: port %d was taken before rank 0 could bind it; starting again on another one\n" % port)
    outs = ["".join(o) for o in outs]
    if args.launch_only:
        ranks = []
        for o in outs:
            ranks += [json.loads(l) for l in o.splitlines() if l.strip().startswith("{")]
        print(json.dumps({"launcher": "bench.py", "n_gpus": n, "master_port": port, "ranks": sorted(ranks, key=lambda d: d["rank"])}))
    else:
        sys.stdout.write(outs[0])
    sys.stdout.flush()
    return rc


class Plumbing:
    """What bench.py needs from torch.distributed around the native path: a barrier, the max over ranks of the timing and
    the gather of the proof fields - on CPU tensors (gloo over the loopback interface), so that the only RCCL communicator
    of the process is the library's own (csrc/comm.cpp).  If the gloo side cannot be used the same calls run on device
    tensors (torch's NCCL = RCCL) instead; `self.how` says which."""

    def __init__(self, world, group=False):
        self.world = world
        self.how = "single process"
        self.grouped = world > 1 or group
        if self.grouped:
            import torch.distributed as dist
            if "MASTER_PORT" not in os.environ:  # (one rank started by hand with --native-startup: its own rendezvous)
                import socket
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(so.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                dist.init_process_group()  # default backends: gloo for CPU tensors, nccl (RCCL) for device tensors, the latter made on first use
            except Exception as e:  # noqa: BLE001
                if "OFDG_BENCH_LAUNCHED" in os.environ and any(k in str(e).lower() for k in ("address already in use", "eaddrinuse")):
                    raise SystemExit(EXIT_PORT_TAKEN)  # (bench.py's own launcher picks another port and starts again)
                raise
            self.dist = dist
            self.device = "cpu"
            try:
                self._reduce([1.0], "max")
                self.how = "torch.distributed over gloo (CPU tensors)"
            except Exception as e:  # noqa: BLE001
                self.device = "cuda"
                self.how = "torch.distributed over nccl (gloo unusable: %s)" % str(e)[:120]

    def _reduce(self, values, op):
        import torch
        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN)
        return [float(v) for v in t.tolist()]

    def reduce(self, value, op="max"):
        return value if not self.grouped else self._reduce([value], op)[0]

    def barrier(self):
        import torch
        if self.grouped:
            self._reduce([0.0], "max")
        if torch.cuda.is_available():  # (the CPU test of this class has no device)
            torch.cuda.synchronize()

    def gather_ints(self, mine):
        import torch
        if not self.grouped:
            return [list(mine)]
        t = torch.tensor(mine, dtype=torch.int64, device=self.device)
        out = [t.clone() for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[int(v) for v in o.tolist()] for o in out]

    def gather_floats(self, mine):
        import torch
        if not self.grouped:
            return [list(mine)]
        t = torch.tensor(mine, dtype=torch.float64, device=self.device)
        out = [t.clone() for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]

    def store(self):
        return self.dist.distributed_c10d._get_default_store()

    def close(self):
        if self.grouped:
            self._reduce([0.0], "max")
            self.dist.destroy_process_group()


def make_generator(ofdg, cfg, prm, pl, rank, world, local_rank, allow_fallback, background_prep, native=False):
    """The context + texture pool of this rank.  world > 1: the one collective of this path, native - rank 0's seed /
    stream / pool header + texture index table in ONE ncclBroadcast on the library's own RCCL communicator (the unique id
    travels through the launcher's store).  Success or failure is decided by all ranks together: a root that cannot set
    itself up broadcasts a failure status (bcast_abort) instead of leaving the others in the collective, and after the
    broadcast the ranks agree (ofdg_comm_agree) that every one of them built its context and pool.
    native (--native-startup) with ONE rank: the same path on a one-rank communicator, and the rank then does what a receiver
    does - context from the broadcast header (params_of), pool from the header + table (pool_from_setup) - and runs on THAT
    context (the root's own is closed): an 8-GPU run differs from this by the world size only."""
    if world == 1 and not native:
        gen = ofdg.Generator(prm)
        gen.pool_synthetic(*cfg["pool"], POOL_SEED)
        return gen, prm, "single process", None
    import torch
    err, gen, rccl_ranks, startup, comm = None, None, None, None, None
    try:
        comm = ofdg.Comm.from_store(pl.store(), rank, world, local_rank)
        if rank == 0:
            try:  # EVERYTHING the root does before the broadcast: whatever fails, the receivers are told
                rccl_ranks = comm.nccl_count()
                gen = ofdg.Generator(prm)
                gen.pool_synthetic(*cfg["pool"], POOL_SEED)
            except Exception:
                comm.bcast_abort()
                raise
        setup, table = comm.bcast_setup(gen)
        local = None
        try:  # alone again: the others must not be left in the next collective if this fails
            if rank != 0 or world == 1:
                if world == 1:  # the root as its own receiver: its context goes, the one built from the broadcast stays
                    gen.close()
                    gen = None
                rccl_ranks = comm.nccl_count()
                prm = comm.params_of(setup)
                gen = ofdg.Generator(prm)
                gen.pool_from_setup(setup, table)
        except Exception as e:  # noqa: BLE001
            local = e
        comm.agree(local is None)
        if local is not None:
            raise local
        startup = "ofdg_comm_bcast_setup: one ncclBroadcast of the setup header + %d-entry texture index table" % setup.n_table
        if world == 1:
            startup += "; one rank (--native-startup): the context and pool in use were built from the broadcast, as a receiving rank builds them"
    except Exception as e:  # noqa: BLE001
        err = e
    finally:
        if comm is not None:
            comm.close()
    ok = pl.reduce(0.0 if err else 1.0, "min")  # (every rank gets here: nobody waits in a broadcast the root never entered)
    if ok != 1.0:
        if not allow_fallback:
            sys.stderr.write("rank %d: native multi-GPU start-up failed%s\n" % (rank, ": %s" % err if err else " on another rank"))
            pl.close()
            raise SystemExit(3)
        import torch.distributed as dist
        W, H = cfg["W"], cfg["H"]
        header = torch.tensor([cfg["mode"], W, H, cfg["nobj"], *cfg["pool"], POOL_SEED, SEED], dtype=torch.int64, device=pl.device)
        if rank != 0:
            header.zero_()
        dist.broadcast(header, src=0)
        mode, W, H, nobj, pn, pw, ph, pseed, seed = [int(v) for v in header.tolist()]
        prm = ofdg.default_params(width=W, height=H, mode=mode, num_objects=nobj, batch_size=cfg["batch"], rank=rank, world_size=world,
                                  device=local_rank, sampler=prm.sampler, seed=seed, background_prep=background_prep)
        gen = ofdg.Generator(prm)
        gen.pool_synthetic(pn, pw, ph, pseed)
        rccl_ranks = None
        startup = "FALLBACK: torch.distributed broadcast (--allow-fallback; the native start-up failed: %s)" % str(err)[:200]
    return gen, prm, startup, rccl_ranks


def timed_pass(ofdg, gen, cfg, pl, outs, steps, warmup, rank, world, stream):
    """W untimed warm-up steps, then exactly K steps between barrier + device-wide synchronisation on both sides; the max
    over ranks of the time.  Returns (seconds, step function, host sampler rate)."""
    import torch
    BATCH, NBUF = cfg["batch"], len(outs)
    host_sampler_rate = None
    outs = [ofdg.device_pointers(o) for o in outs]  # (the buffer sets' device addresses, resolved once)
    if cfg["sampler"] in ("counter", "ref"):
        def step(i):  # counter: samples (step*world + rank)*B + [0, B) on the device, then renders; ref (config 1): the
            gen.forward(*outs[i % NBUF], ofdg.STREAM_OWN)  # reference-stream sampler on the host inside the step, like load_batch
    else:
        # every rank walks the same reference stream and keeps its own block of each B*world tasks
        sampler = ofdg.HostSampler(cfg["mode"], cfg["W"], cfg["H"], cfg["nobj"])
        t_s = time.perf_counter()
        for slot in range(NSLOT):
            tasks, bps, n_bps = sampler.next(BATCH * world, cap=BATCH * world * 64)
            mine = (ofdg.Task * BATCH)(*[tasks[rank * BATCH + i] for i in range(BATCH)])
            gen.upload_slot(slot, mine, BATCH, bps, n_bps, stream)
        host_sampler_rate = NSLOT * BATCH * world / (time.perf_counter() - t_s)

        def step(i):
            gen.render_slot(i % NSLOT, *outs[i % NBUF], ofdg.STREAM_OWN)
    gen.synchronize(stream)
    for i in range(warmup):
        step(i)
    gen.synchronize(stream)
    gen.set_profiling(1)  # HIP events around the compose kernel, on the launch stream
    pl.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pl.barrier()
    gen.synchronize(stream)  # raises if a kernel flagged a capacity error
    return pl.reduce(dt, "max"), step, host_sampler_rate, dt


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ
    if args.launch_only and launched:
        if os.environ.get("OFDG_BENCH_TEST_FAIL_RANK") == os.environ["RANK"]:  # (the launcher's test: a rank that dies)
            return 5
        if os.environ.get("OFDG_BENCH_TEST_PORT_TAKEN") and os.environ["RANK"] == "0" and os.environ.get("OFDG_BENCH_ATTEMPT") == "0":
            return EXIT_PORT_TAKEN  # (the launcher's test: rank 0 could not bind the rendezvous port at the first attempt)
        sys.stdout.write("#" * int(os.environ.get("OFDG_BENCH_TEST_PAD", "0")) + "\n")  # (the launcher's test: more output than a pipe holds)
        print(json.dumps({k.lower(): (int(os.environ[k]) if os.environ[k].isdigit() else os.environ[k])
                          for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return 0
    if not launched and (args.gpus > 1 or args.launcher or args.launch_only):
        return launch(args, argv)
    cfg = dict(CONFIGS[args.config])
    if args.sampler:
        cfg["sampler"] = "counter" if args.sampler == "counter" else "resident"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python3 bench.py --gpus N` by itself, or N ranks with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)  # one process per GPU: the device is bound before any HIP work
    ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")

    counter = cfg["sampler"] == "counter"
    W, H, BATCH = cfg["W"], cfg["H"], cfg["batch"]
    bgp = args.background_prep

    def params(background_prep):
        return ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=BATCH, rank=rank,
                                   world_size=world, device=local_rank, sampler=1 if counter else 0, seed=SEED,
                                   background_prep=background_prep, chains=args.chains)

    native = args.native_startup or bool(os.environ.get("OFDG_BENCH_LAUNCHED"))
    pl = Plumbing(world, group=native)
    gen, prm, startup, rccl_ranks = make_generator(ofdg, cfg, params(bgp), pl, rank, world, local_rank, args.allow_fallback, bgp, native=native)
    if cfg["mode"] == 9:
        gen.warp_generate(2, SEED)  # seeded displacer lists: every rank generates the same fields
    stream = torch.cuda.current_stream().cuda_stream
    NBUF = 2 * gen.num_chains()
    outs = [ofdg.alloc_outputs(BATCH, H, W) for _ in range(NBUF)]

    dt, step, host_sampler_rate, dt_mine = timed_pass(ofdg, gen, cfg, pl, outs, args.steps, args.warmup, rank, world, stream)
    # every rank's own rate beside the aggregate (which divides by the SLOWEST rank's time): a straggler shows
    rank_rates = [args.steps * BATCH / t[0] for t in pl.gather_floats([dt_mine])]

    # every rank's first global sample index of steps 0 and 1 (the sharding rule the library applies, gathered)
    shards = pl.gather_ints([ofdg.shard_first_index(k, BATCH, world, rank) for k in (0, 1)])

    compose_ms = gen.kernel_ms("compose")
    parts = alone = cpu_base = None
    # the background preparation's launch is timed where it runs behind raster (batches the library prepares itself)
    names = ("geom", "raster", "compose") + (("background_prep",) if bgp and counter else ())
    prep_ms = gen.kernel_ms("background_prep") if "background_prep" in names else None
    if rank == 0:
        # second short pass with all three kernels timed (not part of `value`)
        gen.set_profiling(2)
        for i in range(min(args.steps, 48)):
            step(i)
        gen.synchronize(stream)
        parts = {k: gen.kernel_ms(k) for k in names}
        # third short pass, one batch at a time (device idle between the steps): the kernels' durations with nothing
        # else in flight - what one launch of the compose kernel takes when it has the GPU to itself
        gen.set_profiling(2)
        for i in range(min(args.steps, 32)):
            step(i)
            gen.synchronize(stream)
        alone = {k: gen.kernel_ms(k) for k in names}
        gen.set_profiling(0)
        if world == 1 and not args.no_cpu_baseline:
            cpu_base = cpu_baseline(ofdg, gen, cfg, host_pool=args.cpu_pool, background_prep=1 if bgp else 0)
    ctx_info, n_chains = gen.info(), gen.num_chains()
    # The headline context is closed before the secondary one is made: two contexts would own ten streams on the
    # process's eight hardware queues, and chains that share a queue run one after the other.
    gen.synchronize(stream)
    gen.close()
    del gen

    # the secondary pass: the same workload with the other background mode (headline 1 or 2 -> centre crops; 0 -> mode 1)
    secondary = None
    bgp2 = 0 if bgp else 1
    if not args.no_secondary and counter:
        gen2 = ofdg.Generator(params(bgp2))
        gen2.pool_synthetic(*cfg["pool"], POOL_SEED)
        if cfg["mode"] == 9:
            gen2.warp_generate(2, SEED)
        steps2 = min(args.steps, 1000)
        dt2, _, _, _ = timed_pass(ofdg, gen2, cfg, pl, outs, steps2, args.warmup, rank, world, stream)
        v2 = steps2 * BATCH * world / dt2
        secondary = {"value": v2, "unit": "samples/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3,
                     "whole_step_frac": v2 / world * 38 * W * H / 1e9 / HBM_PEAK_GBS, "background_prep": bgp2,
                     "note": ("background_prep = 0: every background is the centre 2W x 2H crop of its pool image (no rotation / zoom / shift "
                              "of the texture: less work than the reference does per sample; the headline of rounds 1-3)") if bgp2 == 0 else
                             ("background_prep = 1: getRandomizedCrop(2W, 2H, rot, zoom, shift) per sample (DataGenerator.cpp:1186-1192), "
                              "the CImg chain stage by stage on the device")}
        if bgp2 == 1 and world == 1 and not args.no_cpu_baseline:
            secondary["cpu_baseline"] = cpu_baseline(ofdg, gen2, cfg, budget_s=12.0, host_pool=args.cpu_pool, background_prep=1)
        gen2.close()
        del gen2

    if rank == 0:
        alg_bytes_per_sample = 38 * W * H  # 32 B/px written (8 fp32 planes) + 6 B/px background read
        samples = args.steps * BATCH * world
        value = samples / dt
        kernel = ("compose_deform" if cfg["mode"] == 9 else "compose_rigid") + ("_pow2_kernel" if W & (W - 1) == 0 else "_kernel")
        # From the committed profiles of THIS configuration (tools/profile_round.sh -> profiles/traffic.json): HBM-side bytes per
        # launch of the step's kernels (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script) and the kernel
        # that holds the largest share of GPU time in the rocprofv3 --kernel-trace --stats run; null when no pass of this
        # config is committed.
        traffic, traffic_src, prep_pmc, traffic_step, dominant = None, None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get("config%d_background_prep_%d" % (args.config, bgp))
            if ent and ent.get("kernel") == kernel and ent.get("background_prep", 0) == bgp:
                traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
                prep_pmc = ent.get("background_prep_kernel")
                traffic_step = ent.get("whole_step")
                dominant = ent.get("dominant_kernel_by_gpu_time")
        ms_per_step = dt / args.steps * 1e3
        alg_step = BATCH * alg_bytes_per_sample            # one step = one compose launch per rank = the batch's algorithmic bytes
        step_gbs = value / world * alg_bytes_per_sample / 1e9  # per GPU
        per_launch = BATCH * alg_bytes_per_sample / (compose_ms * 1e-3) / 1e9
        out = {
            "metric": "training samples/sec (img0+img1+flow, %dx%d)" % (W, H),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 blends / fp64 affines -> f32 planes", "data": "synthetic",
            "samples_per_s_by_rank": rank_rates,
            "config": {"workload": cfg["name"] + (
                           "; backgrounds prepared per sample like the reference: getRandomizedCrop(2W, 2H, rot, zoom, shift) (background_prep = 1)" if bgp == 1 else
                           "; CENTRE-CROP backgrounds (background_prep = 0: lighter than the reference's per-sample getRandomizedCrop)" if bgp == 0 else
                           "; backgrounds prepared by one resampling (background_prep = 2)"),
                       "baseline_config": args.config, "batch_per_gpu": BATCH,
                       "background_prep": bgp, "startup": startup, "rccl_ranks": rccl_ranks, "plumbing": pl.how,
                       "launched_by": "bench.py launcher" if os.environ.get("OFDG_BENCH_LAUNCHED") else ("torch.distributed.run / caller" if launched else "direct"),
                       "shards": {"first_index_of_steps_0_and_1_by_rank": shards}, "context": ctx_info, "output_buffer_sets": NBUF,
                       "chains": n_chains, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "sampler": ("counter (Philox, on the device, inside the timed region; every step renders new samples)"
                                   if counter else "ref (host mt19937 streams) inside the timed region" if cfg["sampler"] == "ref" else
                                   "ref (host mt19937 streams) outside the timed region; %d resident batches rotated" % NSLOT)},
            # The roofline of the WHOLE STEP: exactly one compose launch moves a step's algorithmic bytes, and the launches of
            # neighbouring steps overlap each other and the other kernels, so the sustained rate of those bytes is
            # algorithmic bytes per step / ms_per_step - recomputable from this line alone.  The figures of ONE launch of the
            # compose kernel (live HIP events on its stream: a span that contains the overlap, so it can exceed ms_per_step) and
            # of that launch with the device to itself are under `per_launch`.
            "roofline": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": alg_step, "ms_per_step": ms_per_step,
                         "whole_step_frac": step_gbs / HBM_PEAK_GBS,
                         # the two halves of the algorithmic bytes by themselves (32 B/px written, 6 B/px read)
                         "whole_step_write_frac": value / world * 32 * W * H / 1e9 / HBM_PEAK_GBS,
                         "whole_step_read_frac": value / world * 6 * W * H / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_whole_step": traffic_step, "dominant_kernel_by_gpu_time": dominant,
                         "kernel": kernel,
                         "per_launch": {"kernel": kernel, "algorithmic_bytes_per_launch": alg_step,
                                        "kernel_ms": compose_ms, "achieved": per_launch, "frac": per_launch / HBM_PEAK_GBS,
                                        # the same launch with the device to itself (serialised pass after the timed region)
                                        "kernel_ms_alone": alone["compose"],
                                        "achieved_alone": alg_step / (alone["compose"] * 1e-3) / 1e9,
                                        "frac_alone": alg_step / (alone["compose"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        # launches in flight on average = kernel_ms / ms_per_step
                                        "launches_in_flight": compose_ms / ms_per_step,
                                        "traffic": traffic},
                         "note": "achieved = algorithmic bytes of a step (38 B/px x W x H x batch) / ms_per_step, per GPU; frac = achieved / peak; "
                                 "per_launch: the compose launch by live HIP events (overlapped span) and with the device to itself"},
            "kernel_ms": parts, "kernel_ms_alone": alone,
            "hbm_gbs_whole_step": step_gbs,
        }
        if prep_ms is not None:
            # The step's other heavy kernel is bound by neither HBM nor MFMA (the instruction issue of its own waves - 6 300 instructions per
            # 64 x 32 tile - and what the co-running kernels leave of the wave slots); reported: its launch, and its vector instructions against
            # the chip's issue rate - 256 CUs x 4 SIMDs, one wave64 vector instruction per 2 cycles, 2.4 GHz - as a utilisation.
            # Instruction counts per launch come from the committed PMC pass of this configuration (null without one).
            valu = prep_pmc.get("valu_instructions_per_launch") if prep_pmc else None
            out["background_prep_kernel"] = {
                "kernel": (prep_pmc or {}).get("kernel", "bgprep_stream_kernel"), "bound": "instruction issue of its own waves: 73 % of a wave's life is issue, 27 % s_waitcnt (neither hbm nor mfma; DESIGN.md section 4)",
                "kernel_ms": prep_ms, "kernel_ms_alone": alone["background_prep"],
                "valu_instructions_per_launch": valu, "peak": VALU_PEAK_GINST, "unit": "G wave instructions/s",
                "achieved": valu / (prep_ms * 1e-3) / 1e9 if valu else None,
                "achieved_alone": valu / (alone["background_prep"] * 1e-3) / 1e9 if valu else None,
                "frac_alone": valu / (alone["background_prep"] * 1e-3) / 1e9 / VALU_PEAK_GINST if valu else None,
                "hbm_bytes_per_launch": (prep_pmc or {}).get("hbm_bytes_per_launch"),
                "source": (prep_pmc or {}).get("source"),
                "note": "runs at issue priority 0 beside the other chains' kernels (priority 3): in the pipeline its launch stretches over their gaps"}
        out["centre_crop_backgrounds" if bgp2 == 0 else "reference_equivalent"] = secondary
        if host_sampler_rate is not None:
            out["host_ref_sampler_samples_per_s"] = host_sampler_rate
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    pl.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
