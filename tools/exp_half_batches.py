#!/usr/bin/env python3
"""Developer experiment: a call's 32 samples as ONE batch on one chain against two half batches (or four quarters) on
neighbouring chains - the 20-call run from an empty pipeline and the long run, us per 32 samples.
Usage on the GPU box: python3 tools/exp_half_batches.py [reps]"""
import importlib, os, statistics, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
W, H = 512, 384
for B, chains in ((32, 4), (16, 4), (16, 6), (16, 8), (8, 8)):
    parts = 32 // B
    outs = [ofdg.alloc_outputs(B, H, W) for _ in range(16)]
    g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=5, num_objects=16, batch_size=B, sampler=1, seed=20261003, background_prep=1, chains=chains))
    g.pool_synthetic(1000, 1024, 768, 2024)
    k = 0
    def run(n):
        global k
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n * parts):
            g.forward(*outs[k % 16], g.next_stream()); k += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t) / n * 1e6, (time.perf_counter() - t) / n * 1e6
    run(50)
    short = [run(20) for _ in range(REPS)]
    long_ = [run(1000) for _ in range(2)]
    print("batches of %2d on %d chains: 20 x 32 samples %.1f us per 32 (min %.1f, host issue %.1f) | 1000 x 32: %.1f us (host issue %.1f)" % (
        B, chains, statistics.median(s[1] for s in short), min(s[1] for s in short), statistics.median(s[0] for s in short),
        min(l[1] for l in long_), min(l[0] for l in long_)), flush=True)
    g.synchronize(torch.cuda.current_stream().cuda_stream)
    del g
