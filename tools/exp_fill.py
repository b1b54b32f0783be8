#!/usr/bin/env python3
"""Developer experiment: the driver's 20-step run (pipeline empty at the start of the timed region) against the long run,
by number of chains and look-ahead.  Usage on the GPU box: python3 tools/exp_fill.py [reps]"""
import importlib, os, statistics, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 15
W, H, B = 512, 384, 32
arms = [dict(chains=c, lookahead=l) for c in (3, 4, 6, 8) for l in (0,)] + [dict(chains=4, lookahead=l) for l in (1, 2, 4)]
if os.environ.get("ARMS"):
    arms = [dict(chains=int(a.split(":")[0]), lookahead=int(a.split(":")[1])) for a in os.environ["ARMS"].split(",")]
outs = [ofdg.alloc_outputs(B, H, W) for _ in range(16)]
st = torch.cuda.current_stream().cuda_stream
for arm in arms:
    g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=5, num_objects=16, batch_size=B, sampler=1, seed=20261003, background_prep=int(os.environ.get("BGPREP", "0")), **arm))
    g.pool_synthetic(1000, 1024, 768, 2024)
    nb = min(16, 2 * g.num_chains())
    k = 0
    def run(n):
        global k
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            g.forward(*outs[k % nb], g.next_stream()); k += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (t1 - t) / n * 1e6, (time.perf_counter() - t) / n * 1e6
    run(50)
    short = [run(20) for _ in range(REPS)]
    long_ = [run(1000) for _ in range(2)]
    print("chains %d lookahead %d: 20 calls %.1f us/step (min %.1f, host issue %.1f) | 1000 calls %.1f us/step (host issue %.1f)" % (
        arm["chains"], arm["lookahead"], statistics.median(s[1] for s in short), min(s[1] for s in short),
        statistics.median(s[0] for s in short), min(l[1] for l in long_), min(l[0] for l in long_)), flush=True)
    g.synchronize(st)
    del g
