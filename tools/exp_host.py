#!/usr/bin/env python3
"""Developer experiment: host time per ofdg_forward call (how fast the host can fill the pipeline)."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
g = ofdg.Generator(ofdg.default_params(mode=5, batch_size=32, num_objects=16, sampler=1, seed=5))
g.pool_synthetic(1000, 1024, 768, 1)
outs = [ofdg.alloc_outputs(32, 384, 512) for _ in range(8)]
for i in range(16): g.forward(*outs[i % 8], g.next_stream())
g.synchronize()
for n in (4, 8, 20, 100):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): g.forward(*outs[i % 8], g.next_stream())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%3d calls: host issue %.1f us/call, until done %.1f us/call" % (n, (t1 - t) / n * 1e6, (t2 - t) / n * 1e6))
