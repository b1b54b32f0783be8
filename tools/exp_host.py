#!/usr/bin/env python3
"""Developer experiment: host time per call (how fast the host can fill the pipeline), config 2 (counter sampler, batch 32)
and config 1 (reference-stream sampler on the host inside the call, batch 1).  OFDG_LIB selects the library build."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("OFDG_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
for name, kw, B in (("config 2", dict(mode=5, batch_size=32, num_objects=16, sampler=1, seed=5), 32),
                    ("config 1", dict(mode=7, batch_size=1, num_objects=1, sampler=0), 1)):
    g = ofdg.Generator(ofdg.default_params(**kw))
    g.pool_synthetic(1000, 1024, 768, 1)
    outs = [ofdg.alloc_outputs(B, 384, 512) for _ in range(8)]
    for i in range(64): g.forward(*outs[i % 8], g.next_stream())
    g.synchronize()
    for n in (20, 2000):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n): g.forward(*outs[i % 8], g.next_stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s, %4d calls: host issue %.1f us/call, until done %.1f us/call" % (name, n, (t1 - t) / n * 1e6, (t2 - t) / n * 1e6))
    del g
