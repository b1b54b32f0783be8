#!/usr/bin/env python3
"""Developer experiment: short runs (20 steps) against the number of warm-up steps and the state of the output buffers."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
W, H, B = 512, 384, 32
gen = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=5, num_objects=16, batch_size=B, sampler=1, seed=20261003))
gen.pool_synthetic(1000, 1024, 768, 2024)
st = torch.cuda.current_stream().cuda_stream
def run(steps, warm, nbuf, touch):
    outs = [ofdg.alloc_outputs(B, H, W) for _ in range(nbuf)]
    if touch:
        for o in outs:
            for t in o: t.zero_()
    torch.cuda.synchronize()
    for i in range(warm): gen.forward(*outs[i % nbuf], gen.next_stream())
    gen.synchronize(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): gen.forward(*outs[i % nbuf], gen.next_stream())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del outs
    return steps * B / dt
for rep in range(2):
    for (warm, nbuf, touch) in ((5, 8, 0), (5, 8, 1), (50, 8, 0), (5, 4, 0), (5, 5, 0), (8, 8, 0)):
        print("20 steps after %2d warm-up steps, %d output sets, %s: %.0f samples/s" % (warm, nbuf, "zero-filled" if touch else "untouched", run(20, warm, nbuf, touch)))
