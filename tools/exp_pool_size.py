#!/usr/bin/env python3
"""Developer experiment (round 5): the headline step (config 2, counter sampler, background_prep 1 or 0) against the SIZE of the
texture pool - 1000 images (3 GB: every texel read is a cold HBM read, what bench.py runs) down to a pool that stays in the
Infinity Cache / L2.  If the step does not care, the cold pool reads are not what its kernels wait for.
Usage on the GPU box: python3 tools/exp_pool_size.py [steps]"""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = bench.CONFIGS[2]
W, H, B = cfg["W"], cfg["H"], cfg["batch"]
outs = None
for prep in (1, 0):
    for n_pool in (1000, 128, 32, 8):
        g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=B, sampler=1,
                                               seed=bench.SEED, background_prep=prep))
        g.pool_synthetic(n_pool, cfg["pool"][1], cfg["pool"][2], bench.POOL_SEED)
        if outs is None:
            outs = [ofdg.device_pointers(ofdg.alloc_outputs(B, H, W)) for _ in range(2 * g.num_chains())]
        for i in range(20):
            g.forward(*outs[i % len(outs)], ofdg.STREAM_OWN)
        g.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g.forward(*outs[i % len(outs)], ofdg.STREAM_OWN)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g.synchronize()
        print("background_prep %d  pool %4d images (%5.0f MB)  %7.1f us/step  %8.0f samples/s" % (prep, n_pool, n_pool * cfg["pool"][1] * cfg["pool"][2] * 4 / 1e6, dt / steps * 1e6, steps * B / dt), flush=True)
        g.close()
