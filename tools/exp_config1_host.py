#!/usr/bin/env python3
"""Developer experiment: where the host's time goes in config 1 (batch 1, reference-stream sampler inside the call): the
sampler alone, ofdg_render (realize + one copy + launches) on its blueprints, and ofdg_forward (both)."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
for bgp in (1, 0):
    g = ofdg.Generator(ofdg.default_params(mode=7, batch_size=1, num_objects=1, sampler=0, background_prep=bgp))
    g.pool_synthetic(1000, 1024, 768, 1)
    outs = [ofdg.device_pointers(ofdg.alloc_outputs(1, 384, 512)) for _ in range(8)]
    hs = ofdg.HostSampler(7, 512, 384, 1)
    N = 3000
    t = time.perf_counter()
    for i in range(N): tasks, bps, n = hs.next(1, cap=64)
    t_s = (time.perf_counter() - t) / N * 1e6
    for i in range(200): g.render(tasks, 1, bps, n, *outs[i % 8], ofdg.STREAM_OWN)
    g.synchronize()
    t = time.perf_counter()
    for i in range(N): g.render(tasks, 1, bps, n, *outs[i % 8], ofdg.STREAM_OWN)
    t_r = (time.perf_counter() - t) / N * 1e6
    g.synchronize()
    t = time.perf_counter()
    for i in range(N): g.forward(*outs[i % 8], ofdg.STREAM_OWN)
    t_f = (time.perf_counter() - t) / N * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t) / N * 1e6
    print("background_prep %d: host sampler (python object) %.1f us | ofdg_render %.1f us | ofdg_forward %.1f us host, %.1f us until done" % (bgp, t_s, t_r, t_f, t_all), flush=True)
    g.close()
