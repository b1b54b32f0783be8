#!/usr/bin/env python3
"""Developer experiment: standalone (no stream overlap) per-kernel times of the bench workload."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
W, H, MODE, B = 512, 384, int(os.environ.get("MODE", "5")), 32
NOBJ = int(os.environ.get("NOBJ", "16"))
g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=MODE, num_objects=NOBJ, serial=1))  # one kernel after the other on the caller's stream
g.pool_synthetic(int(os.environ.get("POOLN", "1000")), int(os.environ.get("POOLW", "1024")), int(os.environ.get("POOLH", "768")), 2024)
if MODE == 9:
    g.warp_generate(2, 2024)
hs = ofdg.HostSampler(MODE, W, H, NOBJ)
st = torch.cuda.current_stream().cuda_stream
NS = 8
for slot in range(NS):
    tasks, bps, n = hs.next(B, cap=B * 64)
    if os.environ.get("BGONLY"):
        for t in tasks: t.n_objects = 0
    g.upload_slot(slot, tasks, B, bps, n, st)
i0, i1, fl = ofdg.alloc_outputs(B, H, W)
for i in range(int(os.environ.get("WARM", "20"))): g.render_slot(i % NS, i0, i1, fl, st)
g.synchronize(st)
g.set_profiling(2)
for i in range(int(os.environ.get("ITERS", "128"))): g.render_slot(i % NS, i0, i1, fl, st)
g.synchronize(st)
print("serial, mode=%d geom=%.1f raster=%.1f compose=%.1f us" % (MODE,
    g.kernel_ms("geom") * 1e3, g.kernel_ms("raster") * 1e3, g.kernel_ms("compose") * 1e3))
