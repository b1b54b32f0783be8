#!/usr/bin/env python3
"""Developer diagnostic: per-wave cycle stamps of compose_rigid (a build with -DOFDG_STAMPS, see tools/patches/).
Where does a strip's wave spend its life - scalar record, background loads, visits (tap issue / tap wait), stores -
alone on the device and inside the pipeline?  Usage on the GPU box:
    OFDG_LIB=.../libofdg_stamps.so python3 tools/exp_stamps.py [pipeline|alone] [hold]
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
mode_run = sys.argv[1] if len(sys.argv) > 1 else "pipeline"
hold = 1 if (len(sys.argv) > 2 and sys.argv[2] == "hold") else 0
W, H, B, MODE, NOBJ = 512, 384, 32, int(os.environ.get("MODE", "5")), 16
prm = ofdg.default_params(width=W, height=H, mode=MODE, num_objects=NOBJ, batch_size=B, sampler=1, seed=20261003)
gen = ofdg.Generator(prm)
gen.pool_synthetic(int(os.environ.get("POOLN", "1000")), 1024, 768, 2024)
NBUF = 2 * gen.num_chains()
outs = [ofdg.alloc_outputs(B, H, W) for _ in range(NBUF)]
n_strips = (W // 64) * (H // 16) * B * 4
NL = 8
buf = torch.zeros(NL * n_strips * 8, dtype=torch.int64, device="cuda")
L = ofdg.lib()
L.ofdg_debug_set_stamps.argtypes = [C.c_void_p, C.c_uint, C.c_int]
st = torch.cuda.current_stream().cuda_stream
for i in range(40):
    gen.forward(*outs[i % NBUF], gen.next_stream())
gen.synchronize(st)
assert L.ofdg_debug_set_stamps(C.c_void_p(buf.data_ptr()), NL, hold) == 0
import time
t0 = time.perf_counter()
N = 400 if mode_run == "pipeline" else 64
for i in range(N):
    gen.forward(*outs[i % NBUF], gen.next_stream())
    if mode_run == "alone":
        gen.synchronize(st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s hold=%d: %.1f us/step (%d steps)" % (mode_run, hold, dt / N * 1e6, N))
a = buf.cpu().numpy().view(np.uint64).reshape(NL, n_strips, 8)
# clock: cycles per realtime tick (100 MHz) from the wave start pairs of one launch
for l in (NL - 3,):
    r = a[l]
    real, t0c = r[:, 0].astype(np.float64), r[:, 1].astype(np.float64)
    # per XCD the s_memtime counters differ: fit within one XCD
    xcc = ((r[:, 7] >> np.uint64(32)) & np.uint64(0xF)).astype(int)
    dreal = (r[:, 7] >> np.uint64(36)).astype(np.float64)      # wave start -> after the stamps were taken, 10 ns ticks
    dcyc = (r[:, 4] >> np.uint64(32)).astype(np.float64) + (r[:, 5] >> np.uint64(32)).astype(np.float64)
    slope = dcyc.sum() / dreal.sum()
    ghz = slope * 0.1
    print("launch slot %d: shader clock %.2f GHz (s_memtime ticks per 10 ns: %.1f)" % (l, ghz, slope))
    cyc = lambda v: v / (ghz * 1e3)  # cycles -> us
    d1 = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.float64); d2 = (r[:, 2] >> np.uint64(32)).astype(np.float64)
    d3 = (r[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64); d4 = (r[:, 3] >> np.uint64(32)).astype(np.float64)
    d5 = (r[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.float64); d6 = (r[:, 4] >> np.uint64(32)).astype(np.float64)
    vis = (r[:, 5] & np.uint64(0xFFFF)).astype(int); rvis = ((r[:, 5] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(int)
    tiss = (r[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.float64); twait = (r[:, 6] >> np.uint64(32)).astype(np.float64)
    d7 = (r[:, 5] >> np.uint64(32)).astype(np.float64)
    wall = (real.max() - real.min()) * 0.01
    print("  waves %d, first->last wave start %.1f us; sum of lifetimes (to stores issued) %.0f us -> %.0f waves in flight if the launch takes %.1f us"
          % (len(r), wall, cyc(d6).sum(), cyc(d6).sum() / max(wall, 1e-9), wall))
    def row(name, sel):
        if sel.sum() == 0:
            return
        f = lambda v: "%6.2f" % np.mean(cyc(v[sel]))
        print("  %-22s n=%6d | record %s | bg issue %s | bg wait %s | bg math %s | visits %s | stores %s | total %s (p50 %5.2f p90 %5.2f p99 %5.2f) | ack %s"
              % (name, sel.sum(), f(d1), f(d2 - d1), f(d3 - d2), f(d4 - d3), f(d5 - d4), f(d6 - d5), f(d6),
                 np.percentile(cyc(d6[sel]), 50), np.percentile(cyc(d6[sel]), 90), np.percentile(cyc(d6[sel]), 99), f(d7)))
    row("all", np.ones(len(r), bool))
    row("no mask bits", vis == 0)
    row("mask bits, no coverage", (vis > 0) & (rvis == 0))
    for n in (1, 2, 3):
        row("%d real visit(s)" % n, rvis == n)
    row(">=4 real visits", rvis >= 4)
    k = rvis > 0
    print("  per real visit: tap issue %.2f us, tap wait %.2f us; visits/wave %.2f (real %.2f); waves with a real visit %.1f %%"
          % (cyc(tiss[k]).sum() / rvis[k].sum(), cyc(twait[k]).sum() / rvis[k].sum(), vis.mean(), rvis.mean(), 100.0 * k.mean()))
    # per XCD: number of waves, sum of lifetimes, last end
    for x in range(0):
        kx = xcc == x
        if kx.sum():
            print("  xcd %d: waves %5d, real visits %5d, sum life %.0f us, start span %.1f..%.1f us" % (
                x, kx.sum(), rvis[kx].sum(), cyc(d6[kx]).sum(), (real[kx].min() - real.min()) * 0.01, (real[kx].max() - real.min()) * 0.01))
