#!/usr/bin/env python3
"""Developer experiment: when the workgroups of bgprep_fused_kernel start and how long they run, alone (one chain) and in the
pipeline (four chains).  Needs a library built with tools/patches/r04_bgprep_workgroup_times.patch applied."""
import ctypes as C, importlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
outs = [ofdg.alloc_outputs(32, 384, 512) for _ in range(8)]
for chains in (1, 4):
    g = ofdg.Generator(ofdg.default_params(width=512, height=384, mode=5, num_objects=16, batch_size=32, sampler=1, seed=20261003, background_prep=1, chains=chains))
    g.pool_synthetic(1000, 1024, 768, 2024)
    res = []
    for rep in range(6):
        for i in range(40 + rep): g.forward(*outs[i % 8], g.next_stream())
        g.synchronize()
        buf = (C.c_ulonglong * (2048 * 2))()
        ofdg.lib().ofdg_debug_wg_times(buf, 2048)
        t = np.array(buf, dtype=np.uint64).reshape(2048, 2).astype(np.int64)
        t0 = t[:, 0].min()
        st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
        res.append((np.percentile(st, [10, 50, 90, 100]), np.percentile(en - st, [10, 50, 90]), en.max()))
    for st, du, tot in res[-3:]:
        print("chains %d: workgroup start after the first one (us) p10 %.1f p50 %.1f p90 %.1f max %.1f | workgroup duration p10 %.1f p50 %.1f p90 %.1f | kernel %.1f" % (chains, *st, *du, tot), flush=True)
    g.close()
