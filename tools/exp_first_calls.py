#!/usr/bin/env python3
"""Developer experiment: host time of each of the first calls after an idle device (what paces the start of a 20-step run)."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
B = 32
g = ofdg.Generator(ofdg.default_params(width=512, height=384, mode=5, num_objects=16, batch_size=B, sampler=1, seed=5, background_prep=int(os.environ.get("BGPREP", "1"))))
g.pool_synthetic(1000, 1024, 768, 1)
outs = [ofdg.alloc_outputs(B, 384, 512) for _ in range(8)]
for i in range(5): g.forward(*outs[i % 8], ofdg.STREAM_OWN)
g.synchronize()
for prof in (0, 1):
    g.set_profiling(prof)
    for rep in range(3):
        torch.cuda.synchronize()
        time.sleep(float(os.environ.get("IDLE", "0.002")))
        ts = [time.perf_counter()]
        for i in range(12):
            g.forward(*outs[i % 8], ofdg.STREAM_OWN)
            ts.append(time.perf_counter())
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        print("profiling %d: per call (us): %s | all done after %.0f us" % (prof, " ".join("%.0f" % ((ts[i + 1] - ts[i]) * 1e6) for i in range(12)), (t_end - ts[0]) * 1e6), flush=True)
