"""Counter-sampler forward path timing (device sample+realize -> geom -> raster -> compose)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
W, H, B = 512, 384, 32
MODE = int(os.environ.get("MODE", "5")); NOBJ = int(os.environ.get("NOBJ", "16"))
g = ofdg.Generator(ofdg.default_params(mode=MODE, batch_size=B, width=W, height=H, num_objects=NOBJ, sampler=1, seed=5, background_prep=int(os.environ.get("BGPREP", "0"))))
g.pool_synthetic(1000, 1024, 768, seed=1)
if MODE == 9: g.warp_generate(2, 1)
NBUF = int(os.environ.get("NBUF", "4"))  # output buffer sets the caller cycles (prefetch ring)
outs = [ofdg.alloc_outputs(B, H, W) for _ in range(NBUF)]
OWN = os.environ.get("OWNSTREAM")  # 1: pass torch's stream (cross-stream hand-over) instead of the chain's own
st0 = torch.cuda.current_stream().cuda_stream
st = (lambda: st0) if OWN else g.next_stream
for i in range(20): g.forward_counter(i * B, B, *outs[i % NBUF], st())
torch.cuda.synchronize()
N = int(os.environ.get("N", "300"))
PROF = not os.environ.get("NOPROF")
if PROF: g.set_profiling(1)
t = time.perf_counter()
for i in range(N): g.forward_counter((20 + i) * B, B, *outs[i % NBUF], st())
t_host = (time.perf_counter() - t) / N
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
print(f"counter-sampler forward mode={MODE} nbuf={NBUF}: step={dt*1e6:.1f} us (compose kernel {(g.kernel_ms('compose') if PROF else 0)*1e3:.1f} us, host enqueue {t_host*1e6:.1f} us/step) -> {B/dt:.0f} samples/s")
