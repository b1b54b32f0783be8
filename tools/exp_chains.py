"""Developer experiment: K independent in-order chains (sampler -> geom -> raster -> compose on ONE stream each,
OFDG_OVERLAP=0), K streams round-robin from one host thread — is a work-conserving K-chain pipeline faster than
the three-stream pipeline with cross-stream waits?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
W, H, B = 512, 384, 32
K = int(os.environ.get("K", "2")); MODE = int(os.environ.get("MODE", "5")); NOBJ = int(os.environ.get("NOBJ", "16"))
gens = []
for k in range(K):
    g = ofdg.Generator(ofdg.default_params(mode=MODE, batch_size=B, width=W, height=H, num_objects=NOBJ, sampler=1, seed=5))
    g.pool_synthetic(int(os.environ.get("POOLN", "1000")), 1024, 768, seed=1)
    gens.append(g)
streams = [torch.cuda.Stream() for _ in range(K)]
outs = [ofdg.alloc_outputs(B, H, W) for _ in range(K)]
for i in range(20 * K): gens[i % K].forward_counter(i * B, B, *outs[i % K], streams[i % K].cuda_stream)
torch.cuda.synchronize()
N = int(os.environ.get("N", "300"))
for g in gens: g.set_profiling(2 if os.environ.get("OFDG_OVERLAP") == "0" else 1)
t = time.perf_counter()
for i in range(N): gens[i % K].forward_counter((20 * K + i) * B, B, *outs[i % K], streams[i % K].cuda_stream)
t_host = (time.perf_counter() - t) / N
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
km = lambda n: sum(g.kernel_ms(n) for g in gens) / K * 1e3
print("kernels: compose %.1f us" % km("compose") + (" geom %.1f raster %.1f" % (km("geom"), km("raster")) if os.environ.get("OFDG_OVERLAP") == "0" else ""))
print(f"K={K} chains overlap={os.environ.get('OFDG_OVERLAP','1')}: step={dt*1e6:.1f} us host {t_host*1e6:.1f} us -> {B/dt:.0f} samples/s")
