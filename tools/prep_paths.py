import importlib, sys, torch
sys.path.insert(0, '.')
import bench
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
for cfgn in (2, 3, 4, 5):
    cfg = bench.CONFIGS[cfgn]; B = cfg["batch"]
    prm = ofdg.default_params(width=cfg["W"], height=cfg["H"], mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=B, sampler=1, seed=bench.SEED, background_prep=1)
    g = ofdg.Generator(prm); g.pool_synthetic(*cfg["pool"], bench.POOL_SEED)
    if cfg["mode"] == 9: g.warp_generate(2, bench.SEED)
    got = ofdg.alloc_outputs(B, cfg["H"], cfg["W"])
    g.debug_bgprep_paths()
    tot = [[0]*3 for _ in range(3)]
    for step in range(8):
        g.forward_counter(step * B, B, *got, ofdg.STREAM_OWN); g.synchronize()
        p = g.debug_bgprep_paths()
        for r in range(3):
            for k in range(3): tot[r][k] += p[r][k]
    n = sum(map(sum, tot))
    print("config %d: %d tiles; [resize per-row / enlarge / shrink] x [rotation general / inside / by side] = %s; fast both: %.1f %%" % (cfgn, n, tot, 100.0 * (tot[1][2] + tot[2][2]) / n))
    g.close()
