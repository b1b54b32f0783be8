"""What bench.py TIMES, compared with the oracle (VERDICT r04 "next" #1).

The workloads are bench.CONFIGS themselves (imported: no second copy of the numbers), with the background preparation
the headline runs (background_prep = 1, Texture::getRandomizedCrop(2W, 2H, rot, zoom, shift), DataGenerator.cpp:87-109,
1186-1192), the pool bench.py builds (same size, same seed), the seed bench.py uses, one rank's whole batch rendered the
way a bench step renders it (ofdg_forward_counter on the context's own stream) - and EVERY sample of the batch, for two
different steps per config, compared with oracle.render under oracle.detmath(): frames 0 LSB, flow <= 1 ULP.  The oracle
renders the samples on the host's threads (one sample per thread, the reference's first_level_threads, DG:1023-1027);
OFDG_PARITY_SUBSET=1 in the environment falls back to the first, the middle and the last two samples (a box with few cores).

The one-launch form of the preparation (bgprep_stream_kernel) hands the batch's tiles (64 x 32 texels of a sample's
2W x 2H texture) to a fixed number of single-wave workgroups grid-stride.  A bench batch holds fewer tiles than there are
workgroups (3 400 - 6 100 against 4 096: one tile per wave, two for some in config 3), so the grid-stride part of the
kernel's loop - a workgroup's SECOND, third ... tile, with the placement and the bounding table entries of the next tile
requested a turn ahead - is compared with the oracle on a batch made large enough for it
(test_preparation_tile_loop_beyond_the_grid: config 2's shape with 128 samples), which asserts through
ofdg_debug_bgprep_tiles that it holds more than three times as many tiles as there are workgroups, so that the property
cannot lapse silently when a grid constant or the tile size changes.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def blueprints_of(task, bps):
    """indices of every blueprint a task refers to (background, objects, their components)"""
    used = [task.background] + list(range(task.first_object, task.first_object + task.n_objects))
    for k in range(task.n_objects):
        b = bps[task.first_object + k]
        used += list(range(b.first_component, b.first_component + b.n_components))
    return used


def check_samples(ofdg, oracle, g, prm, tasks, bps, n_bps, got, which, pool_n, crops=None, detmath=True, flow_ulp=1):
    """Samples `which` of a rendered batch against the oracle, one oracle call per sample, the calls spread over the host's
    threads (ctypes releases the GIL; the oracle's mode switches are process-wide and set once around all of them).  Each
    call gets a host pool of just the images its sample uses (texture ids re-indexed: tex_id % len(images) picks the same
    image in the small pool)."""
    import concurrent.futures as cf
    W, H = prm.width, prm.height
    which = list(which)
    i0, i1, fl = (a.cpu().numpy() for a in got)   # one copy of the whole batch

    def one(sidx, images, host_pool):
        t = tasks[sidx]
        sub = (type(bps[0]) * n_bps)()
        C.memmove(sub, bps, C.sizeof(sub))
        for i in blueprints_of(t, bps):
            sub[i].tex_id = images.index(bps[i].tex_id % pool_n)
        q = oracle.default_params(W, H, prm.mode, prm.use_antialiasing, 1, prm.num_objects)
        q.background_prep = prm.background_prep
        e0, e1, ef = oracle.render(q, (type(t) * 1)(t), 1, sub, n_bps, host_pool, warp_crops=crops, reuse=-1)
        a0, a1, af = i0[sidx], i1[sidx], fl[sidx]
        assert np.array_equal(a0, e0[0]), "sample %d image0: %d values differ, max %g" % (sidx, (a0 != e0[0]).sum(), np.abs(a0 - e0[0]).max())
        assert np.array_equal(a1, e1[0]), "sample %d image1: %d values differ, max %g" % (sidx, (a1 != e1[0]).sum(), np.abs(a1 - e1[0]).max())
        assert np.array_equal(np.isnan(af), np.isnan(ef[0]))
        ok = ~np.isnan(ef[0])
        assert ulp_diff(af[ok], ef[0][ok]).max() <= flow_ulp, "sample %d flow" % sidx
        return sidx

    import contextlib, os
    workers = max(1, min(len(which), os.cpu_count() or 1))
    mode = oracle.detmath() if detmath else contextlib.nullcontext()  # the device sampler builds its affines and the preparation record with include/ofdg_detmath.h
    with mode, cf.ThreadPoolExecutor(workers) as ex:
        chunk = 8   # (the pool images of a chunk's samples are downloaded on this thread: the GPU calls stay on one thread)
        for c in range(0, len(which), chunk):
            futs = []
            for sidx in which[c:c + chunk]:
                images = sorted({bps[i].tex_id % pool_n for i in blueprints_of(tasks[sidx], bps)})
                futs.append(ex.submit(one, sidx, images, np.stack([g.pool_download(i) for i in images])))
            for f in futs:
                f.result()
    return len(which)


def subset_only():
    import os
    return os.environ.get("OFDG_PARITY_SUBSET", "0") not in ("", "0")


def bench_generator(ofdg, bench, cfg, sampler):
    W, H, B = cfg["W"], cfg["H"], cfg["batch"]
    prm = ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=B, sampler=sampler,
                              seed=bench.SEED, background_prep=1)
    g = ofdg.Generator(prm)
    g.pool_synthetic(*cfg["pool"], bench.POOL_SEED)
    crops = None
    if cfg["mode"] == 9:
        g.warp_generate(2, bench.SEED)
        crops = np.stack([g.warp_download(i) for i in range(g.warp_count())])
    return g, prm, crops


def assert_tiles_beyond_the_grid(g, k=3):
    tiles, groups = g.debug_bgprep_tiles()
    assert tiles > k * groups, "the batch's %d tiles do not reach a workgroup's tile %d (%d workgroups): this test no longer checks the tile loop" % (tiles, k + 1, groups)
    return tiles, groups


@pytest.mark.parametrize("step", [2, 7])
@pytest.mark.parametrize("config", [2, 3, 4, 5])
def test_bench_batch_with_background_preparation_matches_oracle(ofdg, oracle, config, step):
    """One rank's batch of BASELINE configs 2-5 exactly as `python bench.py --config N` renders its steps (counter sampler,
    background_prep = 1, the config's pool), steps 2 and 7 of rank 0: EVERY sample at 0 LSB / <= 1 ULP (Process_TaskBucket,
    DG:1175-1254)."""
    import bench
    import time
    torch = pytest.importorskip("torch")
    cfg = bench.CONFIGS[config]
    B = cfg["batch"]
    g, prm, crops = bench_generator(ofdg, bench, cfg, sampler=1)
    got = ofdg.alloc_outputs(B, cfg["H"], cfg["W"])
    first = step * B   # (any step is a pure function of (seed, index))
    g.forward_counter(first, B, *got, ofdg.STREAM_OWN)
    g.synchronize()
    torch.cuda.synchronize()
    tiles, groups = g.debug_bgprep_tiles()
    assert tiles > 0                        # (the one-launch form of the preparation rendered this batch)
    tasks, bps, n = g.sample_counter(first, B)
    which = set(range(B))
    if subset_only():
        which = {0, B // 2, B - 2, B - 1}
        if cfg["mode"] == 9:
            # a background that is re-sampled through a warp field reads frame 1 at displaced positions (the device sampler once
            # kept the rigid read region for it - found by this test): every such sample stays in the subset
            which.update(i for i, t in enumerate(tasks) if bps[t.background].do_warpfield_deformation)
    if cfg["mode"] == 9 and step == 2:
        assert sum(1 for t in tasks if bps[t.background].do_warpfield_deformation) >= 2, "too few deforming backgrounds in the batch: pick another step"
    t0 = time.time()
    n_checked = check_samples(ofdg, oracle, g, prm, tasks, bps, n, got, sorted(which), cfg["pool"][0], crops)
    print("config %d step %d: %d of %d samples against the oracle in %.1f s" % (config, step, n_checked, B, time.time() - t0))
    g.close()


def test_preparation_tile_loop_beyond_the_grid(ofdg, oracle):
    """Config 2's workload with 128 samples in the batch: more than three times as many tiles as the preparation has
    workgroups, so every workgroup walks its grid-stride loop to a fourth tile (the next tile's sample, placement and bounding
    table entries requested a turn ahead).  First, middle and last two samples - the highest tile numbers - against the oracle."""
    import bench
    torch = pytest.importorskip("torch")
    cfg = dict(bench.CONFIGS[2], batch=128)
    B = cfg["batch"]
    g, prm, _ = bench_generator(ofdg, bench, cfg, sampler=1)
    got = ofdg.alloc_outputs(B, cfg["H"], cfg["W"])
    g.forward_counter(5 * B, B, *got, ofdg.STREAM_OWN)
    g.synchronize()
    torch.cuda.synchronize()
    assert_tiles_beyond_the_grid(g)
    tasks, bps, n = g.sample_counter(5 * B, B)
    which = [0, B // 2, B - 2, B - 1] if subset_only() else range(B)
    check_samples(ofdg, oracle, g, prm, tasks, bps, n, got, which, cfg["pool"][0])
    g.close()


def test_host_sampled_batch_with_background_preparation_matches_oracle(ofdg, oracle):
    """The same for the host-sampled path (ofdg_render: reference-stream blueprints realised on the host, the preparation's
    records uploaded with the batch, the preparation behind raster): config 2's shape, 32 samples of 512 x 384."""
    import bench
    cfg = bench.CONFIGS[2]
    W, H, B = cfg["W"], cfg["H"], cfg["batch"]
    g, prm, _ = bench_generator(ofdg, bench, cfg, sampler=0)
    tasks, bps, n = oracle.Sampler(cfg["mode"], W, H, cfg["nobj"]).next(B)
    got = ofdg.alloc_outputs(B, H, W)
    g.render(tasks, B, bps, n, *got)
    g.synchronize()
    assert g.debug_bgprep_tiles()[0] > 0
    which = [0, B // 2, B - 2, B - 1] if subset_only() else range(B)
    check_samples(ofdg, oracle, g, prm, tasks, bps, n, got, which, cfg["pool"][0], detmath=False, flow_ulp=0)  # (libm arithmetic: the host path's own)
    g.close()


def test_bench_batches_take_the_fast_forms_of_the_preparation(ofdg):
    """Parity cannot tell a batch whose tiles all fell back to the preparation's general forms (mirrored crop coordinates, resize
    decided per row) from one that took the fast ones - the kernel counts its tiles by form (ofdg_debug_bgprep_paths): of config
    2's batch at least 85 % must take the rotation specialised by the side of the shift's mirror lines (a mirror line crosses about
    one tile in ten) AND one of the two specialised resize loops (getRandomizedCrop's zoom makes both axes enlarge or both shrink,
    DG:87-109), and every tile is counted once."""
    import bench
    torch = pytest.importorskip("torch")
    cfg = bench.CONFIGS[2]
    B = cfg["batch"]
    g, prm, _ = bench_generator(ofdg, bench, cfg, sampler=1)
    got = ofdg.alloc_outputs(B, cfg["H"], cfg["W"])
    assert sum(map(sum, g.debug_bgprep_paths())) == 0      # (switches the counting on)
    fast = total = 0
    for step in (2, 7, 11):
        g.forward_counter(step * B, B, *got, ofdg.STREAM_OWN)
        g.synchronize()
        torch.cuda.synchronize()
        tiles, _ = g.debug_bgprep_tiles()
        paths = g.debug_bgprep_paths()
        assert sum(map(sum, paths)) == tiles, (paths, tiles)
        fast += paths[1][2] + paths[2][2]
        total += tiles
    assert fast >= 0.85 * total, "only %d of %d tiles took the specialised forms" % (fast, total)
    g.close()
