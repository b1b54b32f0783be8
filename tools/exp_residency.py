#!/usr/bin/env python3
"""Developer experiment (VERDICT r04 "next" #3): what the co-running kernels of a step contend for, measured from inside the
pipeline.  Needs a library built with tools/patches/r05_residency_stamps.patch (-DOFDG_RESIDENCY=1): every wave of the step's
five kernels leaves {start, lifetime, HW_ID (wave slot, SIMD, CU, SH, SE), XCC_ID, kernel, chain}.  This script renders the
headline workload (config 2, counter sampler, background_prep 1) with four chains and with one (every kernel alone), reads
the records of the last STEPS steps and reconstructs, per SIMD and per CU, over the window the steps ran in:

  * wave lifetimes per kernel, in the pipeline against alone;
  * time-averaged resident waves per SIMD by kernel, the VGPRs they hold (allocation = registers rounded up to 8, 512 per
    SIMD lane) and the LDS their workgroups hold per CU (160 KB);
  * how much of the time a SIMD / CU had room for ONE MORE wave of each kernel (registers, a wave slot of the 8, LDS) - and
    how much of THAT time the kernel had workgroups waiting to be placed (a launch's dispatch window = first to last wave
    start): room while workgroups wait = the dispatcher, not a resource, is what holds them back;
  * per launch: dispatch window against duration.

Usage on the GPU box: OFDG_LIB=.../libofdg_resid.so python3 tools/exp_residency.py [steps] > profiles/r05_residency.txt"""
import ctypes as C, importlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
L = ofdg.lib()
L.ofdg_debug_residency_reset.argtypes = [C.c_void_p]
L.ofdg_debug_residency_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
L.ofdg_debug_residency_read.restype = C.c_longlong

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
KERNELS = ["sampler", "geom", "raster", "background_prep", "compose"]
# registers (allocated in steps of 8) and LDS per single-wave workgroup of this build (-Rpass-analysis=kernel-resource-usage)
VGPR = {"sampler": 72, "geom": 64, "raster": 64, "background_prep": 88, "compose": 88}
LDS = {"sampler": 14040, "geom": 14080, "raster": 10272, "background_prep": 5760 + 132, "compose": 0}
SIMD_VGPRS, SIMD_SLOTS, CU_LDS = 512, 8, 160 * 1024
TICK_US = 0.01  # s_memrealtime: 100 MHz

cfg = bench.CONFIGS[2]
W, H, B = cfg["W"], cfg["H"], cfg["batch"]


def run(chains):
    g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=cfg["mode"], num_objects=cfg["nobj"], batch_size=B, sampler=1,
                                           seed=bench.SEED, background_prep=1, chains=chains))
    g.pool_synthetic(*cfg["pool"], bench.POOL_SEED)
    outs = [ofdg.device_pointers(ofdg.alloc_outputs(B, H, W)) for _ in range(2 * g.num_chains())]
    for i in range(40):
        g.forward(*outs[i % len(outs)], ofdg.STREAM_OWN)
    g.synchronize()
    L.ofdg_debug_residency_reset(g.h)
    for i in range(STEPS + 16):      # (the first and last steps of the run fill and drain the pipeline: cut off below)
        g.forward(*outs[i % len(outs)], ofdg.STREAM_OWN)
        if chains == 1:
            g.synchronize()
    g.synchronize()
    cap = 1 << 22
    buf = np.zeros((cap, 6), np.uint32)
    n = L.ofdg_debug_residency_read(g.h, buf.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n < cap, n    # (negative: records lost to a full ring)
    g.close()
    r = buf[:n]
    t0 = r[:, 0].astype(np.int64) | (r[:, 1].astype(np.int64) << 32)
    rec = dict(t0=t0 * TICK_US, dt=r[:, 2].astype(np.float64) * TICK_US, hw=r[:, 3], kernel=(r[:, 4] & 15).astype(np.int64),
               xcc=((r[:, 4] >> 4) & 15).astype(np.int64), chain=r[:, 5].astype(np.int32))
    rec["t0"] -= rec["t0"].min()
    return rec


def place(rec):
    hw = rec["hw"].astype(np.int64)
    wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cu_key = ((rec["xcc"] * 8 + se) * 2 + sh) * 16 + cu      # (sorted by XCC first: np.unique keeps the CUs of an XCC together)
    keys, cu_idx = np.unique(cu_key, return_inverse=True)
    place.xcc_of_cu = keys // (8 * 2 * 16)
    return cu_idx, simd, wave


def launches(rec):
    """(kernel, chain) streams are in order: a launch = a run of waves of one (kernel, chain) whose starts are closer than the
    gap to the next launch of that pair.  Returns per wave a launch number."""
    lid = np.full(len(rec["t0"]), -1, np.int64)
    nxt = 0
    info = []
    for k in range(5):
        for ch in np.unique(rec["chain"]):
            idx = np.nonzero((rec["kernel"] == k) & (rec["chain"] == ch))[0]
            if len(idx) == 0:
                continue
            idx = idx[np.argsort(rec["t0"][idx])]
            end_run = np.maximum.accumulate(rec["t0"][idx] + rec["dt"][idx])
            # a new launch starts where a wave starts after every earlier wave of the pair has ended AND a gap of > 3 us follows
            brk = np.nonzero(rec["t0"][idx][1:] > end_run[:-1] + 3.0)[0] + 1
            for seg in np.split(idx, brk):
                lid[seg] = nxt
                info.append((k, int(ch), rec["t0"][seg].min(), rec["t0"][seg].max(), (rec["t0"][seg] + rec["dt"][seg]).max(), len(seg)))
                nxt += 1
    return lid, info


def timeline(rec, cu_idx, simd, lo, hi, step=0.25):
    """resident waves per (CU, SIMD, kernel) on a time grid, by difference arrays"""
    n_cu = cu_idx.max() + 1
    nt = int((hi - lo) / step) + 1
    occ = np.zeros((5, n_cu * 4, nt + 1), np.int32)
    a = np.clip(((rec["t0"] - lo) / step).astype(np.int64), 0, nt)
    b = np.clip(((rec["t0"] + rec["dt"] - lo) / step).astype(np.int64) + 1, 0, nt)
    s = cu_idx * 4 + simd
    for k in range(5):
        m = rec["kernel"] == k
        np.add.at(occ[k], (s[m], a[m]), 1)
        np.add.at(occ[k], (s[m], b[m]), -1)
    return np.cumsum(occ, axis=2)[:, :, :nt], nt, n_cu


def main():
    print("# what the kernels of a step hold and what they wait for: wave records of %d steps (config 2, counter sampler, background_prep 1)" % STEPS)
    alone = run(1)
    pipe = run(0)
    for name, rec in (("alone (one chain, a device-wide wait after every step)", alone), ("in the pipeline (four chains)", pipe)):
        print("\n## wave lifetimes, %s" % name)
        print("%-16s %9s %9s %9s %9s %9s" % ("kernel", "waves", "p10 us", "median", "p90", "mean"))
        for k, kn in enumerate(KERNELS):
            d = rec["dt"][rec["kernel"] == k]
            if len(d):
                print("%-16s %9d %9.1f %9.1f %9.1f %9.1f" % (kn, len(d), *np.percentile(d, [10, 50, 90]), d.mean()))
    cu_idx, simd, wave = place(pipe)
    print("\n## placement: %d compute units, SIMDs %s, wave slots %s seen" % (cu_idx.max() + 1, sorted(set(simd.tolist())), sorted(set(wave.tolist()))))
    lid, info = launches(pipe)
    # the window: the middle STEPS compose launches
    comp = sorted([i for i in info if i[0] == 4], key=lambda i: i[2])
    assert len(comp) >= STEPS + 8, len(comp)
    lo, hi = comp[8][2], comp[8 + STEPS][2]
    print("window: %.0f us = %d steps of %.1f us" % (hi - lo, STEPS, (hi - lo) / STEPS))
    print("\n## launches in the window: dispatch window (first to last wave start) against duration (first start to last end), us, medians")
    print("%-16s %8s %9s %10s %10s" % ("kernel", "launches", "waves", "dispatch", "duration"))
    for k, kn in enumerate(KERNELS):
        sel = [i for i in info if i[0] == k and lo <= i[2] < hi]
        if sel:
            print("%-16s %8d %9.0f %10.1f %10.1f" % (kn, len(sel), np.median([i[5] for i in sel]), np.median([i[3] - i[2] for i in sel]), np.median([i[4] - i[2] for i in sel])))
    occ, nt, n_cu = timeline(pipe, cu_idx, simd, lo, hi)
    print("\n## resident waves per SIMD, time average over the window (%d SIMDs)" % (n_cu * 4))
    tot_w, tot_v = 0.0, 0.0
    for k, kn in enumerate(KERNELS):
        w = occ[k].mean()
        tot_w += w; tot_v += w * VGPR[kn]
        print("%-16s %6.2f waves  %6.1f VGPRs (%4.1f %% of 512)" % (kn, w, w * VGPR[kn], 100 * w * VGPR[kn] / SIMD_VGPRS))
    print("%-16s %6.2f waves  %6.1f VGPRs (%4.1f %% of 512)" % ("all", tot_w, tot_v, 100 * tot_v / SIMD_VGPRS))
    vg = sum(occ[k].astype(np.int64) * VGPR[kn] for k, kn in enumerate(KERNELS))           # [simd, t]
    slots = occ.sum(axis=0)
    lds = sum(occ[k].reshape(n_cu, 4, nt).sum(axis=1).astype(np.int64) * LDS[kn] for k, kn in enumerate(KERNELS))  # [cu, t]
    print("LDS held per CU: mean %.1f KB, p90 %.1f KB of 160; wave slots per SIMD: mean %.2f, p90 %.0f of 8; VGPRs per SIMD: p10 %d, median %d, p90 %d of 512" % (
        lds.mean() / 1024, np.percentile(lds, 90) / 1024, slots.mean(), np.percentile(slots, 90), *np.percentile(vg, [10, 50, 90])))
    print("waves resident on a SIMD, share of SIMD-time: " + "  ".join("%d: %.1f %%" % (n, 100 * (slots == n).mean()) for n in range(0, slots.max() + 1)))
    print("largest number of waves seen together on a SIMD: %d (with %d VGPRs); SIMD-time with >= 5 waves AND >= 64 VGPRs free (room for a sixth wave of raster / geom): %.2f %%" % (
        slots.max(), vg[slots == slots.max()].min(), 100 * ((slots >= 5) & (vg + 64 <= SIMD_VGPRS)).mean()))
    # pending: a kernel has workgroups waiting between the first and the last wave start of a launch
    step = (hi - lo) / nt
    print("\n## room for one more wave of a kernel (registers on the SIMD, a wave slot, LDS on the CU) - share of SIMD-time in the window")
    print("%-16s %12s %22s %30s" % ("kernel", "room", "workgroups waiting", "room WHILE workgroups wait"))
    lds_s = np.repeat(lds, 4, axis=0)
    for k, kn in enumerate(KERNELS):
        room = (vg + VGPR[kn] <= SIMD_VGPRS) & (slots < SIMD_SLOTS) & (lds_s + LDS[kn] <= CU_LDS)
        pend = np.zeros(nt, bool)
        for i in info:
            if i[0] == k:
                a, b = int(max(0, (i[2] - lo) / step)), int(min(nt, (i[3] - lo) / step + 1))
                if b > a:
                    pend[a:b] = True
        both = room[:, pend].mean() if pend.any() else float("nan")
        print("%-16s %10.1f %% %20.1f %% %28.1f %%" % (kn, 100 * room.mean(), 100 * pend.mean(), 100 * both))
    # per XCD: workgroups are dealt to the eight XCDs round-robin, so a launch advances at the pace of its fullest XCD
    print("\n## per XCD (workgroup i of a launch goes to XCD i mod 8): VGPRs held per SIMD (mean), room for one more compose wave, SIMDs of the XCD with NO such room at the same time")
    xcc_of_simd = np.repeat(place.xcc_of_cu, 4)
    room_c = (vg + VGPR["compose"] <= SIMD_VGPRS) & (slots < SIMD_SLOTS)
    for x in np.unique(xcc_of_simd):
        m = xcc_of_simd == x
        full_share = 1.0 - room_c[m].mean(axis=0)           # share of the XCD's SIMDs without room, over time
        print("XCD %d: %3d SIMDs  %5.1f VGPRs  room %5.1f %%  | share of its SIMDs without room: mean %4.1f %%, p90 %4.1f %%, time with none free %4.1f %%" % (
            x, m.sum(), vg[m].mean(), 100 * room_c[m].mean(), 100 * full_share.mean(), 100 * np.percentile(full_share, 90), 100 * (full_share >= 0.999).mean()))
    # how fast waves start, by kernel, while the kernel has workgroups waiting
    print("\n## wave starts per microsecond while a kernel's launch is being dispatched (window medians), alone against in the pipeline")
    _, info_a = launches(alone)
    for k, kn in enumerate(KERNELS):
        ra = [i[5] / max(i[3] - i[2], 0.05) for i in info_a if i[0] == k]
        rp = [i[5] / max(i[3] - i[2], 0.05) for i in info if i[0] == k and lo <= i[2] < hi]
        if ra and rp:
            print("%-16s alone %8.0f / us   in the pipeline %8.0f / us" % (kn, np.median(ra), np.median(rp)))
    # does a launch that cannot place its next workgroup hold the OTHERS' workgroups back?  wave starts of each kernel per
    # microsecond of its own dispatch windows, split by whether a compose / background_prep launch is being dispatched too
    def pending_mask(k):
        m = np.zeros(nt, bool)
        for i in info:
            if i[0] == k:
                a_, b_ = int(max(0, (i[2] - lo) / step)), int(min(nt, (i[3] - lo) / step + 1))
                if b_ > a_:
                    m[a_:b_] = True
        return m
    pend = [pending_mask(k) for k in range(5)]
    starts = np.zeros((5, nt))
    for k in range(5):
        m = (pipe["kernel"] == k) & (pipe["t0"] >= lo) & (pipe["t0"] < hi)
        np.add.at(starts[k], np.clip(((pipe["t0"][m] - lo) / step).astype(np.int64), 0, nt - 1), 1)
    print("\n## wave starts per microsecond of a kernel's own dispatch windows, by what else is being dispatched at that moment")
    print("%-16s %26s %26s %26s" % ("kernel", "compose waiting too", "compose not waiting", "neither compose nor prep"))
    for k, kn in enumerate(KERNELS[:4]):
        row = []
        for other in (pend[4], ~pend[4], ~pend[4] & (~pend[3] if k != 3 else True)):
            m = pend[k] & other
            row.append("%8.0f / us (%4.1f %% of it)" % (starts[k][m].sum() / max(m.sum() * step, 1e-9), 100 * m.sum() / max(pend[k].sum(), 1)))
        print("%-16s %26s %26s %26s" % (kn, *row))
    tot = starts.sum(axis=0)
    print("all kernels together: %.0f wave starts / us over the window; %.0f while a compose launch is being dispatched, %.0f otherwise" % (
        tot.sum() / (nt * step), tot[pend[4]].sum() / max(pend[4].sum() * step, 1e-9), tot[~pend[4]].sum() / max((~pend[4]).sum() * step, 1e-9)))
    print("\n## what keeps a SIMD from taking one more wave, share of SIMD-time in the window (several can hold at once)")
    print("%-16s %12s %12s %12s" % ("kernel", "registers", "wave slots", "LDS"))
    for k, kn in enumerate(KERNELS):
        print("%-16s %10.1f %% %10.1f %% %10.1f %%" % (kn, 100 * (vg + VGPR[kn] > SIMD_VGPRS).mean(), 100 * (slots >= SIMD_SLOTS).mean(), 100 * (lds_s + LDS[kn] > CU_LDS).mean()))


if __name__ == "__main__":
    main()
