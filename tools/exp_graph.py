#!/usr/bin/env python3
"""Developer experiment: a step as ONE submission.  The four (five with the background preparation) kernels
of a step as the product submits them - in-order launches on one of four chains - against one hipGraphLaunch of the same
kernels captured with frozen parameters (ofdg_debug_graph_capture: replays render the same batches again).  Arms:
  launches, new samples     gen.forward(): what bench.py times
  launches, same batches    gen.forward_counter(first index of (chain, buffer set)): the graph arm's work, submitted by launches
  graphs, same batches      one hipGraphLaunch per step
Host time to issue a step and wall time per step, for the driver's run length (20 steps from an idle device, median of REPS)
and a long run.  Needs a library built with tools/patches/r04_graph_capture.patch (the capture entry points are an experiment,
not part of the product's C-ABI).  Usage on the GPU box: python3 tools/exp_graph.py [reps]"""
import ctypes as C, importlib, os, statistics, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
W, H = 512, 384
L = ofdg.lib()
L.ofdg_debug_graph_capture.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
L.ofdg_debug_graph_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.ofdg_debug_graph_destroy.argtypes = [C.c_void_p, C.c_void_p]


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t) / n * 1e6, (time.perf_counter() - t) / n * 1e6


for name, B, nobj, prep in (("config 2, background_prep 0", 32, 16, 0), ("config 2, background_prep 1 (the headline)", 32, 16, 1),
                            ("batch 1, 1 object, background_prep 1", 1, 1, 1)):
    g = ofdg.Generator(ofdg.default_params(width=W, height=H, mode=5, num_objects=nobj, batch_size=B, sampler=1, seed=20261003, background_prep=prep))
    g.pool_synthetic(1000, 1024, 768, 2024)
    nch = g.num_chains()
    nbuf = 2 * nch
    outs = [ofdg.alloc_outputs(B, H, W) for _ in range(nbuf)]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    for i in range(3 * nbuf):
        g.forward(*outs[i % nbuf], g.next_stream())
    g.synchronize()
    execs = []
    for i in range(nbuf):
        ge = C.c_void_p()
        rc = L.ofdg_debug_graph_capture(g.h, i % nch, i * B, B, ptr(outs[i][0]), ptr(outs[i][1]), ptr(outs[i][2]), C.byref(ge))
        assert rc == 0, (rc, L.ofdg_last_error(g.h))
        execs.append(ge)
    arms = {
        "launches, new samples": lambda i: g.forward(*outs[i % nbuf], g.next_stream()),
        "launches, same batches": lambda i: g.forward_counter((i % nbuf) * B, B, *outs[i % nbuf], g.next_stream()),
        "graphs, same batches": lambda i: L.ofdg_debug_graph_launch(g.h, execs[i % nbuf], (i % nbuf) % nch),
    }
    print(name, "(%d chains)" % nch, flush=True)
    for arm, fn in arms.items():
        timed(fn, 4 * nbuf)
        short = [timed(fn, 20) for _ in range(REPS)]
        long_ = [timed(fn, 2000) for _ in range(2)]
        print("  %-24s 20 steps: issue %5.1f us/step, done %6.1f us/step (min %6.1f) | 2000 steps: issue %5.1f, done %6.1f us/step" % (
            arm, statistics.median(s[0] for s in short), statistics.median(s[1] for s in short), min(s[1] for s in short),
            min(l[0] for l in long_), min(l[1] for l in long_)), flush=True)
    g.synchronize()
    # the replays rendered what the launches render: same frames for the same batch
    ref = ofdg.alloc_outputs(B, H, W)
    L.ofdg_debug_graph_launch(g.h, execs[1], 1 % nch)
    g.synchronize()
    g.forward_counter(1 * B, B, *ref)
    g.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, outs[1])), "a replayed graph renders something else than the launches"
    for ge in execs:
        L.ofdg_debug_graph_destroy(g.h, ge)
    g.close()
