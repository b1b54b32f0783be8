#!/usr/bin/env python3
"""End-to-end fixture of the ORACLE itself: sha256 of image0 / image1 / flow for tiny scenes of
several data modes, rendered by oracle/ with the reference-stream sampler and a deterministic texture pool.
It pins the restatement against drift from round to round (the GPU tests compare the HIP path with the live
oracle; this file makes sure the oracle of today is the oracle of the day the fixture was written).

    python tests/golden/gen_e2e_goldens.py        # rewrites tests/golden/e2e_hashes.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as oracle  # noqa: E402

W, H, B = 64, 48, 2
SCENES = [(mode, aa) for mode in (1, 3, 5, 7, 9, 13) for aa in (1, 0)]


def pool(n=3, w=160, h=120, seed=77):
    """A smooth deterministic BGR pool (integer arithmetic only)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    out = np.zeros((n, 3, h, w), np.uint8)
    for i in range(n):
        for c in range(3):
            out[i, c] = ((x * (3 + i + c) + y * (5 + 2 * c + i) + ((x * y) >> (3 + c)) + seed * (i + 1) * (c + 2)) % 251).astype(np.uint8)
    return out


def scene(mode, aa):
    prm = oracle.default_params(W, H, mode, use_aa=aa)
    s = oracle.Sampler(mode, W, H)
    tasks, bps, n = s.next(B)
    crops = oracle.warp_crops(W, H, seed=5)[:4] if mode == 9 else None
    return oracle.render(prm, tasks, B, bps, n, pool(), warp_crops=crops)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {"size": [W, H], "batch": B, "scenes": []}
    for mode, aa in SCENES:
        i0, i1, fl = scene(mode, aa)
        out["scenes"].append({"mode": mode, "use_antialiasing": aa, "image0": digest(i0), "image1": digest(i1),
                              "flow": digest(fl), "mean_image1": float(i1.mean())})
    with open(os.path.join(HERE, "e2e_hashes.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d scenes" % len(out["scenes"]))


if __name__ == "__main__":
    main()
