#!/usr/bin/env python3
"""Generate tests/golden/agg_goldens.npz from matplotlib's COMPILED Anti-Grain
Geometry (backends/_backend_agg*.so, _path*.so, _image*.so).

The reference renders its masks and textures with AGG 2.4, which is not vendored
in the reference checkout (cmake/Dependencies.cmake:4-19) and is absent from this
image; matplotlib ships a compiled AGG whose scanline rasteriser, curve3
subdivision and span interpolator are the same algorithms.  These vectors pin the
oracle's (and through it the HIP path's) restatement of those three pieces.

  G3  raw AA coverage (alpha channel) of polygons fully inside a 128x96 canvas:
      star polygons (incl. self-intersecting), 100-gon ellipses, thin slivers.
  G4  conv_curve/curve3_div point lists for random quadratic Beziers.
  G5  span_interpolator_linear + dda2 sample positions, observed through
      _image.resample(NEAREST) on an index image, for random affines.

Run in the build container only (needs matplotlib):  python tests/golden/gen_agg_goldens.py
"""
import os

import numpy as np
import matplotlib
matplotlib.use("Agg")
from matplotlib import rcParams
from matplotlib.backends.backend_agg import RendererAgg
from matplotlib.path import Path
from matplotlib.transforms import Affine2D
from matplotlib import _image

W, H = 128, 96
rcParams["path.simplify"] = False


def agg_coverage(verts, codes=None):
    r = RendererAgg(W, H, 72)
    gc = r.new_gc()
    gc.set_antialiased(True)
    gc.set_linewidth(0)
    gc.set_snap(False)
    path = Path(np.asarray(verts, float), codes)
    # cancel matplotlib's own y flip so that y points down like the reference
    r.draw_path(gc, path, Affine2D().scale(1, -1).translate(0, H), rgbFace=(1, 1, 1, 1))
    return np.asarray(r.buffer_rgba())[:, :, 3].copy()


def closed(verts):
    v = np.vstack([verts, verts[:1]])
    codes = [Path.MOVETO] + [Path.LINETO] * (len(verts) - 1) + [Path.CLOSEPOLY]
    return v, codes


def main():
    rng = np.random.RandomState(20240607)
    polys, covs = [], []
    # sanity value from SURVEY Appendix E.2
    v, c = closed(np.array([[0.5, 0.5], [2.25, 0.5], [2.25, 1.5], [0.5, 1.5]]))
    cov = agg_coverage(v, c)
    assert list(cov[0, :3]) == [64, 128, 32] and list(cov[1, :3]) == [64, 128, 32], cov[:2, :4]

    def add(p):
        v, c = closed(p)
        polys.append(np.asarray(p, float))
        covs.append(agg_coverage(v, c))

    for i in range(120):  # star polygons, 3..20 spokes, jittered like the reference's sampler
        n = rng.randint(3, 21)
        phi = (np.arange(n) * 360.0 / n + rng.uniform(-10, 10, n)) * np.pi / 180
        r = rng.uniform(4, 40, n)
        cx, cy = rng.uniform(45, W - 45), rng.uniform(42, H - 42)
        sx, sy = rng.uniform(0.5, 1.0, 2)
        add(np.stack([cx + sx * r * np.cos(phi), cy + sy * r * np.sin(phi)], 1))
    for i in range(40):  # self-intersecting (pentagram-like): non-zero winding matters
        n = rng.choice([5, 7, 9, 11])
        k = rng.choice([2, 3])
        phi = (np.arange(n) * k * 2 * np.pi / n) + rng.uniform(0, 6.28)
        r = rng.uniform(10, 40)
        cx, cy = rng.uniform(45, W - 45), rng.uniform(42, H - 42)
        add(np.stack([cx + r * np.cos(phi), cy + r * np.sin(phi)], 1))
    for i in range(60):  # 100-gon ellipses (agg::ellipse with 100 steps), rotated
        rx, ry = rng.uniform(1.0, 40, 2)
        a = rng.uniform(-np.pi, np.pi)
        ang = np.arange(100) / 100.0 * 2.0 * np.pi
        x, y = np.cos(ang) * rx, np.sin(ang) * ry
        cx, cy = rng.uniform(45, W - 45), rng.uniform(42, H - 42)
        add(np.stack([cx + x * np.cos(a) - y * np.sin(a), cy + x * np.sin(a) + y * np.cos(a)], 1))
    for i in range(40):  # slivers / needles (thin objects, DataGenerator.cpp:2462, 2496)
        n = rng.randint(3, 12)
        phi = (np.arange(n) * 360.0 / n + rng.uniform(-10, 10, n)) * np.pi / 180
        r = rng.uniform(10, 40, n)
        a = rng.uniform(-np.pi, np.pi)
        x, y = 0.05 * r * np.cos(phi), r * np.sin(phi)
        cx, cy = rng.uniform(45, W - 45), rng.uniform(42, H - 42)
        add(np.stack([cx + x * np.cos(a) - y * np.sin(a), cy + x * np.sin(a) + y * np.cos(a)], 1))
    for i in range(20):  # axis-aligned boxes incl. integer and half-integer edges (mode 1)
        x0, x1 = sorted(rng.choice(np.arange(8, 240), 2, replace=False) / 2.0)
        y0, y1 = sorted(rng.choice(np.arange(8, 180), 2, replace=False) / 2.0)
        add(np.array([[x1, y0], [x1, y1], [x0, y1], [x0, y0]]))

    # G4: curve3 flattening through conv_curve (Path.cleaned(curves=False))
    curves, curve_pts = [], []
    for i in range(200):
        p = rng.uniform(-150, 150, (3, 2)) if i % 4 else rng.uniform(-3, 3, (3, 2))
        if i % 17 == 0:  # collinear control point
            p[1] = p[0] + (p[2] - p[0]) * rng.uniform(-0.5, 1.5)
        path = Path(np.array([p[0], p[1], p[2]]), [Path.MOVETO, Path.CURVE3, Path.CURVE3])
        cl = path.cleaned(simplify=False, curves=False)
        pts = cl.vertices[cl.codes != Path.STOP]
        curves.append(p)
        curve_pts.append(pts)
    # G4b: coverage of polygons with curve3 segments
    cpolys, ccodes, ccovs = [], [], []
    for i in range(60):
        n = rng.randint(4, 14)
        phi = (np.arange(n) * 360.0 / n + rng.uniform(-10, 10, n)) * np.pi / 180
        r = rng.uniform(6, 40, n)
        cx, cy = rng.uniform(45, W - 45), rng.uniform(42, H - 42)
        pts = np.stack([cx + r * np.cos(phi), cy + r * np.sin(phi)], 1)
        types = [0]
        j = 1
        while j < n:  # reference's curve trigger logic, DataGenerator.cpp:2307-2315
            if j < n - 1 and rng.uniform() < 0.33:
                types += [3, 0]
                j += 2
            else:
                types += [1]
                j += 1
        codes = [Path.MOVETO]
        for t in types[1:]:
            codes.append(Path.LINETO if t == 1 else Path.CURVE3)
        v = np.vstack([pts, pts[:1]])
        cov = agg_coverage(v, codes + [Path.CLOSEPOLY])
        cpolys.append(pts)
        ccodes.append(np.array(types))
        ccovs.append(cov)

    # G5: interpolator sample positions via NEAREST resampling of an index image
    SW, SH = 64, 48
    src = np.arange(SW * SH, dtype=np.float64).reshape(SH, SW)
    mats, pos = [], []
    for i in range(40):
        a = rng.uniform(-0.5, 0.5)
        s = rng.uniform(0.8, 1.25)
        tx, ty = rng.uniform(-6, 6, 2)
        # forward transform input->output; keep the output inside the transformed input
        t = Affine2D().translate(-SW / 2, -SH / 2).rotate(a).scale(s * 1.9).translate(SW / 2 + tx, SH / 2 + ty)
        out = np.full((24, 32), -1.0)
        _image.resample(src, out, t, _image.NEAREST, False, 1.0, False, 1.0)
        if (out < 0).any():
            continue
        # partially covered border pixels are alpha-blended: keep only affines whose
        # output rectangle is fully covered (a constant image comes back unchanged)
        chk = np.zeros((24, 32))
        _image.resample(np.full((SH, SW), 1000.0), chk, t, _image.NEAREST, False, 1.0, False, 1.0)
        if not (chk == 1000.0).all():
            continue
        mats.append(t.get_matrix())
        pos.append(out.astype(np.int32))

    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "agg_goldens.npz")
    np.savez_compressed(
        out_path,
        canvas=np.array([W, H]),
        poly_len=np.array([len(p) for p in polys]), poly_xy=np.vstack(polys), poly_cov=np.stack(covs),
        curve_ctrl=np.stack(curves), curve_len=np.array([len(p) for p in curve_pts]), curve_pts=np.vstack(curve_pts),
        cpoly_len=np.array([len(p) for p in cpolys]), cpoly_xy=np.vstack(cpolys), cpoly_types=np.concatenate(ccodes),
        cpoly_cov=np.stack(ccovs),
        dda_src=np.array([SW, SH]), dda_mats=np.stack(mats), dda_pos=np.stack(pos),
    )
    print("wrote", out_path, os.path.getsize(out_path), "bytes;", len(polys), "polys,", len(curves), "curves,",
          len(cpolys), "curve polys,", len(mats), "affines")


if __name__ == "__main__":
    main()
