"""ctypes binding of the parity oracle (oracle/libofdg_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_SEG = 20


class Blueprint(C.Structure):
    """Mirror of ofdg_blueprint (include/ofdg.h)."""
    _fields_ = [
        ("obj_id", C.c_int32), ("obj_type", C.c_int32),
        ("init_rot", C.c_float), ("init_scale", C.c_float),
        ("init_trans_x", C.c_float), ("init_trans_y", C.c_float),
        ("rot", C.c_float), ("scale", C.c_float),
        ("trans_x", C.c_float), ("trans_y", C.c_float),
        ("tex_id", C.c_int32), ("tex_rot", C.c_float), ("tex_scale", C.c_float),
        ("tex_shift_x", C.c_int32), ("tex_shift_y", C.c_int32),
        ("ellipse_scale_x", C.c_float), ("ellipse_scale_y", C.c_float),
        ("n_segments", C.c_int32),
        ("segment_type", C.c_int32 * MAX_SEG),
        ("segment_x", C.c_float * MAX_SEG),
        ("segment_y", C.c_float * MAX_SEG),
        ("first_component", C.c_int32), ("n_components", C.c_int32),
        ("is_additive_component", C.c_int32),
        ("do_warpfield_deformation", C.c_int32),
    ]


class Task(C.Structure):
    _fields_ = [("background", C.c_int32), ("first_object", C.c_int32),
                ("n_objects", C.c_int32), ("reserved", C.c_int32)]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32),
        ("use_antialiasing", C.c_int32), ("batch_size", C.c_int32), ("prefetch", C.c_int32),
        ("first_level_threads", C.c_int32), ("second_level_threads", C.c_int32),
        ("num_objects", C.c_int32), ("sampler", C.c_int32), ("seed", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("device", C.c_int32),
        ("max_shapes_per_sample", C.c_int32), ("background_prep", C.c_int32), ("reserved", C.c_int32 * 8),
    ]


def default_params(width=512, height=384, mode=7, use_aa=1, batch=1, num_objects=0):
    p = Params()
    p.width, p.height, p.mode, p.use_antialiasing = width, height, mode, use_aa
    p.batch_size, p.prefetch = batch, 1
    p.first_level_threads, p.second_level_threads = 16, 1
    p.num_objects = num_objects
    p.world_size = 1
    return p


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "libofdg_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ofdg_oracle_sampler_create.restype = C.c_void_p
        L.ofdg_oracle_sampler_create.argtypes = [C.c_int] * 4
        L.ofdg_oracle_sampler_destroy.argtypes = [C.c_void_p]
        L.ofdg_oracle_sampler_next.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ofdg_oracle_rng_draws.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.ofdg_oracle_rasterize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ofdg_oracle_curve3.argtypes = [C.c_double] * 6 + [C.c_void_p, C.c_int]
        L.ofdg_oracle_outline.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ofdg_oracle_dda_row.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ofdg_oracle_transformed_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ofdg_oracle_tables.argtypes = [C.c_void_p] * 3
        L.ofdg_oracle_draw_image_value.restype = C.c_uint8
        L.ofdg_oracle_draw_image_value.argtypes = [C.c_uint8] * 3
        L.ofdg_oracle_flowfield.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ofdg_oracle_displacers.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_int]
        L.ofdg_oracle_warp_crops.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p, C.c_int]
        L.ofdg_oracle_render.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ofdg_oracle_shape_masks.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ofdg_oracle_set_detmath.argtypes = [C.c_int]
        L.ofdg_oracle_set_lean.argtypes = [C.c_int]
        L.ofdg_oracle_det_sincos.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ofdg_oracle_det_expf.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class detmath:
    """Context manager: the oracle builds its affines / Gaussian supports with include/ofdg_detmath.h (what the
    device counter-sampler path is defined with) instead of libm (the reference's arithmetic, default)."""

    def __enter__(self):
        self.old = lib().ofdg_oracle_set_detmath(1)

    def __exit__(self, *a):
        lib().ofdg_oracle_set_detmath(self.old)


class lean:
    """Context manager: the oracle's "lean" CPU-baseline cost model (one rasterisation per frame, work restricted
    to the outlines' boxes; same output).  Default and parity tests: the reference's work pattern ("faithful")."""

    def __enter__(self):
        self.old = lib().ofdg_oracle_set_lean(1)

    def __exit__(self, *a):
        lib().ofdg_oracle_set_lean(self.old)


def det_sincos(a):
    a = np.ascontiguousarray(a, np.float64)
    s, c = np.zeros_like(a), np.zeros_like(a)
    lib().ofdg_oracle_det_sincos(_ptr(a), len(a), _ptr(s), _ptr(c))
    return s, c


def det_expf(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().ofdg_oracle_det_expf(_ptr(x), len(x), _ptr(y))
    return y


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def rng_draws(kind, seed, a=0.0, b=0.0, c=0.0, n=16):
    out = np.zeros(n, np.float64)
    lib().ofdg_oracle_rng_draws(kind, seed, a, b, c, n, _ptr(out))
    return out


class Sampler:
    """oracle::Sampler -- the reference's ObjectParametersGenerator driven like load_batch."""

    def __init__(self, mode, W=512, H=384, num_objects=0):
        self.h = lib().ofdg_oracle_sampler_create(mode, W, H, num_objects)
        if not self.h:
            raise ValueError("BAD MODE")

    def next(self, n_tasks, cap=None):
        cap = cap or n_tasks * 256
        tasks = (Task * n_tasks)()
        bps = (Blueprint * cap)()
        n = lib().ofdg_oracle_sampler_next(self.h, n_tasks, C.cast(tasks, C.c_void_p), C.cast(bps, C.c_void_p), cap)
        if n < 0:
            raise RuntimeError("blueprint capacity: need %d" % -n)
        return tasks, bps, n

    def __del__(self):
        if getattr(self, "h", None):
            lib().ofdg_oracle_sampler_destroy(self.h)
            self.h = None


def rasterize(xy, w, h):
    xy = np.ascontiguousarray(xy, np.float64)
    cov = np.zeros((h, w), np.uint8)
    rc = lib().ofdg_oracle_rasterize(_ptr(xy), len(xy), w, h, _ptr(cov))
    assert rc == 0
    return cov


def curve3(p1, p2, p3, cap=4096):
    out = np.zeros((cap, 2), np.float64)
    n = lib().ofdg_oracle_curve3(p1[0], p1[1], p2[0], p2[1], p3[0], p3[1], _ptr(out), cap)
    assert n >= 0
    return out[:n].copy()


def outline(bp, m, cap=8192):
    m = np.ascontiguousarray(m, np.float64)
    out = np.zeros((cap, 2), np.float64)
    n = lib().ofdg_oracle_outline(C.byref(bp), _ptr(m), _ptr(out), cap)
    assert n >= 0
    return out[:n].copy()


def dda_row(inv, y, length):
    inv = np.ascontiguousarray(inv, np.float64)
    out = np.zeros((length, 2), np.int32)
    lib().ofdg_oracle_dda_row(_ptr(inv), y, length, _ptr(out))
    return out


def transformed_texture(img, m):
    img = np.ascontiguousarray(img, np.uint8)
    _, th, tw = img.shape
    m = np.ascontiguousarray(m, np.float64)
    out = np.zeros_like(img)
    lib().ofdg_oracle_transformed_texture(_ptr(img), tw, th, _ptr(m), _ptr(out))
    return out


def tables():
    add = np.zeros((256, 256), np.uint8)
    sub = np.zeros((256, 256), np.uint8)
    aa = np.zeros(256, np.uint8)
    lib().ofdg_oracle_tables(_ptr(add), _ptr(sub), _ptr(aa))
    return add, sub, aa


def flowfield(size, displacers, iters=17):
    d = np.zeros((len(displacers), 11), np.float64)
    d[:, :9] = np.asarray(displacers, np.float64)
    flow = np.zeros((2, size, size), np.float32)
    iflow = np.zeros((2, size, size), np.float32)
    lib().ofdg_oracle_flowfield(size, _ptr(d), len(d), iters, _ptr(flow), _ptr(iflow))
    return flow, iflow


def displacers(W, H, seed):
    out = np.zeros((1024, 9), np.float64)
    n = lib().ofdg_oracle_displacers(W, H, seed, _ptr(out), 1024)
    assert n >= 0
    return out[:n].copy()


def warp_crops(W, H, seed, iters=17, cap=64):
    """All crops of one seeded big field: float32 [n, 4, H+1, W+1] (flow x, y, iflow x, y)."""
    out = np.zeros((cap, 4, H + 1, W + 1), np.float32)
    n = lib().ofdg_oracle_warp_crops(W, H, seed, iters, _ptr(out), cap)
    assert n >= 0, -n
    return out[:n].copy()


def render(params, tasks, n_tasks, bps, n_bps, pool, warp_crops=None, reuse=2, n_threads=1):
    """pool: uint8 [n, 3, h, w] planar BGR. Returns (img0, img1, flow) float32."""
    pool = np.ascontiguousarray(pool, np.uint8)
    pn, _, ph, pw = pool.shape
    W, H = params.width, params.height
    img0 = np.zeros((n_tasks, 3, H, W), np.float32)
    img1 = np.zeros((n_tasks, 3, H, W), np.float32)
    flow = np.zeros((n_tasks, 2, H, W), np.float32)
    if warp_crops is not None:
        warp_crops = np.ascontiguousarray(warp_crops, np.float32)
        wc, ncrops = _ptr(warp_crops), len(warp_crops)
    else:
        wc, ncrops = None, 0
    rc = lib().ofdg_oracle_render(C.byref(params), C.cast(tasks, C.c_void_p), n_tasks, C.cast(bps, C.c_void_p), n_bps, _ptr(pool), pn, pw, ph,
                                  wc, ncrops, reuse, _ptr(img0), _ptr(img1), _ptr(flow), n_threads)
    if rc != 0:
        raise RuntimeError("oracle render failed: %d" % rc)
    return img0, img1, flow


def shape_masks(params, task, bps, pool, max_shapes=64):
    pool = np.ascontiguousarray(pool, np.uint8)
    pn, _, ph, pw = pool.shape
    W, H = params.width, params.height
    masks = np.zeros((max_shapes, 4, H, W), np.uint8)
    n = lib().ofdg_oracle_shape_masks(C.byref(params), C.byref(task), C.cast(bps, C.c_void_p), _ptr(pool), pn, pw, ph, _ptr(masks), max_shapes)
    if n < 0:
        raise RuntimeError("oracle shape_masks failed: %d" % n)
    return masks[:min(n, max_shapes)]
