"""Python binding of libofdg.so (the C-ABI in include/ofdg.h) for tests, bench.py
and PyTorch users.  PyTorch is only plumbing here (device buffers, streams); all
rendering happens in the HIP kernels behind the C-ABI.  There is no CPU fallback:
if the shared library or a HIP device is missing, calls raise.

Import with importlib (the directory name is not a Python identifier):
    ofdg = importlib.import_module("optical-flow-2d-data-generation_amd")
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFDG_LIB") or os.path.join(HERE, "lib", "libofdg.so")  # OFDG_LIB: A/B-test another build
MAX_SEG = 20

OK, EBADMODE, ETEXTURES, EOBJTYPE, EHIP, ECAPACITY, EINVAL, ESTARTUP = 0, -1, -2, -3, -4, -5, -6, -7
OBJ_ELLIPSE, OBJ_POLYGON, OBJ_COMPOSITE = 1, 2, 3
STREAM_OWN = (1 << 64) - 1  # OFDG_STREAM_OWN: as a call's `stream`, the internal stream that call works on (= next_stream())
SEG_DUMMY, SEG_LINE, SEG_CURVE3 = 0, 1, 3


class Blueprint(C.Structure):
    """ofdg_blueprint == DataGenerator::ObjectBlueprint (DataGenerator.h:388-421)."""
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
    """ofdg_task == DataGenerator::TaskBucket (DataGenerator.h:423-437)."""
    _fields_ = [("background", C.c_int32), ("first_object", C.c_int32),
                ("n_objects", C.c_int32), ("reserved", C.c_int32)]


class Params(C.Structure):
    """ofdg_params: data_param + data_generation_param (+ extension keys)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32),
        ("use_antialiasing", C.c_int32), ("batch_size", C.c_int32), ("prefetch", C.c_int32),
        ("first_level_threads", C.c_int32), ("second_level_threads", C.c_int32),
        ("num_objects", C.c_int32), ("sampler", C.c_int32), ("seed", C.c_int32),
        ("rank", C.c_int32), ("world_size", C.c_int32), ("device", C.c_int32),
        ("max_shapes_per_sample", C.c_int32), ("background_prep", C.c_int32),
        ("chains", C.c_int32), ("lookahead", C.c_int32), ("serial", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


class Setup(C.Structure):
    """ofdg_setup: what rank 0 broadcasts at start-up (stream + pool description)."""
    _fields_ = [(k, C.c_int32) for k in ("seed", "mode", "width", "height", "num_objects", "use_antialiasing", "batch_size",
                                         "sampler", "background_prep", "n_tex", "pool_kind", "pool_w", "pool_h")] + \
               [("pool_seed", C.c_uint32), ("n_table", C.c_int32), ("status", C.c_int32), ("max_shapes_per_sample", C.c_int32),
                ("reserved", C.c_int32)]


class TexEntry(C.Structure):
    """ofdg_tex_entry: one texture of the index table."""
    _fields_ = [("offset", C.c_uint64), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("reserved", C.c_uint32)]


UNIQUE_ID_BYTES = 128
POOL_SYNTHETIC, POOL_UNIFORM, POOL_MIXED = 0, 1, 2


class OfdgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ofdg error %d: %s" % (code, msg))
        self.code = code


_lib = None

# every symbol include/ofdg.h declares
EXPORTS = [
    "ofdg_default_params", "ofdg_create", "ofdg_destroy", "ofdg_last_error", "ofdg_ctx_info",
    "ofdg_host_bg_prep", "ofdg_ctx_params", "ofdg_pool_alloc_mixed", "ofdg_pool_upload_mixed", "ofdg_pool_synthetic", "ofdg_pool_alloc", "ofdg_pool_upload", "ofdg_pool_download", "ofdg_pool_info", "ofdg_pool_device",
    "ofdg_sample", "ofdg_render", "ofdg_render_resident", "ofdg_upload_slot", "ofdg_render_slot", "ofdg_forward", "ofdg_shard_first_index", "ofdg_synchronize", "ofdg_stream", "ofdg_get_step", "ofdg_set_step",
    "ofdg_debug_rasterize", "ofdg_debug_rasterize_path", "ofdg_debug_dda_rows", "ofdg_debug_coverage", "ofdg_debug_num_shapes", "ofdg_debug_item_count", "ofdg_debug_bgprep_tiles", "ofdg_debug_bgprep_paths", "ofdg_debug_tables", "ofdg_debug_detmath",
    "ofdg_set_profiling", "ofdg_kernel_ms",
    "ofdg_forward_counter", "ofdg_sample_counter", "ofdg_warp_generate", "ofdg_warp_upload", "ofdg_warp_info", "ofdg_warp_download", "ofdg_host_displacers",
    "ofdg_host_sampler_create", "ofdg_host_sampler_next", "ofdg_host_sampler_destroy", "ofdg_host_realize",
    "ofdg_parse_prototxt", "ofdg_host_last_error", "ofdg_host_decode_image", "ofdg_layer_create", "ofdg_layer_forward", "ofdg_layer_destroy",
    "ofdg_layer_in_flight", "ofdg_poll_errors", "ofdg_poll_errors_of", "ofdg_last_ticket", "ofdg_num_chains",
    "ofdg_comm_unique_id", "ofdg_comm_init", "ofdg_comm_adopt", "ofdg_comm_destroy", "ofdg_comm_rank", "ofdg_comm_world_size",
    "ofdg_comm_last_error", "ofdg_comm_bcast_setup", "ofdg_comm_bcast_abort", "ofdg_comm_nccl_count", "ofdg_comm_bcast_pool", "ofdg_comm_agree", "ofdg_setup_of", "ofdg_setup_params",
    "ofdg_setup_alloc_pool", "ofdg_pool_device_mixed", "ofdg_pool_device_image", "ofdg_layer_create_dist",
]


def build(verbose=False):
    """Compile libofdg.so for gfx950 with hipcc (in-tree, optical-flow-2d-data-generation_amd/lib)."""
    cmd = ["make", "-C", HERE] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)


def lib():
    """Load libofdg.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfdgError(EHIP, "libofdg.so is not built (run __graft_entry__.build()); "
                                  "the HIP extension is the only render path")
        # PyTorch wheels bundle their own libamdhip64.so.7; a process must hold ONE HIP
        # runtime, so when torch is installed let it load first and share its copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.ofdg_default_params.argtypes = [C.POINTER(Params)]
        L.ofdg_default_params.restype = None
        L.ofdg_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.ofdg_destroy.argtypes = [vp]
        L.ofdg_destroy.restype = None
        L.ofdg_last_error.argtypes = [vp]
        L.ofdg_last_error.restype = C.c_char_p
        L.ofdg_ctx_info.argtypes = [vp]
        L.ofdg_ctx_info.restype = C.c_char_p
        L.ofdg_pool_synthetic.argtypes = [vp, i32, i32, i32, C.c_uint32]
        L.ofdg_pool_alloc.argtypes = [vp, i32, i32, i32]
        L.ofdg_pool_upload.argtypes = [vp, i32, vp, i32, i32]
        L.ofdg_pool_download.argtypes = [vp, i32, vp]
        L.ofdg_pool_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_ulonglong), i32]
        L.ofdg_pool_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.ofdg_sample.argtypes = [vp, i32, vp, vp, i32, C.POINTER(i32)]
        L.ofdg_render.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp]
        L.ofdg_render_resident.argtypes = [vp, vp, vp, vp, vp]
        L.ofdg_upload_slot.argtypes = [vp, i32, vp, i32, vp, i32, vp]
        L.ofdg_render_slot.argtypes = [vp, i32, vp, vp, vp, vp]
        L.ofdg_forward.argtypes = [vp, vp, vp, vp, vp]
        L.ofdg_synchronize.argtypes = [vp, vp]
        L.ofdg_stream.argtypes = [vp]
        L.ofdg_get_step.argtypes = [vp]
        L.ofdg_get_step.restype = C.c_longlong
        L.ofdg_set_step.argtypes = [vp, C.c_longlong]
        L.ofdg_stream.restype = vp
        L.ofdg_debug_rasterize.argtypes = [vp, vp, i32, vp]
        L.ofdg_debug_rasterize_path.argtypes = [vp, vp, vp, i32, vp]
        L.ofdg_debug_dda_rows.argtypes = [vp, vp, i32, i32, vp]
        L.ofdg_debug_coverage.argtypes = [vp, i32, i32, i32, vp]
        L.ofdg_debug_num_shapes.argtypes = [vp, i32]
        L.ofdg_debug_tables.argtypes = [vp, vp, vp, vp, vp, i32]
        L.ofdg_debug_bgprep_tiles.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.ofdg_debug_bgprep_paths.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.ofdg_debug_detmath.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp]
        L.ofdg_set_profiling.argtypes = [vp, i32]
        L.ofdg_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float)]
        L.ofdg_forward_counter.argtypes = [vp, C.c_longlong, i32, vp, vp, vp, vp]
        L.ofdg_sample_counter.argtypes = [vp, C.c_longlong, i32, vp, vp]
        L.ofdg_warp_generate.argtypes = [vp, i32, C.c_uint32]
        L.ofdg_warp_upload.argtypes = [vp, vp, i32]
        L.ofdg_warp_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.ofdg_warp_download.argtypes = [vp, i32, vp]
        L.ofdg_host_displacers.argtypes = [i32, i32, C.c_uint32, vp, i32]
        L.ofdg_host_sampler_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
        L.ofdg_host_sampler_next.argtypes = [vp, i32, vp, vp, i32, C.POINTER(i32)]
        L.ofdg_host_sampler_destroy.argtypes = [vp]
        L.ofdg_host_sampler_destroy.restype = None
        L.ofdg_host_realize.argtypes = [C.POINTER(Params), i32, i32, i32, vp, i32, vp, i32, vp, i32, C.POINTER(i32),
                                        vp, i32, C.POINTER(i32)]
        L.ofdg_parse_prototxt.argtypes = [C.c_char_p, C.POINTER(Params), C.c_char_p, i32, C.POINTER(i32)]
        L.ofdg_host_last_error.restype = C.c_char_p
        L.ofdg_layer_create.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.ofdg_layer_forward.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i32 * 4)]
        L.ofdg_layer_destroy.argtypes = [vp]
        L.ofdg_layer_destroy.restype = None
        L.ofdg_layer_in_flight.argtypes = [vp]
        L.ofdg_poll_errors.argtypes = [vp]
        L.ofdg_poll_errors_of.argtypes = [vp, C.c_longlong]
        L.ofdg_last_ticket.argtypes = [vp]
        L.ofdg_last_ticket.restype = C.c_longlong
        L.ofdg_num_chains.argtypes = [vp]
        L.ofdg_comm_unique_id.argtypes = [vp]
        L.ofdg_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
        L.ofdg_comm_adopt.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
        L.ofdg_comm_destroy.argtypes = [vp]
        L.ofdg_comm_destroy.restype = None
        L.ofdg_comm_rank.argtypes = [vp]
        L.ofdg_comm_world_size.argtypes = [vp]
        L.ofdg_comm_last_error.argtypes = [vp]
        L.ofdg_comm_last_error.restype = C.c_char_p
        L.ofdg_comm_bcast_setup.argtypes = [vp, i32, C.POINTER(Setup), vp, i32]
        L.ofdg_comm_bcast_pool.argtypes = [vp, i32, vp]
        L.ofdg_comm_agree.argtypes = [vp, i32]
        L.ofdg_comm_bcast_abort.argtypes = [vp, i32, i32, i32]
        L.ofdg_comm_nccl_count.argtypes = [vp]
        L.ofdg_setup_of.argtypes = [vp, C.POINTER(Setup), vp, i32]
        L.ofdg_setup_params.argtypes = [C.POINTER(Setup), vp, C.POINTER(Params)]
        L.ofdg_setup_alloc_pool.argtypes = [vp, C.POINTER(Setup), vp]
        L.ofdg_pool_device_image.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_ulonglong)]
        L.ofdg_pool_device_mixed.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_ulonglong), C.POINTER(vp), C.POINTER(C.c_ulonglong)]
        L.ofdg_layer_create_dist.argtypes = [C.c_char_p, vp, C.POINTER(vp)]
        _lib = L
    return _lib


def shard_first_index(step, batch, world_size, rank):
    """First global sample index of step `step` on rank `rank` (the rule ofdg_forward shards the stream by)."""
    L = lib()
    L.ofdg_shard_first_index.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int]
    L.ofdg_shard_first_index.restype = C.c_longlong
    return int(L.ofdg_shard_first_index(step, batch, world_size, rank))


def default_params(**kw):
    p = Params()
    lib().ofdg_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Generator:
    """Thin object wrapper over an ofdg_ctx*."""

    def __init__(self, params=None, **kw):
        self.params = params if params is not None else default_params(**kw)
        h = C.c_void_p()
        rc = lib().ofdg_create(C.byref(self.params), C.byref(h))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_last_error(None).decode())
        self.h = h

    def _check(self, rc):
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_last_error(self.h).decode())

    def info(self):
        """How the context is set up (chains and what decided their number, look-ahead, serial mode)."""
        return lib().ofdg_ctx_info(self.h).decode()

    def close(self):
        if getattr(self, "h", None):
            lib().ofdg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- texture pool --
    def pool_synthetic(self, n, w, h, seed=0):
        self._check(lib().ofdg_pool_synthetic(self.h, n, w, h, seed))

    def pool_alloc(self, n, w, h):
        self._check(lib().ofdg_pool_alloc(self.h, n, w, h))

    def pool_from_setup(self, setup, table=None):
        """Allocate (synthetic: also fill) the pool a broadcast Setup describes (ofdg_setup_alloc_pool)."""
        self._check(lib().ofdg_setup_alloc_pool(self.h, C.byref(setup), table))

    def pool_upload(self, index, bgr_planar):
        import numpy as np
        a = np.ascontiguousarray(bgr_planar, np.uint8)
        _, h, w = a.shape
        self._check(lib().ofdg_pool_upload(self.h, index, a.ctypes.data_as(C.c_void_p), w, h))

    def pool_alloc_mixed(self, n):
        """A pool of n images of different sizes (each reduced at upload to its W x H / 2W x 2H textures)."""
        self._check(lib().ofdg_pool_alloc_mixed(self.h, n))

    def pool_upload_mixed(self, index, bgr_planar):
        import numpy as np
        a = np.ascontiguousarray(bgr_planar, np.uint8)
        _, h, w = a.shape
        self._check(lib().ofdg_pool_upload_mixed(self.h, index, a.ctypes.data_as(C.c_void_p), w, h))

    def pool_from_list(self, list_path):
        """TextureCollection (DataGenerator.cpp:117-149) for any image format Pillow decodes: `list_path` is the
        reference's texture_dbases file, one image path per line (a last line without a trailing newline is
        dropped, DG:124-126).  Images are decoded on the host to 8-bit RGB, stored as planar B, G, R (the
        reference swaps R and B after CImg::load, DG:128-131) and uploaded; images of one size are kept whole,
        a list with different sizes becomes a mixed pool.  Returns the number of images."""
        import numpy as np
        from PIL import Image
        with open(list_path, "r") as f:
            text = f.read()
        paths = [ln for ln in text.split("\n")[:-1] if ln.strip()]
        if not paths:
            raise OfdgError(ETEXTURES, "Could not open texture collection (%s lists no image)" % list_path)
        imgs = []
        for pth in paths:
            try:
                rgb = np.asarray(Image.open(pth).convert("RGB"), np.uint8)
            except Exception as e:
                raise OfdgError(ETEXTURES, "Could not open texture collection (cannot read %s: %s)" % (pth, e))
            imgs.append(np.ascontiguousarray(rgb[:, :, ::-1].transpose(2, 0, 1)))
        if len({im.shape for im in imgs}) == 1:
            _, h, w = imgs[0].shape
            self.pool_alloc(len(imgs), w, h)
            for i, im in enumerate(imgs):
                self.pool_upload(i, im)
        else:
            self.pool_alloc_mixed(len(imgs))
            for i, im in enumerate(imgs):
                self.pool_upload_mixed(i, im)
        return len(imgs)

    def pool_broadcast(self, src=0, group=None, chunk_bytes=1 << 30):
        """Multi-GPU start-up: rank `src` has loaded the pool (pool_from_list / pool_upload / pool_synthetic), every
        other rank has allocated one of the same shape (pool_alloc) - one torch.distributed broadcast (RCCL over
        xGMI with the nccl backend) fills the replicas.  The reference has no counterpart (one process, one pool)."""
        import torch
        import torch.distributed as dist
        ptr, nbytes = C.c_void_p(), C.c_ulonglong()
        me = dist.get_rank(group)
        self._check(lib().ofdg_pool_device(self.h, C.byref(ptr), C.byref(nbytes), 0 if me == src else 1))

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (int(nbytes.value),), "typestr": "|u1", "data": (int(ptr.value), False), "version": 2}
        t = torch.as_tensor(h, device="cuda")
        for off in range(0, t.numel(), chunk_bytes):
            dist.broadcast(t[off:off + chunk_bytes], src=src, group=group)
        torch.cuda.synchronize()

    def pool_download(self, index):
        import numpy as np
        n, w, h = C.c_int(), C.c_int(), C.c_int()
        lib().ofdg_pool_info(self.h, C.byref(n), C.byref(w), C.byref(h))
        a = np.zeros((3, h.value, w.value), np.uint8)
        self._check(lib().ofdg_pool_download(self.h, index, a.ctypes.data_as(C.c_void_p)))
        return a

    def pool_info(self):
        """(n, w, h) of the resident texture pool."""
        n, w, h = C.c_int32(), C.c_int32(), C.c_int32()
        lib().ofdg_pool_info(self.h, C.byref(n), C.byref(w), C.byref(h))
        return n.value, w.value, h.value

    def pool_download_all(self):
        import numpy as np
        n, w, h = C.c_int(), C.c_int(), C.c_int()
        lib().ofdg_pool_info(self.h, C.byref(n), C.byref(w), C.byref(h))
        return np.stack([self.pool_download(i) for i in range(n.value)])

    # -- sampler --
    def sample(self, n_tasks, cap=None):
        cap = cap or max(64, n_tasks * 256)
        tasks = (Task * n_tasks)()
        bps = (Blueprint * cap)()
        n = C.c_int()
        self._check(lib().ofdg_sample(self.h, n_tasks, C.cast(tasks, C.c_void_p), C.cast(bps, C.c_void_p), cap, C.byref(n)))
        return tasks, bps, n.value

    # -- hot path --
    def render(self, tasks, n_tasks, bps, n_bps, img0, img1, flow, stream=0):
        """img0/img1/flow: device pointers (int) or torch CUDA tensors."""
        self._check(lib().ofdg_render(self.h, C.cast(tasks, C.c_void_p), n_tasks, C.cast(bps, C.c_void_p), n_bps,
                                      _dptr(img0), _dptr(img1), _dptr(flow), C.c_void_p(stream)))

    def render_resident(self, img0, img1, flow, stream=0):
        self._check(lib().ofdg_render_resident(self.h, _dptr(img0), _dptr(img1), _dptr(flow), C.c_void_p(stream)))

    def upload_slot(self, slot, tasks, n_tasks, bps, n_bps, stream=0):
        self._check(lib().ofdg_upload_slot(self.h, slot, C.cast(tasks, C.c_void_p), n_tasks, C.cast(bps, C.c_void_p), n_bps,
                                           C.c_void_p(stream)))

    def render_slot(self, slot, img0, img1, flow, stream=0):
        self._check(lib().ofdg_render_slot(self.h, slot, _dptr(img0), _dptr(img1), _dptr(flow), C.c_void_p(stream)))

    def forward(self, img0, img1, flow, stream=0):
        self._check(lib().ofdg_forward(self.h, _dptr(img0), _dptr(img1), _dptr(flow), C.c_void_p(stream)))

    def forward_counter(self, first_index, n, img0, img1, flow, stream=0):
        self._check(lib().ofdg_forward_counter(self.h, first_index, n, _dptr(img0), _dptr(img1), _dptr(flow), C.c_void_p(stream)))

    def sample_counter(self, first_index, n):
        """Blueprints of the device counter sampler: (tasks, bps, n_bps) in the fixed layout."""
        tasks = (Task * n)()
        bps = (Blueprint * (n * 257))()
        self._check(lib().ofdg_sample_counter(self.h, first_index, n, C.cast(tasks, C.c_void_p), C.cast(bps, C.c_void_p)))
        return tasks, bps, n * 257

    def synchronize(self, stream=0):
        self._check(lib().ofdg_synchronize(self.h, C.c_void_p(stream)))

    def last_ticket(self):
        """Number of the batch the last render / forward call enqueued (its own device error word: poll_errors_of)."""
        return int(lib().ofdg_last_ticket(self.h))

    def poll_errors(self):
        """Device error flags of every batch rendered so far, without waiting for anything in flight."""
        self._check(lib().ofdg_poll_errors(self.h))

    def poll_errors_of(self, ticket):
        """Was THAT batch truncated (raises ECAPACITY)?  Call it after the batch's own completion event."""
        self._check(lib().ofdg_poll_errors_of(self.h, ticket))

    @property
    def step(self):
        """Batches produced by forward() so far = the sampler state to checkpoint."""
        return int(lib().ofdg_get_step(self.h))

    @step.setter
    def step(self, k):
        self._check(lib().ofdg_set_step(self.h, int(k)))

    def num_chains(self):
        return lib().ofdg_num_chains(self.h)

    def next_stream(self):
        """The internal hipStream_t (int) the next render / forward call works on; pass it as that call's
        `stream` to be ordered on it directly (consecutive calls then overlap, see include/ofdg.h)."""
        return int(lib().ofdg_stream(self.h) or 0)

    # -- mode 9 warp fields --
    def warp_generate(self, n_fields=1, seed=0):
        self._check(lib().ofdg_warp_generate(self.h, n_fields, seed))

    def warp_upload(self, crops):
        import numpy as np
        a = np.ascontiguousarray(crops, np.float32)
        assert a.ndim == 4 and a.shape[1:] == (4, self.params.height + 1, self.params.width + 1), a.shape
        self._check(lib().ofdg_warp_upload(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0]))

    def warp_count(self):
        n = C.c_int()
        self._check(lib().ofdg_warp_info(self.h, C.byref(n), None, None))
        return n.value

    def warp_download(self, index):
        import numpy as np
        a = np.zeros((4, self.params.height + 1, self.params.width + 1), np.float32)
        self._check(lib().ofdg_warp_download(self.h, index, a.ctypes.data_as(C.c_void_p)))
        return a

    # -- inspection --
    def debug_rasterize(self, xy):
        import numpy as np
        xy = np.ascontiguousarray(xy, np.float64)
        cov = np.zeros((self.params.height, self.params.width), np.uint8)
        self._check(lib().ofdg_debug_rasterize(self.h, xy.ctypes.data_as(C.c_void_p), len(xy), cov.ctypes.data_as(C.c_void_p)))
        return cov

    def debug_rasterize_path(self, xy, types):
        """A path with curve3 segments through the device's flattening and rasteriser (ofdg_debug_rasterize_path)."""
        import numpy as np
        xy = np.ascontiguousarray(xy, np.float64)
        types = np.ascontiguousarray(types, np.int32)
        cov = np.zeros((self.params.height, self.params.width), np.uint8)
        self._check(lib().ofdg_debug_rasterize_path(self.h, xy.ctypes.data_as(C.c_void_p), types.ctypes.data_as(C.c_void_p), len(xy),
                                                    cov.ctypes.data_as(C.c_void_p)))
        return cov

    def debug_dda_rows(self, inv, rows, length):
        """(x, y) in 24.8 fixed point of every pixel of `rows` output rows under the inverse affine (device interpolator)."""
        import numpy as np
        inv = np.ascontiguousarray(inv, np.float64)
        out = np.zeros((rows, length, 2), np.int32)
        self._check(lib().ofdg_debug_dda_rows(self.h, inv.ctypes.data_as(C.c_void_p), rows, length, out.ctypes.data_as(C.c_void_p)))
        return out

    def debug_num_shapes(self, sample):
        n = lib().ofdg_debug_num_shapes(self.h, sample)
        if n < 0:
            raise OfdgError(n, "debug_num_shapes")
        return n

    def debug_coverage(self, sample, shape, frame):
        import numpy as np
        cov = np.zeros((self.params.height, self.params.width), np.uint8)
        self._check(lib().ofdg_debug_coverage(self.h, sample, shape, frame, cov.ctypes.data_as(C.c_void_p)))
        return cov

    def debug_bgprep_tiles(self):
        """(tiles, workgroups) of the last batch's one-launch background preparation (0 tiles: it took another form)."""
        t, w = C.c_int(0), C.c_int(0)
        self._check(lib().ofdg_debug_bgprep_tiles(self.h, C.byref(t), C.byref(w)))
        return t.value, w.value

    def debug_bgprep_paths(self):
        """Tiles of the one-launch preparation since the last call by form: a 3 x 3 list [resize][rotation] (see include/ofdg.h);
        the first call switches the counting on."""
        c = (C.c_uint32 * 9)()
        self._check(lib().ofdg_debug_bgprep_paths(self.h, c))
        return [[int(c[3 * r + k]) for k in range(3)] for r in range(3)]

    def debug_tables(self, s_fixed=200):
        import numpy as np
        add = np.zeros((256, 256), np.uint8)
        sub = np.zeros((256, 256), np.uint8)
        aa = np.zeros(256, np.uint8)
        bl = np.zeros((256, 256), np.uint8)
        vp = C.c_void_p
        self._check(lib().ofdg_debug_tables(self.h, add.ctypes.data_as(vp), sub.ctypes.data_as(vp), aa.ctypes.data_as(vp),
                                            bl.ctypes.data_as(vp), s_fixed))
        return add, sub, aa, bl

    def debug_detmath(self, angles, x):
        """include/ofdg_detmath.h evaluated on the device: (sin, cos) of float64 angles, expf of float32 x."""
        import numpy as np
        a = np.ascontiguousarray(angles, np.float64)
        x = np.ascontiguousarray(x, np.float32)
        s, c, e = np.zeros_like(a), np.zeros_like(a), np.zeros_like(x)
        vp = C.c_void_p
        self._check(lib().ofdg_debug_detmath(self.h, a.ctypes.data_as(vp), len(a), s.ctypes.data_as(vp), c.ctypes.data_as(vp),
                                             x.ctypes.data_as(vp), len(x), e.ctypes.data_as(vp)))
        return s, c, e

    def set_profiling(self, mode=2):
        self._check(lib().ofdg_set_profiling(self.h, int(mode)))

    def kernel_ms(self, name):
        ms = C.c_float()
        self._check(lib().ofdg_kernel_ms(self.h, name.encode(), C.byref(ms)))
        return ms.value


def decode_image(path):
    """One image file as the layer's native loader decodes it (binary PPM, or PNG through the system's libpng):
    uint8 array [3, h, w] in B, G, R order.  No GPU needed."""
    import numpy as np
    w, h = C.c_int(), C.c_int()
    L = lib()
    L.ofdg_host_decode_image.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ofdg_host_last_error.restype = C.c_char_p
    rc = L.ofdg_host_decode_image(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if rc != OK:
        raise OfdgError(rc, L.ofdg_host_last_error().decode())
    out = np.empty((3, h.value, w.value), dtype=np.uint8)
    rc = L.ofdg_host_decode_image(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h))
    if rc != OK:
        raise OfdgError(rc, L.ofdg_host_last_error().decode())
    return out


class Comm:
    """One process per GPU on an RCCL communicator (ofdg_comm): the native multi-GPU start-up.

    rank 0 draws the ncclUniqueId (Comm.unique_id()) and hands it to the other processes through whatever
    rendezvous the launcher offers (`exchange`: a callable bytes-or-None -> bytes, e.g. a torch.distributed store);
    Comm(id, rank, world, device) binds the device and joins.  bcast_setup is THE start-up collective."""

    TABLE_CAP = 65536

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib().ofdg_comm_unique_id(buf)
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_comm_last_error(None).decode())
        return buf.raw

    def __init__(self, uid, rank, world_size, device):
        h = C.c_void_p()
        rc = lib().ofdg_comm_init(C.c_char_p(uid), rank, world_size, device, C.byref(h))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_comm_last_error(None).decode())
        self.h, self.rank, self.world_size, self.device = h, rank, world_size, device

    @classmethod
    def from_store(cls, store, rank, world_size, device, key="ofdg_unique_id"):
        """Join through a key-value store (torch.distributed's TCPStore / FileStore ...): rank 0 sets the id."""
        if rank == 0:
            uid = cls.unique_id()
            store.set(key, uid)
        else:
            uid = bytes(store.get(key))
        return cls(uid, rank, world_size, device)

    def _check(self, rc):
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_comm_last_error(self.h).decode())

    def bcast_setup(self, gen=None, root=0):
        """Root passes its Generator (stream + pool description are read off it); returns (Setup, table).
        A root that has no Generator to pass (its own set-up failed) calls bcast_abort instead: success or failure of
        the start-up is decided by all ranks together, nobody is left waiting in the broadcast."""
        su = Setup()
        table = (TexEntry * self.TABLE_CAP)()
        if self.rank == root:
            rc = lib().ofdg_setup_of(gen.h, C.byref(su), table, self.TABLE_CAP)
            if rc != OK:
                su.status = rc
        self._check(lib().ofdg_comm_bcast_setup(self.h, root, C.byref(su), table, self.TABLE_CAP))
        return su, table

    def bcast_abort(self, code=EINVAL, root=0):
        """The root's side of a failed start-up: the receivers' bcast_setup raises with `code`."""
        return lib().ofdg_comm_bcast_abort(self.h, root, int(code), self.TABLE_CAP)

    def nccl_count(self):
        """Number of ranks RCCL itself reports for the communicator (ncclCommCount)."""
        n = lib().ofdg_comm_nccl_count(self.h)
        if n < 0:
            self._check(n)
        return n

    def params_of(self, setup):
        p = Params()
        lib().ofdg_setup_params(C.byref(setup), self.h, C.byref(p))
        return p

    def bcast_pool(self, gen, root=0):
        self._check(lib().ofdg_comm_bcast_pool(self.h, root, gen.h))

    def agree(self, local_ok=True):
        """Every rank calls it after a step it did alone (context, pool ...), with local_ok = False if that step
        failed: raises ESTARTUP on EVERY rank unless all passed True - nobody is left in the next collective."""
        self._check(lib().ofdg_comm_agree(self.h, 1 if local_ok else 0))

    def close(self):
        if getattr(self, "h", None):
            lib().ofdg_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _dptr(x):
    if isinstance(x, C.c_void_p):  # (device_pointers(): resolved once, outside a hot loop)
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("output tensors must be contiguous device tensors")
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def device_pointers(tensors):
    """The device addresses of output tensors as ctypes pointers, checked once: what a loop that renders into the same buffer
    sets again and again passes instead of the tensors (the per-call checks of three tensors cost the host ~2 us)."""
    return tuple(_dptr(t) for t in tensors)


def alloc_outputs(n, height, width, device="cuda"):
    """The three top blobs: image0 [n,3,H,W], image1 [n,3,H,W], flow [n,2,H,W] (float32)."""
    import torch
    return (torch.zeros((n, 3, height, width), dtype=torch.float32, device=device),
            torch.zeros((n, 3, height, width), dtype=torch.float32, device=device),
            torch.zeros((n, 2, height, width), dtype=torch.float32, device=device))


class HostSampler:
    """The reference-stream blueprint sampler on its own (host only, no GPU needed)."""

    def __init__(self, mode, width=512, height=384, num_objects=0):
        h = C.c_void_p()
        rc = lib().ofdg_host_sampler_create(mode, width, height, num_objects, C.byref(h))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_host_last_error().decode())
        self.h = h

    def next(self, n_tasks, cap=None):
        cap = cap or max(64, n_tasks * 256)
        tasks = (Task * n_tasks)()
        bps = (Blueprint * cap)()
        n = C.c_int()
        rc = lib().ofdg_host_sampler_next(self.h, n_tasks, C.cast(tasks, C.c_void_p), C.cast(bps, C.c_void_p), cap, C.byref(n))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_host_last_error().decode())
        return tasks, bps, n.value

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().ofdg_host_sampler_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown
            pass


def host_realize(params, pool_n, pool_w, pool_h, tasks, n_tasks, bps, n_bps, cap=4096):
    """Returns (shape_mats [n,2,6], object_mats [m,2,6]) float64."""
    import numpy as np
    sm = np.zeros((cap, 2, 6), np.float64)
    om = np.zeros((cap, 2, 6), np.float64)
    ns, no = C.c_int(), C.c_int()
    rc = lib().ofdg_host_realize(C.byref(params), pool_n, pool_w, pool_h, C.cast(tasks, C.c_void_p), n_tasks,
                                 C.cast(bps, C.c_void_p), n_bps, sm.ctypes.data_as(C.c_void_p), cap, C.byref(ns),
                                 om.ctypes.data_as(C.c_void_p), cap, C.byref(no))
    if rc != OK:
        raise OfdgError(rc, lib().ofdg_host_last_error().decode())
    return sm[:ns.value].copy(), om[:no.value].copy()


def host_displacers(width, height, seed):
    import numpy as np
    out = np.zeros((1024, 9), np.float64)
    n = lib().ofdg_host_displacers(width, height, seed, out.ctypes.data_as(C.c_void_p), 1024)
    if n < 0:
        raise OfdgError(ECAPACITY, "displacer capacity")
    return out[:n].copy()


def parse_prototxt(text):
    """Returns (Params, texture_dbases, n_top) for one `layer { ... }` block."""
    p = Params()
    buf = C.create_string_buffer(4096)
    ntop = C.c_int()
    rc = lib().ofdg_parse_prototxt(text.encode(), C.byref(p), buf, 4096, C.byref(ntop))
    if rc != OK:
        raise OfdgError(rc, lib().ofdg_host_last_error().decode())
    return p, buf.value.decode(), ntop.value


class DataGenerationLayer:
    """Python handle on the C++ ofdg::DataGenerationLayer (the mirror of the reference's
    Caffe layer): constructed from prototxt text, Forward() returns the three top blobs
    as torch tensors that alias the layer's device memory."""

    def __init__(self, prototxt, comm=None):
        h = C.c_void_p()
        rc = lib().ofdg_layer_create_dist(prototxt.encode(), comm.h if comm is not None else None, C.byref(h))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_host_last_error().decode())
        self.h = h

    def type(self):
        return "DataGeneration"

    def Forward(self):
        import torch
        p0, p1, p2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        shape = (C.c_int * 4)()
        rc = lib().ofdg_layer_forward(self.h, C.byref(p0), C.byref(p1), C.byref(p2), C.byref(shape))
        if rc != OK:
            raise OfdgError(rc, lib().ofdg_host_last_error().decode())
        n, _, hh, ww = list(shape)
        outs = []
        for ptr, ch in ((p0, 3), (p1, 3), (p2, 2)):
            t = torch.empty((n, ch, hh, ww), dtype=torch.float32, device="cuda")
            # copy out of the layer's blob (device-to-device); the blob stays owned by the layer
            src = _as_tensor(ptr.value, (n, ch, hh, ww))
            t.copy_(src)
            outs.append(t)
        return tuple(outs)

    def in_flight(self):
        """Batches rendered ahead that were still unfinished when the last Forward returned (prefetch > 1)."""
        return lib().ofdg_layer_in_flight(self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().ofdg_layer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlowLoader:
    """Endless iterator of (image0, image1, flow) CUDA tensors - the role of the reference's prefetch
    thread + blocking queue (data_generation_layer.cpp:36-56, 141-172, 266-282; data_param.prefetch).

    `prefetch` output buffer sets are cycled; batch k+1 .. k+prefetch-1 are already enqueued on the GPU
    while the consumer works on batch k.  Every batch is rendered on one of the generator's internal
    in-order streams (Generator.next_stream), so the batches in flight overlap; the hand-over is two events
    per batch: the consumer stream (`stream`, default: torch's current stream) waits for the batch it is given,
    and a buffer set is re-rendered only after the consumer work enqueued up to the next `next()` is done.
    Use the tensors on the consumer stream, or synchronise before touching them elsewhere.  Samples shard
    over ranks by global index (params.rank / params.world_size): no communication."""

    def __init__(self, params=None, pool=None, prefetch=3, stream=None, start=0, **kw):
        import torch
        self.gen = Generator(params, **kw)
        p = self.gen.params
        if pool is not None:
            pool(self.gen)                      # callable that fills the texture pool (pool_synthetic / pool_upload ...)
        if p.mode == 9 and self.gen.warp_count() == 0:
            self.gen.warp_generate(2, p.seed)
        if start:                               # (after the warp fields: resuming replays the crop serving order too)
            self.gen.step = int(start)          # resume: the first batch handed out is batch `start` (see `consumed`)
        self.start = int(start)
        self.prefetch = max(2, int(prefetch))
        self.consumer = torch.cuda.current_stream() if stream is None else torch.cuda.ExternalStream(int(stream))
        self.bufs = [alloc_outputs(p.batch_size, p.height, p.width) for _ in range(self.prefetch)]
        self.ready = [torch.cuda.Event() for _ in range(self.prefetch)]      # batch rendered (internal stream)
        self.released = [None] * self.prefetch                               # consumer done with the set
        self.k = 0
        for j in range(self.prefetch - 1):      # fill the ring
            self._enqueue(j)
        self.head = self.prefetch - 1

    def _enqueue(self, j):
        import torch
        s = self.gen.next_stream()
        chain = torch.cuda.ExternalStream(s)
        if self.released[j] is not None:
            chain.wait_event(self.released[j])
        self.gen.forward(*self.bufs[j], s)
        self.ready[j].record(chain)

    @property
    def consumed(self):
        """Index of the next batch the iterator will hand out: what to store in a checkpoint (`start=` on resume)."""
        return self.start + self.k

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        j = self.k % self.prefetch
        self.consumer.wait_event(self.ready[j])
        # the set handed out last time is free once the consumer work enqueued so far is done: render the
        # batch that will be consumed prefetch-1 iterations from now into it
        f = self.head % self.prefetch
        if f != j:
            ev = torch.cuda.Event()
            ev.record(self.consumer)
            self.released[f] = ev
            self._enqueue(f)
            self.head += 1
        self.k += 1
        return self.bufs[j]


def _as_tensor(ptr, shape):
    """Wrap a raw device pointer as a float32 torch tensor (no ownership)."""
    import numpy as np
    import torch

    class _Holder:
        pass

    h = _Holder()
    n = int(np.prod(shape))
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda").view(*shape)
