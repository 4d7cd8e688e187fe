"""ctypes binding of libnubovca_hip's C ABI (include/nubovca.h).

This is the harness side used by tests/ and bench.py; the product host code is
the C++ GStreamer shim under nubomedia-vca_amd/gst/.  There is no fallback: if
the shared library is missing, or no HIP device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVCA_LIB") or os.path.join(os.path.dirname(_HERE), "libnubovca_hip.so")      # NVCA_LIB: A/B runs of kernel variants (scripts/)

OK = 0
ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_PARSE, ERR_UNSUPPORTED, ERR_OVERFLOW, ERR_NOMEM, ERR_INTERNAL = range(-1, -10, -1)
MEM_HOST, MEM_DEVICE = 0, 1
HAAR_DO_CANNY_PRUNING, HAAR_SCALE_IMAGE, HAAR_FIND_BIGGEST_OBJECT, HAAR_DO_ROUGH_SEARCH = 1, 2, 4, 8
SUM_F32PAIR, SUM_F64 = 0, 1
K_COUNT = 14


class NvcaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nubovca error %d: %s" % (code, msg))
        self.code = code


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int), ("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int), ("bgra", C.c_uint8 * 4)]


SHAPE_RECT3, SHAPE_RING4 = 0, 1


class Overlay(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("channels", C.c_int),
                ("offset_x_percent", C.c_double), ("offset_y_percent", C.c_double), ("width_percent", C.c_double), ("height_percent", C.c_double)]


class Frame(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int),
                ("mem", C.c_int), ("pts", C.c_uint64)]


class FaceParams(C.Structure):
    _fields_ = [("width_to_process", C.c_int), ("process_x_every_4", C.c_int), ("scale_factor_pct", C.c_int),
                ("track_threshold", C.c_int), ("euclidean_threshold", C.c_int), ("area_threshold", C.c_int),
                ("min_neighbors", C.c_int), ("detect_event", C.c_int)]


class PartParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("width_to_process", C.c_int), ("process_x_every_4", C.c_int),
                ("scale_factor_pct", C.c_int), ("detect_event", C.c_int)]


PART_EYE, PART_NOSE, PART_MOUTH, PART_EAR = 0, 1, 2, 3


class TrackerParams(C.Structure):
    _fields_ = [("threshold", C.c_int), ("min_area", C.c_int), ("max_area", C.c_long), ("distance", C.c_int),
                ("mhi_duration", C.c_double), ("seg_thresh", C.c_double)]


# every symbol include/nubovca.h declares
SYMBOLS = [
    "nvca_ctx_create", "nvca_ctx_destroy", "nvca_last_error", "nvca_version", "nvca_ctx_set_hit_capacity",
    "nvca_ctx_set_sum_policy", "nvca_ctx_synchronize", "nvca_ctx_stream", "nvca_ctx_enable_kernel_timing",
    "nvca_ctx_kernel_timing", "nvca_kernel_name", "nvca_cascade_load_xml", "nvca_cascade_load_mem",
    "nvca_cascade_free", "nvca_cascade_info", "nvca_cascade_dump", "nvca_bgr2gray", "nvca_resize_linear",
    "nvca_equalize_hist", "nvca_integral", "nvca_detect_multiscale", "nvca_detect_raw", "nvca_group_rectangles",
    "nvca_face_params_default", "nvca_face_stream_create", "nvca_face_stream_destroy",
    "nvca_face_stream_set_params", "nvca_face_stream_motion_event", "nvca_face_stream_process",
    "nvca_face_batch_process", "nvca_tracker_params_default", "nvca_tracker_create", "nvca_tracker_destroy",
    "nvca_tracker_set_params", "nvca_tracker_process", "nvca_tracker_batch_process", "nvca_flip_horizontal",
    "nvca_part_params_default", "nvca_part_stream_create", "nvca_part_stream_destroy", "nvca_part_stream_set_params",
    "nvca_part_stream_push_faces", "nvca_part_stream_process", "nvca_part_stream_faces",
    "nvca_host_register", "nvca_host_unregister", "nvca_face_batch_submit", "nvca_face_batch_collect",
    "nvca_integral_tilted", "nvca_cascade_kind", "nvca_part_batch_process", "nvca_device_count", "nvca_draw_shapes",
    "nvca_cascade_validate_mem", "nvca_abi_selftest", "nvca_ctx_set_option", "nvca_ctx_get_option", "nvca_overlay_blend",
    "nvca_part_batch_submit", "nvca_part_batch_collect",
]

_lib = None


def _preload_torch_hip():
    """One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as
    /opt/rocm's).  Whichever is mapped first serves both, and torch cannot see the GPU
    through the system copy, so map torch's copy first when torch is installed."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the library and declare prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python __graft_entry__.py build` (hipcc, gfx950)" % LIB_PATH)
    _preload_torch_hip()
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.nvca_version.restype = C.c_char_p
    L.nvca_last_error.restype = C.c_char_p
    L.nvca_last_error.argtypes = [vp]
    L.nvca_kernel_name.restype = C.c_char_p
    L.nvca_kernel_name.argtypes = [C.c_int]
    L.nvca_device_count.argtypes = [ip]
    L.nvca_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nvca_ctx_destroy.argtypes = [vp]
    L.nvca_ctx_destroy.restype = None
    L.nvca_ctx_set_hit_capacity.argtypes = [vp, C.c_int]
    L.nvca_ctx_set_sum_policy.argtypes = [vp, C.c_int]
    L.nvca_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.nvca_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.nvca_overlay_blend.argtypes = [vp, C.POINTER(Frame), C.POINTER(Rect), C.c_int, C.POINTER(Overlay)]
    L.nvca_ctx_synchronize.argtypes = [vp]
    L.nvca_ctx_stream.argtypes = [vp]
    L.nvca_host_register.argtypes = [vp, vp, C.c_size_t]
    L.nvca_host_unregister.argtypes = [vp, vp]
    L.nvca_ctx_stream.restype = vp
    L.nvca_ctx_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.nvca_ctx_kernel_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.nvca_cascade_load_xml.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.nvca_cascade_load_mem.argtypes = [vp, C.c_char_p, C.c_int64, C.POINTER(vp)]
    L.nvca_cascade_free.argtypes = [vp]
    L.nvca_cascade_validate_mem.argtypes = [C.c_char_p, C.c_int64, ip, ip, ip, ip, C.c_char_p, C.c_int]
    L.nvca_abi_selftest.argtypes = [C.c_int]
    L.nvca_cascade_free.restype = None
    L.nvca_cascade_info.argtypes = [vp, ip, ip, ip, ip]
    L.nvca_cascade_dump.argtypes = [vp, ip, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), ip, C.POINTER(C.c_float)]
    L.nvca_bgr2gray.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.nvca_resize_linear.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.nvca_equalize_hist.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.nvca_integral.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.nvca_integral_tilted.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    L.nvca_cascade_kind.argtypes = [vp, ip, ip]
    L.nvca_detect_multiscale.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.c_int, ip]
    L.nvca_detect_raw.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.POINTER(Rect), C.c_int, ip]
    L.nvca_group_rectangles.argtypes = [vp, C.POINTER(Rect), C.c_int, C.c_int, C.c_double, ip]
    L.nvca_face_params_default.argtypes = [C.POINTER(FaceParams)]
    L.nvca_face_params_default.restype = None
    L.nvca_face_stream_create.argtypes = [vp, vp, C.POINTER(FaceParams), C.POINTER(vp)]
    L.nvca_face_stream_destroy.argtypes = [vp]
    L.nvca_face_stream_destroy.restype = None
    L.nvca_face_stream_set_params.argtypes = [vp, C.POINTER(FaceParams)]
    L.nvca_face_stream_motion_event.argtypes = [vp]
    L.nvca_face_stream_process.argtypes = [vp, C.POINTER(Frame), C.POINTER(Rect), ip, C.c_int, ip]
    L.nvca_face_batch_process.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(Frame), C.POINTER(Rect), ip, C.c_int, ip]
    L.nvca_face_batch_submit.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(Frame), ip]
    L.nvca_face_batch_collect.argtypes = [vp, C.c_int, C.POINTER(Rect), ip, C.c_int, ip]
    L.nvca_tracker_params_default.argtypes = [C.POINTER(TrackerParams)]
    L.nvca_tracker_params_default.restype = None
    L.nvca_tracker_create.argtypes = [vp, C.POINTER(TrackerParams), C.POINTER(vp)]
    L.nvca_tracker_destroy.argtypes = [vp]
    L.nvca_tracker_destroy.restype = None
    L.nvca_tracker_set_params.argtypes = [vp, C.POINTER(TrackerParams)]
    L.nvca_tracker_process.argtypes = [vp, C.POINTER(Frame), C.c_double, C.POINTER(Rect), C.c_int, ip]
    L.nvca_tracker_batch_process.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(Frame), C.POINTER(C.c_double),
                                             C.POINTER(Rect), C.c_int, ip]
    L.nvca_flip_horizontal.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.nvca_draw_shapes.argtypes = [vp, C.POINTER(Frame), C.c_int, C.POINTER(Shape), C.c_int]
    L.nvca_part_params_default.argtypes = [C.POINTER(PartParams), C.c_int]
    L.nvca_part_params_default.restype = None
    L.nvca_part_stream_create.argtypes = [vp, C.POINTER(PartParams), vp, vp, vp, C.POINTER(vp)]
    L.nvca_part_stream_destroy.argtypes = [vp]
    L.nvca_part_stream_destroy.restype = None
    L.nvca_part_stream_set_params.argtypes = [vp, C.POINTER(PartParams)]
    L.nvca_part_stream_push_faces.argtypes = [vp, C.POINTER(Rect), C.c_int]
    L.nvca_part_stream_faces.argtypes = [vp, C.POINTER(Rect), C.c_int, ip]
    L.nvca_part_batch_process.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(Frame), C.POINTER(Rect), C.c_int, ip, C.POINTER(Rect), C.c_int, ip]
    L.nvca_part_stream_process.argtypes = [vp, C.POINTER(Frame), C.POINTER(Rect), C.c_int, ip, C.POINTER(Rect), C.c_int, ip]
    L.nvca_part_batch_submit.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(Frame), ip]
    L.nvca_part_batch_collect.argtypes = [vp, C.c_int, C.POINTER(Rect), C.c_int, ip, C.POINTER(Rect), C.c_int, ip]
    _lib = L
    return L


def _rects(buf, n):
    return np.array([[buf[i].x, buf[i].y, buf[i].w, buf[i].h] for i in range(n)], dtype=np.int32).reshape(n, 4)


class Context:
    """nvca_ctx: one per GPU."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.nvca_ctx_create(device, C.byref(h))
        if rc != OK:
            raise NvcaError(rc, "nvca_ctx_create failed (no HIP device? this library has no CPU fallback)")
        self.h = h

    def check(self, rc):
        if rc != OK:
            raise NvcaError(rc, self.L.nvca_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.nvca_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_hit_capacity(self, cap):
        self.check(self.L.nvca_ctx_set_hit_capacity(self.h, cap))

    def set_sum_policy(self, policy):
        self.check(self.L.nvca_ctx_set_sum_policy(self.h, policy))

    def set_option(self, name, value):
        self.check(self.L.nvca_ctx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        self.check(self.L.nvca_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """context manager: the given switches for the duration of a with-block; afterwards every one of them holds what it held
        on entry (the process environment's value or an earlier set_option), not a table of defaults"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = {k: self.get_option(k) for k in kw}
            for k, v in kw.items():
                self.set_option(k, v)
            try:
                yield self
            finally:
                for k, v in before.items():
                    self.set_option(k, v)
        return scope()

    def synchronize(self):
        self.check(self.L.nvca_ctx_synchronize(self.h))

    def host_register(self, arr):
        """page-lock a numpy frame (zero-copy ingest); pair with host_unregister before the array goes away"""
        self.check(self.L.nvca_host_register(self.h, arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr):
        self.check(self.L.nvca_host_unregister(self.h, arr.ctypes.data))

    def enable_kernel_timing(self, on=True):
        self.check(self.L.nvca_ctx_enable_kernel_timing(self.h, int(on)))

    def kernel_timing(self):
        ms = (C.c_double * K_COUNT)()
        n = (C.c_int64 * K_COUNT)()
        self.check(self.L.nvca_ctx_kernel_timing(self.h, ms, n))
        return {self.L.nvca_kernel_name(k).decode(): (ms[k], n[k]) for k in range(K_COUNT) if n[k]}

    # ---- cascade
    def load_cascade_xml(self, text):
        if isinstance(text, str):
            text = text.encode()
        h = C.c_void_p()
        self.check(self.L.nvca_cascade_load_mem(self.h, text, len(text), C.byref(h)))
        return Cascade(self, h)

    def load_cascade_file(self, path):
        h = C.c_void_p()
        self.check(self.L.nvca_cascade_load_xml(self.h, path.encode(), C.byref(h)))
        return Cascade(self, h)

    # ---- primitives on host numpy arrays
    def bgr2gray(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w, cn = img.shape
        out = np.empty((h, w), np.uint8)
        self.check(self.L.nvca_bgr2gray(self.h, img.ctypes.data, w, h, img.strides[0], cn, MEM_HOST, out.ctypes.data, w))
        return out

    def resize_linear(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((dh, dw) if cn == 1 else (dh, dw, cn), np.uint8)
        self.check(self.L.nvca_resize_linear(self.h, img.ctypes.data, w, h, img.strides[0], cn, MEM_HOST, out.ctypes.data,
                                             dw, dh, dw * cn))
        return out

    def equalize_hist(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty_like(img)
        self.check(self.L.nvca_equalize_hist(self.h, img.ctypes.data, w, h, img.strides[0], MEM_HOST, out.ctypes.data, w))
        return out

    def integral(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        s = np.empty((h + 1, w + 1), np.int32)
        q = np.empty((h + 1, w + 1), np.float64)
        self.check(self.L.nvca_integral(self.h, img.ctypes.data, w, h, img.strides[0], MEM_HOST,
                                        s.ctypes.data_as(C.POINTER(C.c_int32)), q.ctypes.data_as(C.POINTER(C.c_double))))
        return s, q

    def draw_shapes(self, frame, channels, shapes):
        """nvca_draw_shapes: shapes = [(kind, x, y, w, h, (b, g, r, a))]; frame: a Frame (device memory) or a writable numpy image (drawn in place)"""
        fr = frame if isinstance(frame, Frame) else make_frame(frame)
        arr = (Shape * max(len(shapes), 1))()
        for i, (kind, x, y, w, h, col) in enumerate(shapes):
            arr[i].kind, arr[i].x, arr[i].y, arr[i].w, arr[i].h = kind, x, y, w, h
            for k in range(4):
                arr[i].bgra[k] = col[k]
        self.check(self.L.nvca_draw_shapes(self.h, C.byref(fr), channels, arr, len(shapes)))

    def integral_tilted(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        t = np.empty((h + 1, w + 1), np.int32)
        self.check(self.L.nvca_integral_tilted(self.h, img.ctypes.data, w, h, img.strides[0], MEM_HOST, t.ctypes.data_as(C.POINTER(C.c_int32))))
        return t

    def detect_multiscale(self, casc, gray, scale_factor=1.1, min_neighbors=3, flags=0, min_size=(0, 0),
                          max_size=(0, 0), cap=4096):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        buf = (Rect * cap)()
        n = C.c_int()
        self.check(self.L.nvca_detect_multiscale(self.h, casc.h, gray.ctypes.data, w, h, gray.strides[0], MEM_HOST,
                                                 scale_factor, min_neighbors, flags, min_size[0], min_size[1],
                                                 max_size[0], max_size[1], buf, cap, C.byref(n)))
        return _rects(buf, min(n.value, cap))

    def detect_raw(self, casc, gray, scale_factor=1.1, flags=0, min_size=(0, 0), max_size=(0, 0), cap=1 << 18):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        buf = (Rect * cap)()
        n = C.c_int()
        self.check(self.L.nvca_detect_raw(self.h, casc.h, gray.ctypes.data, w, h, gray.strides[0], MEM_HOST, scale_factor,
                                          flags, min_size[0], min_size[1], max_size[0], max_size[1], buf, cap, C.byref(n)))
        return _rects(buf, min(n.value, cap))

    def group_rectangles(self, rects, group_threshold, eps=0.2):
        rects = np.asarray(rects, np.int32).reshape(-1, 4)
        buf = (Rect * max(len(rects), 1))()
        for i, r in enumerate(rects):
            buf[i] = Rect(*[int(v) for v in r])
        n = C.c_int()
        self.check(self.L.nvca_group_rectangles(self.h, buf, len(rects), group_threshold, eps, C.byref(n)))
        return _rects(buf, n.value)

    # ---- batched frontend
    def face_batch_process(self, streams, frames, cap=64):
        """frames: list of Frame (see make_frame); streams: list of FaceStream (same length)."""
        n = len(frames)
        sh = (C.c_void_p * n)(*[s.h for s in streams])
        fr = (Frame * n)(*frames)
        out = (Rect * (n * cap))()
        ids = (C.c_int * (n * cap))()
        cnt = (C.c_int * n)()
        self.check(self.L.nvca_face_batch_process(self.h, n, sh, fr, out, ids, cap, cnt))
        boxes = np.frombuffer(out, dtype=np.int32).reshape(n, cap, 4)       # views of the ctypes buffers: no per-box python work
        idv = np.frombuffer(ids, dtype=np.int32).reshape(n, cap)
        res = []
        for i in range(n):
            k = min(cnt[i], cap)
            res.append((boxes[i, :k].copy(), idv[i, :k].copy()))
        return res


    def prepare_face_batch(self, streams, frames, cap=64):
        """argument and result arrays of a batch that is handed in again and again (a serving loop over the same buffers):
        the ctypes marshalling is done once, process() is then just the C call"""
        return PreparedFaceBatch(self, streams, frames, cap)

    def face_batch_submit(self, streams, frames):
        """first half of face_batch_process: queues the batch and returns a ticket (at most two in flight)"""
        n = len(frames)
        sh = (C.c_void_p * n)(*[s.h for s in streams])
        fr = (Frame * n)(*frames)
        tk = C.c_int()
        self.check(self.L.nvca_face_batch_submit(self.h, n, sh, fr, C.byref(tk)))
        return (tk.value, n, sh, fr)          # the argument arrays stay referenced until the batch is collected

    def face_batch_collect(self, ticket, cap=64):
        tk, n = ticket[0], ticket[1]
        out = (Rect * (n * cap))()
        ids = (C.c_int * (n * cap))()
        cnt = (C.c_int * n)()
        self.check(self.L.nvca_face_batch_collect(self.h, tk, out, ids, cap, cnt))
        boxes = np.frombuffer(out, dtype=np.int32).reshape(n, cap, 4)
        idv = np.frombuffer(ids, dtype=np.int32).reshape(n, cap)
        return [(boxes[i, :min(cnt[i], cap)].copy(), idv[i, :min(cnt[i], cap)].copy()) for i in range(n)]


class PreparedFaceBatch:
    def __init__(self, ctx, streams, frames, cap):
        self.ctx, self.n, self.cap = ctx, len(frames), cap
        self.keep = (list(streams), list(frames))
        self.sh = (C.c_void_p * self.n)(*[s.h for s in streams])
        self.fr = (Frame * self.n)(*frames)
        self.out = (Rect * (self.n * cap))()
        self.ids = (C.c_int * (self.n * cap))()
        self.cnt = (C.c_int * self.n)()
        self.boxes = np.frombuffer(self.out, dtype=np.int32).reshape(self.n, cap, 4)
        self.idv = np.frombuffer(self.ids, dtype=np.int32).reshape(self.n, cap)
        self.counts = np.frombuffer(self.cnt, dtype=np.int32)

    def process(self):
        """nvca_face_batch_process on the prepared arrays; results stay in self.boxes / self.idv / self.counts"""
        self.ctx.check(self.ctx.L.nvca_face_batch_process(self.ctx.h, self.n, self.sh, self.fr, self.out, self.ids, self.cap, self.cnt))
        return self

    def results(self):
        """[(boxes[k,4], ids[k])] per frame, as face_batch_process returns them"""
        return [(self.boxes[i, :min(self.counts[i], self.cap)].copy(), self.idv[i, :min(self.counts[i], self.cap)].copy()) for i in range(self.n)]


class Cascade:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def info(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.L.nvca_cascade_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def kind(self):
        """(has_tilted, has_trees)"""
        t, r = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.L.nvca_cascade_kind(self.h, C.byref(t), C.byref(r)))
        return bool(t.value), bool(r.value)

    def dump(self):
        ow, oh, ns, nw = self.info()
        rects = np.zeros((nw, 3, 4), np.int32)
        wts = np.zeros((nw, 3), np.float32)
        thr, lv, rv = np.zeros(nw, np.float32), np.zeros(nw, np.float32), np.zeros(nw, np.float32)
        ss, st = np.zeros(ns, np.int32), np.zeros(ns, np.float32)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        self.ctx.check(self.ctx.L.nvca_cascade_dump(self.h, rects.ctypes.data_as(ip), wts.ctypes.data_as(fp),
                                                    thr.ctypes.data_as(fp), lv.ctypes.data_as(fp), rv.ctypes.data_as(fp),
                                                    ss.ctypes.data_as(ip), st.ctypes.data_as(fp)))
        return dict(size=(ow, oh), rects=rects, weights=wts, thr=thr, left=lv, right=rv, stage_sizes=ss, stage_thr=st)

    def free(self):
        if self.h:
            if self.ctx.h:
                self.ctx.L.nvca_cascade_free(self.h)
            self.h = None


def overlay_blend(ctx, frame, boxes, image, offset_x=0.0, offset_y=0.0, width=1.0, height=1.0):
    """nvca_overlay_blend: image (HxW, HxWx3 or HxWx4 uint8, host) scaled onto every box of the BGR frame, in place.
    frame: a writable numpy image (host: ctx may be None) or a Frame (device memory)"""
    L = load()
    fr = frame if isinstance(frame, Frame) else make_frame(frame)
    image = np.ascontiguousarray(image, np.uint8)
    cn = 1 if image.ndim == 2 else image.shape[2]
    ov = Overlay(image.ctypes.data, image.shape[1], image.shape[0], image.strides[0], cn, offset_x, offset_y, width, height)
    boxes = np.asarray(boxes, np.int32).reshape(-1, 4)
    buf = (Rect * max(len(boxes), 1))()
    for i, r in enumerate(boxes):
        buf[i] = Rect(*[int(v) for v in r])
    rc = L.nvca_overlay_blend(ctx.h if ctx is not None else None, C.byref(fr), buf, len(boxes), C.byref(ov))
    if rc != 0:
        raise NvcaError(rc, L.nvca_last_error(ctx.h).decode() if ctx is not None else "nvca_overlay_blend")


def draw_shapes_host(img, channels, shapes):
    """nvca_draw_shapes on a host image without a context (no device needed): drawn in place"""
    L = load()
    fr = make_frame(img)
    arr = (Shape * max(len(shapes), 1))()
    for i, (kind, x, y, w, h, col) in enumerate(shapes):
        arr[i].kind, arr[i].x, arr[i].y, arr[i].w, arr[i].h = kind, x, y, w, h
        for k in range(4):
            arr[i].bgra[k] = col[k]
    rc = L.nvca_draw_shapes(None, C.byref(fr), channels, arr, len(shapes))
    if rc != 0:
        raise NvcaError(rc, "nvca_draw_shapes")


def make_frame(arr_or_ptr, width=None, height=None, stride=None, mem=MEM_HOST, pts=0):
    """Frame from a numpy HxWxC uint8 array (host) or a raw device pointer."""
    if isinstance(arr_or_ptr, np.ndarray):
        a = arr_or_ptr
        assert a.dtype == np.uint8 and a.flags.c_contiguous
        f = Frame(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], MEM_HOST, pts)
        f._keep = a
        return f
    return Frame(int(arr_or_ptr), width, height, stride, mem, pts)


class FaceStream:
    """nvca_face_stream: mirrors one `nubofacedetector` element instance
    (properties of FACE/kmsfacedetect.cpp:1043-1102 by their reference names)."""

    PROPS = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4",
             "multi_scale_factor": "scale_factor_pct", "track_threshold": "track_threshold",
             "euclidean_distance": "euclidean_threshold", "area_threshold": "area_threshold",
             "min_neighbors": "min_neighbors", "detect_event": "detect_event"}

    def __init__(self, ctx, cascade, **props):
        self.ctx, self.cascade = ctx, cascade
        self.p = FaceParams()
        ctx.L.nvca_face_params_default(C.byref(self.p))
        for k, v in props.items():
            setattr(self.p, self.PROPS[k], int(v))
        h = C.c_void_p()
        ctx.check(ctx.L.nvca_face_stream_create(ctx.h, cascade.h, C.byref(self.p), C.byref(h)))
        self.h = h

    def set_property(self, name, value):
        setattr(self.p, self.PROPS[name], int(value))
        self.ctx.check(self.ctx.L.nvca_face_stream_set_params(self.h, C.byref(self.p)))

    def motion_event(self):
        self.ctx.check(self.ctx.L.nvca_face_stream_motion_event(self.h))

    def process(self, bgr, cap=64):
        """One transform_frame_ip on a host BGR frame -> (boxes[n,4], ids[n])."""
        return self.ctx.face_batch_process([self], [make_frame(np.ascontiguousarray(bgr, np.uint8))], cap)[0]

    def close(self):
        if getattr(self, "h", None):
            if self.ctx.h:                    # a closed context took its device state with it
                self.ctx.L.nvca_face_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tracker:
    """nvca_tracker: mirrors one `nubotracker` element instance (properties of
    TRK/gstnubotracker.cpp:504-542 by their reference names)."""

    PROPS = {"set_threshold": "threshold", "set_min_area": "min_area", "set_max_area": "max_area",
             "set_distance": "distance", "mhi_duration": "mhi_duration", "seg_thresh": "seg_thresh"}

    def __init__(self, ctx, **props):
        self.ctx = ctx
        self.p = TrackerParams()
        ctx.L.nvca_tracker_params_default(C.byref(self.p))
        for k, v in props.items():
            setattr(self.p, self.PROPS[k], v)
        h = C.c_void_p()
        ctx.check(ctx.L.nvca_tracker_create(ctx.h, C.byref(self.p), C.byref(h)))
        self.h = h

    def set_property(self, name, value):
        setattr(self.p, self.PROPS[name], value)
        self.ctx.check(self.ctx.L.nvca_tracker_set_params(self.h, C.byref(self.p)))

    def process(self, bgra, timestamp_ms, cap=4096):
        return tracker_batch_process(self.ctx, [self], [make_frame(np.ascontiguousarray(bgra, np.uint8))],
                                     [timestamp_ms], cap)[0]

    def close(self):
        if getattr(self, "h", None):
            if self.ctx.h:
                self.ctx.L.nvca_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


import threading
_trk_tls = threading.local()


def tracker_batch_process(ctx, trackers, frames, timestamps, cap=4096):
    n = len(frames)
    th = (C.c_void_p * n)(*[t.h for t in trackers])
    fr = (Frame * n)(*frames)
    ts = (C.c_double * n)(*[float(t) for t in timestamps])
    key = (n, cap)
    if getattr(_trk_tls, "key", None) != key:      # result buffers are reused across calls of a thread (the library overwrites them)
        _trk_tls.key, _trk_tls.bufs = key, ((Rect * (n * cap))(), (C.c_int * n)())
    out, cnt = _trk_tls.bufs
    ctx.check(ctx.L.nvca_tracker_batch_process(ctx.h, n, th, fr, ts, out, cap, cnt))
    boxes = np.frombuffer(out, dtype=np.int32).reshape(n, cap, 4)
    return [boxes[i, :min(cnt[i], cap)].copy() for i in range(n)]


def part_batch_process(ctx, streams, frames, cap=64):
    """nvca_part_batch_process: one frame of each stream; returns [(list A, list B)] per stream"""
    n = len(streams)
    sh = (C.c_void_p * n)(*[s.h for s in streams])
    fr = (Frame * n)(*[f if isinstance(f, Frame) else make_frame(np.ascontiguousarray(f, np.uint8)) for f in frames])
    a, b = (Rect * (n * cap))(), (Rect * (n * cap))()
    na, nb = (C.c_int * n)(), (C.c_int * n)()
    ctx.check(ctx.L.nvca_part_batch_process(ctx.h, n, sh, fr, a, cap, na, b, cap, nb))
    A = np.frombuffer(a, dtype=np.int32).reshape(n, cap, 4)
    B = np.frombuffer(b, dtype=np.int32).reshape(n, cap, 4)
    return [(A[i, :min(na[i], cap)].copy(), B[i, :min(nb[i], cap)].copy()) for i in range(n)]


class PartTicket:
    """a submitted nvca_part_batch_submit call: keeps what the library still reads (the frame records and whatever backs them) alive"""

    def __init__(self, ticket, n, keep):
        self.ticket, self.n, self._keep = ticket, n, keep


def part_batch_submit(ctx, streams, frames):
    """nvca_part_batch_submit: gates, working images and face passes of one frame per stream are queued; -> PartTicket"""
    n = len(streams)
    sh = (C.c_void_p * n)(*[s.h for s in streams])
    fr = (Frame * n)(*[f if isinstance(f, Frame) else make_frame(np.ascontiguousarray(f, np.uint8)) for f in frames])
    t = C.c_int(-1)
    ctx.check(ctx.L.nvca_part_batch_submit(ctx.h, n, sh, fr, C.byref(t)))
    return PartTicket(t.value, n, (sh, fr, frames))


def part_batch_collect(ctx, tk, cap=64):
    """nvca_part_batch_collect: the ticket's results, [(list A, list B)] per stream (tickets in submit order)"""
    n = tk.n
    a, b = (Rect * (n * cap))(), (Rect * (n * cap))()
    na, nb = (C.c_int * n)(), (C.c_int * n)()
    ctx.check(ctx.L.nvca_part_batch_collect(ctx.h, tk.ticket, a, cap, na, b, cap, nb))
    tk._keep = None
    A = np.frombuffer(a, dtype=np.int32).reshape(n, cap, 4)
    B = np.frombuffer(b, dtype=np.int32).reshape(n, cap, 4)
    return [(A[i, :min(na[i], cap)].copy(), B[i, :min(nb[i], cap)].copy()) for i in range(n)]


class PreparedPartBatch:
    """argument and result arrays of a part-detector batch that is handed in again and again (the same streams on the same frame
    buffers, e.g. one frame set of a serving loop): the ctypes marshalling is done once; process() / submit() + collect() are then
    just the C calls, the results stay in the arrays (counts_a / counts_b, boxes_a / boxes_b)"""

    def __init__(self, ctx, streams, frames, cap=64):
        self.ctx, self.n, self.cap = ctx, len(streams), cap
        n = self.n
        self.keep = (list(streams), list(frames))
        self.sh = (C.c_void_p * n)(*[s.h for s in streams])
        self.fr = (Frame * n)(*[f if isinstance(f, Frame) else make_frame(np.ascontiguousarray(f, np.uint8)) for f in frames])
        self.a, self.b = (Rect * (n * cap))(), (Rect * (n * cap))()
        self.na, self.nb = (C.c_int * n)(), (C.c_int * n)()
        self.boxes_a = np.frombuffer(self.a, dtype=np.int32).reshape(n, cap, 4)
        self.boxes_b = np.frombuffer(self.b, dtype=np.int32).reshape(n, cap, 4)
        self.counts_a = np.frombuffer(self.na, dtype=np.int32)
        self.counts_b = np.frombuffer(self.nb, dtype=np.int32)
        self.ticket = C.c_int(-1)

    def process(self):
        self.ctx.check(self.ctx.L.nvca_part_batch_process(self.ctx.h, self.n, self.sh, self.fr, self.a, self.cap, self.na, self.b, self.cap, self.nb))

    def submit(self):
        self.ctx.check(self.ctx.L.nvca_part_batch_submit(self.ctx.h, self.n, self.sh, self.fr, C.byref(self.ticket)))

    def collect(self):
        self.ctx.check(self.ctx.L.nvca_part_batch_collect(self.ctx.h, self.ticket.value, self.a, self.cap, self.na, self.b, self.cap, self.nb))

    def found(self):
        return int(np.minimum(self.counts_a, self.cap).sum() + np.minimum(self.counts_b, self.cap).sum())

    def results(self):
        return [(self.boxes_a[i, :min(self.na[i], self.cap)].copy(), self.boxes_b[i, :min(self.nb[i], self.cap)].copy()) for i in range(self.n)]


class PartStream:
    """nvca_part_stream: mirrors one nuboeyedetector / nubonosedetector / nubomouthdetector / nuboeardetector instance."""

    PROPS = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4",
             "multi_scale_factor": "scale_factor_pct", "detect_event": "detect_event"}

    def __init__(self, ctx, kind, face, a, b=None, **props):
        self.ctx = ctx
        self.p = PartParams()
        ctx.L.nvca_part_params_default(C.byref(self.p), kind)
        for k, v in props.items():
            setattr(self.p, self.PROPS[k], int(v))
        h = C.c_void_p()
        ctx.check(ctx.L.nvca_part_stream_create(ctx.h, C.byref(self.p), face.h, a.h, b.h if b is not None else None, C.byref(h)))
        self.h = h
        self._keep = (face, a, b)

    def push_faces(self, faces):
        faces = np.asarray(faces, np.int32).reshape(-1, 4)
        buf = (Rect * max(len(faces), 1))()
        for i, r in enumerate(faces):
            buf[i] = Rect(*[int(v) for v in r])
        self.ctx.check(self.ctx.L.nvca_part_stream_push_faces(self.h, buf, len(faces)))

    def process(self, bgr, cap=64):
        f = bgr if isinstance(bgr, Frame) else make_frame(np.ascontiguousarray(bgr, np.uint8))
        a, b = (Rect * cap)(), (Rect * cap)()
        na, nb = C.c_int(), C.c_int()
        self.ctx.check(self.ctx.L.nvca_part_stream_process(self.h, C.byref(f), a, cap, C.byref(na), b, cap, C.byref(nb)))
        return _rects(a, min(na.value, cap)), _rects(b, min(nb.value, cap))

    def close(self):
        if getattr(self, "h", None):
            if self.ctx.h:
                self.ctx.L.nvca_part_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
