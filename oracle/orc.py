"""ctypes binding of the CPU oracle (oracle/libnvca_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (nubomedia-vca_amd/) never imports it.
PARITY UNPINNED -- see oracle/nvca_oracle.h.

Also holds an independent reader of OpenCV's old-format Haar cascade XML
 built on xml.etree, so that the product's hand-written C++
loader can be cross-checked against a second implementation.
"""
import ctypes as C
import os
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HAAR_DO_CANNY_PRUNING = 1
HAAR_SCALE_IMAGE = 2
HAAR_FIND_BIGGEST_OBJECT = 4
HAAR_DO_ROUGH_SEARCH = 8
SUM_F32PAIR = 0
SUM_F64 = 1


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("windows", C.c_int64), ("stumps", C.c_int64), ("raw_hits", C.c_int64),
                ("n_scales", C.c_int), ("pad_", C.c_int), ("stage_enter", C.c_int64 * 64)]


class CCascade(C.Structure):
    _fields_ = [("ow", C.c_int), ("oh", C.c_int), ("n_stages", C.c_int),
                ("stage_ncls", C.POINTER(C.c_int)), ("stage_thr", C.POINTER(C.c_float)),
                ("n_cls", C.c_int), ("cls_nnodes", C.POINTER(C.c_int)),
                ("n_nodes", C.c_int), ("rects", C.POINTER(C.c_int)),
                ("rweights", C.POINTER(C.c_float)), ("tilted", C.POINTER(C.c_int)),
                ("node_thr", C.POINTER(C.c_float)), ("left", C.POINTER(C.c_int)),
                ("right", C.POINTER(C.c_int)), ("alpha", C.POINTER(C.c_float))]


class FaceParams(C.Structure):
    _fields_ = [("width_to_process", C.c_int), ("process_x_every_4", C.c_int),
                ("scale_factor_pct", C.c_int), ("track_threshold", C.c_int),
                ("euclidean_threshold", C.c_int), ("area_threshold", C.c_int),
                ("full_res", C.c_int), ("min_neighbors", C.c_int), ("policy", C.c_int)]


class PartParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("width_to_process", C.c_int), ("process_x_every_4", C.c_int),
                ("scale_factor_pct", C.c_int), ("detect_event", C.c_int), ("policy", C.c_int)]


PART_EYE, PART_NOSE, PART_MOUTH, PART_EAR = 0, 1, 2, 3


class TrackerParams(C.Structure):
    _fields_ = [("threshold", C.c_int), ("min_area", C.c_int), ("max_area", C.c_long),
                ("distance", C.c_int), ("mhi_duration", C.c_double), ("seg_thresh", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libnvca_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_imgproc.c", "orc_haar.c", "orc_pipe.c", "orc_parts.c",
                                             "nvca_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p = C.POINTER(C.c_uint8)
        L.orc_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orc_resize_linear.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_equalize_hist.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orc_integral.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.orc_integral_tilted.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.orc_flip_h.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orc_detect_multiscale.argtypes = [C.POINTER(CCascade), u8p, C.c_int, C.c_int, C.c_int, C.c_double,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(Rect), C.c_int, C.POINTER(Stats)]
        L.orc_detect_raw.argtypes = [C.POINTER(CCascade), u8p, C.c_int, C.c_int, C.c_int, C.c_double,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(Rect), C.c_int, C.POINTER(Stats)]
        L.orc_group_rectangles.argtypes = [C.POINTER(Rect), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.orc_scale_grid.argtypes = [C.c_int] * 4 + [C.c_double] + [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_int]
        L.orc_face_params_default.argtypes = [C.POINTER(FaceParams)]
        L.orc_face_stream_create.argtypes = [C.POINTER(CCascade), C.POINTER(FaceParams)]
        L.orc_face_stream_create.restype = C.c_void_p
        L.orc_face_stream_destroy.argtypes = [C.c_void_p]
        L.orc_face_stream_process.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(Rect),
                                              C.POINTER(C.c_int), C.c_int]
        L.orc_face_stream_gate.argtypes = [C.c_void_p]
        L.orc_face_frame_detect.argtypes = [C.POINTER(CCascade), C.POINTER(FaceParams), u8p, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(Rect), C.c_int]
        L.orc_face_stream_finish.argtypes = [C.c_void_p, C.c_int, C.POINTER(Rect), C.c_int, C.c_int, C.c_int,
                                             C.POINTER(Rect), C.POINTER(C.c_int), C.c_int]
        L.orc_track_faces.argtypes = [C.POINTER(Rect), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(Rect), C.c_int, C.c_int, C.c_int]
        L.orc_tracker_params_default.argtypes = [C.POINTER(TrackerParams)]
        L.orc_tracker_create.argtypes = [C.POINTER(TrackerParams)]
        L.orc_tracker_create.restype = C.c_void_p
        L.orc_tracker_destroy.argtypes = [C.c_void_p]
        L.orc_tracker_process.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.POINTER(Rect), C.c_int]
        L.orc_update_mhi.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_double, C.c_double]
        L.orc_segment_motion.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.POINTER(Rect), C.c_int]
        L.orc_join_objects.argtypes = [C.POINTER(Rect), C.c_int, C.c_int, C.c_long, C.c_int]
        L.orc_part_params_default.argtypes = [C.POINTER(PartParams), C.c_int]
        L.orc_part_stream_create.argtypes = [C.POINTER(PartParams), C.POINTER(CCascade), C.POINTER(CCascade), C.POINTER(CCascade)]
        L.orc_part_stream_create.restype = C.c_void_p
        L.orc_part_stream_destroy.argtypes = [C.c_void_p]
        L.orc_part_stream_push_faces.argtypes = [C.c_void_p, C.POINTER(Rect), C.c_int]
        L.orc_part_stream_process.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(Rect), C.c_int,
                                              C.POINTER(C.c_int), C.POINTER(Rect), C.c_int, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def rects_to_np(buf, n):
    return np.array([[buf[i].x, buf[i].y, buf[i].w, buf[i].h] for i in range(n)], dtype=np.int32).reshape(n, 4)


def np_to_rects(a):
    a = np.asarray(a, dtype=np.int32).reshape(-1, 4)
    buf = (Rect * max(len(a), 1))()
    for i, r in enumerate(a):
        buf[i] = Rect(int(r[0]), int(r[1]), int(r[2]), int(r[3]))
    return buf


# ---------------------------------------------------------------- imgproc
def bgr2gray(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, cn = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_u8(img), w, h, img.strides[0], cn, _u8(out), w)
    return out


def resize_linear(img, dw, dh):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_resize_linear(_u8(img), w, h, img.strides[0], cn, _u8(out), dw, dh, dw * cn)
    return out


def equalize_hist(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty_like(img)
    lib().orc_equalize_hist(_u8(img), w, h, img.strides[0], _u8(out), w)
    return out


def integral(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    s = np.empty((h + 1, w + 1), np.int32)
    q = np.empty((h + 1, w + 1), np.float64)
    lib().orc_integral(_u8(img), w, h, img.strides[0], s.ctypes.data_as(C.POINTER(C.c_int32)),
                       q.ctypes.data_as(C.POINTER(C.c_double)))
    return s, q


def integral_tilted(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    t = np.empty((h + 1, w + 1), np.int32)
    lib().orc_integral_tilted(_u8(img), w, h, img.strides[0], t.ctypes.data_as(C.POINTER(C.c_int32)))
    return t


def flip_h(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty_like(img)
    lib().orc_flip_h(_u8(img), w, h, img.strides[0], _u8(out), w)
    return out


# ---------------------------------------------------------------- cascade
class Cascade:
    """Flat cascade (python-owned numpy arrays) + the C view of it."""

    def __init__(self, ow, oh, stage_ncls, stage_thr, cls_nnodes, rects, rweights, tilted,
                 node_thr, left, right, alpha):
        self.ow, self.oh = int(ow), int(oh)
        self.stage_ncls = np.ascontiguousarray(stage_ncls, np.int32)
        self.stage_thr = np.ascontiguousarray(stage_thr, np.float32)
        self.cls_nnodes = np.ascontiguousarray(cls_nnodes, np.int32)
        self.rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 3, 4)
        self.rweights = np.ascontiguousarray(rweights, np.float32).reshape(-1, 3)
        self.tilted = np.ascontiguousarray(tilted, np.int32)
        self.node_thr = np.ascontiguousarray(node_thr, np.float32)
        self.left = np.ascontiguousarray(left, np.int32)
        self.right = np.ascontiguousarray(right, np.int32)
        self.alpha = np.ascontiguousarray(alpha, np.float32)
        ip, fp = C.POINTER(C.c_int), C.POINTER(C.c_float)
        self.c = CCascade(self.ow, self.oh, len(self.stage_ncls),
                          self.stage_ncls.ctypes.data_as(ip), self.stage_thr.ctypes.data_as(fp),
                          len(self.cls_nnodes), self.cls_nnodes.ctypes.data_as(ip),
                          len(self.node_thr), self.rects.ctypes.data_as(ip),
                          self.rweights.ctypes.data_as(fp), self.tilted.ctypes.data_as(ip),
                          self.node_thr.ctypes.data_as(fp), self.left.ctypes.data_as(ip),
                          self.right.ctypes.data_as(ip), self.alpha.ctypes.data_as(fp))

    @property
    def n_stages(self):
        return len(self.stage_ncls)

    @property
    def n_nodes(self):
        return len(self.node_thr)


def parse_cascade_xml(text):
    """Old-format OpenCV Haar cascade XML -> Cascade.

    Follows icvReadHaarClassifier (OpenCV 2.4 haar.cpp): numbers are parsed as
    double and stored as float; a <left_val>/<right_val> becomes alpha[last++]
    with child index -last; tree-structured STAGE graphs (parent / next) are rejected, tree-structured weak
    classifiers and tilted features are read.
    """
    root = ET.fromstring(text)
    node = None
    for child in root:
        if child.get("type_id") == "opencv-haar-classifier":
            node = child
            break
    if node is None:
        raise ValueError("no opencv-haar-classifier node")
    ow, oh = [int(v) for v in node.find("size").text.split()]
    stage_ncls, stage_thr, cls_nnodes = [], [], []
    rects, rweights, tilted, node_thr, left, right, alpha = [], [], [], [], [], [], []
    for si, st in enumerate(node.find("stages")):
        trees = st.find("trees")
        stage_ncls.append(len(trees))
        stage_thr.append(np.float32(float(st.find("stage_threshold").text)))
        parent = int(st.find("parent").text)
        nxt = int(st.find("next").text)
        if parent != si - 1 or nxt != -1:
            raise ValueError("tree-structured STAGE graph not supported (weak classifiers may be trees)")
        for tree in trees:
            cls_nnodes.append(len(tree))
            last = 0
            for nd in tree:
                feat = nd.find("feature")
                rr = np.zeros((3, 4), np.int32)
                ww = np.zeros(3, np.float32)
                for k, r in enumerate(feat.find("rects")):
                    vals = r.text.split()
                    rr[k] = [int(v) for v in vals[:4]]
                    ww[k] = np.float32(float(vals[4]))
                rects.append(rr)
                rweights.append(ww)
                tilted.append(int(feat.find("tilted").text))
                node_thr.append(np.float32(float(nd.find("threshold").text)))
                for tag, dst in (("left", left), ("right", right)):
                    nn = nd.find(tag + "_node")
                    if nn is not None:
                        dst.append(int(nn.text))
                    else:
                        dst.append(-last)
                        alpha.append(np.float32(float(nd.find(tag + "_val").text)))
                        last += 1
    return Cascade(ow, oh, stage_ncls, stage_thr, cls_nnodes, rects, rweights, tilted, node_thr,
                   left, right, alpha)


def load_cascade(path):
    with open(path, "r") as f:
        return parse_cascade_xml(f.read())


# ---------------------------------------------------------------- detection
def detect_multiscale(casc, gray, scale_factor=1.1, min_neighbors=3, flags=0, min_size=(0, 0),
                      max_size=(0, 0), policy=SUM_F32PAIR, cap=4096, return_stats=False):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    buf = (Rect * cap)()
    st = Stats()
    n = lib().orc_detect_multiscale(C.byref(casc.c), _u8(gray), w, h, gray.strides[0], scale_factor,
                                    min_neighbors, flags, min_size[0], min_size[1], max_size[0],
                                    max_size[1], policy, buf, cap, C.byref(st))
    r = rects_to_np(buf, n)
    return (r, st) if return_stats else r


def detect_raw(casc, gray, scale_factor=1.1, flags=0, min_size=(0, 0), max_size=(0, 0),
               policy=SUM_F32PAIR, cap=1 << 20, return_stats=False):
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    buf = (Rect * cap)()
    st = Stats()
    n = lib().orc_detect_raw(C.byref(casc.c), _u8(gray), w, h, gray.strides[0], scale_factor, flags,
                             min_size[0], min_size[1], max_size[0], max_size[1], policy, buf, cap,
                             C.byref(st))
    r = rects_to_np(buf, n)
    return (r, st) if return_stats else r


def group_rectangles(rects, group_threshold, eps=0.2):
    buf = np_to_rects(rects)
    n = len(np.asarray(rects).reshape(-1, 4))
    wts = (C.c_int * max(n, 1))()
    m = lib().orc_group_rectangles(buf, n, group_threshold, eps, wts)
    return rects_to_np(buf, m), np.array(wts[:m], dtype=np.int32)


def scale_grid(ow, oh, w, h, scale_factor, min_size=(0, 0), max_size=(0, 0), cap=256):
    f = (C.c_double * cap)()
    n = lib().orc_scale_grid(ow, oh, w, h, scale_factor, min_size[0], min_size[1], max_size[0],
                             max_size[1], f, cap)
    return list(f[:n])


class FaceStream:
    def __init__(self, casc, **kw):
        p = FaceParams()
        lib().orc_face_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self.casc = casc
        self.p = p
        self.h = lib().orc_face_stream_create(C.byref(casc.c), C.byref(p))

    def frame_detect(self, bgr, cap=256):
        """the stateless part of an analysed frame (resize, gray, equalizeHist, detectMultiScale): working-image boxes"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W, _ = bgr.shape
        buf = (Rect * cap)()
        n = lib().orc_face_frame_detect(C.byref(self.casc.c), C.byref(self.p), _u8(bgr), W, H, bgr.strides[0], buf, cap)
        return rects_to_np(buf, n)

    def process_memo(self, key, bgr, memo, cap=256):
        """process(bgr), with the frame's detections taken from / left in memo[key]: for harnesses that feed the same
        frames again and again (the temporal logic still runs frame by frame)"""
        H, W = bgr.shape[:2]
        analysed = lib().orc_face_stream_gate(self.h)
        det = np.zeros((0, 4), np.int32)
        if analysed:
            if key not in memo:
                memo[key] = self.frame_detect(bgr)
            det = memo[key]
        buf = (Rect * cap)()
        ids = (C.c_int * cap)()
        n = lib().orc_face_stream_finish(self.h, analysed, np_to_rects(det), len(det), W, H, buf, ids, cap)
        return rects_to_np(buf, n), np.array(ids[:n], dtype=np.int32)

    def process(self, bgr, cap=256):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W, _ = bgr.shape
        buf = (Rect * cap)()
        ids = (C.c_int * cap)()
        n = lib().orc_face_stream_process(self.h, _u8(bgr), W, H, bgr.strides[0], buf, ids, cap)
        return rects_to_np(buf, n), np.array(ids[:n], dtype=np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_face_stream_destroy(self.h)
            except Exception:       # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


def track_faces(faces, ids, next_id, cur, track_threshold=40, cap=256):
    faces = np.asarray(faces, np.int32).reshape(-1, 4)
    fb = (Rect * cap)()
    for i, r in enumerate(faces):
        fb[i] = Rect(*[int(v) for v in r])
    ib = (C.c_int * cap)(*[int(v) for v in ids])
    nid = C.c_int(next_id)
    cur = np.asarray(cur, np.int32).reshape(-1, 4)
    cb = np_to_rects(cur)
    n = lib().orc_track_faces(fb, ib, len(faces), C.byref(nid), cb, len(cur), track_threshold, cap)
    return rects_to_np(fb, n), np.array(ib[:n], dtype=np.int32), nid.value


class Tracker:
    def __init__(self, **kw):
        p = TrackerParams()
        lib().orc_tracker_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self.h = lib().orc_tracker_create(C.byref(p))

    def process(self, bgra, timestamp_ms, cap=4096):
        bgra = np.ascontiguousarray(bgra, dtype=np.uint8)
        H, W, _ = bgra.shape
        buf = (Rect * cap)()
        n = lib().orc_tracker_process(self.h, _u8(bgra), W, H, bgra.strides[0], float(timestamp_ms), buf, cap)
        return rects_to_np(buf, n)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_tracker_destroy(self.h)
            except Exception:       # interpreter shutdown: the module globals are already gone
                pass
            self.h = None


def update_mhi(silh, mhi, ts, dur):
    silh = np.ascontiguousarray(silh, np.uint8)
    assert mhi.dtype == np.float32 and mhi.flags.c_contiguous
    h, w = silh.shape
    lib().orc_update_mhi(_u8(silh), w, h, mhi.ctypes.data_as(C.POINTER(C.c_float)), ts, dur)
    return mhi


def segment_motion(mhi, ts, seg_thresh=32.0, cap=65536):
    assert mhi.dtype == np.float32 and mhi.flags.c_contiguous
    h, w = mhi.shape
    buf = (Rect * cap)()
    n = lib().orc_segment_motion(mhi.ctypes.data_as(C.POINTER(C.c_float)), w, h, ts, seg_thresh, buf, cap)
    return rects_to_np(buf, n)


def join_objects(rects, min_area=50, max_area=30000, distance=35):
    rects = np.asarray(rects, np.int32).reshape(-1, 4)
    buf = np_to_rects(rects)
    n = lib().orc_join_objects(buf, len(rects), min_area, max_area, distance)
    return rects_to_np(buf, n)


class PartStream:
    def __init__(self, kind, face, a, b=None, **kw):
        p = PartParams()
        lib().orc_part_params_default(C.byref(p), kind)
        for k, v in kw.items():
            setattr(p, k, v)
        self._keep = (face, a, b)
        self.h = lib().orc_part_stream_create(C.byref(p), C.byref(face.c), C.byref(a.c), C.byref(b.c) if b is not None else None)

    def push_faces(self, faces):
        faces = np.asarray(faces, np.int32).reshape(-1, 4)
        lib().orc_part_stream_push_faces(self.h, np_to_rects(faces), len(faces))

    def process(self, bgr, cap=64):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W, _ = bgr.shape
        a, b = (Rect * cap)(), (Rect * cap)()
        na, nb = C.c_int(), C.c_int()
        lib().orc_part_stream_process(self.h, _u8(bgr), W, H, bgr.strides[0], a, cap, C.byref(na), b, cap, C.byref(nb))
        return rects_to_np(a, na.value), rects_to_np(b, nb.value)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().orc_part_stream_destroy(self.h)
            except Exception:       # interpreter shutdown: the module globals are already gone
                pass
            self.h = None
