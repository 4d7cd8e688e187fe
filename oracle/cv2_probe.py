"""Opportunistic real-OpenCV datum.  TEST INFRASTRUCTURE, like everything in oracle/.

The reference's arithmetic lives in OpenCV (FACE/kmsfacedetect.cpp:805-811 calls cv::resize, cv::cvtColor,
cv::equalizeHist and CascadeClassifier::detectMultiScale); OpenCV is absent from the build container and nothing is ever
installed.  If -- and only if -- a `cv2` module is already importable on the box this runs on, the same frames are run
through the real calls, timed, and diffed primitive by primitive against the oracle (and, by the caller, against the HIP
path).  That would be the first datum that pins the oracle to OpenCV itself; OpenCV >= 3 re-implements the old-format
cascade evaluation, so a mismatch there is informational and the first differing primitive is reported.

probe() never raises: whatever goes wrong ends up in the returned dict.
"""
import os
import tempfile
import time

import numpy as np


def _import_cv2():
    try:
        import cv2                      # noqa: F401  (never installed by us; only used if the box already has it)
        return cv2, None
    except Exception as e:              # ImportError, or a broken install
        return None, "%s: %s" % (type(e).__name__, e)


def probe(xml, frames_bgr, width_to_process=0, scale_factor=1.1, min_neighbors=3, budget_s=10.0):
    """frames_bgr: list of HxWx3 uint8.  Returns {"available": bool, ...}."""
    cv2, why = _import_cv2()
    if cv2 is None:
        return {"available": False, "note": "cv2 absent on this box (%s): oracle stays pinned by hand-derived answers only" % why}
    import orc
    out = {"available": True, "version": getattr(cv2, "__version__", "?"), "frames": 0, "fps": None,
           "primitives_equal": {}, "boxes_equal": None, "first_difference": None}
    path = None
    try:
        fd, path = tempfile.mkstemp(suffix=".xml")
        with os.fdopen(fd, "w") as f:
            f.write(xml)
        cc = cv2.CascadeClassifier(path)
        if cc.empty():
            out["first_difference"] = "CascadeClassifier.load refused the old-format XML"
            return out
        oc = orc.parse_cascade_xml(xml)
        prim = {"resize": True, "cvtColor": True, "equalizeHist": True, "integral": True}
        boxes_equal = True
        first = None
        t_cv = 0.0
        n = 0
        t_start = time.perf_counter()
        for bgr in frames_bgr:
            if time.perf_counter() - t_start > budget_s and n > 0:
                break
            H, W = bgr.shape[:2]
            scale = (W // width_to_process) if width_to_process else 1
            cols, rows = int(np.rint(W / scale)), int(np.rint(H / scale))
            t0 = time.perf_counter()
            small = cv2.resize(bgr, (cols, rows), interpolation=cv2.INTER_LINEAR)      # FACE/kmsfacedetect.cpp:805
            gray = cv2.cvtColor(small, cv2.COLOR_BGR2GRAY)                             # :806
            eq = cv2.equalizeHist(gray)                                                # :807
            det = cc.detectMultiScale(eq, scaleFactor=scale_factor, minNeighbors=min_neighbors, flags=0,
                                      minSize=(cols // 20, rows // 20))                # :809-811
            t_cv += time.perf_counter() - t0
            det = np.asarray(det, np.int32).reshape(-1, 4)
            # the same chain through the oracle; each primitive is fed the oracle's own predecessor AND compared on
            # OpenCV's predecessor, so one early difference does not cascade into the later verdicts
            o_small = orc.resize_linear(bgr, cols, rows) if (cols, rows) != (W, H) else np.ascontiguousarray(bgr)
            checks = [("resize", np.array_equal(o_small, small)),
                      ("cvtColor", np.array_equal(orc.bgr2gray(small), gray)),
                      ("equalizeHist", np.array_equal(orc.equalize_hist(gray), eq))]
            s_cv, q_cv = cv2.integral2(eq, sdepth=cv2.CV_32S, sqdepth=cv2.CV_64F) if hasattr(cv2, "integral2") else (None, None)
            if s_cv is not None:
                s_o, q_o = orc.integral(eq)
                checks.append(("integral", np.array_equal(s_o, s_cv) and np.array_equal(q_o, q_cv)))
            for name, ok in checks:
                prim[name] = prim[name] and bool(ok)
                if not ok and first is None:
                    first = name
            o_det = orc.detect_multiscale(oc, eq, scale_factor, min_neighbors, 0, (cols // 20, rows // 20))
            # OpenCV's raw-hit order depends on its thread schedule: compare as sets
            same = sorted(map(tuple, det.tolist())) == sorted(map(tuple, o_det.tolist()))
            boxes_equal = boxes_equal and same
            if not same and first is None:
                first = "detectMultiScale (frame %d: cv2 %d boxes, oracle %d)" % (n, len(det), len(o_det))
            n += 1
        out.update(frames=n, fps=(n / t_cv if t_cv > 0 else None), primitives_equal=prim, boxes_equal=bool(boxes_equal),
                   first_difference=first, threads=int(cv2.getNumThreads()) if hasattr(cv2, "getNumThreads") else None)
    except Exception as e:
        out["first_difference"] = "probe failed: %s: %s" % (type(e).__name__, e)
    finally:
        if path:
            try:
                os.unlink(path)
            except OSError:
                pass
    return out
