"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference holds no fixtures for this path and cannot be run here, so these vectors are
produced by oracle/ (the restatement of OpenCV 2.4) on small seeded inputs.  They pin the oracle AND the HIP
path against silent drift: both must keep reproducing the committed numbers.  Inputs are stored with the
outputs, so the fixtures do not depend on numpy's generators staying stable.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nubomedia-vca_amd")):
    sys.path.insert(0, p)
import orc  # noqa: E402
from nubovca import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # a small 8-stage cascade keeps the fixture (XML text included) small
    xml = synth.synthetic_cascade_xml(seed=99, stages=[3, 6, 9, 12, 15, 18, 21, 24])
    casc = orc.parse_cascade_xml(xml)
    W, H = 160, 120
    gray = synth.make_gray(W, H, 5, "natural", [(40, 25, 64)])
    bgr = [synth.make_bgr(W, H, 20 + i, "natural", [] if i == 2 else [(30 + 4 * i, 20, 70)]) for i in range(5)]
    eq = orc.equalize_hist(gray)
    s, q = orc.integral(eq)
    out = dict(cascade_xml=np.frombuffer(xml.encode(), np.uint8), gray=gray, equalized=eq,
               resized_80x60=orc.resize_linear(gray, 80, 60), resized_53x41=orc.resize_linear(gray, 53, 41),
               integral_sum=s, integral_sqsum=q,
               raw_sc=orc.detect_raw(casc, eq, 1.1, 0, (0, 0)),
               det_sc=orc.detect_multiscale(casc, eq, 1.1, 3, 0, (0, 0)),
               raw_si=orc.detect_raw(casc, eq, 1.1, orc.HAAR_SCALE_IMAGE, (0, 0)),
               det_si=orc.detect_multiscale(casc, eq, 1.1, 2, orc.HAAR_SCALE_IMAGE, (0, 0)),
               det_big=orc.detect_multiscale(casc, eq, 1.1, 3, orc.HAAR_FIND_BIGGEST_OBJECT, (1, 1)),
               frames_bgr=np.stack(bgr))
    fs = orc.FaceStream(casc, width_to_process=160, scale_factor_pct=10)
    for i, f in enumerate(bgr):
        b, ids = fs.process(f)
        out["face_boxes_%d" % i] = b
        out["face_ids_%d" % i] = ids
    bgra = np.concatenate([np.stack(bgr), np.full((5, H, W, 1), 255, np.uint8)], axis=3)
    tr = orc.Tracker(threshold=15, min_area=20)
    for i in range(5):
        out["trk_boxes_%d" % i] = tr.process(bgra[i], 100.0 + 33.0 * i)
    np.savez_compressed(os.path.join(HERE, "face_path_160x120.npz"), **out)
    print("wrote", os.path.join(HERE, "face_path_160x120.npz"), {k: v.shape for k, v in out.items() if k != "cascade_xml"})


if __name__ == "__main__":
    main()
