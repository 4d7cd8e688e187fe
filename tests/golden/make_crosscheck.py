"""Independent cross-checks of the oracle, from libraries the BUILD container happens to have (never installed, never used on
the GPU box; the tests read only the fixtures this script writes):

  crosscheck_ccl.npz    scipy.ndimage.label (4-connectivity) on plateau motion-history images: the components a flood fill
                        with a tolerance finds when every moving region holds one timestamp -- boxes in raster order of their
                        first pixel -- against orc.segment_motion (cvSegmentMotion, TRK/gstnubotracker.cpp:376)
  crosscheck_rects.npz  skimage.transform.integral_image / integrate on random images: the integral image and inclusive
                        rectangle sums, against orc.integral and the corner arithmetic sum[y+h][x+w] - sum[y][x+w] - sum[y+h][x]
                        + sum[y][x] every Haar feature of the oracle is built on

    python3 tests/golden/make_crosscheck.py            (scipy part)
    cd /tmp && /opt/conda/bin/python3.9 /root/repo/tests/golden/make_crosscheck.py      (skimage lives in the conda python)

This does not pin the oracle to OpenCV (parity stays "unpinned"): it removes same-author risk from the two places -- component
order, rectangle corners -- where a misreading shared by oracle and product would be invisible."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def make_ccl():
    from scipy import ndimage
    rng = np.random.default_rng(2024)
    cases = []
    for k in range(24):
        H, W = int(rng.integers(20, 90)), int(rng.integers(20, 120))
        mhi = np.zeros((H, W), np.float32)
        ts = 1000.0 + 33.0 * k
        # moving regions: rectangles, L shapes, diagonal chains (8- but not 4-connected), single pixels, a frame-wide bar
        for _ in range(int(rng.integers(1, 9))):
            y, x = int(rng.integers(0, H - 2)), int(rng.integers(0, W - 2))
            h, w = int(rng.integers(1, max(2, H // 3))), int(rng.integers(1, max(2, W // 3)))
            mhi[y:y + h, x:x + w] = ts
            if rng.random() < 0.4:
                mhi[y:y + 1, x:min(W, x + 2 * w)] = ts
        if k % 3 == 0:
            for d in range(min(H, W) // 2):                 # a diagonal: every pixel its own 4-connected component
                mhi[d * 2 % H, d * 2 % W] = ts
        if k % 5 == 0:
            mhi[H // 2, :] = ts
        lab, n = ndimage.label(mhi > 0)                      # default structure: 4-connectivity; labels in raster order of first pixel
        boxes = []
        for sl in ndimage.find_objects(lab):
            boxes.append([sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start])
        cases.append((mhi, ts, np.array(boxes, np.int32).reshape(-1, 4)))
    out = {}
    for i, (m, ts, b) in enumerate(cases):
        out["mhi_%d" % i] = m
        out["ts_%d" % i] = np.float64(ts)
        out["boxes_%d" % i] = b
    out["n"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "crosscheck_ccl.npz"), **out)
    print("crosscheck_ccl.npz:", len(cases), "cases,", sum(len(c[2]) for c in cases), "components")


def make_rects():
    from skimage.transform import integral_image, integrate
    rng = np.random.default_rng(77)
    out = {}
    n = 10
    for i in range(n):
        H, W = int(rng.integers(5, 70)), int(rng.integers(5, 90))
        img = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
        ii = integral_image(img.astype(np.int64))          # ii[r, c] = sum of img[0..r, 0..c], no padding row / column
        rects, sums = [], []
        for _ in range(60):
            w, h = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
            x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
            rects.append([x, y, w, h])
            sums.append(int(integrate(ii, (y, x), (y + h - 1, x + w - 1))[0]))      # inclusive corners
        out["img_%d" % i] = img
        out["ii_%d" % i] = ii.astype(np.int64)
        out["rects_%d" % i] = np.array(rects, np.int32)
        out["sums_%d" % i] = np.array(sums, np.int64)
    out["n"] = np.int32(n)
    np.savez_compressed(os.path.join(HERE, "crosscheck_rects.npz"), **out)
    print("crosscheck_rects.npz:", n, "images")


if __name__ == "__main__":
    done = 0
    try:
        make_ccl(); done += 1
    except ImportError as e:
        print("scipy part skipped:", e)
    try:
        make_rects(); done += 1
    except ImportError as e:
        print("skimage part skipped:", e)
    sys.exit(0 if done else 1)
