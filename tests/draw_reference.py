"""Reference for nvca_draw_shapes written from the rule include/nubovca.h states (used by the CPU and the GPU test): the
rectangle as the union of its four 3-pixel edge bands plus the 4-neighbourhood of each vertex, the circle as the ring of
distances [r - 2, r + 2]."""
import numpy as np


def _ref(img, shapes):
    H, W, C = img.shape
    yy, xx = np.mgrid[0:H, 0:W]
    for kind, x, y, w, h, col in shapes:
        if kind == 1:
            if w < 0:
                continue
            d2 = (xx - x).astype(np.int64) ** 2 + (yy - y).astype(np.int64) ** 2
            m = (d2 <= (w + 2) ** 2) & (d2 >= max(w - 2, 0) ** 2)
        else:
            x0, x1 = sorted((x, x + w))
            y0, y1 = sorted((y, y + h))
            m = np.zeros((H, W), bool)

            def span(ax0, ax1, ay0, ay1):
                m[max(ay0, 0):max(ay1 + 1, 0), max(ax0, 0):max(ax1 + 1, 0)] = True
            span(x0, x1, y0 - 1, y0 + 1); span(x0, x1, y1 - 1, y1 + 1)
            span(x0 - 1, x0 + 1, y0, y1); span(x1 - 1, x1 + 1, y0, y1)
            for cx, cy in ((x0, y0), (x1, y0), (x1, y1), (x0, y1)):
                for dx, dy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    if 0 <= cx + dx < W and 0 <= cy + dy < H:
                        m[cy + dy, cx + dx] = True
        img[m] = np.asarray(col[:C], np.uint8)
    return img


def _shapes(rng, W, H, n):
    out = []
    for i in range(n):
        col = tuple(int(v) for v in rng.integers(0, 256, 4))
        if i % 3 == 2:
            out.append((1, int(rng.integers(-20, W + 20)), int(rng.integers(-20, H + 20)), int(rng.integers(-2, H // 2)), 0, col))
        else:
            out.append((0, int(rng.integers(-30, W + 10)), int(rng.integers(-30, H + 10)), int(rng.integers(-W // 2, W)), int(rng.integers(-H // 2, H)), col))
    return out
