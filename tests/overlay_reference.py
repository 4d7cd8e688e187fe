"""Reference for nvca_overlay_blend, written with numpy from the reference's own loop
(kms_face_detect_display_detections_overlay_img, FACE/kmsfacedetect.cpp:427-502) and cv::resize's 8-bit bilinear rules
 -- used by the CPU test (host frames) and the GPU test (device frames).  The resize for C channels is pinned
against the oracle's 1- and 3-channel resize in tests/test_overlay_cpu.py before it is trusted for 4 channels."""
import numpy as np


def resize_linear_cn(img, dw, dh):
    """cv::resize(img, (dw, dh), INTER_LINEAR) for uint8 images with any number of interleaved channels"""
    img = np.asarray(img, np.uint8)
    if img.ndim == 2:
        return resize_linear_cn(img[:, :, None], dw, dh)[:, :, 0]
    sh, sw, cn = img.shape
    if (sw, sh) == (dw, dh):
        return img.copy()
    scale_x, scale_y = sw / dw, sh / dh          # 1 / (dw / sw) in double
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if abs(scale_x - round(scale_x)) < np.finfo(np.float64).eps and abs(scale_y - round(scale_y)) < np.finfo(np.float64).eps and round(scale_x) == 2 and round(scale_y) == 2:
        s = img.astype(np.int32)
        return ((s[0:2 * dh:2, 0:2 * dw:2] + s[0:2 * dh:2, 1:2 * dw:2] + s[1:2 * dh:2, 0:2 * dw:2] + s[1:2 * dh:2, 1:2 * dw:2] + 2) >> 2).astype(np.uint8)
    dx = np.arange(dw)
    fx = ((dx + 0.5) * scale_x - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    neg = sx < 0
    fx[neg] = 0; sx[neg] = 0
    xmax = dw
    edge = sx + 1 >= sw
    if edge.any():
        xmax = int(np.argmax(edge))
    last = sx >= sw - 1
    fx[last] = 0; sx[last] = sw - 1
    a0 = np.clip(np.rint((np.float32(1) - fx) * np.float32(2048)), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint(fx * np.float32(2048)), -32768, 32767).astype(np.int64)
    inner = dx < xmax
    a0 = np.where(inner, a0, 2048); a1 = np.where(inner, a1, 0)
    dy = np.arange(dh)
    fy = ((dy + 0.5) * scale_y - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(np.float32)).astype(np.float32)
    b0 = np.clip(np.rint((np.float32(1) - fy) * np.float32(2048)), -32768, 32767).astype(np.int64)
    b1 = np.clip(np.rint(fy * np.float32(2048)), -32768, 32767).astype(np.int64)
    sy0 = np.clip(sy, 0, sh - 1); sy1 = np.clip(sy + 1, 0, sh - 1)
    s = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    h = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]          # horizontal pass of every source row
    h0, h1 = h[sy0], h[sy1]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def overlay_blend(frame, boxes, image, off_x=0.0, off_y=0.0, wp=1.0, hp=1.0):
    """in place on `frame` (H x W x 3 uint8)"""
    if wp == 0 or hp == 0:
        return frame
    H, W, _ = frame.shape
    image = np.asarray(image, np.uint8)
    cn = 1 if image.ndim == 2 else image.shape[2]
    for (bx, by, bw, bh) in np.asarray(boxes, np.int64).reshape(-1, 4):
        x = int(float(bx) + float(bw) * off_x)          # C: int = int + int * double (the SUM is truncated)
        y = int(float(by) + float(bh) * off_y)
        h = int(float(bh) * hp)
        w = int(float(bw) * wp)
        if w <= 0 or h <= 0:
            continue
        small = resize_linear_cn(image, w, h)
        if small.ndim == 2:
            small = small[:, :, None]
        for r in range(h):
            yy = r + y
            if yy < 0 or yy >= H:
                continue
            c0, c1 = max(0, -x), min(w, W - x)
            if c1 <= c0:
                continue
            src = small[r, c0:c1].astype(np.float64)
            dst = frame[yy, x + c0:x + c1]
            if cn == 1:
                dst[:] = small[r, c0:c1, :1]
            elif cn == 3:
                dst[:] = small[r, c0:c1]
            else:
                prop = src[:, 3:4] / 255.0
                overlay = 1.0 * prop
                original = 1 - overlay
                dst[:] = (src[:, :3] * overlay + dst.astype(np.float64) * original).astype(np.uint8)          # C cast: truncation
    return frame
