"""diagnostic: the tracker on a frame of isolated moving pixels (more tile roots than the root list holds), step by step with prints"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nubomedia-vca_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from nubovca import capi
import orc
from test_gpu_tracker import moving_scene
W, H = 640, 480
ctx = capi.Context(0)
trk, otr = capi.Tracker(ctx), orc.Tracker()
seq = moving_scene(W, H, 4, 6, 77, noise=20)
dots = np.zeros((H, W, 4), np.uint8); dots[..., 3] = 255
lit = dots.copy(); lit[::2, ::2, :3] = 255
which = sys.argv[1] if len(sys.argv) > 1 else "all"
seq = seq[:2] + ([dots, lit, dots] if which != "plain" else []) + seq[2:]
for i, f in enumerate(seq):
    ts = 2000.0 + 33.3 * i
    print("frame", i, "...", flush=True)
    t0 = time.time()
    got = trk.process(f, ts)
    exp = otr.process(f, ts, cap=1 << 16)
    print("frame", i, len(got), len(exp), np.array_equal(np.asarray(got).reshape(-1, 4), np.asarray(exp).reshape(-1, 4)), round(time.time() - t0, 3), flush=True)
