"""cost of a plan-cache miss: first call on a geometry vs the second (1) for whole frames, (2) for ROI-sized images in the
three detectMultiScale variants (the part detectors meet new ROI sizes all the time)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np
from nubovca import capi, synth
ctx = capi.Context(0); casc = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
for (W, H, w2p) in [(1920, 1080, 1920), (1280, 720, 1280), (640, 480, 160)]:
    fs = capi.FaceStream(ctx, casc, width_to_process=w2p, multi_scale_factor=10)
    f = synth.make_bgr(W, H, 1, "noise")
    t0 = time.perf_counter(); fs.process(f); t1 = time.perf_counter(); fs.process(f); t2 = time.perf_counter()
    print(W, H, w2p, 'first call %.1f ms, second %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
rng = np.random.RandomState(3)
for name, flags, ms in [("scale-cascade", 0, (0, 0)), ("SCALE_IMAGE", capi.HAAR_SCALE_IMAGE, (20, 20)), ("FIND_BIGGEST", capi.HAAR_FIND_BIGGEST_OBJECT, (1, 1))]:
    first, second = [], []
    for k in range(12):
        w, h = int(rng.randint(90, 200)), int(rng.randint(60, 140))
        g = synth.make_gray(w, h, k, "natural")
        t0 = time.perf_counter(); ctx.detect_multiscale(casc, g, 1.1, 2, flags, ms); t1 = time.perf_counter()
        ctx.detect_multiscale(casc, g, 1.1, 2, flags, ms); t2 = time.perf_counter()
        first.append((t1 - t0) * 1e3); second.append((t2 - t1) * 1e3)
    print('%-14s ROI ~150x100: new size %.2f ms, cached %.2f ms' % (name, np.median(first), np.median(second)))
