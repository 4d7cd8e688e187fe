"""Randomised parity run: the HIP path against the CPU oracle on random configurations (frame sizes, properties,
host / device frames, synchronous / per-stream / submit-collect calls, the three detectMultiScale variants with random
size limits) for a given number of seconds.  TEST TOOL (it loads oracle/).  Usage: fuzz_parity.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nubomedia-vca_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
import orc
from nubovca import capi, synth

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0)
xml_full, xml_small = synth.synthetic_cascade_xml(), synth.synthetic_cascade_xml(stages=[3, 8, 12, 16, 20, 24])
cascs = [(ctx.load_cascade_xml(x), orc.parse_cascade_xml(x)) for x in (xml_full, xml_small)]
KW = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4", "multi_scale_factor": "scale_factor_pct",
      "min_neighbors": "min_neighbors"}
t_end = time.time() + SECS
rounds = {"face": 0, "detect": 0}
while time.time() < t_end:
    if rng.rand() < 0.5:
        # ---- face streams
        gc, oc = cascs[0]
        W, H = int(rng.randint(80, 360)), int(rng.randint(64, 260))
        ns, ticks = int(rng.randint(1, 6)), int(rng.randint(2, 6))
        props = []
        for _ in range(ns):
            p = {"width_to_process": int(W // rng.randint(1, 4)), "multi_scale_factor": int(rng.choice([10, 15, 20, 25, 40])),
                 "min_neighbors": int(rng.randint(1, 5)), "process_x_every_4_frames": int(rng.randint(1, 5))}
            props.append(p)
        gs = [capi.FaceStream(ctx, gc, **p) for p in props]
        os_ = [orc.FaceStream(oc, **{KW[k]: v for k, v in p.items()}) for p in props]
        mode = int(rng.randint(0, 3))
        frames = [[synth.make_bgr(W, H, int(rng.randint(1 << 30)), "natural", [(int(rng.randint(0, W // 2)), int(rng.randint(0, H // 3)), int(min(W, H) * rng.uniform(0.3, 0.6)))] if rng.rand() < 0.7 else [])
                   for _ in range(ns)] for _ in range(ticks)]
        keep = []
        def fr(t):
            out = []
            for i in range(ns):
                if (i + t) % 3 == 0:
                    d = torch.from_numpy(frames[t][i]).cuda(); keep.append(d)
                    out.append(capi.make_frame(d.data_ptr(), W, H, W * 3, capi.MEM_DEVICE))
                else:
                    out.append(capi.make_frame(frames[t][i]))
            torch.cuda.synchronize()
            return out
        got = []
        if mode == 0:
            for t in range(ticks): got.append(ctx.face_batch_process(gs, fr(t)))
        elif mode == 1:
            for t in range(ticks): got.append([gs[i].process(frames[t][i]) for i in range(ns)])
        else:
            pend = ctx.face_batch_submit(gs, fr(0))
            for t in range(1, ticks):
                nxt = ctx.face_batch_submit(gs, fr(t)); got.append(ctx.face_batch_collect(pend)); pend = nxt
            got.append(ctx.face_batch_collect(pend))
        for t in range(ticks):
            for i in range(ns):
                eb, eid = os_[i].process(frames[t][i])
                if not (np.array_equal(got[t][i][0], eb) and np.array_equal(got[t][i][1], eid)):
                    print("MISMATCH face", W, H, props[i], "mode", mode, "tick", t, got[t][i], eb, eid); sys.exit(1)
        for s in gs: s.close()
        rounds["face"] += 1
    else:
        # ---- detectMultiScale variants
        gc, oc = cascs[int(rng.randint(0, 2))]
        w, h = int(rng.randint(30, 420)), int(rng.randint(30, 300))
        sf = float(rng.choice([1.1, 1.2, 1.25, 1.5]))
        flags = int(rng.choice([0, capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT, capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_DO_ROUGH_SEARCH]))
        mn = int(rng.randint(0, 5))
        ms = (int(rng.randint(0, 40)), int(rng.randint(0, 40)))
        mx = (0, 0) if rng.rand() < 0.6 else (int(rng.randint(20, w + 1)), int(rng.randint(20, h + 1)))
        s = int(min(w, h) * rng.uniform(0.3, 0.9))
        faces = [(int(rng.randint(0, max(1, w - s))), int(rng.randint(0, max(1, h - s))), s)] if s >= 24 else []
        g = orc.equalize_hist(synth.make_gray(w, h, int(rng.randint(1 << 30)), str(rng.choice(["natural", "noise", "gradient"])), faces))
        a, b = ctx.detect_multiscale(gc, g, sf, mn, flags, ms, mx), orc.detect_multiscale(oc, g, sf, mn, flags, ms, mx)
        if not np.array_equal(a, b):
            print("MISMATCH detect", w, h, sf, mn, flags, ms, mx, a, b); sys.exit(1)
        rounds["detect"] += 1
print("rounds", rounds, "no mismatch")
