"""Randomised parity run: the HIP path against the CPU oracle on random configurations (frame sizes, properties,
host / device frames, synchronous / per-stream / submit-collect calls, the three detectMultiScale variants with random
size limits) for a given number of seconds.  Test tool (lives under tests/ because it loads oracle/).
Usage: python tests/fuzz_parity.py [seconds] [seed]; tests/test_gpu_fuzz.py runs a short burst of it."""
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
rounds = {"face": 0, "detect": 0, "tracker": 0, "parts": 0, "part_batch": 0, "generic": 0}
part_xml = {n: synth.synthetic_part_cascade_xml(n) for n in ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")}
part_dev = {n: ctx.load_cascade_xml(x) for n, x in part_xml.items()}
part_cpu = {n: orc.parse_cascade_xml(x) for n, x in part_xml.items()}
KINDS = {0: ("righteye", "lefteye"), 1: ("nose", None), 2: ("mouth", None), 3: ("leftear", "rightear")}
gen_cache = {}


def generic_pair(ow, oh, seed, tilt, tree):
    """a tree / tilted cascade on both sides, cached (parsing dominates otherwise)"""
    key = (ow, oh, seed, tilt, tree)
    if key not in gen_cache:
        if len(gen_cache) > 24:
            gen_cache.clear()
        x = synth.generic_cascade_xml(ow=ow, oh=oh, seed=seed, stage_sizes=(3, 6, 9, 12, 15), tilt_frac=tilt, tree_frac=tree)
        gen_cache[key] = (ctx.load_cascade_xml(x), orc.parse_cascade_xml(x))
    return gen_cache[key]


while time.time() < t_end:
    u = rng.rand()
    if u < 0.08:
        # ---- nvca_part_batch_process: streams of random kinds / properties, several of them on the same frame (host or device),
        # tree / tilted cascades now and then; every stream against its own oracle stream
        nv, ns = int(rng.randint(1, 4)), int(rng.randint(2, 9))
        W, H = int(rng.choice([320, 400, 480, 640])), int(rng.choice([240, 300, 360, 480]))
        generic = rng.rand() < 0.3
        face_pair = generic_pair(20, 20, int(rng.randint(3)), 0.3, 0.3) if generic else cascs[0]
        gps, ops, vid, evt = [], [], [], []
        names = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4", "multi_scale_factor": "scale_factor_pct", "detect_event": "detect_event"}
        for i in range(ns):
            kind = int(rng.randint(0, 4)); na, nbn = KINDS[kind]
            pr = {"process_x_every_4_frames": int(rng.randint(1, 5)), "multi_scale_factor": int(rng.choice([10, 25, 25, 40]))}
            if kind != 3 and rng.rand() < 0.3: pr["detect_event"] = 1
            if rng.rand() < 0.3: pr["width_to_process"] = int(rng.choice([160, 320, W]))
            if generic:
                sizes = {"righteye": (18, 12), "lefteye": (18, 12), "nose": (18, 15), "mouth": (25, 15), "leftear": (12, 20), "rightear": (12, 20)}
                pa = generic_pair(sizes[na][0], sizes[na][1], 40 + kind, 0.2, 0.4)
                pb = generic_pair(sizes[nbn][0], sizes[nbn][1], 50 + kind, 0.4, 0.2) if nbn else (None, None)
            else:
                pa, pb = (part_dev[na], part_cpu[na]), ((part_dev[nbn], part_cpu[nbn]) if nbn else (None, None))
            gps.append(capi.PartStream(ctx, kind, face_pair[0], pa[0], pb[0], **pr))
            ops.append(orc.PartStream(kind, face_pair[1], pa[1], pb[1], **{names[k]: v for k, v in pr.items()}))
            vid.append(int(rng.randint(nv))); evt.append(bool(pr.get("detect_event")))
        sfc = int(H * rng.uniform(0.35, 0.6))
        keep = []
        inflight = rng.rand() < 0.5            # nvca_part_batch_submit / _collect with two calls outstanding instead of one call per tick

        def make_tick():
            fset, handed = [], []
            for v in range(nv):
                faces = [(int(rng.randint(0, max(1, W - sfc))), int(rng.randint(0, max(1, H - sfc))), sfc)] if rng.rand() < 0.8 else []
                f = synth.make_bgr(W, H, int(rng.randint(1 << 30)), "natural", faces)
                fset.append((f, faces))
                if rng.rand() < 0.5:
                    d = torch.from_numpy(f).cuda(); keep.append(d)
                    handed.append(capi.make_frame(d.data_ptr(), W, H, W * 3, capi.MEM_DEVICE))
                else:
                    handed.append(capi.make_frame(f))
            torch.cuda.synchronize()
            pushes = []
            for i in range(ns):
                if evt[i] and rng.rand() < 0.8:
                    pushes.append((i, [(x, y, s_, s_) for (x, y, s_) in fset[vid[i]][1]]))
            return fset, handed, pushes

        def check(t, fset, pushes, res):
            for i, boxes in pushes:
                ops[i].push_faces(boxes)
            for i in range(ns):
                ea, eb = ops[i].process(fset[vid[i]][0])
                if not (np.array_equal(res[i][0], ea) and np.array_equal(res[i][1], eb)):
                    print("MISMATCH part batch", "in flight" if inflight else "", W, H, "generic" if generic else "stumps", "stream", i, "tick", t, res[i], ea, eb); sys.exit(1)
        nt = int(rng.randint(2, 5))
        pending = []                           # (tick, frame set, pushes, ticket)
        for t in range(nt):
            fset, handed, pushes = make_tick()
            for i, boxes in pushes:
                gps[i].push_faces(boxes)
            if inflight:
                pending.append((t, fset, pushes, capi.part_batch_submit(ctx, gps, [handed[vid[i]] for i in range(ns)])))
                if len(pending) == 2:
                    t0, f0, p0, tk0 = pending.pop(0)
                    check(t0, f0, p0, capi.part_batch_collect(ctx, tk0))
            else:
                check(t, fset, pushes, capi.part_batch_process(ctx, gps, [handed[vid[i]] for i in range(ns)]))
        for t0, f0, p0, tk0 in pending:
            check(t0, f0, p0, capi.part_batch_collect(ctx, tk0))
        if inflight: rounds["part_inflight"] = rounds.get("part_inflight", 0) + 1
        for g_ in gps: g_.close()
        rounds["part_batch"] += 1
    elif u < 0.16:
        # ---- detectMultiScale variants on tree / tilted cascades
        ow, oh = [(20, 20), (25, 15), (12, 20), (18, 12)][int(rng.randint(4))]
        gc, oc = generic_pair(ow, oh, int(rng.randint(4)), float(rng.choice([0.0, 0.2, 0.4])), float(rng.choice([0.0, 0.3, 0.5])))
        w, h = int(rng.randint(40, 360)), int(rng.randint(40, 260))
        sf = float(rng.choice([1.1, 1.2, 1.25]))
        flags = int(rng.choice([0, capi.HAAR_SCALE_IMAGE, capi.HAAR_FIND_BIGGEST_OBJECT, capi.HAAR_FIND_BIGGEST_OBJECT | capi.HAAR_DO_ROUGH_SEARCH]))
        mn = int(rng.randint(0, 4))
        g = orc.equalize_hist(synth.make_gray(w, h, int(rng.randint(1 << 30)), str(rng.choice(["natural", "noise", "gradient"]))))
        # a random cascade may let (nearly) every window through: the call re-runs its launch set with lists of the exact size
        # and must still return the reference's boxes (counted: how often the first set would not have fitted)
        a = ctx.detect_multiscale(gc, g, sf, mn, flags, (ow, oh), cap=1 << 16)
        if not (flags & capi.HAAR_FIND_BIGGEST_OBJECT) and len(orc.detect_raw(oc, g, sf, flags & capi.HAAR_SCALE_IMAGE, (ow, oh), cap=1 << 20)) > 16384:
            rounds["overflow_answered"] = rounds.get("overflow_answered", 0) + 1
        b = orc.detect_multiscale(oc, g, sf, mn, flags, (ow, oh), cap=1 << 16)
        if not np.array_equal(a, b):
            print("MISMATCH generic detect", ow, oh, w, h, sf, mn, flags, a, b); sys.exit(1)
        rounds["generic"] += 1
    elif u < 0.26:
        # ---- tracker: blobs moving over a noisy background, random parameters and sizes
        W, H = int(rng.choice([160, 200, 320, 322, 401])), int(rng.randint(90, 260))
        tp = {"set_threshold": int(rng.randint(5, 60)), "set_min_area": int(rng.randint(5, 200)), "set_max_area": int(rng.randint(500, 40000)),
              "set_distance": int(rng.randint(5, 80))}
        gt = capi.Tracker(ctx, **tp)
        ot = orc.Tracker(threshold=tp["set_threshold"], min_area=tp["set_min_area"], max_area=tp["set_max_area"], distance=tp["set_distance"])
        bg = rng.randint(0, 256, size=(H, W, 4)).astype(np.uint8)
        dense = rng.rand() < 0.25                       # now and then the whole scene changes from frame to frame
        nb = int(rng.randint(1, 5))
        blobs = [(rng.randint(0, W - 20), rng.randint(0, H - 20), rng.randint(6, 60), rng.randint(6, 60), rng.randint(-9, 10), rng.randint(-9, 10), rng.randint(0, 256)) for _ in range(nb)]
        for t in range(int(rng.randint(3, 9))):
            f = rng.randint(0, 256, size=(H, W, 4)).astype(np.uint8) if dense else bg.copy()
            if rng.rand() < 0.3:
                f[:, :, :3] = np.clip(f[:, :, :3].astype(int) + rng.randint(-3, 4), 0, 255).astype(np.uint8)
            for (x, y, w_, h_, vx, vy, c) in blobs:
                x0, y0 = int(np.clip(x + vx * t, 0, W - 2)), int(np.clip(y + vy * t, 0, H - 2))
                f[y0:min(H, y0 + h_), x0:min(W, x0 + w_), :3] = c
            ts = 1000.0 + 33.3 * t
            a, b = gt.process(f, ts, cap=1 << 14), ot.process(f, ts, cap=1 << 16)
            if not np.array_equal(a, b):
                print("MISMATCH tracker", W, H, tp, "tick", t, len(a), len(b)); sys.exit(1)
        gt.close()
        rounds["tracker"] += 1
    elif u < 0.32:
        # ---- part detectors (own face pass or detect-event mode)
        kind = int(rng.randint(0, 4)); na, nbn = KINDS[kind]
        W, H = int(rng.choice([320, 400, 480, 640])), int(rng.choice([240, 300, 360, 480]))
        pr = {"process_x_every_4_frames": int(rng.randint(1, 5)), "multi_scale_factor": int(rng.choice([10, 25, 40]))}
        if kind != 3 and rng.rand() < 0.5: pr["detect_event"] = 1
        if rng.rand() < 0.3: pr["width_to_process"] = int(rng.choice([160, 320, W]))
        names = {"width_to_process": "width_to_process", "process_x_every_4_frames": "process_x_every_4", "multi_scale_factor": "scale_factor_pct", "detect_event": "detect_event"}
        gp = capi.PartStream(ctx, kind, cascs[0][0], part_dev[na], part_dev[nbn] if nbn else None, **pr)
        op = orc.PartStream(kind, cascs[0][1], part_cpu[na], part_cpu[nbn] if nbn else None, **{names[k]: v for k, v in pr.items()})
        sfc = int(H * rng.uniform(0.35, 0.6))
        for t in range(int(rng.randint(2, 6))):
            faces = [(int(rng.randint(0, max(1, W - sfc))), int(rng.randint(0, max(1, H - sfc))), sfc)] if rng.rand() < 0.8 else []
            f = synth.make_bgr(W, H, int(rng.randint(1 << 30)), "natural", faces)
            if pr.get("detect_event"):
                boxes = [(x, y, s_, s_) for (x, y, s_) in faces]
                gp.push_faces(boxes); op.push_faces(boxes)
            (a1, b1), (a2, b2) = gp.process(f), op.process(f)
            if not (np.array_equal(a1, a2) and np.array_equal(b1, b2)):
                print("MISMATCH parts", kind, W, H, pr, "tick", t, a1, a2, b1, b2); sys.exit(1)
        gp.close()
        rounds["parts"] += 1
    elif u < 0.65:
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
