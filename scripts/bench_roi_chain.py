"""BASELINE config 3: 1080p face -> eye + nose + mouth + ear ROI chain on one stream, frames resident in
HBM; the part detectors run in detect-event mode on the face boxes of the same frame (the reference's
`nubofacedetector ! nuboeyedetector detect-event=1 ! ...` pipeline).  Prints frames/s and the per-element cost."""
import json, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth

W, H, N = 1920, 1080, 16
ctx = capi.Context(0)
# calibrated cascades (bench.py's since round 4: every early stage rejects about half -- face -- or a third -- parts -- of what reaches it);
# --standin: rounds 1-3's uncalibrated ones
STANDIN = "--standin" in sys.argv
FACE_XML = synth.synthetic_cascade_xml() if STANDIN else synth.calibrated_cascade_xml()
PART_XML = synth.synthetic_part_cascade_xml if STANDIN else synth.calibrated_part_cascade_xml
face_c = ctx.load_cascade_xml(FACE_XML)
pc = {n: ctx.load_cascade_xml(PART_XML(n)) for n in ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")}
face = capi.FaceStream(ctx, face_c, width_to_process=W, multi_scale_factor=10)
parts = {"eye": capi.PartStream(ctx, 0, face_c, pc["righteye"], pc["lefteye"], detect_event=1),
         "nose": capi.PartStream(ctx, 1, face_c, pc["nose"], None, detect_event=1),
         "mouth": capi.PartStream(ctx, 2, face_c, pc["mouth"], None, detect_event=1),
         "ear": capi.PartStream(ctx, 3, face_c, pc["leftear"], pc["rightear"], detect_event=1)}
base = [(150, 200, 560), (1100, 260, 620)]          # as bench.py's roi_chain: faces whose parts reach the part cascades' windows on the 320-pixel working image
JITTER = "--jitter" in sys.argv          # faces change size from frame to frame: the part detectors meet new ROI sizes all the time
N = 48 if JITTER else N
frames = [synth.make_bgr(W, H, 40 + i, "natural", [(x + 8 * (i % 16), y, s + ((7 * i) % 23 if JITTER else 0)) for x, y, s in base]) for i in range(N)]
keep = [torch.from_numpy(f).cuda() for f in frames]
torch.cuda.synchronize()
fr = [capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in keep]
acc = {k: 0.0 for k in ["face"] + list(parts)}


def step(i, timed):
    t0 = time.perf_counter()
    boxes, _ = face.process(fr[i % N]) if False else ctx.face_batch_process([face], [fr[i % N]])[0]
    t1 = time.perf_counter()
    if timed:
        acc["face"] += t1 - t0
    found = 0
    for k, p in parts.items():
        t0 = time.perf_counter()
        p.push_faces(boxes)
        a, b = p.process(fr[i % N])
        found += len(a) + len(b)
        if timed:
            acc[k] += time.perf_counter() - t0
    return len(boxes), found


for i in range(4):
    step(i, False)
K = 48
t0 = time.perf_counter()
tot = [0, 0]
for i in range(K):
    nb, nf = step(i, True)
    tot[0] += nb; tot[1] += nf
ctx.synchronize()
dt = time.perf_counter() - t0
print(("roi chain 1080p, face sizes changing every frame: " if JITTER else "roi chain 1080p: ") + "%.1f frames/s (%.3f ms/frame); faces/frame %.2f, parts/frame %.2f; per element ms: %s" %
      (K / dt, dt / K * 1e3, tot[0] / K, tot[1] / K, {k: round(v / K * 1e3, 3) for k, v in acc.items()}))


# ---- the same chain batched: V video streams x (face detector + the four part detectors, own face pass each) per tick,
# one nvca_face_batch_process + one nvca_part_batch_process call per tick (BASELINE config 3 on several concurrent streams)
V = 8
for a in sys.argv:
    if a.startswith("--streams="):
        V = int(a.split("=")[1])
faces_v = [capi.FaceStream(ctx, face_c, width_to_process=W, multi_scale_factor=10) for _ in range(V)]
kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
parts_v = [[capi.PartStream(ctx, k, face_c, pc[a], pc[b] if b else None) for k, a, b in kinds] for _ in range(V)]
flat = [p for row in parts_v for p in row]


def tick(i):
    fb = [fr[(i + 3 * v) % N] for v in range(V)]
    ctx.face_batch_process(faces_v, fb)
    res = capi.part_batch_process(ctx, flat, [fb[v] for v in range(V) for _ in range(4)])
    return sum(len(a) + len(b) for a, b in res)


def tick_pipelined(i):
    """what a pipeline with a queue between the face detector and the part detectors does: the face detector's batch is
    queued (nvca_face_batch_submit), the part detectors' call runs meanwhile, then the face results are collected"""
    fb = [fr[(i + 3 * v) % N] for v in range(V)]
    tk = ctx.face_batch_submit(faces_v, fb)
    res = capi.part_batch_process(ctx, flat, [fb[v] for v in range(V) for _ in range(4)])
    ctx.face_batch_collect(tk)
    return sum(len(a) + len(b) for a, b in res)


# detect-event chain: the part detectors take the faces the face detector published (no face pass of their own); the face
# detector works on tick i + 1 while they work on tick i
faces_e = [capi.FaceStream(ctx, face_c, width_to_process=W, multi_scale_factor=10) for _ in range(V)]
parts_e = [capi.PartStream(ctx, k, face_c, pc[a], pc[b] if b else None, detect_event=1) for _ in range(V) for k, a, b in kinds]
state = {}


def tick_event(i):
    fb = [fr[(i + 3 * v) % N] for v in range(V)]
    if "boxes" not in state:
        state["boxes"] = [b for b, _ in ctx.face_batch_process(faces_e, fb)]
    nxt = [fr[(i + 1 + 3 * v) % N] for v in range(V)]
    tk = ctx.face_batch_submit(faces_e, nxt)
    for v in range(V):
        for j in range(4):
            parts_e[4 * v + j].push_faces(state["boxes"][v])
    res = capi.part_batch_process(ctx, parts_e, [fb[v] for v in range(V) for _ in range(4)])
    state["boxes"] = [b for b, _ in ctx.face_batch_collect(tk)]
    return sum(len(a) + len(b) for a, b in res)


# where a pipelined tick's wall time goes (the three calls timed separately over a few ticks)
for i in range(3):
    tick_pipelined(i)
acc3 = [0.0, 0.0, 0.0]
for i in range(3, 15):
    fb = [fr[(i + 3 * v) % N] for v in range(V)]
    t0 = time.perf_counter()
    tk = ctx.face_batch_submit(faces_v, fb)
    t1 = time.perf_counter()
    res = capi.part_batch_process(ctx, flat, [fb[v] for v in range(V) for _ in range(4)])
    t2 = time.perf_counter()
    ctx.face_batch_collect(tk)
    t3 = time.perf_counter()
    acc3[0] += t1 - t0; acc3[1] += t2 - t1; acc3[2] += t3 - t2
print(json.dumps({"tick_breakdown_ms": {"face_batch_submit": acc3[0] / 12 * 1e3, "part_batch_process": acc3[1] / 12 * 1e3, "face_batch_collect": acc3[2] / 12 * 1e3}}))

K2 = 24
for name, fn in (("pipelined", tick_pipelined), ("detect_event_pipelined", tick_event)):
    for i in range(3):
        fn(i)
    t0 = time.perf_counter()
    found = 0
    for i in range(3, 3 + K2):
        found += fn(i)
    ctx.synchronize()
    dtv = time.perf_counter() - t0
    print(json.dumps({"roi_chain_batched_" + name: {"video_streams": V, "frames_per_s": V * K2 / dtv, "ms_per_tick": dtv / K2 * 1e3, "parts_per_frame": found / (V * K2)}}))

for i in range(3):
    tick(i)
t0 = time.perf_counter()
found = 0
for i in range(K2):
    found += tick(i)
ctx.synchronize()
dt2 = time.perf_counter() - t0
# a second, separate pass with an event pair around every launch (they serialise the launches: not part of the rate above)
ctx.enable_kernel_timing(1)
for i in range(K2):
    tick(i)
ctx.synchronize()
kt = ctx.kernel_timing()
ctx.enable_kernel_timing(0)
busy_ms = sum(v[0] for v in kt.values())
# algorithmic bytes of the chain per video frame (SURVEY.md 8d formula applied to each working image): the face detector's
# 60.21 MB plus, per part detector, BGR in + gray out/in + the small working images' integrals
alg = 60.21e6 + 4 * (3 * W * H + 2 * W * H) + 4 * 25 * (320 * 180 + 160 * 90)
print(json.dumps({"roi_chain_batched": {"video_streams": V, "frames_per_s": V * K2 / dt2, "ms_per_tick": dt2 / K2 * 1e3, "parts_per_frame": found / (V * K2),
      "gpu_kernel_ms_per_tick": busy_ms / K2, "gpu_busy_frac": busy_ms / (dt2 * 1e3),
      "kernel_ms_per_tick": {k: round(v[0] / K2, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1]},
      "launch_sets_per_tick": {k: round(v[1] / K2, 1) for k, v in kt.items() if v[1]},
      "roofline": {"bound": "hbm", "achieved": alg * V * K2 / dt2 / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg * V * K2 / dt2 / 1e9 / 8000.0,
                   "note": "launch-bound small-image work: %d kernel launches per tick" % int(sum(v[1] for v in kt.values()) / K2)}}}))


# ---- the same workload with TWO part batches in flight (nvca_part_batch_submit / _collect): tick k + 1's gates, working images and face
# passes are queued before tick k's searches are collected -- what a serving loop does; results are tick k's when collect(k) returns
def run_inflight():
    """frame sets repeat every N ticks: a prepared batch per set (the ctypes marshalling of 32 streams and frames is a third of a
    millisecond a tick in Python, which is the harness, not the library)"""
    found = 0
    pbs, fbs = [], []
    for i in range(N):
        fb = [fr[(i + 3 * v) % N] for v in range(V)]
        pbs.append(capi.PreparedPartBatch(ctx, flat, [fb[v] for v in range(V) for _ in range(4)]))
        fbs.append(fb)
    for i in range(4):
        tick_pipelined(i)
    ctx.synchronize()
    K3 = 4 * K2
    for mode in ("one call per tick", "two in flight"):
        found = 0
        ctx.synchronize()
        t0 = time.perf_counter()
        if mode == "two in flight":
            pbs[0].submit(); ft = ctx.face_batch_submit(faces_v, fbs[0])
            for i in range(K3):
                nxt = None
                if i + 1 < K3:
                    pbs[(i + 1) % N].submit(); nxt = ctx.face_batch_submit(faces_v, fbs[(i + 1) % N])
                pbs[i % N].collect()
                ctx.face_batch_collect(ft)
                found += pbs[i % N].found()
                ft = nxt
        else:
            for i in range(K3):
                ft = ctx.face_batch_submit(faces_v, fbs[i % N])
                pbs[i % N].process()
                ctx.face_batch_collect(ft)
                found += pbs[i % N].found()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"roi_chain_batched_prepared, " + mode: {"video_streams": V, "frames_per_s": V * K3 / dt, "ms_per_tick": dt / K3 * 1e3, "parts_per_frame": found / (V * K3)}}))


run_inflight()


# ---- the same V video streams spread over C contexts on the one GPU, one host thread per context (what the GStreamer shim's
# per-GPU frontend does with NVCA_VIRTUAL_GPUS): the launch-bound chains of different contexts are queued in parallel
import threading
for C in (() if "--no-contexts" in sys.argv else (2, 4)):
    if V % C:
        continue
    ctxs = [capi.Context(0) for _ in range(C)]
    per = V // C
    setups = []
    for cx in ctxs:
        fc = cx.load_cascade_xml(FACE_XML)
        pcs = {n: cx.load_cascade_xml(PART_XML(n)) for n in ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")}
        fv = [capi.FaceStream(cx, fc, width_to_process=W, multi_scale_factor=10) for _ in range(per)]
        pv = [capi.PartStream(cx, k, fc, pcs[a], pcs[b] if b else None) for _ in range(per) for k, a, b in kinds]
        setups.append((cx, fv, pv))

    def work(ci, ticks):
        cx, fv, pv = setups[ci]
        for i in range(ticks):
            fb = [fr[(i + 3 * (ci * per + v)) % N] for v in range(per)]
            tk = cx.face_batch_submit(fv, fb)             # as tick_pipelined: the face detector's batch under the part detectors' call
            capi.part_batch_process(cx, pv, [fb[v] for v in range(per) for _ in range(4)])
            cx.face_batch_collect(tk)

    def run(ticks):
        th = [threading.Thread(target=work, args=(ci, ticks)) for ci in range(C)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    run(3)
    t0 = time.perf_counter()
    run(K2)
    for cx, _, _ in setups:
        cx.synchronize()
    dt3 = time.perf_counter() - t0
    print(json.dumps({"roi_chain_batched_contexts": {"contexts": C, "video_streams": V, "frames_per_s": V * K2 / dt3, "ms_per_tick": dt3 / K2 * 1e3}}))
    for cx, _, _ in setups:
        cx.close()
