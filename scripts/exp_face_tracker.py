"""BASELINE configs[4] on one GPU (8 x 1080p streams through NuboFaceDetector + NuboTracker per tick): where a tick's wall time goes.
usage (GPU box): python3 scripts/exp_face_tracker.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth

W, H, S, T = 1920, 1080, 8, 4
ctx = capi.Context(0)
casc = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
bgs = [synth.make_gray(W, H, 9000 + s, "natural") for s in range(S)]
rows, rows4 = [], []
for t in range(T):
    row = []
    for s in range(S):
        faces = [(200 + 8 * t + 16 * (s % 7), 150, 300), (900 + 8 * t, 400, 180)]
        row.append(torch.from_numpy(synth.gray_to_bgr(synth.paste_faces(bgs[s], faces, s), 9000 + s)).cuda())
    rows.append(row)
    rows4.append([torch.cat([x, torch.full((H, W, 1), 255, dtype=torch.uint8, device="cuda")], dim=2).contiguous() for x in row])
torch.cuda.synchronize()
sts = [capi.FaceStream(ctx, casc, width_to_process=W, multi_scale_factor=25) for _ in range(S)]
trk = [capi.Tracker(ctx) for _ in range(S)]
frs = [[capi.make_frame(x.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for x in row] for row in rows]
fr4 = [[capi.make_frame(x.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for x in row] for row in rows4]
infl = [ctx.face_batch_submit(sts, frs[0])]
acc = [0.0, 0.0, 0.0]
nrect = 0


def tick(i, timed):
    global nrect
    t0 = time.perf_counter()
    nxt = ctx.face_batch_submit(sts, frs[(i + 1) % T])
    t1 = time.perf_counter()
    r = capi.tracker_batch_process(ctx, trk, fr4[i % T], [33.3 * i] * S, cap=256)
    t2 = time.perf_counter()
    ctx.face_batch_collect(infl[0], cap=64)
    t3 = time.perf_counter()
    infl[0] = nxt
    if timed:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
        nrect += sum(len(x) for x in r)


for i in range(8):
    tick(i, False)
K = 40
t0 = time.perf_counter()
for i in range(8, 8 + K):
    tick(i, True)
dt = time.perf_counter() - t0
print(json.dumps({"frames_per_s": S * K / dt, "ms_per_tick": dt / K * 1e3, "ms": {"face_submit": acc[0] / K * 1e3, "tracker_call": acc[1] / K * 1e3, "face_collect": acc[2] / K * 1e3},
                  "tracker_rects_per_frame": nrect / (S * K)}))
# trackers alone, and face detectors alone (serving loop)
t0 = time.perf_counter()
for i in range(K):
    capi.tracker_batch_process(ctx, trk, fr4[i % T], [33.3 * (100 + i)] * S, cap=256)
dtt = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(K):
    nxt = ctx.face_batch_submit(sts, frs[(i + 1) % T]); ctx.face_batch_collect(infl[0], cap=64); infl[0] = nxt
dtf = time.perf_counter() - t0
print(json.dumps({"tracker_alone_ms_per_tick": dtt / K * 1e3, "face_alone_ms_per_tick": dtf / K * 1e3}))
ctx.enable_kernel_timing(1)
for i in range(8):
    tick(200 + i, False)
ctx.synchronize()
kt = ctx.kernel_timing()
print(json.dumps({"kernel_ms_per_tick": {k: round(v[0] / 8, 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1]}}))
