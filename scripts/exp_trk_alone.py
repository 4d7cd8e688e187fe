"""NuboTracker alone on 8 x 1080p (BASELINE configs[4]'s tracker half): wall time per tick; run under rocprofv3 --kernel-trace --stats for
the per-kernel durations without a face batch beside them.  usage (GPU box): python3 scripts/exp_trk_alone.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth
W, H, S, T = 1920, 1080, 8, 4
ctx = capi.Context(0)
bgs = [synth.make_gray(W, H, 9000 + s, "natural") for s in range(S)]
rows4 = []
for t in range(T):
    row = []
    for s in range(S):
        faces = [(200 + 8 * t + 16 * (s % 7), 150, 300), (900 + 8 * t, 400, 180)]
        x = torch.from_numpy(synth.gray_to_bgr(synth.paste_faces(bgs[s], faces, s), 9000 + s)).cuda()
        row.append(torch.cat([x, torch.full((H, W, 1), 255, dtype=torch.uint8, device="cuda")], dim=2).contiguous())
    rows4.append(row)
torch.cuda.synchronize()
trk = [capi.Tracker(ctx) for _ in range(S)]
fr4 = [[capi.make_frame(x.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for x in row] for row in rows4]
for i in range(8):
    capi.tracker_batch_process(ctx, trk, fr4[i % T], [33.3 * i] * S, cap=256)
K = 60
t0 = time.perf_counter()
n = 0
for i in range(8, 8 + K):
    n += sum(len(x) for x in capi.tracker_batch_process(ctx, trk, fr4[i % T], [33.3 * i] * S, cap=256))
dt = time.perf_counter() - t0
print(json.dumps({"tracker_alone_ms_per_tick": dt / K * 1e3, "frames_per_s": S * K / dt, "rects_per_frame": n / (S * K)}))
