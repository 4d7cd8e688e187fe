"""Soak: many ticks of the batched entry points with changing content; device memory and results must stay put.
usage (GPU box): python scripts/soak.py [ticks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ctx = capi.Context(0)
face_c = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
pc = {n: ctx.load_cascade_xml(synth.synthetic_part_cascade_xml(n)) for n in ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")}
V, W, H = 4, 1280, 720
faces = [capi.FaceStream(ctx, face_c, width_to_process=W, multi_scale_factor=10) for _ in range(V)]
kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
parts = [capi.PartStream(ctx, k, face_c, pc[a], pc[b] if b else None) for _ in range(V) for k, a, b in kinds]
trks = [capi.Tracker(ctx) for _ in range(V)]
N = 24
frames = [synth.make_bgr(W, H, 100 + i, "natural", [(100 + 13 * (i % 11), 80 + 7 * (i % 5), 200 + 9 * (i % 7)), (700, 300, 150 + 5 * (i % 13))]) for i in range(N)]
keep = [torch.from_numpy(f).cuda() for f in frames]
keep4 = [torch.cat([k, torch.full((H, W, 1), 255, dtype=torch.uint8, device="cuda")], dim=2).contiguous() for k in keep]
torch.cuda.synchronize()
fr = [capi.make_frame(k.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for k in keep]
fr4 = [capi.make_frame(k.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for k in keep4]


DO_PARTS, DO_TRK = os.environ.get("SOAK_PARTS", "1") == "1", os.environ.get("SOAK_TRK", "1") == "1"


def tick(i):
    fb = [fr[(i + 5 * v) % N] for v in range(V)]
    tk = ctx.face_batch_submit(faces, fb)
    res = capi.part_batch_process(ctx, parts, [fb[v] for v in range(V) for _ in range(4)]) if DO_PARTS else []
    boxes = ctx.face_batch_collect(tk)
    tr = capi.tracker_batch_process(ctx, trks, [fr4[(i + 5 * v) % N] for v in range(V)], [33.3 * i] * V, cap=256) if DO_TRK else []
    return sum(len(b) for b, _ in boxes), sum(len(a) + len(b) for a, b in res), sum(len(x) for x in tr)


for i in range(N):
    tick(i)
ctx.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
tot = [0, 0, 0]
first = None
for i in range(N, N + T):
    r = tick(i)
    tot = [a + b for a, b in zip(tot, r)]
    if i == 2 * N - 1:
        first = list(tot)
ctx.synchronize()
dt = time.time() - t0
free1 = torch.cuda.mem_get_info()[0]
print("soak: %d ticks x %d streams in %.1f s (%.0f frames/s), boxes %d, parts %d, tracker rects %d; device memory delta %.1f MB"
      % (T, V, dt, T * V / dt, tot[0], tot[1], tot[2], (free0 - free1) / 1e6))
assert abs(free0 - free1) < 64e6, "device memory grew"
