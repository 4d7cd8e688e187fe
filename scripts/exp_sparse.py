"""bench.py's roi_chain_sparse workload alone (8 x 1080p, four small faces a frame, uncalibrated part cascades): ms per tick + host phase stats"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth
V, ticks, reps = 8, 4, 6
base = [(150, 200, 560), (1100, 260, 620)] if os.environ.get("BIG") else [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)]
ctx = capi.Context(0)
xml_face = synth.calibrated_cascade_xml()
casc = ctx.load_cascade_xml(xml_face)
names = ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")
kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
pcs = {nm: ctx.load_cascade_xml((synth.calibrated_part_cascade_xml if os.environ.get("BIG") else synth.synthetic_part_cascade_xml)(nm)) for nm in names}
keep = [[torch.from_numpy(synth.make_bgr(1920, 1080, 40 + 5 * t + v, "natural", [(x + 8 * t + 6 * v, y + 3 * v, sz + 4 * ((t + v) % 3)) for x, y, sz in base])).cuda() for v in range(V)] for t in range(ticks)]
torch.cuda.synchronize()
frs = [[capi.make_frame(x.data_ptr(), 1920, 1080, 1920 * 3, capi.MEM_DEVICE) for x in row] for row in keep]
fcs = [capi.FaceStream(ctx, casc, width_to_process=1920, multi_scale_factor=10) for _ in range(V)]
parts = [capi.PartStream(ctx, k, casc, pcs[a], pcs[b] if b else None) for _ in range(V) for k, a, b in kinds]
found = 0
def tick(i):
    global found
    fb = frs[i % ticks]
    tk = ctx.face_batch_submit(fcs, fb)
    res = capi.part_batch_process(ctx, parts, [f for f in fb for _ in range(4)])
    ctx.face_batch_collect(tk)
    found += sum(len(a) + len(b) for a, b in res)
for i in range(2 * ticks):
    tick(i)
ctx.synchronize(); found = 0
t0 = time.perf_counter()
for i in range(reps * ticks):
    tick(i)
ctx.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"sparse_ms_per_tick": dt / (reps * ticks) * 1e3, "frames_per_s": V * reps * ticks / dt, "parts_per_frame": found / (V * reps * ticks)}))
