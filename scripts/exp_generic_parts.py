"""ROI chain with tree / tilted part cascades (what the real haarcascade_mcs_* / profileface files may hold): rate of the
batched call on 8 x 1080p streams x 4 detectors, next to the stump cascades of the same sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth
ctx = capi.Context(0)
W, H, V, N = 1920, 1080, 8, 16
base = [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)]
frames = [synth.make_bgr(W, H, 40 + i, "natural", [(x + 8 * (i % 16), y, s) for x, y, s in base]) for i in range(N)]
keep = [torch.from_numpy(f).cuda() for f in frames]
torch.cuda.synchronize()
fr = [capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in keep]
kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
sizes = {"righteye": (18, 12), "lefteye": (18, 12), "nose": (18, 15), "mouth": (25, 15), "leftear": (12, 20), "rightear": (12, 20)}
for label, gen in (("stump cascades", None), ("tree + tilted cascades", dict(tilt_frac=0.3, tree_frac=0.4))):
    if gen is None:
        face_c = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
        pc = {n: ctx.load_cascade_xml(synth.synthetic_part_cascade_xml(n)) for n in sizes}
    else:
        face_c = ctx.load_cascade_xml(synth.generic_cascade_xml(seed=11, stage_sizes=(3, 8, 12, 16, 20, 24, 28, 30), **gen))
        pc = {n: ctx.load_cascade_xml(synth.generic_cascade_xml(ow=s[0], oh=s[1], seed=20 + i, stage_sizes=(3, 6, 9, 12, 15, 18), **gen)) for i, (n, s) in enumerate(sizes.items())}
    parts = [capi.PartStream(ctx, k, face_c, pc[a], pc[b] if b else None) for _ in range(V) for k, a, b in kinds]
    def tick(i):
        fb = [fr[(i + 3 * v) % N] for v in range(V)]
        return sum(len(a) + len(b) for a, b in capi.part_batch_process(ctx, parts, [fb[v] for v in range(V) for _ in range(4)]))
    for i in range(4): tick(i)
    ctx.synchronize()
    t0 = time.perf_counter(); found = 0
    for i in range(24): found += tick(i)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %.0f frames/s (%.2f ms per tick of %d streams x 4 detectors), parts/frame %.2f" % (label, V * 24 / dt, dt / 24 * 1e3, V, found / (V * 24)))
    ctx.enable_kernel_timing(1)
    for i in range(8): tick(i)
    ctx.synchronize()
    kt = ctx.kernel_timing(); ctx.enable_kernel_timing(0)
    print("   kernel ms per tick:", {k: round(v[0] / 8, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1]}, "launch sets:", {k: v[1] / 8 for k, v in kt.items() if v[1]})
    for p in parts: p.close()
