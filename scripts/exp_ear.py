import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth
ctx = capi.Context(0)
face_c = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
pc = {n: ctx.load_cascade_xml(synth.synthetic_part_cascade_xml(n)) for n in ("leftear", "rightear", "nose")}
kind = sys.argv[1] if len(sys.argv) > 1 else "ear"
p = capi.PartStream(ctx, 3, face_c, pc["leftear"], pc["rightear"]) if kind == "ear" else capi.PartStream(ctx, 1, face_c, pc["nose"], None)
W, H = 1920, 1080
base = [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)]
frames = [synth.make_bgr(W, H, 40 + i, "natural", [(x + 8 * i, y, s) for x, y, s in base]) for i in range(8)]
keep = [torch.from_numpy(f).cuda() for f in frames]
torch.cuda.synchronize()
fr = [capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in keep]
for i in range(8): p.process(fr[i % 8])
t0 = time.perf_counter()
for i in range(64): a, b = p.process(fr[i % 8])
dt = time.perf_counter() - t0
print(kind, "ms/frame", dt / 64 * 1e3, len(a), len(b))
ctx.enable_kernel_timing(1)
for i in range(16): p.process(fr[i % 8])
kt = ctx.kernel_timing()
print({k: (round(v[0] / 16, 4), v[1] / 16) for k, v in kt.items() if v[1]})
