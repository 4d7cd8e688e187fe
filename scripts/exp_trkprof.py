import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import numpy as np, torch
from nubovca import capi, synth
ctx = capi.Context(0)
V, W, H, N = 4, 1280, 720, 12
trks = [capi.Tracker(ctx) for _ in range(V)]
frames = [synth.make_bgr(W, H, 100 + i, "natural", [(100 + 13 * (i % 11), 80, 200)]) for i in range(N)]
keep4 = [torch.cat([torch.from_numpy(f).cuda(), torch.full((H, W, 1), 255, dtype=torch.uint8, device="cuda")], dim=2).contiguous() for f in frames]
torch.cuda.synchronize()
fr4 = [capi.make_frame(k.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for k in keep4]
def tick(i):
    return capi.tracker_batch_process(ctx, trks, [fr4[(i + 5 * v) % N] for v in range(V)], [33.3 * i] * V, cap=4096)
for i in range(5): tick(i)
ctx.synchronize()
t0 = time.time()
for i in range(5, 45): r = tick(i)
ctx.synchronize(); dt = time.time() - t0
print("wall ms/tick", dt / 40 * 1e3, "rects", [len(x) for x in r])
ctx.enable_kernel_timing(1)
for i in range(45, 65): tick(i)
ctx.synchronize()
kt = ctx.kernel_timing()
print({k: (round(v[0] / 20, 3), v[1] / 20) for k, v in kt.items() if v[1]})
ctx.set_option("host_profile", 1)
# determinism: two tracker sets, same frames and timestamps -> identical rect lists every tick
A = [capi.Tracker(ctx) for _ in range(V)]
B = [capi.Tracker(ctx) for _ in range(V)]
bad = 0
for i in range(30):
    fa = capi.tracker_batch_process(ctx, A, [fr4[(i + 5 * v) % N] for v in range(V)], [33.3 * i] * V, cap=4096)
    fb = capi.tracker_batch_process(ctx, B, [fr4[(i + 5 * v) % N] for v in range(V)], [33.3 * i] * V, cap=4096)
    for x, y in zip(fa, fb):
        if not np.array_equal(x, y):
            bad += 1
print("determinism: differing (tick, stream) pairs:", bad)
sys.path.insert(0, os.path.join(ROOT, 'oracle')); import orc
ot = orc.Tracker()
gt = capi.Tracker(ctx)
worst = 0
for i in range(6):
    f = keep4[i % N].cpu().numpy()
    e = ot.process(f, 33.3 * i)
    g = capi.tracker_batch_process(ctx, [gt], [fr4[i % N]], [33.3 * i], cap=4096)[0]
    print("oracle vs gpu tick", i, len(e), len(g), np.array_equal(np.asarray(e).reshape(-1, 4), np.asarray(g).reshape(-1, 4)))
