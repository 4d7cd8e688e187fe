"""k_roi's phase sums over the batched ROI chain (8 video streams x 4 part detectors, as scripts/bench_roi_chain.py's pipelined tick).
Diagnostic build only:  NVCA_BUILD_STAMPS=1 python nubomedia-vca_amd/build.py --force  (keep it out of the shipped library: copy it to
nubomedia-vca_amd/variants/stamps.so and rebuild), then on the GPU box
    NVCA_LIB=$PWD/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=gpurun_out/x/stamps python3 scripts/exp_roi_stamps.py
writes gpurun_out/x/stamps.roi.txt when the context synchronises."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nubomedia-vca_amd"))
import torch
from nubovca import capi, synth

W, H, N, V = 1920, 1080, 16, 8
ctx = capi.Context(0)
STANDIN = "--standin" in sys.argv          # rounds 1-3's uncalibrated cascades
face_c = ctx.load_cascade_xml(synth.synthetic_cascade_xml() if STANDIN else synth.calibrated_cascade_xml())
part_xml = synth.synthetic_part_cascade_xml if STANDIN else synth.calibrated_part_cascade_xml
pc = {n: ctx.load_cascade_xml(part_xml(n)) for n in ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")}
base = [(150, 200, 560), (1100, 260, 620)]
frames = [synth.make_bgr(W, H, 40 + i, "natural", [(x + 8 * (i % 16), y, s) for x, y, s in base]) for i in range(N)]
keep = [torch.from_numpy(f).cuda() for f in frames]
torch.cuda.synchronize()
fr = [capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in keep]
faces_v = [capi.FaceStream(ctx, face_c, width_to_process=W, multi_scale_factor=10) for _ in range(V)]
kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
flat = [capi.PartStream(ctx, k, face_c, pc[a], pc[b] if b else None) for _ in range(V) for k, a, b in kinds]
found = 0
for i in range(24):
    fb = [fr[(i + 3 * v) % N] for v in range(V)]
    tk = ctx.face_batch_submit(faces_v, fb)
    res = capi.part_batch_process(ctx, flat, [fb[v] for v in range(V) for _ in range(4)])
    ctx.face_batch_collect(tk)
    found += sum(len(a) + len(b) for a, b in res)
ctx.synchronize()
print("ticks 24, parts per frame %.2f" % (found / (24 * V)))
