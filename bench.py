#!/usr/bin/env python3
"""bench.py -- NuboFaceDetector Haar hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): NuboFaceDetector, one 1920x1080 BGR stream per
GPU, full-resolution mode (width-to-process = 1920), scaleFactor 1.1, minNeighbors 3,
minSize (96,54): 25 scales, <= 355 162 windows per frame.  A "step" is one pass of
the hot path (resize/gray -> equalizeHist -> integral -> cascade -> groupRectangles
-> track_faces) over one batch of F consecutive frames of the stream, frames already
resident in HBM.  N > 1: one process per GPU, one stream per rank (streams are the
independent units), no data-path collective; one RCCL all_gather of the fixed-size
box table per step (the result gather).  value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "nubomedia-vca_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

torch = None            # imported by the worker only: the launcher parent of `--gpus N` touches neither torch nor the GPU
dist = None

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MAX_BOXES = 64


def algorithmic_bytes(W, H, w, h, n_boxes=0):
    """SURVEY.md 8d, per frame, split by kernel group."""
    gray = 3 * W * H + w * h                           # BGR in, gray out
    integral = w * h + 12 * (w + 1) * (h + 1)          # gray in, sum i32 + sqsum 8 B out
    cascade = 12 * (w + 1) * (h + 1) + 16 * n_boxes    # integral pair read once, boxes out
    return {"gray_resize_hist": gray, "integral": integral, "cascade": cascade, "tracker": 15 * W * H,
            "total": gray + integral + cascade}


def source_hash():
    """sha256 (16 hex digits) over the kernel and host sources of the library: what profiles/pmc_traffic.json was collected on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nubomedia-vca_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(xml, frames_np, params, budget_s=12.0, max_frames=48):
    """The CPU oracle (a restatement of the reference's OpenCV-2.4 path, NOT OpenCV itself) timed on this box's host
    cores on a bounded sample of the same workload: one stream per thread (the reference runs one element per streaming
    thread; streams are the independent units), all available cores; the single-thread rate is reported alongside."""
    import threading
    if os.path.join(ROOT, "oracle") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    oc = orc.parse_cascade_xml(xml)
    kw = dict(width_to_process=params["width_to_process"], scale_factor_pct=params["multi_scale_factor"])
    orc.FaceStream(oc, **kw).process(frames_np[0])                    # warm caches / page in

    def run(nthreads, budget, cap):
        counts = [0] * nthreads
        t0 = time.perf_counter()

        def work(k):
            s = orc.FaceStream(oc, **kw)
            while counts[k] < cap and (time.perf_counter() - t0) < budget:
                s.process(frames_np[(counts[k] + 7 * k) % len(frames_np)])       # ctypes releases the GIL inside the call
                counts[k] += 1
        th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return sum(counts), time.perf_counter() - t0

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    n1, dt1 = run(1, budget_s * 0.4, max_frames // 2)
    nall, dtall = run(cores, budget_s * 0.6, max(2, max_frames // 4)) if cores > 1 else (n1, dt1)
    return {"value": nall / dtall, "unit": "frames/s", "cores": cores, "kind": "port", "single_core_value": n1 / dt1,
            "sample": "%d frames on %d threads in %.1f s (one stream per thread) and %d frames on 1 thread in %.1f s, same 1080p "
                      "full-res workload; CPU restatement of the reference's OpenCV-2.4 path (OpenCV itself is not "
                      "available offline)" % (nall, cores, dtall, n1, dt1)}


def oracle_expected(xml, frames_np, params, rows, multi_stream):
    """What the oracle says the LAST step's boxes must be.  The harness feeds the same frames again and again: every
    distinct frame's detections are computed once (threads: the call is stateless and releases the GIL), then the temporal
    logic (gating, Faces::track_faces, hysteresis) is replayed over the whole sequence of steps, frame by frame, exactly
    as the timed loop fed them.  rows: the frame-set index of every batch handed in so far, in order."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    oc = orc.parse_cascade_xml(xml)
    kw = dict(width_to_process=params["width_to_process"], scale_factor_pct=params["multi_scale_factor"])
    F = len(frames_np[0])
    probe = orc.FaceStream(oc, **kw)
    keys = [(r, s) for r in sorted(set(rows)) for s in range(F)]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    with ThreadPoolExecutor(max(1, min(cores, 32))) as ex:
        dets = list(ex.map(lambda k: probe.frame_detect(frames_np[k[0]][k[1]]), keys))
    memo = dict(zip(keys, dets))
    streams = [orc.FaceStream(oc, **kw) for _ in range(F if multi_stream else 1)]
    last = None
    for r in rows:
        last = [streams[s if multi_stream else 0].process_memo((r, s), frames_np[r][s], memo) for s in range(F)]
    return last


def secondary_table(ctx, casc, props, frames_np, W, H, F, dev, args, xml_face):
    """Figures a pipeline sees that the headline (device-resident, F frames per call) does not show; each a few untimed-
    warm-up + timed calls, never `value`: one frame per call (latency mode), host frames incl. the PCIe copy (pageable and
    page-locked), and the content sweep of SURVEY.md 8d (uniform noise; gradient + K pasted templates)."""
    from nubovca import capi, synth

    def rate(frames, reps, host=False, pinned=False, per_call=None, use=None):
        per_call = per_call or len(frames)
        st = capi.FaceStream(ctx, use or casc, **props)
        if host:
            fr = [capi.make_frame(f) for f in frames]
            if pinned:
                for f in frames:
                    ctx.host_register(f)
        else:
            keep = [torch.from_numpy(f).to(dev) for f in frames]
            torch.cuda.synchronize()
            fr = [capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in keep]
        batches = [ctx.prepare_face_batch([st] * per_call, fr[i:i + per_call], cap=MAX_BOXES) for i in range(0, len(fr), per_call)]
        try:
            for b in batches[:max(1, len(batches) // 4)]:
                b.process()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                for b in batches:
                    b.process()
            ctx.synchronize()
            dt = time.perf_counter() - t0
        finally:
            if host and pinned:
                for f in frames:
                    ctx.host_unregister(f)
            st.close()
        return reps * len(frames) / dt, dt / (reps * len(batches)) * 1e3

    tab = {}
    fps1, ms1 = rate(frames_np[:8], 4, per_call=1)
    tab["single_frame"] = {"ms_per_frame": ms1, "frames_per_s": fps1, "note": "one frame per nvca_face_batch_process call (no batching latency)"}
    fps_p, _ = rate(frames_np, 3, host=True)
    fps_l, _ = rate(frames_np, 3, host=True, pinned=True)
    tab["host_frames"] = {"pageable_frames_per_s": fps_p, "pinned_frames_per_s": fps_l, "frames_per_call": F,
                          "note": "frames start in host memory: the rate includes the PCIe copy (6.2 MB per 1080p frame)"}
    sx, sy = W / 1920.0, H / 1080.0
    spots = [(60 + 230 * (k % 8), 40 + 500 * (k // 8), 200 if k < 8 else 160) for k in range(16)]
    content = {}
    for name, kind, K in (("noise", "noise", 0), ("gradient+0", "gradient", 0), ("gradient+1", "gradient", 1),
                          ("gradient+4", "gradient", 4), ("gradient+16", "gradient", 16)):
        distinct = [synth.make_bgr(W, H, 777 + i, kind, [(int((x + 8 * i) * sx), int(y * sy), int(s * min(sx, sy))) for (x, y, s) in spots[:K]])
                    for i in range(8)]
        fr = [distinct[i % 8] for i in range(F)]
        fps, ms = rate(fr, 3)
        content[name] = {"frames_per_s": fps, "ms_per_step": ms}
    tab["content"] = content
    tab["content_note"] = "device-resident, %d frames per call (8 distinct frames cycled), same cascade and parameters as the headline" % F
    # rounds 1-3 quoted the headline on the UNcalibrated stand-in cascade (stage thresholds from i.i.d. noise: on this content its
    # stages 0-3 let 51 / 64 / 82 / 94 % through and stage 4 rejects 97.5 %): the same frames through it, for continuity
    other = synth.synthetic_cascade_xml() if args.cascade == "calibrated" else synth.calibrated_cascade_xml()
    oc = ctx.load_cascade_xml(other)
    fps_o, ms_o = rate(frames_np, 3, use=oc)
    tab["other_cascade"] = {"cascade": "standin (rounds 1-3)" if args.cascade == "calibrated" else "calibrated", "frames_per_s": fps_o, "ms_per_call": ms_o,
                            "note": "synchronous calls (not the serving loop), same frames and parameters as the headline"}

    # the other BASELINE configs on this GPU, a few ticks each (their own `--workload` runs give the full line)
    def multi(Wm, Hm, S, with_tracker, ticks=3, reps=4):
        bgs = [synth.make_gray(Wm, Hm, 9000 + s, "natural") for s in range(S)]
        kx, ky = Wm / 1920.0, Hm / 1080.0
        rows, rows4 = [], []
        for t in range(ticks):
            row = []
            for s in range(S):
                faces = [(int((200 + 8 * t + 16 * (s % 7)) * kx), int(150 * ky), int(300 * min(kx, ky))), (int((900 + 8 * t) * kx), int(400 * ky), int(180 * min(kx, ky)))]
                row.append(torch.from_numpy(synth.gray_to_bgr(synth.paste_faces(bgs[s], faces, s), 9000 + s)).to(dev))
            rows.append(row)
            if with_tracker:
                rows4.append([torch.cat([x, torch.full((Hm, Wm, 1), 255, dtype=torch.uint8, device=dev)], dim=2).contiguous() for x in row])
        torch.cuda.synchronize()
        sts = [capi.FaceStream(ctx, casc, width_to_process=Wm, multi_scale_factor=props["multi_scale_factor"]) for _ in range(S)]
        trk = [capi.Tracker(ctx) for _ in range(S)] if with_tracker else None
        prep = [ctx.prepare_face_batch(sts, [capi.make_frame(x.data_ptr(), Wm, Hm, Wm * 3, capi.MEM_DEVICE) for x in row], cap=MAX_BOXES) for row in rows]
        fr4 = [[capi.make_frame(x.data_ptr(), Wm, Hm, Wm * 4, capi.MEM_DEVICE) for x in row] for row in rows4]

        def tick(i):
            prep[i % ticks].process()
            if trk:
                capi.tracker_batch_process(ctx, trk, fr4[i % ticks], [33.3 * i] * S, cap=256)
        for i in range(ticks):
            tick(i)
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(ticks, ticks + reps * ticks):
            tick(i)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        # the same ticks through the serving loop the headline uses: tick i + 1 of the face detectors is submitted before tick i
        # is collected (nvca_face_batch_submit / _collect, two batches in flight); the trackers' call runs beside it on its lane
        frs = [[capi.make_frame(x.data_ptr(), Wm, Hm, Wm * 3, capi.MEM_DEVICE) for x in row] for row in rows]
        infl = [ctx.face_batch_submit(sts, frs[0])]

        def tick_p(i):
            nxt = ctx.face_batch_submit(sts, frs[(i + 1) % ticks])
            if trk:
                capi.tracker_batch_process(ctx, trk, fr4[i % ticks], [33.3 * (ticks + reps * ticks + i)] * S, cap=256)
            ctx.face_batch_collect(infl[0], cap=MAX_BOXES)
            infl[0] = nxt
        for i in range(ticks):
            tick_p(i)
        t0 = time.perf_counter()
        for i in range(ticks, ticks + reps * ticks):
            tick_p(i)
        dtp = time.perf_counter() - t0
        ctx.face_batch_collect(infl[0], cap=MAX_BOXES)
        ctx.synchronize()
        for st in sts:
            st.close()
        return S * reps * ticks / dtp, dtp / (reps * ticks) * 1e3, S * reps * ticks / dt, dt / (reps * ticks) * 1e3
    f720, ms720, f720s, ms720s = multi(1280, 720, 32, False)
    ftrk, mstrk, ftrks, mstrks = multi(1920, 1080, 8, True)
    # BASELINE configs[2]: the face -> eye / nose / mouth / ear chain on V concurrent 1080p streams, batched entry points
    def roi_chain(base, V=8, ticks=4, reps=6, contexts=1, calibrated=True, inflight=False):
        """V video streams x (face detector + eye + nose + mouth + ear detectors, own face pass each) per tick.  contexts = 2: the streams
        are dealt to two contexts of this GPU, a serving thread each (what the GStreamer shim does with NVCA_VIRTUAL_GPUS=2: a
        context per slot) -- one context's host work between its three waits runs beside the other's kernels."""
        import threading
        part_xml = synth.calibrated_part_cascade_xml if calibrated else synth.synthetic_part_cascade_xml
        names = ("righteye", "lefteye", "nose", "mouth", "leftear", "rightear")
        kinds = [(0, "righteye", "lefteye"), (1, "nose", None), (2, "mouth", None), (3, "leftear", "rightear")]
        keep = [[torch.from_numpy(synth.make_bgr(1920, 1080, 40 + 5 * t + v, "natural", [(x + 8 * t + 6 * v, y + 3 * v, sz + 4 * ((t + v) % 3)) for x, y, sz in base])).to(dev) for v in range(V)]
                for t in range(ticks)]
        torch.cuda.synchronize()
        frs = [[capi.make_frame(x.data_ptr(), 1920, 1080, 1920 * 3, capi.MEM_DEVICE) for x in row] for row in keep]
        cx = [ctx] + [capi.Context(dev.index or 0) for _ in range(contexts - 1)]
        share = [list(range(c, V, contexts)) for c in range(contexts)]           # video streams of context c
        S = []
        loaded = []
        for c, cc in enumerate(cx):
            fcas = casc if c == 0 else cc.load_cascade_xml(xml_face)
            pcs = {nm: cc.load_cascade_xml(part_xml(nm)) for nm in names}
            loaded.append(list(pcs.values()) + ([fcas] if c else []))
            fcs = [capi.FaceStream(cc, fcas, width_to_process=1920, multi_scale_factor=props["multi_scale_factor"]) for _ in share[c]]
            parts = [capi.PartStream(cc, k, fcas, pcs[a], pcs[b] if b else None) for _ in share[c] for k, a, b in kinds]
            S.append((cc, fcs, parts))
        found = [0] * contexts

        def tick(c, i):
            cc, fcs, parts = S[c]
            fb = [frs[i % ticks][v] for v in share[c]]
            tk = cc.face_batch_submit(fcs, fb)            # the face detector's batch runs under the part detectors' call
            res = capi.part_batch_process(cc, parts, [f for f in fb for _ in range(4)])
            cc.face_batch_collect(tk)
            found[c] += sum(len(a) + len(b) for a, b in res)

        def submit(c, i):
            cc, fcs, parts = S[c]
            fb = [frs[i % ticks][v] for v in share[c]]
            return capi.part_batch_submit(cc, parts, [f for f in fb for _ in range(4)]), cc.face_batch_submit(fcs, fb)

        def run(i0, i1):
            if contexts == 1 and inflight:
                # two part batches in flight: tick i + 1's gates, images and face passes are queued before tick i is collected
                cc = S[0][0]
                pt, ft = submit(0, i0)
                for i in range(i0, i1):
                    nxt = submit(0, i + 1) if i + 1 < i1 else None
                    res = capi.part_batch_collect(cc, pt)
                    cc.face_batch_collect(ft)
                    found[0] += sum(len(a) + len(b) for a, b in res)
                    if nxt:
                        pt, ft = nxt
            elif contexts == 1:
                for i in range(i0, i1):
                    tick(0, i)
            else:
                th = [threading.Thread(target=lambda c=c: [tick(c, i) for i in range(i0, i1)]) for c in range(contexts)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            for cc, _, _ in S:
                cc.synchronize()
        run(0, 2 * ticks)                                  # two turns through the frame sets: plans, tables and buffers are in place
        found[:] = [0] * contexts
        t0 = time.perf_counter()
        run(2 * ticks, 2 * ticks + reps * ticks)
        dt = time.perf_counter() - t0
        nparts = sum(found) / (V * reps * ticks)
        # launches per tick: a separate pass with an event pair on every launch (they serialise the launches: not the rate above)
        launches = 0
        for c, (cc, _, _) in enumerate(S):
            cc.enable_kernel_timing(1)
        for i in range(ticks):
            for c in range(contexts):
                tick(c, i)
        for cc, _, _ in S:
            cc.synchronize()
            kt = cc.kernel_timing()
            cc.enable_kernel_timing(0)
            launches += sum(v[1] for v in kt.values()) / ticks
        for cc, fcs, parts in S:
            for st in fcs:
                st.close()
            for pt in parts:
                pt.close()
        for pcs_c in loaded:                     # the workload's cascades go with it (their stump tables would otherwise fill the context's cache for the next workload)
            for cs in pcs_c:
                cs.free()
        for cc in cx[1:]:
            cc.close()
        # algorithmic bytes per video frame (SURVEY.md 8d's formula per working image): the face detector's 60.21 MB plus, per part
        # detector, BGR in + gray out / in, plus the small working images' integral pairs written and read
        alg = 60.21e6 + 4 * (3 * 1920 * 1080 + 2 * 1920 * 1080) + 4 * 25 * (320 * 180 + 160 * 90)
        ach = alg * V * reps * ticks / dt / 1e9
        return V * reps * ticks / dt, dt / (reps * ticks) * 1e3, nparts, {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                                                           "launches_per_tick": launches, "alg_bytes_per_frame": alg,
                                                                           "note": "launch- and latency-bound small-image work next to one 8-frame face-detector batch"}
    # two faces per frame, large enough that their parts (a quarter of the face wide, synth.part_template) reach the part cascades'
    # 20-pixel windows on the 320-pixel working image: the search phase is loaded (11 parts per frame).  Round 2's workload -- four
    # faces of 120-300 pixels, whose parts stay below those windows: 0.3 parts per frame, the search phase idle -- is kept as
    # `roi_chain_sparse` for continuity with the figures quoted then
    big = [(150, 200, 560), (1100, 260, 620)]
    on1080 = (W, H) == (1920, 1080)
    froi, msroi, nparts, roiroof = roi_chain(big) if on1080 else (None, None, None, None)
    froi2, msroi2, nparts2, roiroof2 = roi_chain(big, contexts=2) if on1080 else (None, None, None, None)
    froi3, msroi3, nparts3, roiroof3 = roi_chain(big, inflight=True) if on1080 else (None, None, None, None)
    fold, msold, npold, rold = roi_chain(big, calibrated=False) if on1080 else (None, None, None, None)
    fsp, mssp, npsp, rsp = roi_chain([(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)], calibrated=False) if on1080 else (None, None, None, None)
    tab["workloads"] = {"roi_chain": {"frames_per_s": froi, "ms_per_tick": msroi, "streams": 8, "parts_per_frame": nparts, "roofline": roiroof,
                                      "note": "BASELINE configs[2]: 8 x 1080p streams x (face detector + eye + nose + mouth + ear, own face pass each), nvca_face_batch_submit/collect around nvca_part_batch_process, one context, one serving thread; part cascades calibrated on face regions (every early stage lets ~2/3 through); scripts/bench_roi_chain.py gives the breakdown"},
                        "roi_chain_2ctx": {"frames_per_s": froi2, "ms_per_tick": msroi2, "streams": 8, "parts_per_frame": nparts2, "roofline": roiroof2,
                                           "note": "the same 8 streams dealt to two contexts of this GPU with a serving thread each (the GStreamer shim's NVCA_VIRTUAL_GPUS=2): one context's host work between its waits runs beside the other's kernels"},
                        "roi_chain_two_in_flight": {"frames_per_s": froi3, "ms_per_tick": msroi3, "streams": 8, "parts_per_frame": nparts3, "roofline": roiroof3,
                                                    "note": "the same 8 streams, one context, one serving thread, nvca_part_batch_submit / _collect: tick k + 1's gates, working images and face passes are queued before tick k's searches are collected (results arrive one tick later)"},
                        "roi_chain_standin_parts": {"frames_per_s": fold, "ms_per_tick": msold, "streams": 8, "parts_per_frame": npold, "roofline": rold,
                                                    "note": "as roi_chain with rounds 2-3's uncalibrated part cascades (their early stages let ~150 windows per face region through to the late stages)"},
                        "roi_chain_sparse": {"frames_per_s": fsp, "ms_per_tick": mssp, "streams": 8, "parts_per_frame": npsp, "roofline": rsp,
                                             "note": "the same chain on round 2's frames (faces of 120-300 pixels: their parts never reach the part cascades' windows, the searches find almost nothing), uncalibrated part cascades"},
                        "streams720p": {"frames_per_s": f720, "ms_per_tick": ms720, "streams": 32, "frames_per_s_sync": f720s, "ms_per_tick_sync": ms720s,
                                        "note": "BASELINE configs[3] per GPU: 32 independent 1280x720 streams, one frame each per tick; serving loop as the headline (the next tick submitted before this one is collected); _sync: one synchronous call per tick"},
                        "face_tracker": {"frames_per_s": ftrk, "ms_per_tick": mstrk, "streams": 8, "frames_per_s_sync": ftrks, "ms_per_tick_sync": mstrks,
                                         "note": "BASELINE configs[4] per GPU: 8 x 1080p streams through NuboFaceDetector + NuboTracker per tick; face detectors in the serving loop, the trackers' batched call beside them; _sync: both calls synchronous"}}
    return tab


def boxes_equal(got, exp):
    if len(got) != len(exp):
        return False
    for (gb, gi), (eb, ei) in zip(got, exp):
        if not (np.array_equal(np.asarray(gb).reshape(-1, 4), eb) and np.array_equal(np.asarray(gi).reshape(-1), ei)):
            return False
    return True


def launch_ranks(n, argv):
    """`python bench.py --gpus N` typed as it stands: this process becomes a launcher that makes NO GPU call (it imports neither
    torch nor the library), starts N ranks of this same script -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, exactly what `python -m torch.distributed.run --nproc-per-node N`
    would give them -- relays rank 0's JSON line and exits with the worst child status.  Under torchrun (WORLD_SIZE already
    set) the script is a rank itself and never comes here."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    child_cmd = os.environ.get("NVCA_BENCH_CHILD_CMD")          # tests: a stub rank instead of this script
    cmd = child_cmd.split() if child_cmd else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is not None:
        print(line, flush=True)
    worst = max((abs(rc) for rc in rcs), default=0)
    if line is None and worst == 0:
        worst = 1                                          # every rank exited cleanly and nobody printed the line: still a failure
    if worst:
        print("bench.py launcher: rank exit codes %s" % rcs, file=sys.stderr)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=32)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width-to-process", type=int, default=0, help="0 = full resolution (benchmark mode)")
    ap.add_argument("--scale-factor-pct", type=int, default=10)
    ap.add_argument("--content", default="natural", choices=["natural", "noise", "gradient"])
    ap.add_argument("--faces", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline, boxes_match, the OpenCV probe)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary table (single-frame latency, host-frame rates, content sweep)")
    ap.add_argument("--host-frames", action="store_true", help="feed host buffers (PCIe-inclusive rate; not `value`)")
    ap.add_argument("--pipeline", action="store_true", help="(the default since round 2) nvca_face_batch_submit / _collect with two batches in flight")
    ap.add_argument("--sync", action="store_true", help="one synchronous nvca_face_batch_process call per step instead of the submit / collect serving loop")
    ap.add_argument("--pinned", action="store_true", help="with --host-frames: page-lock the frame buffers (nvca_host_register)")
    ap.add_argument("--workload", default="face1080p", choices=["face1080p", "streams720p", "face_tracker"],
                    help="face1080p: BASELINE configs[1] (default, the headline metric); streams720p: configs[3], "
                         "--streams-per-gpu 720p streams, one frame each per step; face_tracker: configs[4], "
                         "--streams-per-gpu 1080p streams through NuboFaceDetector and NuboTracker")
    ap.add_argument("--streams-per-gpu", type=int, default=0)
    ap.add_argument("--cascade", default="calibrated", choices=["calibrated", "standin"],
                    help="calibrated (default since round 4): the synthetic cascade with every early stage's threshold set on windows of "
                         "the bench's own content so that each stage rejects about half of what reaches it (a trained cascade's profile); "
                         "standin: rounds 1-3's cascade (thresholds from i.i.d. noise: non-monotone stage selectivities on this content)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    global torch, dist
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal hook for a one-GPU box: NVCA_BENCH_REHEARSAL=1 runs every rank on device 0 with the gloo backend, so the
    # N > 1 code path (barriers, max-over-ranks clock, per-tick gather) can be exercised without N GPUs
    rehearsal = os.environ.get("NVCA_BENCH_REHEARSAL") is not None
    if rehearsal:
        local_rank = 0
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if rehearsal else dev

    # helper threads of the library (per-job host work, host-frame copies): a rank takes its share of the box's cores, no more --
    # N ranks x (a serving loop + up to 7 spinning helpers) would oversubscribe the host at N = 8
    try:
        _cores = len(os.sched_getaffinity(0))
    except AttributeError:
        _cores = os.cpu_count() or 1
    os.environ.setdefault("NVCA_HOST_THREADS", str(max(0, min(7, _cores // max(world, 1) - 1))))
    from nubovca import capi, synth
    if args.workload == "streams720p":
        args.width, args.height = 1280, 720
        args.frames_per_step = args.streams_per_gpu or 32
    elif args.workload == "face_tracker":
        args.frames_per_step = args.streams_per_gpu or 8
    multi_stream = args.workload != "face1080p"
    W, H, F = args.width, args.height, args.frames_per_step
    w2p = args.width_to_process or W
    xml = synth.calibrated_cascade_xml() if args.cascade == "calibrated" else synth.synthetic_cascade_xml()
    ctx = capi.Context(local_rank)
    casc = ctx.load_cascade_xml(xml)
    props = {"width_to_process": w2p, "multi_scale_factor": args.scale_factor_pct}
    stream = capi.FaceStream(ctx, casc, **props)
    scale = W // w2p
    w, h = int(np.rint(W / scale)), int(np.rint(H / scale))

    base_faces = [(200, 150, 300), (900, 400, 180), (1400, 100, 120), (1500, 700, 240)][:args.faces]
    sx, sy = W / 1920.0, H / 1080.0
    TICKS = 4
    frames_np = []          # [tick][slot]
    if not multi_stream:
        # 4 x F consecutive synthetic frames of this rank's stream (stream id = rank), faces drifting 8 px/frame: four DISTINCT sets of F
        # frames cycled through the timed steps -- 4 x 199 MB of input at 32 x 1080p, beyond the 256 MB Infinity Cache, so no step
        # finds its frames on-die.  Sets 1 .. 3 are the 1/f fields of set 0 mirrored (left-right, top-bottom, both: other bytes at
        # every address, the same statistics, a quarter of the generation time) with the templates pasted afterwards.
        base = [synth.make_gray(W, H, synth.frame_seed(rank, i), args.content) for i in range(F)]
        for r in range(TICKS):
            row = []
            for i in range(F):
                g = base[i]
                if r & 1:
                    g = g[:, ::-1]
                if r & 2:
                    g = g[::-1]
                k = r * F + i
                faces = [(int(((x + 8 * k) % 1500) * sx), int(y * sy), int(s * min(sx, sy))) for (x, y, s) in base_faces]
                row.append(synth.gray_to_bgr(synth.paste_faces(np.ascontiguousarray(g), faces, synth.frame_seed(rank, k)), synth.frame_seed(rank, k)))
            frames_np.append(row)
    else:
        # F streams (ids rank*F..), static per-stream background, templates moving 8 px per tick
        bgs = [synth.make_gray(W, H, synth.frame_seed(rank * F + s, 0), args.content) for s in range(F)]
        for t in range(TICKS):
            row = []
            for s in range(F):
                faces = [(int((x + 8 * t + 16 * (s % 7)) * sx), int(y * sy), int(sz * min(sx, sy))) for (x, y, sz) in base_faces]
                row.append(synth.gray_to_bgr(synth.paste_faces(bgs[s], faces, s), synth.frame_seed(rank * F + s, 0)))
            frames_np.append(row)
    if args.host_frames:
        frames_t = [[capi.make_frame(f) for f in row] for row in frames_np]
        keep = frames_np
        if args.pinned:
            for row in frames_np:
                for f in row:
                    ctx.host_register(f)
    else:
        keep = [[torch.from_numpy(f).to(dev) for f in row] for row in frames_np]
        frames_t = [[capi.make_frame(t.data_ptr(), W, H, W * 3, capi.MEM_DEVICE) for t in row] for row in keep]
    torch.cuda.synchronize()
    streams = [capi.FaceStream(ctx, casc, **props) for _ in range(F)] if multi_stream else [stream] * F
    trackers, bgra_frames, bgra_keep = None, None, None
    if args.workload == "face_tracker":     # the same RGB field as BGRA for the tracker
        trackers = [capi.Tracker(ctx) for _ in range(F)]
        bgra_keep = [[torch.cat([torch.as_tensor(t, device=dev), torch.full((H, W, 1), 255, dtype=torch.uint8, device=dev)], dim=2).contiguous()
                      for t in row] for row in keep]
        bgra_frames = [[capi.make_frame(t.data_ptr(), W, H, W * 4, capi.MEM_DEVICE) for t in row] for row in bgra_keep]
        torch.cuda.synchronize()
    tick = [0]
    from nubovca import sharding
    gather = sharding.TableGather(device=coll_dev) if world > 1 else None

    # serving loop: two batches in flight -- the next batch is queued before the previous one is unpacked, so the host
    # work between batches overlaps the GPU.  K steps = K submits + K collects; one batch stays in flight across steps.
    pipelined = not args.sync              # the serving loop: the next batch is queued before the previous one is unpacked
    rows = [0] if pipelined else []          # frame-set index of every batch handed in, in order (the oracle replays them)
    inflight = [ctx.face_batch_submit(streams, frames_t[0])] if pipelined else [None]
    # the synchronous loop hands the same frame buffers in again and again: marshal the ctypes arguments once per buffer set
    prepared = None if pipelined else [ctx.prepare_face_batch(streams, fr, cap=MAX_BOXES) for fr in frames_t]

    def step():
        tick[0] += 1
        rows.append(tick[0] % TICKS)
        if pipelined:
            nxt = ctx.face_batch_submit(streams, frames_t[tick[0] % TICKS])
            res = ctx.face_batch_collect(inflight[0], cap=MAX_BOXES)
            inflight[0] = nxt
        else:
            res = prepared[tick[0] % TICKS].process()
        if trackers is not None:
            capi.tracker_batch_process(ctx, trackers, bgra_frames[tick[0] % TICKS], [33.3 * tick[0]] * F, cap=256)
        if world > 1:           # result gather (the only collective): fixed-size box table per stream tick, over RCCL;
            tab = sharding.pack_box_arrays(res.boxes, res.counts, rank=rank) if hasattr(res, "counts") else sharding.pack_boxes(res, MAX_BOXES, rank=rank)
            gather.submit(tab)                                      # asynchronous: it overlaps the next tick's kernels
        return res

    def fence():
        if world > 1:
            gather.finish()
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def as_list(r):
        return r.results() if hasattr(r, "results") else r

    for _ in range(args.warmup):
        res = step()
    n_boxes = float(np.mean([len(b) for b, _ in as_list(res)])) if args.warmup else 0.0
    # per-kernel HIP events ride on every 4th step of the timed region (they keep consecutive launches from overlapping:
    # ~6 us per launch); NVCA_BENCH_TIMING_STRIDE=1 puts them on every step, NVCA_BENCH_NOTIMING=1 on none
    stride = 0 if os.environ.get("NVCA_BENCH_NOTIMING") is not None else max(1, int(os.environ.get("NVCA_BENCH_TIMING_STRIDE", "4")))
    ctx.enable_kernel_timing(stride)
    sampled_steps = (args.steps + stride - 1) // stride if stride else 0     # steps whose launches carried events
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t0
    if pipelined:               # the batch queued by the last step has finished inside the timed region; unpack it now
        res = ctx.face_batch_collect(inflight[0], cap=MAX_BOXES)
        inflight[0] = None
    ktimes = ctx.kernel_timing()
    ctx.enable_kernel_timing(False)
    n_boxes = float(np.mean([len(b) for b, _ in as_list(res)]))
    ranks_seen = [0]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the last gathered box table carries every rank's stamp: N ranks really ran and their rows arrived where they belong
        g = gather.last()
        stamps = sharding.table_ranks(g)
        assert g.shape[0] == world and all((stamps[r] == r).all() for r in range(world)), stamps
        ranks_seen = sorted(set(int(v) for v in stamps.reshape(-1)))
        assert ranks_seen == list(range(world)), ranks_seen

    if rank == 0:
        total_frames = world * F * args.steps
        fps = total_frames / dt
        ab = algorithmic_bytes(W, H, w, h, n_boxes)
        groups = {"gray_resize_hist": ["gray_resize_hist"], "equalize_lut": ["equalize_lut"],
                  "integral": ["integral_colsum", "integral_bandscan", "integral_rows"],
                  "cascade": ["cascade_stage0", "cascade_strip", "cascade_deep", "cascade_tile", "cascade_band", "cascade_roi", "group_rects"],
                  "tracker": ["tracker"]}
        kern = {}
        for gname, members in groups.items():
            ms = sum(ktimes.get(m, (0.0, 0))[0] for m in members)
            launches = max([ktimes.get(m, (0.0, 0))[1] for m in members] + [0])
            if launches:
                per_launch_ms = ms / launches
                bytes_per_launch = ab.get(gname, 0) * F * sampled_steps / launches  # frames per launch set (host frames go through in chunks)
                kern[gname] = {"ms_per_launch": per_launch_ms, "launches": launches,
                               "alg_bytes_per_launch": bytes_per_launch,
                               "achieved_GBs": bytes_per_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0}
        dom = max(kern, key=lambda k: kern[k]["ms_per_launch"]) if kern else None
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # The PMC-derived per-launch figures are NOT counters of this run: they come from profiles/pmc_traffic.json (separate
        # rocprofv3 --pmc passes of this command line).  They are quoted only for the configuration AND the kernel sources they were
        # collected on (a hash of csrc/ is stored with them), with their source named; otherwise traffic is null.
        pmc_ok, pmc_doc = False, {}
        if os.path.exists(tpath):
            try:
                pmc_doc = json.load(open(tpath))
                cfg = pmc_doc.get("_config", {})
                pmc_ok = (cfg.get("workload") == args.workload and cfg.get("width") == W and cfg.get("height") == H and
                          cfg.get("frames_per_launch") == F and not args.host_frames and (w, h) == (W, H) and
                          cfg.get("cascade") == args.cascade and cfg.get("src_sha16") == source_hash())
            except Exception:
                pmc_ok = False
        if dom and pmc_ok:
            traffic = pmc_doc.get(dom, {}).get("hbm_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %s; kernel sources %s)" % (pmc_doc.get("_config", {}).get("collected", "?"), source_hash())
        elif dom:
            traffic_source = "none: profiles/pmc_traffic.json was collected on another configuration or on other kernel sources"
        lds = None
        if dom and pmc_ok:
            try:        # the dominant kernel works out of LDS: also price it against the LDS peak
                lb = pmc_doc.get(dom, {}).get("lds_bytes_per_launch")
                if lb:
                    peak = 128.0 * 256 * 2.4                     # B/clk/CU x CUs x GHz = GB/s
                    ach = lb / (kern[dom]["ms_per_launch"] * 1e-3) / 1e9
                    lds = {"achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                           "source": "LDS wave instructions per launch from profiles/ (PMC) x 64 lanes x 3 B mean access, / live kernel time"}
            except Exception:
                lds = None
        roofline = None
        if dom:
            a = kern[dom]["achieved_GBs"]
            roofline = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": a / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                        "pipeline_achieved": ab["total"] * (fps / world) / 1e9,
                        "pipeline_frac": ab["total"] * (fps / world) / 1e9 / HBM_PEAK_GBS,
                        "kernels": kern, "lds": lds,
                        "dominant_launch": None,
                        "timed_steps": sampled_steps,      # steps of the timed region whose launches carried HIP events (every `stride`-th)
                        "detail_ms_per_launch": {k: v[0] / v[1] for k, v in ktimes.items() if v[1]}}
        if roofline and dom == "cascade":
            # `frac` prices the whole cascade group (band / tile + late stages + grouping) against its algorithmic bytes; the single
            # longest launch of the group on its own (what `rocprofv3 --stats` lists first):
            det = roofline["detail_ms_per_launch"]
            name = max((k for k in det if k.startswith("cascade_")), key=lambda k: det[k], default=None)
            if name and det[name] > 0:
                ach = kern[dom]["alg_bytes_per_launch"] / (det[name] * 1e-3) / 1e9
                roofline["dominant_launch"] = {"name": name, "ms": det[name], "achieved": ach, "frac": ach / HBM_PEAK_GBS}
        out = {
            "metric": "1080p frames/sec/node (NuboFaceDetector); achieved HBM GB/s vs peak",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32 rect sums / f32 products / f64 stage sums (u8 pixels)",
            "data": "synthetic (%s field + %d pasted templates; seeded synthetic stump cascade shaped like "
                    "haarcascade_frontalface_alt: 22 stages, 2135 stumps, %s stage thresholds)" % (args.content, args.faces, args.cascade),
            "config": {"workload": ("NuboFaceDetector %dx%d single stream per GPU" if not multi_stream else
                                    ("NuboFaceDetector %dx%d, one frame of each of %d streams per GPU per step" % (W, H, F) if trackers is None else
                                     "NuboFaceDetector + NuboTracker %dx%d, one frame of each of %d streams per GPU per step" % (W, H, F))
                                    ) % ((W, H) if not multi_stream else ()) +
                                   ", working image %dx%d, scaleFactor %.2f, minNeighbors 3, minSize (w/20,h/20)" % (w, h, 1 + args.scale_factor_pct / 100.0),
                       "frames_per_step": F, "frame_sets": TICKS, "input_bytes_cycled": TICKS * F * W * H * 3, "streams": world * (F if multi_stream else 1), "frames_resident": ("host-pinned" if args.pinned else "host") if args.host_frames else "hbm",
                       "boxes_per_frame": n_boxes, "parallelism": "stream-sharded x%d" % world, "ranks_seen": ranks_seen,
                       "batches_in_flight": 2 if pipelined else 1,
                       "cascade": ("calibrated: stage thresholds set on windows of this content, every early stage rejects about half of what reaches it"
                                   if args.cascade == "calibrated" else "standin: rounds 1-3's thresholds from i.i.d. noise")},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            # parity at the timed size, through the timed entry point: the boxes and ids of the LAST timed step against the
            # oracle replaying the same sequence of batches (FACE/kmsfacedetect.cpp:805-826)
            try:
                exp = oracle_expected(xml, frames_np, props, rows, multi_stream)
                out["boxes_match"] = bool(boxes_equal(as_list(res), exp))
                out["boxes_checked"] = {"frames": len(exp), "boxes": int(sum(len(b) for b, _ in exp)), "batches_replayed": len(rows),
                                        "against": "oracle (CPU restatement) on the same frames, same sequence of batches"}
            except Exception as e:          # the check must never cost the line
                out["boxes_match"] = None
                out["boxes_checked"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["cpu_baseline"] = cpu_baseline(xml, frames_np[0], props)
            try:                             # the real library, if this box happens to have it (never installed): SURVEY.md 8d(2)
                import cv2_probe
                out["cpu_baseline"]["opencv"] = cv2_probe.probe(xml, frames_np[0][:8], width_to_process=0 if w2p == W else w2p,
                                                                 scale_factor=1 + args.scale_factor_pct / 100.0, budget_s=10.0)
            except Exception as e:
                out["cpu_baseline"]["opencv"] = {"available": False, "note": "probe not run: %s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_secondary and args.workload == "face1080p" and not args.host_frames:
            try:
                out["secondary"] = secondary_table(ctx, casc, props, frames_np[0], W, H, F, dev, args, xml)
            except Exception as e:
                out["secondary"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
