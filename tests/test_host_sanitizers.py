"""The product's pure-host sources -- the cascade loader (cascade_xml.cpp), the table builders (plan.cpp) and the host glue
(host_logic.cpp: groupRectangles, Faces::track_faces, __join_objects, the part detectors' merging heuristics) -- built under
AddressSanitizer + UndefinedBehaviorSanitizer on the CPU and driven by tests/san/san_driver.cpp; what the driver prints is
checked against the oracle.  (The GPU pool refuses sanitizer runs, so host memory errors are looked for here.)"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "san")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def driver():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with sanitizer runtimes")
    csrc = os.path.join(ROOT, "nubomedia-vca_amd", "csrc")
    out = os.path.join(SAN, "build", "san_driver")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(SAN, "san_driver.cpp")] + [os.path.join(csrc, f) for f in ("cascade_xml.cpp", "plan.cpp", "host_logic.cpp")]
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = [CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
               "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"), "-w"] + srcs + ["-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return out


def _run(driver, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver] + list(args), capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_loader_under_sanitizers(driver, tmp_path, synth_xml):
    from nubovca import synth
    files = {"face.xml": synth_xml, "generic.xml": synth.generic_cascade_xml(seed=3), "mouth.xml": synth.synthetic_part_cascade_xml("mouth")}
    gold = os.path.join(ROOT, "tests", "golden")
    paths = []
    for n, x in files.items():
        p = tmp_path / n
        p.write_text(x)
        paths.append(str(p))
    paths += [os.path.join(gold, f) for f in sorted(os.listdir(gold)) if f.startswith("oldformat_") and f.endswith(".xml")]
    out = _run(driver, "loader", *paths)
    loaded = [o for o in out if "loader" in o]
    assert len(loaded) == len(paths) and all(o["rc"] == 0 for o in loaded), loaded
    assert loaded[0]["stages"] == 22 and loaded[0]["cls"] == 2135 and loaded[0]["stumps"] == 1
    assert loaded[1]["tilted"] == 1 and loaded[1]["stumps"] == 0
    fuzz = [o for o in out if "loader_fuzz" in o]
    assert all(o["other"] == 0 and o["parse"] > 0 for o in fuzz), fuzz          # damaged files end as NVCA_ERR_PARSE, nothing else


def test_plans_under_sanitizers_match_the_oracle_grid(driver, tmp_path, synth_xml):
    """every plan the driver builds passed its structural checks (tiles cover the grid once, every sample a window can touch is
    staged and inside the plane, keys round-trip); the factors and grids are OpenCV's, as the oracle computes them"""
    import orc
    p = tmp_path / "face.xml"
    p.write_text(synth_xml)
    out = [o for o in _run(driver, "plans", str(p)) if "plan" in o]
    assert len(out) == 33
    tiled = 0
    for o in out:
        w, h, sf, minw, minh = o["plan"]
        assert o["rc"] == 0, o
        exp = orc.scale_grid(20, 20, w, h, sf, (minw, minh), (w, h))
        assert o["factors"] == exp, (o["plan"], o["factors"][:3], exp[:3])
        for f, (ex, ey) in zip(o["factors"], o["grid"]):
            ystep = max(2.0, f)
            ww = int(np.rint(20 * f))
            assert ex == int(np.rint((w - ww) / ystep)) and ey == int(np.rint((h - ww) / ystep))
        if o["variant"] == 0 and o["factors"]:
            assert o["strips"] == 0 and o["tiles"] > 0 and o["bands"] > 0, o
            tiled += 1
        if o["variant"] == 1:
            assert o["tiles"] == 0
    assert tiled >= 8
    full = [o for o in out if o["plan"][:2] == [1920, 1080] and o["variant"] == 0][0]
    assert len(full["factors"]) == 25 and sum(a * b for a, b in full["grid"]) == 355162          # SURVEY 8a, derived independently


def test_glue_under_sanitizers_matches_the_oracle(driver, tmp_path):
    import orc
    rng = np.random.default_rng(11)
    lines, expect = [], {}

    def rects(n, span=400, smax=120):
        r = np.stack([rng.integers(0, span, n), rng.integers(0, span, n), rng.integers(1, smax, n), rng.integers(1, smax, n)], 1).astype(np.int32)
        for i in range(1, n):                       # clusters: near-copies of earlier boxes, so that classes form
            if rng.random() < 0.6:
                r[i] = r[rng.integers(0, i)] + rng.integers(-3, 4, 4)
                r[i, 2:] = np.maximum(r[i, 2:], 1)
        return r

    def fmt(r):
        return "%d %s" % (len(r), " ".join(str(int(v)) for v in r.reshape(-1)))

    for i in range(150):
        # mostly a few dozen boxes; every tenth case a FIND_BIGGEST-sized list (hundreds of near-copies: the x-window partition's case)
        r = rects(int(rng.integers(200, 700))) if i % 10 == 9 else rects(int(rng.integers(0, 40)))
        thr = int(rng.integers(1, 5))
        lines.append("G %d %d 0.2 %s" % (i, thr, fmt(r)))
        expect[("group", i)] = orc.group_rectangles(r, thr, 0.2)
    for i in range(150):
        r = rects(int(rng.integers(0, 30)), span=300, smax=200)
        mn, mx, dist = int(rng.integers(0, 200)), int(rng.integers(500, 40000)), int(rng.integers(5, 80))
        lines.append("J %d %d %d %d %s" % (i, mn, mx, dist, fmt(r)))
        expect[("join", i)] = orc.join_objects(r, mn, mx, dist)
    for i in range(100):
        nf = int(rng.integers(1, 8))
        thr = int(rng.integers(5, 60))
        faces, ids, nid = np.zeros((0, 4), np.int32), np.zeros(0, np.int32), 0
        seq = []
        for k in range(nf):
            cur = rects(int(rng.integers(0, 5)), span=150, smax=90)
            seq.append(fmt(cur))
            if len(cur):
                faces, ids, nid = orc.track_faces(faces, ids, nid, cur, thr)
            elif k % 3 == 2:
                faces, ids = np.zeros((0, 4), np.int32), np.zeros(0, np.int32)
        lines.append("T %d %d %d %s" % (i, thr, nf, " ".join(seq)))
        expect[("track", i)] = (faces, ids)
    p = tmp_path / "cases.txt"
    p.write_text("\n".join(lines) + "\n")
    out = _run(driver, "glue", str(p))
    seen = 0
    for o in out:
        for tag in ("group", "join", "track"):
            if tag in o:
                got = np.array(o["out"], np.int32).reshape(-1, 4)
                e = expect[(tag, o[tag])]
                if tag == "join":
                    assert np.array_equal(got, e), (tag, o[tag], got, e)
                else:
                    assert np.array_equal(got, e[0]) and np.array_equal(np.array(o["extra"], np.int32), e[1]), (tag, o[tag], got, e)
                seen += 1
    assert seen == 400
    assert any("merges_checksum" in o for o in out)


def test_work_pool_under_thread_sanitizer():
    """the helper-thread pool of the per-job host work (csrc/work_pool.cpp), thousands of back-to-back runs under TSan"""
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with sanitizer runtimes")
    out = os.path.join(SAN, "build", "pool_driver")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(SAN, "pool_driver.cpp"), os.path.join(ROOT, "nubomedia-vca_amd", "csrc", "work_pool.cpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in srcs):
        r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread"] + srcs + ["-o", out], capture_output=True, text=True)
        if r.returncode != 0 and "tsan" in r.stderr.lower():
            pytest.skip("no TSan runtime")
        assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "pool ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_host_range_bookkeeping_under_sanitizers():
    """which caller memory may reach the runtime as a raw pointer, and which streams nvca_host_unregister drains before the pages go
    (csrc/host_ranges.h): direct copies only inside a range that is registered NOW, released ranges are bounced like any memory,
    unregister waits for the copy stream and the lanes that carried copies of the range, not only the context's own stream"""
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with sanitizer runtimes")
    out = os.path.join(SAN, "build", "ranges_driver")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(SAN, "ranges_driver.cpp")
    r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-w", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ranges ok" in r.stdout, (r.stdout, r.stderr[-3000:])
