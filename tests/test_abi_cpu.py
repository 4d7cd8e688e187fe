"""CPU-side checks of the drop-in boundary: the library builds, loads, exports
every symbol include/nubovca.h declares, and refuses to work without a GPU
(no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from nubovca import capi
    return capi.load()


def test_header_symbols_all_exported(lib):
    from nubovca import capi
    hdr = open(os.path.join(ROOT, "include", "nubovca.h")).read()
    declared = set(re.findall(r"\b(nvca_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(lib, s), s


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.nvca_ctx_create(0, C.byref(h))
    assert rc == -2 and not h.value          # NVCA_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nubomedia-vca_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "synth.py" and False, os.path.join(dp, f)


# ---------------------------------------------------------------------------------------------------------------
# the exception barrier (include/nubovca.h: "never throws"; the reference never lets an error out of the element,
# FACE/kmsfacedetect.cpp:897) and the loader on hostile input -- all on the host, through the ABI
def test_exception_barrier_maps_every_kind(lib):
    NOMEM, INTERNAL = -8, -9
    assert lib.nvca_abi_selftest(0) == NOMEM          # std::bad_alloc
    assert lib.nvca_abi_selftest(1) == NOMEM          # std::length_error out of vector::resize
    assert lib.nvca_abi_selftest(2) == INTERNAL       # std::runtime_error
    assert lib.nvca_abi_selftest(3) == INTERNAL       # throw 42
    assert lib.nvca_abi_selftest(4) == INTERNAL       # std::out_of_range out of vector::at
    assert lib.nvca_abi_selftest(99) == 0


def test_every_entry_point_has_the_barrier():
    """every extern "C" definition with a body of its own is a function-try-block that ends in NVCA_API_CATCH*"""
    src = os.path.join(ROOT, "nubomedia-vca_amd", "csrc")
    trivial = {"nvca_version", "nvca_kernel_name", "nvca_last_error", "nvca_ctx_stream", "nvca_face_stream_destroy"}   # one-liners that allocate nothing
    seen = set()
    for f in ("api.cpp", "parts.cpp", "tracker.cpp"):
        lines = open(os.path.join(src, f)).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r"^(?:int|void|const char \*|void \*)\s*(nvca_[a-z0-9_]+)\(", ln)
            if not m or ln.rstrip().endswith(";"):
                continue
            name = m.group(1)
            if name in trivial:
                seen.add(name)
                continue
            j = i
            while "{" not in lines[j]:
                j += 1
            assert lines[j].strip() == "try {", (f, name, lines[j])
            k = j
            while lines[k] != "}":
                k += 1
            assert lines[k + 1].startswith("NVCA_API_CATCH"), (f, name, lines[k + 1])
            seen.add(name)
    from nubovca import capi
    assert seen == set(capi.SYMBOLS), seen ^ set(capi.SYMBOLS)


def _validate(lib, xml):
    if isinstance(xml, str):
        xml = xml.encode()
    err = C.create_string_buffer(256)
    w, h, ns, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.nvca_cascade_validate_mem(xml, len(xml), C.byref(w), C.byref(h), C.byref(ns), C.byref(nw), err, 256)
    return rc, (w.value, h.value, ns.value, nw.value), err.value.decode(errors="replace")


def test_loader_accepts_the_synthetic_cascades(lib):
    from nubovca import synth
    rc, shape, err = _validate(lib, synth.synthetic_cascade_xml())
    assert rc == 0 and shape[:2] == (20, 20) and shape[2] == 22 and shape[3] == 2135, (rc, shape, err)
    rc, shape, err = _validate(lib, synth.synthetic_part_cascade_xml("mouth"))
    assert rc == 0, err


def test_loader_refuses_garbage_with_a_code(lib):
    """hostile / damaged cascade files end as NVCA_ERR_PARSE (or _UNSUPPORTED / _NOMEM), never as a crash"""
    import numpy as np
    from nubovca import synth
    good = synth.synthetic_cascade_xml(seed=7, stages=[3, 4])
    PARSE, UNSUP = -5, -6
    cases = {
        "empty root": "<opencv_storage></opencv_storage>",
        "not xml": "\x00\x01\x02 garbage <<<>>>",
        "unterminated": good[: len(good) // 2],
        "huge size": good.replace("<size>20 20</size>", "<size>2147483647 2147483647</size>"),
        "negative size": good.replace("<size>20 20</size>", "<size>-5 20</size>"),
        "deep nesting": "<a>" * 200 + "</a>" * 200,
        "2^31 in a rect": re.sub(r"<_>(\d+) (\d+) (\d+) (\d+) ", "<_>2147483647 2147483647 2147483647 2147483647 ", good, count=1),
        "rect beyond the window": re.sub(r"<_>(\d+) (\d+) (\d+) (\d+) ", "<_>19 19 5 5 ", good, count=1),
        "stage graph": good.replace("<next>-1</next>", "<next>1</next>", 1),
    }
    for name, xml in cases.items():
        rc, _, err = _validate(lib, xml)
        assert rc in (PARSE, UNSUP), (name, rc, err)
        assert err, name
    # a child index that does not follow its parent (the device's tree walk would never end): refused by the loader
    tree = synth.generic_cascade_xml(seed=3) if hasattr(synth, "generic_cascade_xml") else None
    if tree and "<left_node>" in tree:
        bad = re.sub(r"<left_node>\d+</left_node>", "<left_node>0</left_node>", tree, count=1)
        rc, _, err = _validate(lib, bad)
        assert rc == PARSE, (rc, err)
    # random byte damage: any status is fine, returning is the point
    rng = np.random.default_rng(5)
    raw = bytearray(good.encode())
    for _ in range(300):
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 12))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        rc, _, _ = _validate(lib, bytes(b))
        assert rc in (0, PARSE, UNSUP, -8, -9), rc
    assert lib.nvca_cascade_validate_mem(None, 5, None, None, None, None, None, 0) == -1


def test_loader_reads_the_handwritten_oldformat_files(lib):
    """tests/golden/oldformat_*.xml are written by hand in cvSave's layout (every other XML the loader sees comes out of
    synth.cascade_to_xml): shapes through the ABI on the host; the values are compared on the GPU box (nvca_cascade_dump)"""
    gold = os.path.join(ROOT, "tests", "golden")
    rc, shape, err = _validate(lib, open(os.path.join(gold, "oldformat_stumps_24x24.xml")).read())
    assert rc == 0 and shape == (24, 24, 3, 7), (rc, shape, err)
    rc, shape, err = _validate(lib, open(os.path.join(gold, "oldformat_trees_tilted_20x20.xml")).read())
    assert rc == 0 and shape == (20, 20, 2, 4), (rc, shape, err)


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Every kernel of the library keeps its state in registers and LDS: a private segment (scratch memory) brings the runtime's
    scratch set-up into every launch, and the one kernel that once had one (56 bytes a thread, a struct indexed by a flag) was the
    kernel of the call that hit a GPU memory access fault under PyTorch's bundled ROCm 7.0 runtime.  Read from the
    code objects inside the built objects."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    build = os.path.join(ROOT, "nubomedia-vca_amd", "build")
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")) or not os.path.isdir(build):
        pytest.skip("no LLVM tools / no build directory")
    seen = 0
    for f in sorted(os.listdir(build)):
        if not f.endswith(".hip.o"):
            continue
        fb, co = str(tmp_path / (f + ".fatbin")), str(tmp_path / (f + ".co"))
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, os.path.join(build, f)], check=True, capture_output=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fb, "--output=" + co], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
        name = None
        for ln in notes.splitlines():
            ln = ln.strip()
            if ln.startswith(".name:"):
                name = ln.split()[-1]
            elif ln.startswith(".private_segment_fixed_size:"):
                seen += 1
                assert int(ln.split()[-1]) == 0, (f, name, ln)
            elif ln.startswith(".uses_dynamic_stack:"):
                assert ln.split()[-1] == "false", (f, name, ln)
    assert seen >= 20, seen
