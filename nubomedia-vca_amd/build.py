"""Builds libnubovca_hip.so (hand-written HIP kernels + C ABI) for gfx950, in tree.

    python nubomedia-vca_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
-ffp-contract=off on host AND device code: the reference arithmetic has no FMA
contraction, and box exactness depends on every rounding.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# NVCA_BUILD_VARIANT=name: a side build (own object directory, variants/name.so, extra flags from NVCA_BUILD_FLAGS) for A/B
# runs through NVCA_LIB; the shipped library is never touched by it
VARIANT = os.environ.get("NVCA_BUILD_VARIANT")
OBJ = os.path.join(HERE, "build" if not VARIANT else "build_" + VARIANT)
LIB = os.path.join(HERE, "libnubovca_hip.so") if not VARIANT else os.path.join(HERE, "variants", VARIANT + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]
if os.environ.get("NVCA_BUILD_STAMPS"):      # diagnostic build: in-kernel phase stamps (scripts/stamps.py); never the shipped library
    COMMON.append("-DNVCA_STAMPS")
COMMON += os.environ.get("NVCA_BUILD_FLAGS", "").split()
ARCH = ["--offload-arch=gfx950"]


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    hdrs.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            cmd = [HIPCC] + COMMON + (ARCH + ["-x", "hip"] if f.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC] + ARCH + ["-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
