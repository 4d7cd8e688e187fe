"""The parity tests of the primitives and the detectMultiScale variants once more, in a child process whose device buffers are
mapped between unmapped guard ranges and end where their mappings end (NVCA_ALLOC_GUARD=2: csrc/api.cpp, "electric fence"; released
buffers are unmapped too): a kernel that reads or writes past one of the library's buffers, or touches a released one, faults at
that access and takes the child down -- every time, not now and then as on an ordinary heap.  (In a child: a fault ends the process.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_kernels_stay_inside_their_buffers():
    env = dict(os.environ, NVCA_ALLOC_GUARD="2", NVCA_ALLOC_LOG="1", HSA_ENABLE_VM_FAULT_MESSAGE="1", NVCA_GUARD_CHILD="1")
    cmd = [sys.executable, "-m", "pytest", "-s", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           "-k", "detect or resize or integral or gray or equalize or flip or group"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    if "guard unavailable" in out:
        pytest.skip("this runtime refuses the virtual-memory calls the guard allocator needs")
    tail = "\n".join(out.splitlines()[-40:])
    summary = [ln for ln in out.splitlines() if " passed" in ln or " failed" in ln]
    print("guarded child:", summary[-1] if summary else "(no summary line)", "| guarded allocations:", out.count("guarded)"))
    assert "Memory access fault" not in out, tail
    assert r.returncode == 0, tail
    assert " passed" in out and out.count("guarded)") > 20, tail
