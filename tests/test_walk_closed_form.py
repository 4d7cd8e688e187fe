"""OpenCV's adaptive x step (cvHaarDetectObjectsForROC: `ix += result != 0 ? 1 : 2`, result == 0 = rejected by stage 0) as the closed
form every evaluator of this repository uses instead of the serial walk -- the band / tile / small-image kernels per 64-window chunk
with a carried parity, and the host when it replays a FIND_BIGGEST search's narrowed re-scan from the stage-0 reject bits of the full
grid (api.cpp, fb_visited): window ix is visited iff the run of stage-0 rejects immediately left of it, not reaching below the walk's
start column, has even length.  Checked here against the walk itself on random reject patterns, start columns and row lengths
(pure Python: the statement the GPU parity tests rely on, separated from any kernel)."""
import numpy as np


def serial_walk(rej, start, end):
    seen = np.zeros(len(rej), bool)
    x = start
    while x < end:
        seen[x] = True
        x += 2 if rej[x] else 1
    return seen


def closed_form(rej, start, end):
    seen = np.zeros(len(rej), bool)
    for ix in range(start, end):
        run, x = 0, ix - 1
        while x >= start and rej[x]:
            run += 1
            x -= 1
        seen[ix] = run % 2 == 0
    return seen


def chunked(rej, start, end, chunk=64):
    """the kernels' form: 64 windows at a time, the parity of the reject run that ends at a chunk's right edge carried into the next"""
    seen = np.zeros(len(rej), bool)
    carry = 0
    for c0 in range(start, end, chunk):
        n = min(chunk, end - c0)
        for lane in range(n):
            ones = 0
            while ones < lane and rej[c0 + lane - 1 - ones]:
                ones += 1
            parity = (lane + carry) & 1 if ones == lane else ones & 1
            seen[c0 + lane] = not parity
        tail = 0
        while tail < n and rej[c0 + n - 1 - tail]:
            tail += 1
        carry = (n + carry) & 1 if tail == n else tail & 1
    return seen


def test_closed_form_equals_the_serial_walk():
    rng = np.random.RandomState(5)
    for it in range(600):
        n = int(rng.randint(1, 200))
        p = float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0]))
        rej = rng.rand(n) < p
        start = int(rng.randint(0, n))
        end = int(rng.randint(start, n + 1))
        w = serial_walk(rej, start, end)
        assert np.array_equal(w, closed_form(rej, start, end)), (it, n, p, start, end)
        assert np.array_equal(w, chunked(rej, start, end)), (it, n, p, start, end)


def test_a_narrowed_walk_is_not_a_subset_of_the_full_one():
    """why the narrowed re-scan cannot be cut out of the full scan's candidates: started further right, the walk visits windows the full
    walk skipped (the dense first launch evaluates them all, the reject bits say which walk sees which)"""
    rej = np.array([1, 0, 0, 1, 1, 0], bool)
    full = serial_walk(rej, 0, 6)
    narrowed = serial_walk(rej, 1, 6)
    assert not full[1] and narrowed[1]
    assert np.array_equal(narrowed, closed_form(rej, 1, 6))
