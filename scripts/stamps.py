#!/usr/bin/env python3
"""Phase timeline of k_band from a diagnostic build (NVCA_BUILD_VARIANT=stamps NVCA_BUILD_STAMPS=1 python nubomedia-vca_amd/build.py;
NVCA_LIB=.../variants/stamps.so NVCA_STAMPS_OUT=file at run time): thread 0 (wave 0) of the first 64 workgroups stamps s_memtime
at every phase boundary of its first 16 tiles.  Prints the median cycles per phase.  Since round 4 the waves of a tile walk the
stages on their own: the stage rows are WAVE 0's walk (its window count k at the top of each stage)."""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 64).astype(np.int64)
rows = []
for b in range(64):
    for t in range(15):
        s = d[b, t]
        if s[0] == 0 or s[4] == 0 or d[b, t + 1][0] == 0:
            continue
        r = {"fill_coords_sq": s[1] - s[0], "fill_commit": s[2] - s[1], "fill_wait_barrier": s[3] - s[2], "var_stage0_list": s[4] - s[3], "k_after_stage0": s[5]}
        prev = s[4]
        for st in range(1, 22):
            o = 8 + 2 * st
            if s[o] == 0:
                break
            if st > 1:
                r["s%d_cycles" % (st - 1)] = s[o] - prev
            r["s%d_k" % st] = s[o + 1]
            prev = s[o]
            last = st
        else:
            last = 21
        if s[6] and prev != s[4]:
            r["s%d_cycles" % last] = s[6] - prev
        r["walk_total"] = (s[6] - s[4]) if s[6] else 0
        r["append"] = (s[7] - s[6]) if s[6] and s[7] else 0
        r["wave0_busy"] = (s[7] if s[7] else s[6]) - s[0]
        r["tile_total"] = d[b, t + 1][0] - s[0]
        r["wait_for_other_waves"] = r["tile_total"] - r["wave0_busy"]
        rows.append(r)
keys = []
for r in rows:
    for k in r:
        if k not in keys:
            keys.append(k)
print("tiles:", len(rows))
for k in keys:
    v = np.array([r[k] for r in rows if k in r])
    print("%-22s median %8.0f  mean %8.0f  (n=%d)" % (k, np.median(v), v.mean(), len(v)))
