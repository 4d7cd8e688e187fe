#!/usr/bin/env python3
"""Phase timeline of k_band from a diagnostic build (NVCA_BUILD_STAMPS=1 python nubomedia-vca_amd/build.py --force;
NVCA_STAMPS_OUT=file at run time): thread 0 of the first 64 workgroups stamps s_memtime at every phase boundary of its
first 16 tiles.  Prints the median cycles per phase."""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 16, 64).astype(np.int64)
names = {0: "tile top (after barrier)", 1: "coords loaded + sq gathers issued", 2: "DMA issued + maps scattered", 3: "samples landed (wait + barrier)",
         4: "variance + stage 0 (barrier)", 5: "adaptive + push (barrier) + carry", 6: "stage loop done (barrier)", 7: "list append done"}
rows = []
for b in range(64):
    for t in range(15):
        s = d[b, t]
        if s[0] == 0 or s[5] == 0 or d[b, t + 1][0] == 0:
            continue
        r = {"fill_coords": s[1] - s[0], "fill_commit": s[2] - s[1], "fill_wait": s[3] - s[2], "stage0": s[4] - s[3], "adaptive": s[5] - s[4]}
        prev = s[5]
        # the stages in the order this tile walked them (the band kernel orders stages 1 .. 5 by what the previous tile saw)
        walked = sorted((st for st in range(1, 7) if s[8 + 8 * st] != 0), key=lambda st: s[8 + 8 * st])      # stages 1 .. 6 have stamp words (a round behind stage 6 shows up in end_barrier)
        r["order"] = int("".join(str(st) for st in walked) or "0")
        for st in walked:
            o = 8 + 8 * st
            r["s%d_top_barrier" % st] = s[o] - prev
            if s[o + 1]:
                r["s%d_stumps" % st] = s[o + 1] - s[o]
                r["s%d_psum_barrier" % st] = s[o + 2] - s[o + 1]
                r["s%d_reduce_push" % st] = s[o + 3] - s[o + 2]
                prev = s[o + 3]
            else:
                prev = s[o]
            r["s%d_n" % st] = s[o + 7]
        if s[6]:
            r["end_barrier"] = s[6] - prev
            r["append"] = (s[7] - s[6]) if s[7] else 0
        r["tile_total"] = d[b, t + 1][0] - s[0]
        rows.append(r)
keys = []
for r in rows:
    for k in r:
        if k not in keys:
            keys.append(k)
print("tiles:", len(rows))
orders = {}
for r in rows:
    orders[r["order"]] = orders.get(r["order"], 0) + 1
print("stage orders walked (digits = stages in order, tiles):", sorted(orders.items(), key=lambda kv: -kv[1])[:8])
for k in keys:
    if k == "order":
        continue
    v = np.array([r[k] for r in rows if k in r])
    print("%-22s median %8.0f  mean %8.0f  (n=%d)" % (k, np.median(v), v.mean(), len(v)))
