#!/bin/bash
# usage (GPU box): scripts/exp_roi_trace.sh <out-dir>  -- per-launch durations of the ROI chain's kernels (rocprofv3 kernel trace of scripts/bench_roi_chain.py)
OUT=$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/bench_roi_chain.py > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/*/*kernel_trace.csv")
if not f:
    print("no kernel trace"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("nvca::", "").replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-28s calls %5d  total %9.1f us  median %7.1f  p90 %7.1f  max %7.1f" % (n[:28], len(v), sum(v), v2[len(v2) // 2], v2[int(len(v2) * 0.9)], v2[-1]))
# the k_roi launches of the batched ticks, in order: durations of the last 60
k = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, r.get("Workgroup_Size_X", ""), r.get("Grid_Size_X", ""), r.get("LDS_Block_Size", "")) for r in rows if "k_roi" in r["Kernel_Name"]]
k.sort()
print("last k_roi launches (us, grid, lds):", [(round(x[1], 1), x[3], x[4]) for x in k[-18:]])
PY
