#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/exp_pre_cus.sh <out-dir> [n ...]
# NVCA_PRE_CUS sweep on the headline workload: the next batch's pre-processing on a stream confined to n CUs, beside the band kernel
OUT=$1; shift
mkdir -p $OUT
for n in "$@"; do
  for rep in 1 2; do
    NVCA_PRE_CUS=$n python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/pre$n.$rep.json 2> $OUT/pre$n.$rep.err
    python3 - $OUT/pre$n.$rep.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    det = d["roofline"]["detail_ms_per_launch"]
    print("pre_cus %-4s step %.3f ms  %.0f frames/s  band %.3f  deep %.3f  group %.3f  integral %.3f  colsum %.3f gray %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], det.get("cascade_band", 0), det.get("cascade_deep", 0), det.get("group_rects", 0), det.get("integral_rows", 0), det.get("integral_colsum", 0), det.get("gray_resize_hist", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
