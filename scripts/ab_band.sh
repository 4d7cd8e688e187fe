#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/ab_band.sh <out-dir> <variant.so>...
# A/B of cascade-kernel variants on the headline workload: the shipped library first, then every variant library given
# (nubomedia-vca_amd/variants/*.so, loaded through NVCA_LIB); prints step time and the band kernel's time per launch.
OUT=$1; shift
mkdir -p $OUT
run() {   # name, lib ("" = the shipped one)
  for rep in 1 2; do
    NVCA_LIB=$2 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/$1.$rep.json 2> $OUT/$1.$rep.err
    python3 - $OUT/$1.$rep.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    det = d["roofline"]["detail_ms_per_launch"]
    print("%-14s step %.3f ms  %.0f frames/s  band %.3f  deep %.3f  group %.3f  integral %.3f  gray %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], det.get("cascade_band", 0), det.get("cascade_deep", 0), det.get("group_rects", 0), det.get("integral_rows", 0), det.get("gray_resize_hist", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
}
run shipped ""
for v in "$@"; do run $(basename $v .so) $PWD/$v; done
