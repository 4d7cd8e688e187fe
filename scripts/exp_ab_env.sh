#!/bin/bash
# usage (GPU box): scripts/exp_ab_env.sh <out-dir> "<name> <lib-or-"-"> <ENV=VAL ...>" ...   -- bench line per configuration, two repetitions, alternating
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for cfg in "$@"; do
    read -r name lib envs <<< "$cfg"
    [ "$lib" = "-" ] && lib=""
    env NVCA_LIB=$lib $envs python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $BENCH_ARGS > $OUT/$name.$rep.json 2> $OUT/$name.$rep.err
    python3 - $OUT/$name.$rep.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    det = d["roofline"]["detail_ms_per_launch"]
    print("%-16s step %.3f ms  %.0f frames/s  band %.3f  deep %.3f  group %.3f  integral %.3f  colsum %.3f  gray %.3f" % (sys.argv[2], d["ms_per_step"], d["value"], det.get("cascade_band", 0), det.get("cascade_deep", 0), det.get("group_rects", 0), det.get("integral_rows", 0), det.get("integral_colsum", 0), det.get("gray_resize_hist", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
