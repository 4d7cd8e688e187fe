#!/bin/bash
# usage (GPU box): scripts/exp_roi_ab.sh <out-dir> "<name> <lib-or-"-"> <ENV=VAL ...>" ...   -- the batched ROI chain (scripts/bench_roi_chain.py) per configuration, two repetitions, alternating
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for cfg in "$@"; do
    read -r name lib envs <<< "$cfg"
    [ "$lib" = "-" ] && lib=""
    env NVCA_LIB=$lib $envs python3 scripts/bench_roi_chain.py --no-contexts > $OUT/$name.$rep.txt 2> $OUT/$name.$rep.err
    python3 - $OUT/$name.$rep.txt $name <<'PY'
import json, sys
out = {}
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        for k, v in d.items():
            if isinstance(v, dict) and "frames_per_s" in v and "contexts" not in v: out[k.replace("roi_chain_batched", "b")] = round(v["frames_per_s"])
print("%-14s" % sys.argv[2], out)
PY
  done
done
