#!/bin/bash
# usage: [BENCH_ARGS=...] scripts/exp_env_sweep.sh VAR v1 v2 ...   (GPU box) -- bench line per value of an environment switch: step time + per-kernel ms
VAR=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sweep_$VAR
mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $BENCH_ARGS > $OUT/$v.json 2> $OUT/$v.err || { echo "$VAR=$v FAILED"; tail -3 $OUT/$v.err; continue; }
  python3 - "$OUT/$v.json" "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = (d.get("roofline") or {}).get("detail_ms_per_launch", {})
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "fps %.0f" % d["value"], " ".join("%s=%.3f" % (n, v) for n, v in sorted(k.items()) if v > 0.01))
PY
done
