#!/bin/bash
# usage: scripts/r4_sweep.sh <tag> VAR v1 v2 ...   -- the headline bench under one environment switch's values
TAG=$1; VAR=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_${VAR}_$v.json 2> $OUT/bench_${VAR}_$v.err || echo "$v failed"
done
python3 - $OUT <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
        print(f.split('/')[-1], "step %.3f ms %.0f fps"%(d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items() if k.startswith('cascade') or k.startswith('group')})
    except Exception as e: print(f,"FAILED",e)
PY
