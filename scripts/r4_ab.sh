#!/bin/bash
# usage: scripts/r4_ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   -- the headline bench under each environment (first argument may be "" = defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for e in "$@"; do
  i=$((i+1)); name=$(echo "v${i}_$e" | tr ' =' '__')
  env $e timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err || echo "$e failed"
  python3 - $OUT/$name.json "$e" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
    print("%-40s step %.3f ms %.0f fps"%(sys.argv[2] or "(defaults)",d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items() if k.startswith('cascade') or k.startswith('group')}, "match" if d.get("boxes_match") else "")
except Exception as e: print(sys.argv[2],"FAILED",e)
PY
done
