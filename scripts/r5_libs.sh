#!/bin/bash
# usage: scripts/r5_libs.sh <tag> <variant> ...   -- the headline bench with the shipped library ("base") and with variants/<variant>.so
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset NVCA_LIB; else export NVCA_LIB=$GRAFT_REPO_ROOT/nubomedia-vca_amd/variants/$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $BENCH_ARGS > $OUT/$v.json 2> $OUT/$v.err || echo "$v failed"
  python3 - $OUT/$v.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
    print("%-24s step %.3f ms %.0f fps"%(sys.argv[2],d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items() if k.startswith('cascade')}, "match" if d.get("boxes_match") else "")
except Exception as e: print(sys.argv[2],"FAILED",e)
PY
done
