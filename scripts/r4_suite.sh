#!/bin/bash
# usage: scripts/r4_suite.sh <tag>  -- full GPU suite, headline bench, host-frame benches (pageable / page-locked)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?; tail -3 $OUT/tests.txt; echo "tests rc=$rc"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert\|FAILED" $OUT/tests.txt | head -30; exit $rc; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_cal.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames > $OUT/bench_host.json 2>> $OUT/bench.err
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames --pinned > $OUT/bench_host_pinned.json 2>> $OUT/bench.err
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames --width-to-process 160 --scale-factor-pct 25 > $OUT/bench_host_w160.json 2>> $OUT/bench.err
python3 - $OUT <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
        print(f.split('/')[-1], "step %.3f ms %.0f fps"%(d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items() if k.startswith('cascade')})
    except Exception as e: print(f,"FAILED",e)
PY
