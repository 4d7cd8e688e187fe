#!/bin/bash
# usage: scripts/r4_run.sh <tag> [tests-selection...]  -- a parity selection, the headline bench (calibrated + stand-in cascade) and wave 0's phase stamps
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ $# -gt 0 ]; then
  timeout -k 10 600 python3 -m pytest -x -q "$@" > $OUT/tests.txt 2>&1; rc=$?; tail -3 $OUT/tests.txt; echo "tests rc=$rc"
  [ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $OUT/tests.txt | head -20; exit $rc; }
fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_cal.json 2> $OUT/bench_cal.err || { echo bench failed; tail -5 $OUT/bench_cal.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --cascade standin > $OUT/bench_standin.json 2>> $OUT/bench_cal.err
NVCA_LIB=$PWD/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=$PWD/$OUT/stamps.bin timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_stamps.json 2> $OUT/bench_stamps.err
python3 scripts/stamps.py $OUT/stamps.bin > $OUT/stamps_k_band.txt 2>&1
python3 - $OUT <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
        print(f, "step %.3f ms %.0f fps"%(d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items()})
    except Exception as e: print(f,"FAILED",e)
PY
cat $OUT/stamps_k_band.txt
