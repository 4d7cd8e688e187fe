#!/bin/bash
# round 4, first GPU call: suite, calibrated-cascade bench (both stage orders), stand-in bench, phase stamps on the calibrated cascade
OUT=gpurun_out/r4a; mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_cal.json 2> $OUT/bench_cal.err; echo "bench rc=$?"
NVCA_STAGE_ORDER=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_cal_order0.json 2>> $OUT/bench_cal.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --cascade standin > $OUT/bench_standin.json 2>> $OUT/bench_cal.err
NVCA_LIB=$PWD/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=$PWD/$OUT/stamps.bin python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_stamps.json 2> $OUT/bench_stamps.err
python3 scripts/stamps.py $OUT/stamps.bin > $OUT/stamps_k_band.txt 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); det=d["roofline"]["detail_ms_per_launch"]
        print(f, "step %.3f ms %.0f fps"%(d["ms_per_step"],d["value"]), {k:round(v,3) for k,v in det.items()})
    except Exception as e: print(f,"FAILED",e)
PY
cat $OUT/stamps_k_band.txt
