#!/bin/bash
# k_band block -> (frame, band) mapping experiment: time and L2-miss traffic (FETCH_SIZE) of the default mapping against
# NVCA_BAND_MAP variants.  Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/bandmap
mkdir -p $OUT
for m in 0 1 2; do
  export NVCA_BAND_MAP=$m
  python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  python3 - <<PY
import json
d = json.loads(open("$OUT/bench_$m.json").read().strip().splitlines()[-1])
print("map $m: %.0f fps, band %.4f ms, deep %.4f ms" % (d["value"], d["roofline"]["detail_ms_per_launch"]["cascade_band"], d["roofline"]["detail_ms_per_launch"]["cascade_deep"]))
PY
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/m$m/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$m.log 2>&1
  python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT/m$m | grep -A1 "k_band"
done
