#!/bin/bash
# where does k_tile's time go: staging only / + queue build / + stages up to s
for e in 1 2 12 13 14 15 0; do
  NVCA_EXP=$e python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exp=$e', 'tile', round(d['roofline']['detail_ms_per_launch']['cascade_tile'],3), 'step', round(d['ms_per_step'],3))" || exit 1
done
for ds in 4 5 7 8 10; do
  NVCA_DEEP_STAGE=$ds python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['roofline']['detail_ms_per_launch']; print('deep_stage=$ds', 'tile', round(x['cascade_tile'],3), 'deep', round(x['cascade_deep'],3), 'step', round(d['ms_per_step'],3))" || exit 1
done
