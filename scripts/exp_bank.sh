#!/bin/bash
for e in 0 3 4; do
  NVCA_EXP=$e python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['roofline']['detail_ms_per_launch']; print('exp=$e band', round(x['cascade_band'],3), 'step', round(d['ms_per_step'],3))" || exit 1
done
