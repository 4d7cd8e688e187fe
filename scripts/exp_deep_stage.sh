#!/bin/bash
for ds in 5 6 7 8 10 12; do
  NVCA_DEEP_STAGE=$ds python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['roofline']['detail_ms_per_launch']; print('deep_stage=$ds', {k: round(v,3) for k,v in x.items() if k.startswith('cascade')}, 'step', round(d['ms_per_step'],3), 'fps', round(d['value']))" || exit 1
done
