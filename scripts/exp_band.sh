#!/bin/bash
# band kernel vs stage-0 pre-pass + tile kernel, at several batch sizes
for fps in 32 8 1; do
for b in 0 1; do
  NVCA_BAND=$b python bench.py --no-cpu-baseline --steps 10 --warmup 2 --frames-per-step $fps 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['roofline']['detail_ms_per_launch']; print('frames=$fps band=$b', {k: round(v,3) for k,v in x.items() if k.startswith('cascade')}, 'step', round(d['ms_per_step'],3), 'fps', round(d['value']))" || exit 1
done; done
