#!/bin/bash
for cfg in "NVCA_NO_TILES=1" "NVCA_TILE_MIN_TW=6" "NVCA_TILE_MIN_TW=10" "NVCA_TILE_MIN_TW=14"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 5 --warmup 1 --no-cpu-baseline --faces 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps %.0f ms/step %.2f'%(d['value'], d['ms_per_step']), {n:round(v,3) for n,v in d['roofline']['detail_ms_per_launch'].items() if 'cascade' in n})"
done
