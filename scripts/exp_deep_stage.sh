#!/bin/bash
python -m pytest tests -x -q -m gpu -k degenerate 2>&1 | grep -E "Error|error|assert" | head -5
for ds in 4 5 6 8; do
  echo "== NVCA_DEEP_STAGE=$ds"
  for args in "--faces 4" "--faces 0"; do
  NVCA_DEEP_STAGE=$ds python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps %.0f ms/step %.2f'%(d['value'], d['ms_per_step']), {n:round(v,3) for n,v in d['roofline']['detail_ms_per_launch'].items() if 'cascade' in n})"
  done
done
