#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
for cfg in "NVCA_LISTS=0" "NVCA_LIST_FROM=2" "NVCA_LIST_FROM=3" "NVCA_LIST_FROM=4" "NVCA_LIST_FROM=3 NVCA_DEEP_STAGE=8"; do
  for args in "--faces 4" "--faces 4 --frames-per-step 1"; do
  echo "== $cfg $args"
  env $cfg python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps %.0f ms/step %.2f'%(d['value'], d['ms_per_step']), {n:round(v,3) for n,v in d['roofline']['detail_ms_per_launch'].items() if 'cascade' in n})"
  done
done
