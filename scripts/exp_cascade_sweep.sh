#!/bin/bash
# sweep content / batch to see what the cascade kernel's time depends on
for args in "--content natural --faces 4 --frames-per-step 32" "--content natural --faces 0 --frames-per-step 32" "--content noise --faces 0 --frames-per-step 32" "--content natural --faces 4 --frames-per-step 1" "--content natural --faces 4 --frames-per-step 4" "--content natural --faces 0 --frames-per-step 1"; do
  echo "== $args"
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('fps %.0f ms/step %.2f'%(d['value'], d['ms_per_step']), {n:round(v,3) for n,v in d['roofline']['detail_ms_per_launch'].items()})"
done
