#!/bin/bash
for args in "--workload face1080p" "--workload streams720p" "--workload face_tracker" "--workload face1080p --width-to-process 160 --scale-factor-pct 25 --frames-per-step 64"; do
  echo "== $args"
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps %.0f ms/step %.2f'%(d['value'], d['ms_per_step']), d['config']['workload'][:70], {n:round(v,3) for n,v in d['roofline']['detail_ms_per_launch'].items()})"
done
