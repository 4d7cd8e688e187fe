#!/bin/bash
for w in "--workload streams720p" "--workload face_tracker" "--width-to-process 160 --scale-factor-pct 25" "--width-to-process 640 --scale-factor-pct 25" "--workload streams720p --streams-per-gpu 64"; do
  python bench.py --no-cpu-baseline --steps 10 --warmup 2 $w 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['roofline']['detail_ms_per_launch']; print('$w', 'fps', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in x.items()})" || exit 1
done
