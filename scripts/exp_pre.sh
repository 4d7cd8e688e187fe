#!/bin/bash
NVCA_SKIP_CASCADE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $@ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.3f'%d['ms_per_step'], {n:round(v,4) for n,v in d['roofline']['detail_ms_per_launch'].items()})"
