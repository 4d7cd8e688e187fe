#!/bin/bash
# usage: scripts/r5_host.sh <tag>  -- pageable host frames with streaming-store copies into the bounce slots and with plain memcpy (same box)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
for NT in 0 1 0 1; do
  for extra in "" "--width-to-process 160 --scale-factor-pct 25"; do
    NVCA_NT_COPY=$NT timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames $extra 2>> $OUT/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NT=$NT $extra: %.3f ms %.0f fps'%(d['ms_per_step'],d['value']))"
  done
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_hoststage.py -x -q 2>&1 | tail -2
