#!/bin/bash
# usage: scripts/r5_roi.sh <tag>  -- part-detector parity tests, ROI chain rates, k_roi phase stamps (diagnostic build)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parts.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $OUT/tests.txt 2>&1; rc=$?; tail -3 $OUT/tests.txt; echo "tests rc=$rc"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert\|FAILED" $OUT/tests.txt | head -30; exit $rc; }
timeout -k 10 400 python3 scripts/bench_roi_chain.py > $OUT/chain.out 2> $OUT/chain.err || { echo chain failed; tail -5 $OUT/chain.err; exit 1; }
cut -c1-420 $OUT/chain.out
NVCA_LIB=$GRAFT_REPO_ROOT/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=$OUT/stamps timeout -k 10 400 python3 scripts/exp_roi_stamps.py > $OUT/stamps.out 2>&1 && cat $OUT/stamps.roi.txt
