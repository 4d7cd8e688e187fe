#!/bin/bash
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/exp_roi_stamps.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_gaps.py $OUT/trace 0.5 > $OUT/gaps.txt 2>&1; cat $OUT/gaps.txt
find $OUT/trace -name "*.csv" -size +1M -delete
