#!/bin/bash
# usage: scripts/r4_trk.sh <tag>  -- tracker parity tests, face + tracker tick breakdown, per-kernel durations of the trackers alone
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest -x -q tests/test_gpu_tracker.py "tests/test_gpu_timed_path.py::test_face_tracker_batch_8x1080p_vs_oracle" > $OUT/tests.txt 2>&1; rc=$?; tail -3 $OUT/tests.txt
[ $rc -ne 0 ] && { grep -n "Error\|assert" $OUT/tests.txt | head; exit $rc; }
timeout -k 10 300 python3 scripts/exp_face_tracker.py > $OUT/face_tracker.txt 2>&1; cat $OUT/face_tracker.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/exp_face_tracker.py > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ccl' in r['Name'] or 'trk' in r['Name']:
            print("%-40s calls %6s avg %8.1f us total %8.1f ms"%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
find $OUT/trace -name "*.csv" -size +1M -delete
