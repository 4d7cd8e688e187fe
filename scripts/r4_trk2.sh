#!/bin/bash
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/exp_trk_alone.py 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/exp_trk_alone.py > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ccl' in r['Name'] or 'trk' in r['Name']:
            print("%-40s calls %6s avg %8.1f us total %8.1f ms"%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
find $OUT/trace -name "*.csv" -size +1M -delete
