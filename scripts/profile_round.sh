#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (run on the GPU box via gpurun)
# 1. bench JSON (the default command line)  2. rocprofv3 kernel stats of the same command  3. PMC passes (FETCH_SIZE / WRITE_SIZE separately)
# 4. phase stamps of k_band (diagnostic build)  5. the other workloads' scripts
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 600 python3 $B --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames > $OUT/bench_host_frames.json 2>> $OUT/bench.err
timeout -k 10 300 python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --sync --frames-per-step 1 > $OUT/bench_single_frame.json 2>> $OUT/bench.err
timeout -k 10 300 python3 $B --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --cascade standin > $OUT/bench_standin_cascade.json 2>> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $B --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT > $OUT/pmc_summary.txt 2>&1
NVCA_LIB=$GRAFT_REPO_ROOT/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=$OUT/stamps.bin timeout -k 10 300 python3 $B --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_stamps.json 2> $OUT/bench_stamps.err
python3 $GRAFT_REPO_ROOT/scripts/stamps.py $OUT/stamps.bin > $OUT/stamps_k_band.txt 2>&1
# BASELINE config 3 with its breakdown, the trackers alone, and the N > 1 command line as the driver types it (two ranks rehearsed on this one GPU, gloo)
timeout -k 10 600 python3 $GRAFT_REPO_ROOT/scripts/bench_roi_chain.py > $OUT/roi_chain.txt 2> $OUT/roi_chain.err
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/exp_face_tracker.py > $OUT/face_tracker.txt 2> $OUT/face_tracker.err
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/scripts/exp_trk_alone.py > $OUT/tracker_alone.txt 2> $OUT/tracker_alone.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trk_stats -- python3 $GRAFT_REPO_ROOT/scripts/exp_trk_alone.py > $OUT/trk_stats.log 2>&1
python3 - $OUT >> $OUT/tracker_alone.txt <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/trk_stats/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ccl' in r['Name'] or 'trk' in r['Name'] or 'rocclr' in r['Name']:
            print("%-44s calls %6s avg %8.1f us total %8.2f ms"%(r['Name'][:44],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
NVCA_LIB=$GRAFT_REPO_ROOT/nubomedia-vca_amd/variants/stamps.so NVCA_STAMPS_OUT=$OUT/stamps_roi timeout -k 10 400 python3 $GRAFT_REPO_ROOT/scripts/exp_roi_stamps.py > $OUT/stamps_roi.log 2>&1; cp $OUT/stamps_roi.roi.txt $OUT/stamps_k_roi.txt 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/chain_trace -- python3 $GRAFT_REPO_ROOT/scripts/exp_roi_stamps.py > $OUT/chain_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_gaps.py $OUT/chain_trace 0.5 > $OUT/roi_chain_gaps.txt 2>&1
NVCA_BENCH_REHEARSAL=1 timeout -k 10 600 python3 $B --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete; rm -f $OUT/stamps.bin
python3 - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f fps step %.3f ms boxes_match %s"%(d["value"],d["ms_per_step"],d.get("boxes_match")))
print({k:round(v,3) for k,v in d["roofline"]["detail_ms_per_launch"].items()})
print("roofline", {k:d["roofline"][k] for k in ("frac","traffic","traffic_source")}, d["roofline"]["dominant_launch"])
w=d.get("secondary",{}).get("workloads",{}); print({k:(round(v["frames_per_s"]) if v.get("frames_per_s") else None) for k,v in w.items()})
s=d.get("secondary",{}); print({k:s[k] for k in ("single_frame","host_frames","other_cascade") if k in s})
print(d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("single_core_value"))
PY
