#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (run on the GPU box via gpurun)
# 1. bench JSON  2. rocprofv3 kernel stats of the same command  3. PMC passes (FETCH_SIZE / WRITE_SIZE separately)
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --host-frames > $OUT/bench_host_frames.json 2>> $OUT/bench.err
python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --sync --frames-per-step 1 > $OUT/bench_single_frame.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT > $OUT/pmc_summary.txt 2>&1
# BASELINE config 3 with its breakdown, and the N > 1 command line as the driver types it (two ranks rehearsed on this one GPU, gloo)
python3 $GRAFT_REPO_ROOT/scripts/bench_roi_chain.py > $OUT/roi_chain.txt 2> $OUT/roi_chain.err
NVCA_BENCH_REHEARSAL=1 python3 $GRAFT_REPO_ROOT/bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/bench.json
