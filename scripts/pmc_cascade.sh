#!/bin/bash
# PMC passes for the cascade kernel (separate runs; kernel-trace only alongside)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline ${@:2}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
