#!/bin/bash
# usage (GPU box): scripts/r4_pmc.sh <tag> [extra bench args]   -- counters of the headline launch set, separate passes (MI355X_MICROARCH.md: rocprofv3 PMC slots)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary $*"
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summarize.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -A30 "k_band" $OUT/pmc_summary.txt | head -40
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*agent_info.csv" -delete
