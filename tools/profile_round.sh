#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py + separate PMC passes.
# Usage: bash tools/profile_round.sh <round-tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra $@"
echo "== kernel trace" | tee $OUT/log.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $BENCH >> $OUT/log.txt 2>&1 || echo "trace rc=$?" >> $OUT/log.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass" >> $OUT/log.txt
  timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name --output-format csv -- $BENCH >> $OUT/log.txt 2>&1 || echo "pmc $pass rc=$?" >> $OUT/log.txt
done
find $OUT -name "*.csv" | head -50 >> $OUT/log.txt
tail -5 $OUT/log.txt
