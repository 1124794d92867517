#!/bin/bash
# PMC passes for the dominant kernel (separate passes, kernel-trace only: see the brief's rocprofv3 rules).
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_pmc.sh <tag> [bench args...]'
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${FRAMES:+--frames $FRAMES} $*"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "== pass $i: $line"
  timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $BENCH > $OUT/p$i.log 2>&1
  echo "rc=$?"
done <<'PASSES'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
FETCH_SIZE
WRITE_SIZE
PASSES
python $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# drop bulky traces, keep counter CSVs
find $OUT -name "*kernel_trace.csv" -size +5M -delete
