#!/bin/bash
# r03x: batches of frame blocks (filtered evaluation): parity tests + the tax of keeping block partials, A/B of the two options
T=${1:-r03x}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "filtered or overflow or blocks" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for S in 125 50 25; do
  for o in "block_superbatch=0" "block_two_streams=0" ""; do
    timeout 300 python scripts/exp_filtered.py c2 $S $o 2>> $O/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('c2 S=%d %-22s plain %.3f ms  with blocks %.3f ms (+%.1f %%)  filtered [103,897) %.3f ms' % (d['block_frames'], d['options'], d['full_plain_ms'], d['full_with_blocks_ms'], 100 * (d['full_with_blocks_ms'] / d['full_plain_ms'] - 1), d['range_103_897']['filtered_ms']))" | tee -a $O/filtered_ab.txt
  done
done
for o in "block_superbatch=0" ""; do
  timeout 300 python scripts/exp_filtered.py c3 125 $o 2>> $O/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('c3 S=%d %-22s plain %.3f ms  with blocks %.3f ms (+%.1f %%)' % (d['block_frames'], d['options'], d['full_plain_ms'], d['full_with_blocks_ms'], 100 * (d['full_with_blocks_ms'] / d['full_plain_ms'] - 1)))" | tee -a $O/filtered_ab.txt
done
timeout 300 python scripts/exp_filtered.py c4 500 2>> $O/err.log | tee -a $O/filtered_ab.txt
grep -v amdgpu.ids $O/err.log | tail -5
