#!/bin/bash
# r03ab: round-3 fuzzing on the product build: random RDF / SDF / distance scenarios against the oracle, random XTC frames through every
# device decoder against the byte-wise restatement, random filtered evaluations (batches of frame blocks) against plain ones
T=${1:-r03ab}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
{ echo "# python scripts/fuzz_gpu.py 5000 5150 20"; timeout 900 python scripts/fuzz_gpu.py 5000 5150 20 2>&1 | grep -v amdgpu.ids | tail -4
  echo "# python scripts/fuzz_xtc.py 400 21 gpu"; timeout 900 python scripts/fuzz_xtc.py 400 21 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -3 | cut -c1-400
  echo "# python scripts/fuzz_emu_filtered.py 40 5 gpu"; timeout 900 python scripts/fuzz_emu_filtered.py 40 5 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -4 | cut -c1-400
} | tee $O/fuzz.txt
