#!/bin/bash
# c3 with the pair kernel's launch knobs (bench.py --opt): usage on the GPU box: bash scripts/sweep_pair.sh
cd ${GRAFT_REPO_ROOT:-.}
for O in "rdf_blocks=2048" "rdf_blocks=1792" "rdf_blocks=1536" "rdf_shared_hist=1" "rdf_nsub_pct=75" "rdf_nsub_pct=150" "nxf_divisor=12" "nxf_divisor=24" "rdf_blocks=2048"; do
  python bench.py --workload c3 --frames 500 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --opt $O 2>/dev/null > /tmp/s.json
  python - <<PY
import json
d=json.load(open("/tmp/s.json")); print("$O", round(d["value"]), "frames/s", {k: round(v/d["steps"],2) for k,v in d["kernel_ms"].items() if k in ("rdf_pencil","cells_build")})
PY
done
