#!/bin/bash
# A/B of the size of the persistent EM grids (RPVG_HIP_EM_GRID_SCALE)
cd /root/repo
for v in 1.0 0.5 0.25 0.125; do
  for i in 1 2; do
    RPVG_HIP_EM_GRID_SCALE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
    python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); k=d["kernels"]
print("$v", round(d["ms_per_step"],2), "em", round(k["em_sparse_ms_per_step"],2), "collapse", round(k.get("collapse_ms_per_step",0),2), "busy", round(k.get("busy_ms_per_step",0),2))
PY
  done
done
