#!/bin/bash
# A/B of the number of hardware queues the runtime uses (GPU_MAX_HW_QUEUES; the library asks for 8)
cd /root/repo
for v in 8 4 6 12 16; do
  for i in 1 2; do
    GPU_MAX_HW_QUEUES=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
    python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); k=d["kernels"]
print("$v", round(d["ms_per_step"],2), "em", round(k["em_sparse_ms_per_step"],2), "collapse", round(k.get("collapse_ms_per_step",0),2))
PY
  done
done
