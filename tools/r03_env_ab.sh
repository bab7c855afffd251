#!/bin/bash
# same-box A/B of environment knobs: tools/r03_env_ab.sh "<VAR=value ...>" "<VAR=value ...>" ...  ("-" = no knob); 3 rounds, interleaved
cd /root/repo
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
    else env $v timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json; fi
    python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); k=d["kernels"]
print("%-50s" % "$v", round(d["ms_per_step"],2), "resident", round(d.get("ms_per_step_resident",0),2), "em", round(k["em_sparse_ms_per_step"],2), "collapse", round(k.get("collapse_ms_per_step",0),2))
PY
  done
done
