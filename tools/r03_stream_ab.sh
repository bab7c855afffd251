#!/bin/bash
# A/B of the thread count of streamed EM problems (RPVG_HIP_EM_STREAM_SMALL: rows + entries above which a streamed problem gets 1 024 threads)
cd /root/repo
for v in default 0 8192; do
  for i in 1 2; do
    if [ $v = default ]; then python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
    else RPVG_HIP_EM_STREAM_SMALL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json; fi
    python - <<PY
import json
d=json.loads(open("/tmp/b.json").read()); e=d["em_kernels"]
print("$v", round(d["ms_per_step"],2), {k:(round(v["ms_per_launch"],3), round(v["us_per_iteration_of_slowest"],1), v["problems_per_launch"]) for k,v in e.items() if "false" in k})
PY
  done
done
