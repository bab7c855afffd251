#!/bin/bash
# S5 A/B on one box: tools/r03_s5_ab.sh "<env A>" "<env B>" ... ; three interleaved rounds, ms per step (resident)
cd /root/repo
for round in 1 2 3; do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --workload s5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['ms_per_step_resident'],2))")
    echo "round $round [$cfg] $v"
  done
done
