#!/bin/bash
# lanes A/B on one box: host lanes 1, 2, 3 (x host threads default / 4)
cd /root/repo
for rep in 1 2; do
for lanes in 1 2 3; do
  for t in "" 4; do
    RPVG_AMD_LANES=$lanes ${t:+RPVG_AMD_HOST_THREADS=$t} python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $lanes threads ${t:-default}', 'ms_per_step', round(d['ms_per_step'],2), 'resident', round(d['ms_per_step_resident'],2), 'active', round(d.get('gpu_active_frac') or 0,2))"
  done
done
done
