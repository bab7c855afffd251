#!/bin/bash
# same-box A/B: the tree under _ab_old (a build of an earlier commit) against the working tree; alternating runs
cd /root/repo
for i in 1 2 3; do
  for t in _ab_old .; do
    (cd $t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', 'ms_per_step', round(d['ms_per_step'],2), 'resident', round(d.get('ms_per_step_resident', d['ms_per_step']),2))")
  done
done
